// The claim behind ldlt_solve_rows' pivot order (m3t_links.hip): Eigen's LDLT<Lower> selection -- at step k take the
// first largest |diagonal| among positions k .. n-1 and swap it to position k -- leaves DISTINCT values in descending
// order, so "position p holds the row whose rank is p" gives the same transposition result as the step-by-step
// selection.  Checked on random diagonals (and that ties really differ, which is why the kernel detects them).
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

static std::vector<int> by_selection(std::vector<float> d) {
  const int n = (int)d.size();
  std::vector<int> src(n);
  std::iota(src.begin(), src.end(), 0);
  for (int k = 0; k < n; ++k) {
    int piv = k;
    for (int i = k + 1; i < n; ++i)
      if (d[i] > d[piv]) piv = i;  // the first largest
    std::swap(d[k], d[piv]);
    std::swap(src[k], src[piv]);
  }
  return src;
}
static std::vector<int> by_rank(const std::vector<float>& d) {
  const int n = (int)d.size();
  std::vector<int> src(n, -1);
  for (int i = 0; i < n; ++i) {
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += d[j] > d[i] ? 1 : 0;
    src[rank] = i;
  }
  return src;
}

int main() {
  std::mt19937 rng(99);
  std::uniform_real_distribution<float> uni(0.0f, 10.0f);
  long long cases = 0, mismatches = 0, ties_that_differ = 0;
  for (int it = 0; it < 200000; ++it) {
    const int n = 1 + (int)(rng() % 16);
    std::vector<float> d(n);
    for (float& v : d) v = uni(rng);
    bool distinct = true;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j) distinct = distinct && d[i] != d[j];
    if (!distinct) continue;
    ++cases;
    mismatches += by_selection(d) != by_rank(d);
  }
  for (int it = 0; it < 20000; ++it) {  // with ties the swaps matter: the kernel must not use ranks then
    const int n = 3 + (int)(rng() % 6);
    std::vector<float> d(n);
    for (float& v : d) v = (float)(rng() % 3);
    std::vector<int> a = by_selection(d);
    // the stable descending order of the tied values
    std::vector<int> b(n);
    std::iota(b.begin(), b.end(), 0);
    std::stable_sort(b.begin(), b.end(), [&](int x, int y) { return d[x] > d[y]; });
    ties_that_differ += a != b;
  }
  // round 6: the kernel's test for equal values -- two rows with the same value share a rank, so some position finds no
  // row -- and its padding (lanes behind the n rows hold -1: they rank behind every row and take no position below n)
  long long detection_errors = 0;
  for (int it = 0; it < 200000; ++it) {
    const int n = 1 + (int)(rng() % 16), N = 16;
    std::vector<float> d(N, -1.0f);
    const bool coarse = (rng() & 1) != 0;
    for (int i = 0; i < n; ++i) d[i] = coarse ? (float)(rng() % 5) : uni(rng);
    bool ties = false;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j) ties = ties || d[i] == d[j];
    std::vector<int> src(N, -1);
    for (int lane = 0; lane < N; ++lane) {
      int rank = 0;
      for (int j = 0; j < N; ++j) rank += d[j] > d[lane] ? 1 : 0;
      for (int p = 0; p < N; ++p)  // "if (row_lane(rank, j) == lane) src = j", seen from position p
        if (rank == p) src[p] = lane;
    }
    bool unfilled = false;
    for (int p = 0; p < n; ++p) unfilled = unfilled || src[p] < 0;
    detection_errors += unfilled != ties;
    if (!ties) {
      std::vector<float> rows(d.begin(), d.begin() + n);
      std::vector<int> want = by_selection(rows);
      for (int p = 0; p < n; ++p) detection_errors += src[p] != want[p];
    }
  }
  std::printf("cases %lld mismatches %lld ties_that_differ %lld\n", cases, mismatches, ties_that_differ);
  if (detection_errors) std::printf("tie detection errors %lld\n", detection_errors);
  return mismatches == 0 && detection_errors == 0 ? 0 : 1;
}
