// Host check of closest_view_local's exactness claim (3dobjecttracking_amd/csrc/m3t_view_rows.h + the selection rule
// of closest_view_local in m3t_kernels.hip): whenever the row of the previous view vouches for a direction -- its f32
// dot product with the previous view exceeds the row's threshold -- the arg-max over the ROW (largest f32 dot product,
// lowest view index among equals) equals the arg-max over ALL views (RegionModel::GetClosestView's scan: first
// maximum wins).  Directions: small and large moves away from random views, so that both outcomes occur.
//   view_rows_check VIEWS.f32 N_VIEWS [N_DIRECTIONS]     prints "directions N vouched V mismatches M"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../3dobjecttracking_amd/csrc/m3t_view_rows.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int n_views = std::atoi(argv[2]);
  const long n_dir = argc > 3 ? std::atol(argv[3]) : 200000;
  std::vector<float> ori(size_t(n_views) * 3);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(ori.data(), 4, ori.size(), f) != ori.size()) return 3;
  std::fclose(f);
  const std::vector<float> rows = m3t_view_rows(ori.data(), n_views);
  std::mt19937 rng(3);
  std::normal_distribution<float> gauss(0.0f, 1.0f);
  std::uniform_int_distribution<int> pick(0, n_views - 1);
  std::uniform_real_distribution<float> uni(0.0f, 1.0f);
  long vouched = 0, mismatches = 0;
  for (long i = 0; i < n_dir; ++i) {
    const int prev = pick(rng);
    // a unit direction near view `prev`: angular offsets from micro-radians to a few view spacings
    const float spread = std::pow(10.0f, -6.0f + 5.7f * uni(rng));
    float o[3];
    float norm = 0.0f;
    for (int k = 0; k < 3; ++k) { o[k] = ori[size_t(prev) * 3 + k] + spread * gauss(rng); norm += o[k] * o[k]; }
    norm = std::sqrt(norm);
    for (int k = 0; k < 3; ++k) o[k] /= norm;
    // the scan over all views, the kernels' expression: (o0 x + o1 y) + o2 z, first maximum wins
    float best = -1.0f;
    int full = 0;
    for (int v = 0; v < n_views; ++v) {
      const float* n = ori.data() + size_t(v) * 3;
      const float d = (o[0] * n[0] + o[1] * n[1]) + o[2] * n[2];
      if (d > best) { best = d; full = v; }
    }
    // closest_view_local
    const float* row = rows.data() + size_t(prev) * M3T_VIEW_ROW * 4;
    const float d_prev = (o[0] * row[0] + o[1] * row[1]) + o[2] * row[2];
    if (!(d_prev > row[(M3T_VIEW_ROW - 1) * 4])) continue;  // the row does not vouch: the kernel scans all views
    ++vouched;
    float lbest = -2.0f;
    int local = 1 << 30;
    for (int k = 0; k <= M3T_VIEW_NEIGHBORS; ++k) {
      const float d = (o[0] * row[k * 4] + o[1] * row[k * 4 + 1]) + o[2] * row[k * 4 + 2];
      int id;
      std::memcpy(&id, row + k * 4 + 3, 4);
      if (d > lbest || (d == lbest && id < local)) { lbest = d; local = id; }
    }
    mismatches += local != full;
  }
  std::printf("directions %ld vouched %ld mismatches %ld\n", n_dir, vouched, mismatches);
  return mismatches == 0 ? 0 : 1;
}
