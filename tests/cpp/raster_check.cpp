// Host check of 3dobjecttracking_amd/csrc/m3t_raster.h: the row scan of the focused renderers' kernels (raster_row,
// serial over a bounding box and in the 8- / 32-pixel pieces the workgroup paths cut it into) against the per-pixel definition
// (raster_pixel) on random triangles -- slivers, triangles with vertices on pixel centres and on pixel edges,
// axis-parallel edges, triangles that leave the image -- word for word.  Prints "triangles N covered P mismatches M".
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../3dobjecttracking_amd/csrc/m3t_raster.h"

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 200000;
  const int S = 200;
  std::mt19937 rng(12345);
  std::uniform_real_distribution<float> uni(-1.15f, 1.15f);
  std::uniform_int_distribution<int> grid(-10, 2 * S + 10), kind(0, 5);
  RasterM44 trans{};
  trans(0, 0) = trans(1, 1) = trans(2, 2) = 1.0f;
  trans(3, 3) = 1.0f;
  long long covered = 0, mismatches = 0, accepted = 0;
  std::vector<uint32_t> a(S * S), b(S * S), c(S * S);
  for (int it = 0; it < n; ++it) {
    float v[9];
    const int k = kind(rng);
    for (int i = 0; i < 3; ++i) {
      if (k <= 1) {  // anywhere
        v[3 * i] = uni(rng);
        v[3 * i + 1] = uni(rng);
      } else if (k == 2) {  // small, anywhere
        const float cx = uni(rng), cy = uni(rng);
        v[3 * i] = cx + 0.02f * uni(rng);
        v[3 * i + 1] = cy + 0.02f * uni(rng);
        if (i > 0) { v[3 * i] = v[0] + 0.03f * uni(rng); v[3 * i + 1] = v[1] + 0.03f * uni(rng); }
      } else if (k == 3) {  // sliver: third vertex almost on the line through the first two
        v[3 * i] = uni(rng);
        v[3 * i + 1] = uni(rng);
        if (i == 2) {
          const float s = 0.5f + 0.4f * uni(rng);
          v[6] = v[0] + s * (v[3] - v[0]) + 0.004f * uni(rng);
          v[7] = v[1] + s * (v[4] - v[1]) + 0.004f * uni(rng);
        }
      } else {  // on the half-pixel grid: pixel centres (odd multiples of 1/2) and pixel edges
        v[3 * i] = (float)grid(rng) * 0.5f / (0.5f * S) - 1.0f;
        v[3 * i + 1] = (float)grid(rng) * 0.5f / (0.5f * S) - 1.0f;
        if (k == 5 && i == 1) v[4] = v[1];  // a horizontal edge
      }
      v[3 * i + 2] = 0.9f * uni(rng);
    }
    const int idx[3] = {0, 1, 2};
    RasterTriangle t;
    if (!raster_setup(trans, v, idx, 0, (it & 7) == 0, S, t)) continue;
    ++accepted;
    const uint32_t low = (uint32_t)(it & 0xff);
    std::fill(a.begin(), a.end(), 0xffffffffu);
    std::fill(b.begin(), b.end(), 0xffffffffu);
    std::fill(c.begin(), c.end(), 0xffffffffu);
    auto sink_a = [&](int px, int py, uint32_t w) { a[py * S + px] = w < a[py * S + px] ? w : a[py * S + px]; ++covered; };
    auto sink_b = [&](int px, int py, uint32_t w) { b[py * S + px] = w < b[py * S + px] ? w : b[py * S + px]; };
    auto sink_c = [&](int px, int py, uint32_t w) { c[py * S + px] = w < c[py * S + px] ? w : c[py * S + px]; };
    for (int py = t.y0; py <= t.y1; ++py)
      for (int px = t.x0; px <= t.x1; ++px) raster_pixel(t, px, py, low, sink_a);
    for (int py = t.y0; py <= t.y1; ++py) raster_row(t, py, t.x0, t.x1, low, sink_b);
    // (the pieces of the kernels: 8 pixels in focused_resolve_kernel, 32 in focused_raster_kernel)
    const int piece = (it & 1) ? 8 : 32;
    for (int py = t.y0; py <= t.y1; ++py)
      for (int xa = t.x0; xa <= t.x1; xa += piece)
        raster_row(t, py, xa, xa + piece - 1 < t.x1 ? xa + piece - 1 : t.x1, low, sink_c);
    for (int i = 0; i < S * S; ++i) mismatches += (a[i] != b[i]) + (a[i] != c[i]);
  }
  // raster_quotient against the division it replaces: integer operands as the edge functions and areas are
  {
    std::mt19937_64 r64(99);
    for (int bits = 2; bits <= 52; ++bits)
      for (int i = 0; i < 1000000; ++i) {
        const unsigned long long area = (r64() >> (64 - bits)) | 1ull;
        const unsigned long long e = (i & 1) ? r64() % (area + 1) : r64() >> (64 - bits);
        const double b = (double)area, a = (double)e;
        mismatches += raster_quotient(a, b, 1.0 / b) != a / b;
      }
  }
  std::printf("triangles %lld covered %lld mismatches %lld\n", accepted, covered, mismatches);
  return mismatches == 0 ? 0 : 1;
}
