// extern "C" face of m3t_raster.h for tools/raster_stats.py: bounding-box and coverage statistics of a mesh under one
// focused projection (what the rasteriser's work distribution has to cope with)
#include <vector>

#include "../../3dobjecttracking_amd/csrc/m3t_raster.h"

extern "C" {
// out: [0] triangles that survive set-up, [1] sum of box pixels, [2] covered pixels, [3] boxes > 192 px, [4] largest box,
// [5..5+15] histogram of box sizes by power of two (1, 2-3, 4-7, ...), [21..36] the same for covered pixels per triangle
void raster_stats(const float* trans16, const float* vertices, const int* triangles, int n_triangles, int culling, int S,
                  long long* out) {
  RasterM44 trans;
  for (int i = 0; i < 16; ++i) trans.m[i] = trans16[i];
  for (int i = 0; i < 40; ++i) out[i] = 0;
  for (int t = 0; t < n_triangles; ++t) {
    RasterTriangle tri;
    if (!raster_setup(trans, vertices, triangles, t, culling != 0, S, tri)) continue;
    const long long box = (long long)(tri.x1 - tri.x0 + 1) * (tri.y1 - tri.y0 + 1);
    long long covered = 0;
    auto sink = [&](int, int, uint32_t) { ++covered; };
    for (int py = tri.y0; py <= tri.y1; ++py) raster_row(tri, py, tri.x0, tri.x1, 0u, sink);
    out[0] += 1;
    out[1] += box;
    out[2] += covered;
    out[3] += box > 192;
    if (box > out[4]) out[4] = box;
    int b = 0;
    while ((1ll << (b + 1)) <= box && b < 15) ++b;
    out[5 + b] += 1;
    int c = 0;
    while ((1ll << (c + 1)) <= covered && c < 15) ++c;
    if (covered > 0) out[21 + c] += 1;
  }
}
}

// The whole rendering of one body list into a z-buffer the way focused_setup_kernel + focused_resolve_kernel do it
// (set-up, survivors, row scan, minimum of the packed words, unpack), for tests/test_raster_scene.py: against the
// oracle's focused renderer on the same scene.  packed: [S * S] words, 0xffffffff = nothing (the caller clears it).
extern "C" void raster_body(const float* trans16, const float* vertices, const int* triangles, int n_triangles, int culling,
                            int S, unsigned low_bits, unsigned* packed) {
  RasterM44 trans;
  for (int i = 0; i < 16; ++i) trans.m[i] = trans16[i];
  std::vector<RasterTriangle> survivors;
  for (int t = 0; t < n_triangles; ++t) {
    RasterTriangle tri;
    if (raster_setup(trans, vertices, triangles, t, culling != 0, S, tri)) survivors.push_back(tri);
  }
  auto sink = [&](int px, int py, uint32_t word) {
    if (word < packed[py * S + px]) packed[py * S + px] = word;
  };
  for (const RasterTriangle& tri : survivors) {
    const int pixels = (tri.x1 - tri.x0 + 1) * (tri.y1 - tri.y0 + 1);
    if (pixels <= 192) {
      for (int py = tri.y0; py <= tri.y1; ++py) raster_row(tri, py, tri.x0, tri.x1, low_bits, sink);
    } else {  // the workgroup path: 32-pixel pieces of the rows
      const int pieces = (tri.x1 - tri.x0 + 32) / 32, total = pieces * (tri.y1 - tri.y0 + 1);
      for (int k = 0; k < total; ++k) {
        const int row = k / pieces, xa = tri.x0 + (k - row * pieces) * 32;
        raster_row(tri, tri.y0 + row, xa, xa + 31 < tri.x1 ? xa + 31 : tri.x1, low_bits, sink);
      }
    }
  }
}
