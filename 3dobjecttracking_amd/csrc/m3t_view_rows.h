// m3t_view_rows.h -- host side of closest_view_local (m3t_kernels.hip): for every view of a sparse viewpoint model
// its M3T_VIEW_NEIGHBORS nearest views and how far a viewing direction may be from the view for the arg-max of
// RegionModel::GetClosestView (region_model.cpp:105-130) over ALL views to lie among them.  With R = the angle to
// the nearest view that is NOT in the row, a direction closer to the view than R / 2 is closer to it than to any view
// outside the row (triangle inequality on the sphere).  The threshold stored is cos(R / 2 - 2e-3 rad): the slack
// dwarfs every rounding error of the f32 dot products.  Plain C++ (no HIP): tests/cpp/view_rows_check.cpp runs it on
// the host against the scan over all views.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#ifndef M3T_VIEW_NEIGHBORS
#define M3T_VIEW_NEIGHBORS 18
#define M3T_VIEW_ROW (M3T_VIEW_NEIGHBORS + 2)
#endif

// orientations: [n_views][3]; returns [n_views][M3T_VIEW_ROW][4] floats: entry 0 the view itself, entries
// 1 .. M3T_VIEW_NEIGHBORS its nearest views ({x, y, z, id bits}), the last entry {threshold, 0, 0, 0}
// (2.0 = never met).  n_views must exceed M3T_VIEW_ROW.
inline std::vector<float> m3t_view_rows(const float* ori, int n_views) {
  std::vector<float> rows(size_t(n_views) * M3T_VIEW_ROW * 4, 0.0f);
  std::vector<std::pair<float, int>> by_dot(static_cast<size_t>(n_views));
  for (int v = 0; v < n_views; ++v) {
    const float* a = ori + size_t(v) * 3;
    for (int w = 0; w < n_views; ++w) {
      const float* b = ori + size_t(w) * 3;
      by_dot[size_t(w)] = {w == v ? 4.0f : float(double(a[0]) * b[0] + double(a[1]) * b[1] + double(a[2]) * b[2]), w};
    }
    std::partial_sort(by_dot.begin(), by_dot.begin() + M3T_VIEW_NEIGHBORS + 2, by_dot.end(),
                      [](const std::pair<float, int>& x, const std::pair<float, int>& y) {
                        return x.first > y.first || (x.first == y.first && x.second < y.second);
                      });
    float* row = rows.data() + size_t(v) * M3T_VIEW_ROW * 4;
    for (int k = 0; k <= M3T_VIEW_NEIGHBORS; ++k) {  // by_dot[0] is the view itself
      const int w = by_dot[size_t(k)].second;
      std::memcpy(row + k * 4, ori + size_t(w) * 3, 12);
      std::memcpy(row + k * 4 + 3, &w, 4);
    }
    const double first_outside = std::min(1.0, std::max(-1.0, double(by_dot[M3T_VIEW_NEIGHBORS + 1].first)));
    const double half = 0.5 * std::acos(first_outside) - 2.0e-3;
    row[(M3T_VIEW_ROW - 1) * 4] = half > 0.0 ? float(std::cos(half)) : 2.0f;  // 2: never met
  }
  return rows;
}
