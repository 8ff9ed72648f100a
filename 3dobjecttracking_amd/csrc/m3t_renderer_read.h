// m3t_renderer_read.h -- what the modalities read from a focused rendering (region_modality.cpp:1157-1229, 1293-1341,
// 1391-1431; depth_modality.cpp:778-824), free of device-only constructs so that tests/cpp/renderer_read_check.cpp
// can run them on the host beside the loops as the reference writes them (one sample, one decision, next sample:
// tests/cpp/renderer_read_reference.h, test infrastructure):
//   what the kernels call: all samples a loop can touch are requested first (independent, predicated
//                loads: one memory round trip instead of up to 36 in a row), then the reference's decisions are taken
//                in the reference's order on the loaded values.  Same samples, same comparisons, same result.
// Pointer types are template parameters (the kernels pass address-space-1 pointers).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/m3t_types.h"

#if defined(__HIPCC__)
#define M3T_READ_FN __device__ __forceinline__
#else
#define M3T_READ_FN static inline
#endif

M3T_READ_FN int m3t_read_f2i(float v) { return (int)v; }  // truncation like C int(float)

// crop of the rendering: image coordinate -> focused image coordinate (FocusedRenderer, renderer.cpp:348-405)
struct FocusedCrop {
  float corner_u, corner_v, scale;
  int image_size;
};

// ---- IsLineUnoccludedModeled :1391-1431 / IsPointUnoccludedModeled depth_modality.cpp:778-824: the minimum of the
// <= 6 x 6 strided samples of the rendered depth around a point
struct ModeledWindow {
  int u_min, v_min, u_max, v_max, stride;
};
M3T_READ_FN ModeledWindow modeled_window(const FocusedCrop& c, float center_u, float center_v, float diameter) {
  const int size_minus_1 = c.image_size - 1;
  ModeledWindow w;
  w.stride = m3t_read_f2i(diameter / M3T_MAX_N_OCCLUSION_STRIDES + 1.0f);
  int n_strides = m3t_read_f2i(diameter / w.stride + 0.5f);
  int rounded_diameter = n_strides * w.stride;
  float rounded_radius = 0.5f * (float)rounded_diameter;
  float focused_center_u = (center_u - c.corner_u) * c.scale;
  float focused_center_v = (center_v - c.corner_v) * c.scale;
  w.u_min = m3t_read_f2i(focused_center_u - rounded_radius + 0.5f);
  w.v_min = m3t_read_f2i(focused_center_v - rounded_radius + 0.5f);
  w.u_max = w.u_min + rounded_diameter;
  w.v_max = w.v_min + rounded_diameter;
  w.u_min = w.u_min > 0 ? w.u_min : 0;
  w.v_min = w.v_min > 0 ? w.v_min : 0;
  w.u_max = w.u_max < size_minus_1 ? w.u_max : size_minus_1;
  w.v_max = w.v_max < size_minus_1 ? w.v_max : size_minus_1;
  return w;
}
// the window spans at most M3T_MAX_N_OCCLUSION_STRIDES strides (stride > diameter / 5, so diameter / stride + 0.5
// truncates to at most 5): at most 6 samples per row and column, before and after the clamps
template <typename DepthPtr>
M3T_READ_FN unsigned short modeled_window_min(DepthPtr depth_image, const FocusedCrop& c, float center_u, float center_v,
                                              float diameter) {
  const ModeledWindow w = modeled_window(c, center_u, center_v, diameter);
  constexpr int kSamples = M3T_MAX_N_OCCLUSION_STRIDES + 1;
  unsigned short d[kSamples][kSamples];
#pragma unroll
  for (int j = 0; j < kSamples; ++j) {
    const int v = w.v_min + j * w.stride;
#pragma unroll
    for (int i = 0; i < kSamples; ++i) {
      const int u = w.u_min + i * w.stride;
      d[j][i] = 65535;
      if (v <= w.v_max && u <= w.u_max) d[j][i] = depth_image[(size_t)v * c.image_size + u];
    }
  }
  unsigned short min_value = 65535;
#pragma unroll
  for (int j = 0; j < kSamples; ++j)
#pragma unroll
    for (int i = 0; i < kSamples; ++i) min_value = d[j][i] < min_value ? d[j][i] : min_value;
  return min_value;
}

// ---- silhouette id at a focused image coordinate, -1: off the image
template <typename IdPtr>
M3T_READ_FN int silhouette_at(IdPtr silhouette_image, int image_size, float u, float v) {
  const float size = (float)image_size;
  if (u >= size || u < 0.0f || v >= size || v < 0.0f) return -1;
  return silhouette_image[(size_t)m3t_read_f2i(v) * image_size + m3t_read_f2i(u)];
}
// the ids at start, start + step, ... (count samples; the coordinates advance by repeated addition like the reference's
// loops, so every sample is the reference's sample)
template <int COUNT, typename IdPtr>
M3T_READ_FN void silhouette_run(IdPtr silhouette_image, int image_size, float u, float v, float step_u, float step_v,
                                int first, int (&ids)[COUNT]) {
#pragma unroll
  for (int i = 0; i < COUNT; ++i) {
    ids[i] = -2;  // not sampled
    if (i >= first) {
      ids[i] = silhouette_at(silhouette_image, image_size, u, v);
      u += step_u;
      v += step_v;
    }
  }
}

// ---- IsDynamicLineRegionSufficient :1293-1341 (an off-image coordinate in the foreground loop, which the reference
// reads unchecked, counts as another region)
template <typename IdPtr>
M3T_READ_FN bool dynamic_line_region_sufficient(IdPtr silhouette_image, const FocusedCrop& c, int region_id,
                                                float min_continuous_distance, float fscale, float center_u,
                                                float center_v, float normal_u, float normal_v) {
  const float scale = c.scale;
  float focused_min_continuous_distance = min_continuous_distance * fscale * scale;
  float focused_stride = fmaxf((focused_min_continuous_distance - M3T_REGION_OFFSET) / (float)M3T_N_REGION_STRIDE, 0.0f);
  float stride_u = focused_stride * normal_u;
  float stride_v = focused_stride * normal_v;
  float offset_u = M3T_REGION_OFFSET * normal_u;
  float offset_v = M3T_REGION_OFFSET * normal_v;
  float focused_center_u = 0.5f + (center_u - c.corner_u) * scale;
  float focused_center_v = 0.5f + (center_v - c.corner_v) * scale;
  constexpr int kSamples = M3T_N_REGION_STRIDE + 1;
  int fg[kSamples], bg[kSamples];
  silhouette_run<kSamples>(silhouette_image, c.image_size, focused_center_u - offset_u, focused_center_v - offset_v,
                           -stride_u, -stride_v, 0, fg);
  silhouette_run<kSamples>(silhouette_image, c.image_size, focused_center_u + offset_u, focused_center_v + offset_v,
                           stride_u, stride_v, 0, bg);
  bool sufficient = true;
#pragma unroll
  for (int i = 0; i < kSamples; ++i) sufficient = sufficient && fg[i] == region_id;
  bool open = true;  // the background loop has not left through its break
#pragma unroll
  for (int i = 0; i < kSamples; ++i) {
    if (open && bg[i] < 0) open = false;
    if (open && bg[i] == region_id) sufficient = false;
  }
  return sufficient;
}

// ---- DynamicRegionDistance :1157-1229 (with the assignment to the *foreground* distance in the background loop)
template <typename IdPtr>
M3T_READ_FN void dynamic_region_distance(IdPtr silhouette_image, const FocusedCrop& c, int region_id,
                                         float max_considered_line_length, float unconsidered_line_length, float center_u,
                                         float center_v, float normal_u, float normal_v, float* foreground,
                                         float* background) {
  const float scale = c.scale;
  float stride = max_considered_line_length / (float)M3T_N_REGION_STRIDE;
  float focused_stride = stride * scale;
  float focused_stride_u = focused_stride * normal_u;
  float focused_stride_v = focused_stride * normal_v;
  float delta_start = M3T_REGION_OFFSET / scale - unconsidered_line_length;
  int i_start = m3t_read_f2i(delta_start / stride + 1.0f);
  i_start = i_start > 0 ? i_start : 0;
  float offset = unconsidered_line_length + (float)i_start * stride;
  float focused_offset = offset * scale;
  float focused_offset_u = focused_offset * normal_u;
  float focused_offset_v = focused_offset * normal_v;
  float focused_center_u = 0.5f + (center_u - c.corner_u) * scale;
  float focused_center_v = 0.5f + (center_v - c.corner_v) * scale;
  constexpr int kSamples = M3T_N_REGION_STRIDE + 1;
  int fg[kSamples], bg[kSamples];
  silhouette_run<kSamples>(silhouette_image, c.image_size, focused_center_u - focused_offset_u,
                           focused_center_v - focused_offset_v, -focused_stride_u, -focused_stride_v, i_start, fg);
  silhouette_run<kSamples>(silhouette_image, c.image_size, focused_center_u + focused_offset_u,
                           focused_center_v + focused_offset_v, focused_stride_u, focused_stride_v, i_start, bg);
  bool open = true;
#pragma unroll
  for (int i = 0; i < kSamples; ++i) {
    if (open && i >= i_start) {
      if (fg[i] < 0) {
        *foreground = stride * (float)i;
        open = false;
      } else if (fg[i] != region_id) {
        *foreground = i == i_start ? 0.0f : stride * (float)i;
        open = false;
      }
    }
  }
  open = true;
#pragma unroll
  for (int i = 0; i < kSamples; ++i) {
    if (open && i >= i_start) {
      if (bg[i] < 0) {
        *background = max_considered_line_length;
        open = false;
      } else if (bg[i] == region_id) {
        if (i == i_start) *background = 0.0f;
        else *foreground = stride * (float)i;
        open = false;
      }
    }
  }
}
