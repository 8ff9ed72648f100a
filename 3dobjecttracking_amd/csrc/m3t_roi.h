// m3t_roi.h -- which part of a camera frame a tracking step can read for one body (ROI ingest): a
// conservative rectangle from the pose a search runs at, a box around the model's points and the modality's parameters.
// Every pixel the path reads lies within a bounded distance of the projection of a point of the body:
//   RegionModality, colour frame   correspondence line of (function_length + distribution_length - 1) x scale pixels
//                                  centred on a projected contour point (region_modality.cpp:1433-1508); histogram
//                                  lines of unconsidered + max_considered_line_length along the normal (:1640-1780)
//   RegionModality, depth frame    the <= 6 x 6 window of IsLineUnoccludedMeasured around the projected point, diameter
//                                  2 x measured_occlusion_radius x fu / z (:1343-1389)
//   DepthModality, depth frame     the window of FindCorrespondence, diameter 2 x considered_distance x fu / z
//                                  (x z with depth scaling; depth_modality.cpp:826-884), and the measured-occlusion
//                                  window (:736-776)
// and those points are data points of the modality's sparse viewpoint model moved by the pose: the rectangle is the
// bounding rectangle of the projected box around the model's points -- cut down to the outline of the ellipsoid that
// holds them where that is tighter (round 5) --, widened by a reach in pixels plus a reach in metres at the near side.
// Host and device code (plain functions, C++ only for a default argument); tests/test_roi_bound.py checks the rectangles against the oracle: frames
// scrambled outside them leave every pose and histogram of a tracked sequence unchanged.
#pragma once
#include <math.h>

#include "../../include/m3t_types.h"

#if defined(__HIPCC__)
#define M3T_ROI_FN __host__ __device__ __forceinline__
#else
#define M3T_ROI_FN static inline
#endif

typedef struct m3t_roi_rect {
  int x0, y0, x1, y1;  // inclusive pixel bounds; empty when x1 < x0
} m3t_roi_rect;

M3T_ROI_FN m3t_roi_rect m3t_roi_empty(void) {
  m3t_roi_rect r = {1 << 30, 1 << 30, -(1 << 30), -(1 << 30)};
  return r;
}
M3T_ROI_FN m3t_roi_rect m3t_roi_union(m3t_roi_rect a, m3t_roi_rect b) {
  m3t_roi_rect r;
  r.x0 = a.x0 < b.x0 ? a.x0 : b.x0;
  r.y0 = a.y0 < b.y0 ? a.y0 : b.y0;
  r.x1 = a.x1 > b.x1 ? a.x1 : b.x1;
  r.y1 = a.y1 > b.y1 ? a.y1 : b.y1;
  return r;
}
M3T_ROI_FN int m3t_roi_contains(m3t_roi_rect outer, m3t_roi_rect inner) {
  return inner.x1 < inner.x0 || (outer.x0 <= inner.x0 && outer.y0 <= inner.y0 && outer.x1 >= inner.x1 && outer.y1 >= inner.y1);
}

// a window of the reference's strided scans around a centre: rounded_radius + the roundings of its corners
M3T_ROI_FN float m3t_roi_window_reach(float diameter) { return 0.55f * diameter + 2.0f; }

// reach in pixels of the colour frame for a RegionModality: the correspondence lines of search corr_iteration
// (scale of that iteration, region_modality.cpp:1011-1023), and the histogram lines (StartModality, CalculateResults)
M3T_ROI_FN float m3t_roi_region_line_reach(const m3t_region_modality_params* p, int corr_iteration) {
  const int last = p->n_scales - 1;
  const int scale = p->n_scales > 0 ? p->scales[corr_iteration < last ? corr_iteration : last] : 1;
  return 0.5f * (float)((p->function_length + p->distribution_length - 1) * scale) + 2.0f;
}
M3T_ROI_FN float m3t_roi_region_histogram_reach(const m3t_region_modality_params* p) {
  return p->unconsidered_line_length + p->max_considered_line_length + 2.0f;
}
// reach of a RegionModality in its depth frame (measured occlusions): metres at the point's depth, plus pixels
M3T_ROI_FN void m3t_roi_region_depth_reach(const m3t_region_modality_params* p, float* reach_m, float* reach_px) {
  *reach_m = p->measure_occlusions ? 1.1f * p->measured_occlusion_radius : 0.0f;
  *reach_px = 2.0f;
}
// reach of a DepthModality in its depth frame
M3T_ROI_FN void m3t_roi_depth_reach(const m3t_depth_modality_params* p, float fu, float* reach_m, float* reach_px) {
  float distance = 0.0f;
  for (int i = 0; i < p->n_considered_distances; ++i)
    distance = p->considered_distances[i] > distance ? p->considered_distances[i] : distance;
  if (p->measure_occlusions && p->measured_occlusion_radius > distance) distance = p->measured_occlusion_radius;
  if (p->use_depth_scaling) {  // distances are multiples of the point's depth: a fixed number of pixels
    *reach_m = 0.0f;
    *reach_px = 1.1f * distance * fu + 2.0f;
  } else {
    *reach_m = 1.1f * distance;
    *reach_px = 2.0f;
  }
}

// The rectangle of one body in one camera.  body2camera = 4 x 4 column-major pose of the body frame in the camera
// frame; box_min / box_max = a box in the body frame that holds every point the modality works with (the data points
// of its sparse viewpoint model: line and point centres are model points moved by the pose).  A perspective
// projection maps the box into the convex hull of its projected corners as long as the box lies in front of the
// camera; a box that reaches the camera plane gives the whole frame.  reach_px + reach_m (metres, taken at the box's
// nearest depth) widen the hull's bounding rectangle.
// (the two halves of m3t_roi_body: the tracking kernels' ROI guard takes one corner per lane and joins the eight with
// lane exchanges -- min / max are exact whatever their order, so both ways give the same rectangle)
// One corner of the box in the image: returns 0 when it does not lie in front of the camera.
M3T_ROI_FN int m3t_roi_corner(const float* body2camera, const float* box_min, const float* box_max, int corner,
                              const m3t_intrinsics* intr, float* u, float* v, float* z_out) {
  const float bx = (corner & 1) ? box_max[0] : box_min[0];
  const float by = (corner & 2) ? box_max[1] : box_min[1];
  const float bz = (corner & 4) ? box_max[2] : box_min[2];
  const float x = body2camera[0] * bx + body2camera[4] * by + body2camera[8] * bz + body2camera[12];
  const float y = body2camera[1] * bx + body2camera[5] * by + body2camera[9] * bz + body2camera[13];
  const float z = body2camera[2] * bx + body2camera[6] * by + body2camera[10] * bz + body2camera[14];
  *z_out = z;
  if (!(z > 1e-3f)) return 0;
  *u = x * intr->fu / z + intr->ppu;
  *v = y * intr->fv / z + intr->ppv;
  return 1;
}
// A second, independent bound (round 5): the points also lie inside the ellipsoid around the box's centre whose
// semi-axes are rho x the box's half extents (rho: the largest normalised radius of a point, computed where the model is
// loaded; 1 for points on the inscribed ellipsoid -- a convex body seen from all sides --, sqrt(3) at worst).  Its
// outline in the image is a conic whose axis-parallel tangents have a closed form: with c the centre and S = R D R^T
// the shape matrix in the camera frame (D = diag(semi-axes^2)), the dual conic in normalised image coordinates is
// C* = S - c c^T and the tangents u = const are the roots of C*22 u^2 - 2 C*02 u + C*00 = 0 (v alike, index 1).  The box's
// projected corners over-estimate an ellipsoidal body by the corners' overhang (~ 30 % of the radius); the
// intersection of the two bounds is still a bound.  Returns 0 (and leaves the limits alone) when rho <= 0 or the
// ellipsoid does not lie in front of the camera.
M3T_ROI_FN int m3t_roi_ellipsoid(const float* body2camera, const float* box_min, const float* box_max, float rho,
                                 const m3t_intrinsics* intr, float* u_min, float* u_max, float* v_min, float* v_max,
                                 float* z_min) {
  if (!(rho > 0.0f)) return 0;
  float m[3], d[3];
  for (int k = 0; k < 3; ++k) {
    m[k] = 0.5f * (box_min[k] + box_max[k]);
    const float e = rho * 0.5f * (box_max[k] - box_min[k]);
    d[k] = e * e;
  }
  const float* B = body2camera;
  const float cx = B[0] * m[0] + B[4] * m[1] + B[8] * m[2] + B[12];
  const float cy = B[1] * m[0] + B[5] * m[1] + B[9] * m[2] + B[13];
  const float cz = B[2] * m[0] + B[6] * m[1] + B[10] * m[2] + B[14];
  // S_ij = sum_k B(i, k) B(j, k) d_k (rows 0, 1, 2 of the rotation)
  const float s00 = B[0] * B[0] * d[0] + B[4] * B[4] * d[1] + B[8] * B[8] * d[2];
  const float s11 = B[1] * B[1] * d[0] + B[5] * B[5] * d[1] + B[9] * B[9] * d[2];
  const float s22 = B[2] * B[2] * d[0] + B[6] * B[6] * d[1] + B[10] * B[10] * d[2];
  const float s02 = B[0] * B[2] * d[0] + B[4] * B[6] * d[1] + B[8] * B[10] * d[2];
  const float s12 = B[1] * B[2] * d[0] + B[5] * B[6] * d[1] + B[9] * B[10] * d[2];
  const float depth = sqrtf(s22);         // the ellipsoid's half extent along the optical axis
  if (!(cz - depth > 1e-3f)) return 0;    // it has to lie in front of the camera as a whole
  const float c22 = s22 - cz * cz;        // < 0
  const float c00 = s00 - cx * cx, c02 = s02 - cx * cz;
  const float c11 = s11 - cy * cy, c12 = s12 - cy * cz;
  const float du = c02 * c02 - c00 * c22, dv = c12 * c12 - c11 * c22;
  if (!(du >= 0.0f && dv >= 0.0f && c22 < 0.0f)) return 0;
  const float ru = sqrtf(du), rv = sqrtf(dv);
  // (c22 < 0: the root with + ru is the smaller one)
  *u_min = (c02 + ru) / c22 * intr->fu + intr->ppu;
  *u_max = (c02 - ru) / c22 * intr->fu + intr->ppu;
  *v_min = (c12 + rv) / c22 * intr->fv + intr->ppv;
  *v_max = (c12 - rv) / c22 * intr->fv + intr->ppv;
  *z_min = cz - depth;
  return 1;
}
// ... intersected with the limits of the projected corners (both hold, so the tighter one of each holds)
M3T_ROI_FN void m3t_roi_tighten(const float* body2camera, const float* box_min, const float* box_max, float rho,
                                const m3t_intrinsics* intr, float* u_min, float* u_max, float* v_min, float* v_max,
                                float* z_min) {
  float eu0, eu1, ev0, ev1, ez;
  if (!m3t_roi_ellipsoid(body2camera, box_min, box_max, rho, intr, &eu0, &eu1, &ev0, &ev1, &ez)) return;
  // (half a pixel of slack for the f32 evaluation of the roots: their cancellation error is ~1e-3 pixel)
  eu0 -= 0.5f; ev0 -= 0.5f; eu1 += 0.5f; ev1 += 0.5f;
  *u_min = eu0 > *u_min ? eu0 : *u_min;
  *u_max = eu1 < *u_max ? eu1 : *u_max;
  *v_min = ev0 > *v_min ? ev0 : *v_min;
  *v_max = ev1 < *v_max ? ev1 : *v_max;
  *z_min = ez > *z_min ? ez : *z_min;
}
// The bounding rectangle of the projected corners, widened by the reach and cut to the frame.
M3T_ROI_FN m3t_roi_rect m3t_roi_widen(float u_min, float u_max, float v_min, float v_max, float z_min,
                                      const m3t_intrinsics* intr, float reach_px, float reach_m) {
  m3t_roi_rect full = {0, 0, intr->width - 1, intr->height - 1};
  const float f = intr->fu > intr->fv ? intr->fu : intr->fv;
  const float reach = reach_px + reach_m * f / z_min + 1.0f;
  if (!(u_min > -1.0e7f && u_max < 1.0e7f && v_min > -1.0e7f && v_max < 1.0e7f && reach < 1.0e7f)) return full;
  m3t_roi_rect r;
  r.x0 = (int)floorf(u_min - reach);
  r.y0 = (int)floorf(v_min - reach);
  r.x1 = (int)ceilf(u_max + reach);
  r.y1 = (int)ceilf(v_max + reach);
  if (r.x0 < 0) r.x0 = 0;
  if (r.y0 < 0) r.y0 = 0;
  if (r.x1 > intr->width - 1) r.x1 = intr->width - 1;
  if (r.y1 > intr->height - 1) r.y1 = intr->height - 1;
  return r;
}
M3T_ROI_FN m3t_roi_rect m3t_roi_body(const float* body2camera, const float* box_min, const float* box_max,
                                     const m3t_intrinsics* intr, float reach_px, float reach_m, float rho = 0.0f) {
  m3t_roi_rect full = {0, 0, intr->width - 1, intr->height - 1};
  float u_min = 3.0e38f, u_max = -3.0e38f, v_min = 3.0e38f, v_max = -3.0e38f, z_min = 3.0e38f;
  for (int corner = 0; corner < 8; ++corner) {
    float u, v, z;
    if (!m3t_roi_corner(body2camera, box_min, box_max, corner, intr, &u, &v, &z)) return full;
    u_min = u < u_min ? u : u_min;
    u_max = u > u_max ? u : u_max;
    v_min = v < v_min ? v : v_min;
    v_max = v > v_max ? v : v_max;
    z_min = z < z_min ? z : z_min;
  }
  m3t_roi_tighten(body2camera, box_min, box_max, rho, intr, &u_min, &u_max, &v_min, &v_max, &z_min);
  return m3t_roi_widen(u_min, u_max, v_min, v_max, z_min, intr, reach_px, reach_m);
}
