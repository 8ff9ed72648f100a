// m3t_device.h — device-resident layouts shared by the host-side C-ABI
// implementation (m3t_hip_api.hip) and the gfx950 kernels (m3t_kernels.hip).
//
// Data layout in HBM:
//   * one CameraDev row per camera (intrinsics, world2camera, current frame ptr)
//   * one float[16] column-major body2world pose per body
//   * one RegionModDev / DepthModDev row per modality: parameters, pointers to
//     the (shared) sparse viewpoint model, the per-object histogram tables and
//     the per-object line / point state (structure-of-arrays, field-major)
//   * one RigidOptDev row per rigid-body optimizer
#ifndef M3T_DEVICE_H_
#define M3T_DEVICE_H_

#include <stdint.h>

#include "../../include/m3t_types.h"

// ---- per-line state, field-major: state[field * n_lines_max + line] ----------
enum {
  LS_CX = 0, LS_CY, LS_CZ,          // center_f_body
  LS_CENTER_U, LS_CENTER_V, LS_NORMAL_U, LS_NORMAL_V,
  LS_DELTA_R, LS_NCTS,              // normal_component_to_scale
  LS_MEAN, LS_VAR,                  // measured_variance
  LS_CONT,                          // continuous_distance
  LS_WALK_START,                    // int bits: first major-axis pixel coordinate
  LS_WALK_STEP,                     // minor-axis increment per pixel
  LS_VALID,                         // int bits: bit0 valid, bit1 valid w/o occlusion handling, bit2 horizontal walk, bit3 reversed fill
  LS_DIST0,                         // distribution[M3T_MAX_DISTRIBUTION_LENGTH]; LS_VALID .. are the rows a split object's workgroups exchange
  LS_FIELDS = LS_DIST0 + M3T_MAX_DISTRIBUTION_LENGTH
};
// ---- per-point state (depth modality), field-major -------------------------
enum {
  PS_CX = 0, PS_CY, PS_CZ, PS_NX, PS_NY, PS_NZ,
  PS_CENTER_U, PS_CENTER_V, PS_DEPTH,
  PS_CORR_X, PS_CORR_Y, PS_CORR_Z,
  PS_VALID,
  PS_FIELDS
};

struct CameraDev {
  const uint8_t* image;  // current frame (BGR8 or u16), pitch-linear
  uint32_t pitch;        // bytes per row
  int width, height;
  float fu, fv, ppu, ppv;
  float depth_scale;
  float world2camera[16];  // column-major
  int slot;                // which slot of the camera's frame ring `image` is (ROI ingest: whose rectangle applies)
};

// FocusedBasicDepthRenderer / FocusedSilhouetteRenderer (renderer.h:180-330) on the device
enum { RS_CORNER_U = 0, RS_CORNER_V, RS_SCALE, RS_TERM_A, RS_TERM_B, RS_N_VISIBLE, RS_VISIBLE0 = 8,
       RS_FLOATS = 8 + M3T_MAX_RENDERER_BODIES };
struct RendererDev {
  int camera, silhouette, id_type, image_size;
  float z_min, z_max;
  int n_bodies;                                 // RendererGeometry, draw order
  int body[M3T_MAX_RENDERER_BODIES];
  const float* vertices[M3T_MAX_RENDERER_BODIES];
  const int* triangles[M3T_MAX_RENDERER_BODIES];
  int n_triangles[M3T_MAX_RENDERER_BODIES];
  float geometry2body[M3T_MAX_RENDERER_BODIES][16];
  int culling[M3T_MAX_RENDERER_BODIES];
  int id[M3T_MAX_RENDERER_BODIES];              // value written by the silhouette pass
  int n_referenced;
  int referenced[M3T_MAX_RENDERER_BODIES];
  float referenced_diameter[M3T_MAX_RENDERER_BODIES];
  // results of the last rendering
  uint16_t* depth_image;        // [image_size^2], 65535 = nothing
  uint8_t* silhouette_image;    // [image_size^2] ids (0 = nothing)
  uint32_t* packed;             // z-buffer scratch when image_size^2 words do not fit in LDS
  float* state;                 // [RS_FLOATS]
  // triangles that survive set-up (culling, clipping against the crop), appended by focused_setup_kernel and
  // rasterised into an LDS z-buffer by focused_resolve_kernel: [survivor_capacity] x M3T_SURVIVOR_BYTES, and their count
  void* survivors;
  int* n_survivors;
  int survivor_capacity;
};
// closest_view_local: per view a row of M3T_VIEW_ROW float4: entry 0 the view itself, entries 1 .. M3T_VIEW_NEIGHBORS
// its nearest views ({x, y, z, id bits}), the last entry {threshold, 0, 0, 0}
#define M3T_VIEW_NEIGHBORS 18
#define M3T_VIEW_ROW (M3T_VIEW_NEIGHBORS + 2)
#define M3T_SURVIVOR_BYTES 112  /* RasterTriangle (m3t_raster.h: 11 doubles + 4 ints) + the low bits of its words */

// a ColorHistograms object shared by several RegionModalities (region_modality.cpp:168-173)
struct SharedHistogramsDev {
  int n_bins;
  float learning_rate_f, learning_rate_b;
  float* histogram_f;
  float* histogram_b;
  float2* histogram_norm;
  unsigned long long* counts;  // [n_bins^3] foreground count | background count << 32
};

struct RegionModDev {
  int body, camera, depth_camera;
  // sparse viewpoint model (shared between objects using the same model)
  const float* points;        // [n_views][n_points][38]  the .bin layout (depth_offsets are read from here)
  const float4* points8;      // [n_views][n_points][2]: {cx,cy,cz,nx},{ny,nz,foreground_distance,background_distance}
  const float4* orientations4;  // [n_views] xyz + pad
  const float4* view_neighbors; // [n_views][M3T_VIEW_ROW]: closest_view_local (m3t_kernels.hip), or nullptr
  int* last_view;             // the view the last correspondence search of this modality used (-1: none yet)
  const float* extents;       // contour_length [n_views]
  int n_views, n_points;
  float max_extent;
  // parameters (m3t_region_modality_params, precalculated)
  int n_lines_max, use_adaptive_coverage;
  float reference_contour_length, min_continuous_distance;
  int function_length, distribution_length, n_seg;
  float function_lookup_f[M3T_MAX_FUNCTION_LENGTH], function_lookup_b[M3T_MAX_FUNCTION_LENGTH];
  float learning_rate;
  int n_global_iterations;
  int n_scales, scales[M3T_MAX_SCALES];
  int n_standard_deviations;
  float standard_deviations[M3T_MAX_SCALES];
  int n_bins, bitshift;
  float learning_rate_f, learning_rate_b, unconsidered_line_length, max_considered_line_length;
  int measure_occlusions, measured_depth_offset_id;
  float measured_occlusion_radius, measured_occlusion_threshold;
  int n_unoccluded_iterations, min_n_unoccluded_lines;
  float min_expected_variance, distribution_length_minus_1_half, distribution_length_plus_1_half;
  int first_iteration;
  // renderer-fed branches (region_modality.cpp:1157-1229,1293-1341,1391-1431)
  int use_region_checking, model_occlusions, modeled_depth_offset_id, region_id;
  float modeled_occlusion_radius, modeled_occlusion_threshold;
  const RendererDev* depth_renderer;
  const RendererDev* silhouette_renderer;
  int depth_renderer_slot, silhouette_renderer_slot;  // index of the body among the referenced bodies
  // per-object state
  float* histogram_f;     // [n_bins^3]
  float* histogram_b;     // [n_bins^3]
  float2* histogram_norm; // [n_bins^3] (pf/(pf+pb), pb/(pf+pb)) or (0.5,0.5): MultiplyPixelColorProbability hoisted per bin
  uint8_t* occupancy;     // [n_bins^3 / 4] 1: the group of four bins holds a non-zero histogram value (the update only
                          // streams those groups and the ones that received samples; a trained 32^3 table is ~85 % empty)
  uint32_t* count_scratch; // [n_bins^3] packed counts, only when they do not fit in LDS (n_bins = 64)
  unsigned long long* shared_counts;  // count table of the shared ColorHistograms object or nullptr
  float* line_state;      // [LS_FIELDS][n_lines_max]
  float* gradient_hessian;  // [6 + 36] gradient, column-major hessian
};

struct DepthModDev {
  int body, camera;
  const float* points;        // [n_views][n_points][36]  the .bin layout (depth_offsets are read from here)
  const float4* points8;      // [n_views][n_points][2]: {cx,cy,cz,nx},{ny,nz,0,0}
  const float4* orientations4;
  const float* extents;       // surface_area
  int n_views, n_points;
  float max_extent;
  float stride_depth_offset;
  int n_points_max, use_adaptive_coverage, use_depth_scaling;
  float reference_surface_area, stride_length;
  int n_considered_distances;
  float considered_distances[M3T_MAX_SCALES];
  int n_standard_deviations;
  float standard_deviations[M3T_MAX_SCALES];
  int measure_occlusions;
  float measured_depth_offset_radius, measured_occlusion_radius, measured_occlusion_threshold;
  int n_unoccluded_iterations, min_n_unoccluded_points;
  int first_iteration;
  // renderer-fed branches (depth_modality.cpp:728-734,778-824)
  int use_silhouette_checking, model_occlusions, body_id;
  float modeled_depth_offset_radius, modeled_occlusion_radius, modeled_occlusion_threshold;
  const RendererDev* depth_renderer;
  const RendererDev* silhouette_renderer;
  int depth_renderer_slot, silhouette_renderer_slot;
  float* point_state;       // [PS_FIELDS][n_points_max]
  float* gradient_hessian;  // [6 + 36]
  // 1: the region modality on the same rigid optimizer uses a model with the same view orientations and a camera
  // with the same world2camera pose, so both GetClosestView searches give the same index (set by UploadTables)
  int view_search_shared;
};

// One rigid body with an unconstrained 6-dof root link (body2joint = I):
// Optimizer::CalculateOptimization specialised to dof = 6, J = I.
struct RigidOptDev {
  int body;
  int region_modality;  // index into RegionModDev table or -1
  int depth_modality;   // index into DepthModDev table or -1
  float tikhonov_rotation, tikhonov_translation;
  // ROI ingest (m3t_ingest.hip): where the fused kernels leave the poses a step read its frames at --
  // [n_corr_iterations + 2][16]: start of the step, every correspondence search, the final pose -- or null
  float* search_poses;
};

// ROI ingest: one reader of a camera's frames = one modality of one body (m3t_roi.h)
struct RoiItemDev {
  int camera, body;
  int opt;                       // the rigid optimizer whose search poses apply (-1: none recorded)
  float box_min[3], box_max[3];  // around the data points of the modality's model, body frame
  float reach_px, reach_m;       // the modality's reach (m3t_roi.h), without the caller's margin
  float rho;                     // the points also lie in the ellipsoid of rho x the box's half extents (0: not used)
};

// ROI guard (tracking_step_*_guard_kernel): appended to an object's search-pose block, at search_poses + 16 * n_poses.
// flag: the object's last guarded step needed pixels outside its rectangle and was NOT committed (pose, histograms and
// modality state are those before the step); the repeat launch (mode 2) runs the flagged objects only and clears it.
#define M3T_ROI_GUARD_ITEMS 3 /* readers per object: region (colour), region (depth: measured occlusions), depth */
struct RoiGuardDev {
  int flag;
  int n_items;
  int item[M3T_ROI_GUARD_ITEMS];  // rows of the RoiItemDev table
  int reserved[3];
};
struct m3t_roi_rect;
struct RoiGuardArgs {
  const RoiItemDev* items;
  const m3t_roi_rect* rects;  // [slot][camera id]
  int n_cams, n_rect_slots;
  int n_poses;                // floats of an object's search-pose block in front of its RoiGuardDev: 16 * n_poses
  int mode;                   // 1: every object, flag the ones that leave their rectangles; 2: the flagged ones only
  int* misses;                // mapped host memory: [0] count, [1 ..] body ids (mode 1)
  int miss_capacity;
  int* unrecovered;           // the same for objects that miss again in mode 2 (no whole frame could be fetched)
};

// LDS carve-up of the tracking kernels, computed on the host from the maxima
// over all objects of one launch.
struct TrackLdsLayout {
  int nl;        // max n_lines_max / n_points_max
  int ns;        // max n_seg
  int off_state;   // float offset: LS_FIELDS * nl
  int off_chain;   // nl * ns  (aliased by the raw distribution nl * M3T_MAX_DISTRIBUTION_LENGTH)
  int off_seg_f;   // nl * ns
  int off_seg_b;   // nl * ns
  int off_misc;    // reduction scratch
  int off_hist;    // optional LDS-staged histogram_norm (float2[n_bins^3]) or -1
  // product rows of the gradient / Hessian sums ([27][pitch] floats each, 16-byte aligned); they take over the
  // chain / segment buffers, which are dead between two correspondence searches
  int off_rows_r, pitch_r;   // region modality: one column per correspondence line
  int off_rows_d, pitch_d;   // depth modality: one column per correspondence point
  int total_floats;
};

// LDS carve-up of tracking_step_compact_kernel (m3t_compact.hip): what one object keeps between the phases of a
// step when several objects share a CU -- no chain / segment buffers (a thread walks its whole line and keeps a
// window of eight segments in registers), 8 + 1 factor rows per modality instead of 27 product rows.
enum {
  CS_CX = 0, CS_CY, CS_CZ, CS_CENTER_U, CS_CENTER_V, CS_NORMAL_U, CS_NORMAL_V, CS_DELTA_R, CS_NCTS, CS_MEAN, CS_VAR,
  CS_VALID,   // int bits: bit 0 = the line is in data_lines_
  CS_DIST0,   // distribution[12] (the raw products first, normalised in place)
  CS_FIELDS = CS_DIST0 + 12
};
#define M3T_COMPACT_THREADS 256
#define M3T_COMPACT_MISC_FLOATS 672
#define M3T_COMPACT_ROWS 9      /* six Jacobian / direction entries, two weights, one row of constants */
struct CompactLayout {
  int nl, np;                 // lines (<= M3T_COMPACT_THREADS), depth points
  int off_state;              // [CS_FIELDS][nl]
  int off_rows_r, pitch_r;    // [M3T_COMPACT_ROWS][pitch_r]
  int off_points;             // [PS_FIELDS][np]
  int off_rows_d, pitch_d;    // [M3T_COMPACT_ROWS][pitch_d]
  // histogram update in the tail of the launch (the carve-up above is dead by then): 1024 floats of scratch, then
  // either a count word per bin (tail_pass_bins == 0) or a sample list + the count words of one pass over the bins
  int tail_list_row;          // list entries per walker (>= the longest projected line), 0: no list
  int tail_pass_bins;         // bins per pass
  int off_tail_list, off_tail_counts;
  int total_floats;
  // tracking_step_compact_table_kernel (round 6): the region modality's pair table compacted in LDS behind the search's
  // carve-up -- [table_words] x {bits a, bits b} | [3 + table_cap] f32: the smaller float of a pair (the three constants
  // first) | [3 + table_cap] u8: which one it is and the larger one's distance from 1 - smaller | [table_words] u16: 3 + the
  // rank of the word's first mixed bin
  int off_table, table_words, table_cap;
  unsigned* table_overflow;  // mapped host word: the largest number of mixed bins a workgroup could not place (atomic max)
};

#ifndef M3T_BLOCK_THREADS
#define M3T_BLOCK_THREADS 512
#endif
#define M3T_MISC_FLOATS 1024
#define M3T_SPLIT_LANES 256   /* tracking_step_split_kernel: workgroups per object x padded elements per part */
#define M3T_SPLIT_MAX_PARTS 16

#endif  // M3T_DEVICE_H_
