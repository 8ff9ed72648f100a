/* m3t_types.h — plain-old-data contract shared by the C-ABI (include/m3t_hip.h)
 * and the CPU oracle (oracle/m3t_oracle.h).
 *
 * Every struct mirrors member names of the reference class it parameterises
 * (trailing underscore dropped), so a maintainer can fill it 1:1 from an
 * m3t::RegionModality / m3t::DepthModality / m3t::Optimizer instance:
 *   m3t_region_modality_params  <- M3T/include/m3t/region_modality.h:411-443
 *   m3t_depth_modality_params   <- M3T/include/m3t/depth_modality.h:302-321
 *   m3t_intrinsics              <- M3T/include/m3t/common.h (struct Intrinsics)
 *   m3t_region_model_desc       <- M3T/include/m3t/region_model.h:89-110 (+ model.h)
 *   m3t_depth_model_desc        <- M3T/include/m3t/depth_model.h:67-86
 * All poses are 4x4 float, COLUMN-major (Eigen::Transform<float,3,Affine>::data()).
 * All images are borrowed host pointers, valid only for the duration of a call.
 */
#ifndef M3T_TYPES_H_
#define M3T_TYPES_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M3T_MAX_SCALES 8            /* entries kept of scales_/standard_deviations_ */
#define M3T_MAX_FUNCTION_LENGTH 16  /* function_length_ upper bound */
#define M3T_MAX_DISTRIBUTION_LENGTH 16
#define M3T_MAX_SEGMENTS (M3T_MAX_FUNCTION_LENGTH + M3T_MAX_DISTRIBUTION_LENGTH - 1)
#define M3T_N_DEPTH_OFFSETS 30      /* Model::kMaxNDepthOffsets, model.h */
#define M3T_REGION_POINT_FLOATS 38  /* RegionModel::DataPoint = 152 B */
#define M3T_DEPTH_POINT_FLOATS 36   /* DepthModel::DataPoint  = 144 B */
#define M3T_MAX_N_OCCLUSION_STRIDES 5 /* kMaxNOcclusionStrides, region_modality.h:145 */
#define M3T_N_REGION_STRIDE 5          /* kNRegionStride, region_modality.h:146 */
#define M3T_REGION_OFFSET 2.0f         /* kRegionOffset, region_modality.h:147 */

/* status codes (the reference returns bool + std::cerr; 0 == true) */
enum {
  M3T_OK = 0,
  M3T_ERR_INVALID_ARGUMENT = -1,
  M3T_ERR_NOT_SET_UP = -2,     /* reference: "Set up ... first" -> false */
  M3T_ERR_UNSUPPORTED = -3,    /* a limit of this implementation (e.g. more than 8 bodies in a renderer) */
  M3T_ERR_IO = -4,
  M3T_ERR_DEVICE = -5,         /* HIP / RCCL runtime error */
  M3T_ERR_NO_MEMORY = -6
};

typedef struct m3t_intrinsics {
  float fu, fv, ppu, ppv;
  int width, height;
} m3t_intrinsics;

/* RegionModel: n_views views, each n_points DataPoints of 38 floats
 * {center_f_body[3], normal_f_body[3], foreground_distance, background_distance,
 *  depth_offsets[30]} followed (separately here) by orientation[3] and
 * contour_length.  Same bytes as the .bin file. */
typedef struct m3t_region_model_desc {
  int n_views;
  int n_points;
  const float* data_points;      /* [n_views][n_points][38] */
  const float* orientations;     /* [n_views][3] */
  const float* contour_lengths;  /* [n_views] */
  float stride_depth_offset;
  float max_radius_depth_offset;
} m3t_region_model_desc;

/* DepthModel: DataPoint = {center_f_body[3], normal_f_body[3], depth_offsets[30]} */
typedef struct m3t_depth_model_desc {
  int n_views;
  int n_points;
  const float* data_points;    /* [n_views][n_points][36] */
  const float* orientations;   /* [n_views][3] */
  const float* surface_areas;  /* [n_views] */
  float stride_depth_offset;
  float max_radius_depth_offset;
} m3t_depth_model_desc;

typedef struct m3t_region_modality_params {
  /* general distribution */
  int n_lines_max;
  int use_adaptive_coverage;
  float reference_contour_length;
  float min_continuous_distance;
  int function_length;
  int distribution_length;
  float function_amplitude;
  float function_slope;
  float learning_rate;
  int n_global_iterations;
  int n_scales;
  int scales[M3T_MAX_SCALES];
  int n_standard_deviations;
  float standard_deviations[M3T_MAX_SCALES];
  /* histogram calculation */
  int n_histogram_bins;
  float learning_rate_f;
  float learning_rate_b;
  float unconsidered_line_length;
  float max_considered_line_length;
  /* occlusion handling and line validation */
  int use_region_checking;  /* 0 at creation; switched on by *_region_modality_use_region_checking() */
  int measure_occlusions;
  float measured_depth_offset_radius;
  float measured_occlusion_radius;
  float measured_occlusion_threshold;
  int model_occlusions;     /* 0 at creation; switched on by *_region_modality_model_occlusions() */
  int n_unoccluded_iterations;
  int min_n_unoccluded_lines;
  float modeled_depth_offset_radius;
  float modeled_occlusion_radius;
  float modeled_occlusion_threshold;
} m3t_region_modality_params;

typedef struct m3t_depth_modality_params {
  int n_points_max;
  int use_adaptive_coverage;
  int use_depth_scaling;
  float reference_surface_area;
  float stride_length;
  int n_considered_distances;
  float considered_distances[M3T_MAX_SCALES];
  int n_standard_deviations;
  float standard_deviations[M3T_MAX_SCALES];
  int use_silhouette_checking; /* 0 at creation; switched on by *_depth_modality_use_silhouette_checking() */
  int measure_occlusions;
  float measured_depth_offset_radius;
  float measured_occlusion_radius;
  float measured_occlusion_threshold;
  int model_occlusions;        /* 0 at creation; switched on by *_depth_modality_model_occlusions() */
  int n_unoccluded_iterations;
  int min_n_unoccluded_points;
  float modeled_depth_offset_radius;
  float modeled_occlusion_radius;
  float modeled_occlusion_threshold;
} m3t_depth_modality_params;

/* Body geometry for the renderer-fed branches (body.h:46-60, body.cpp:196-250): a triangle mesh in
 * metres (geometry_unit_in_meter already applied), its pose in the body frame and the two ids the
 * silhouette renderer writes (body.h: body_id / region_id). */
typedef struct m3t_body_geometry {
  const float* vertices; /* [n_vertices][3] */
  int n_vertices;
  const int* triangles;  /* [n_triangles][3] vertex indices */
  int n_triangles;
  float geometry2body[16];
  int geometry_counterclockwise;
  int geometry_enable_culling;
  int body_id;   /* 0..255 */
  int region_id; /* 0..255 */
} m3t_body_geometry;
/* RegionModel / DepthModel generation parameters (region_model.h:142-148, depth_model.h:88-94) */
typedef struct m3t_model_generation_params {
  float sphere_radius;
  int n_divides;
  int n_points;
  float max_radius_depth_offset;
  float stride_depth_offset;
  int image_size;
} m3t_model_generation_params;
#define M3T_ID_TYPE_BODY 0   /* IDType::BODY */
#define M3T_ID_TYPE_REGION 1 /* IDType::REGION */
#define M3T_MAX_RENDERER_BODIES 8

/* RegionModality::DataLine, the fields CalculateGradientAndHessian reads
 * (region_modality.h:108-124).  Returned by *_region_modality_get_lines for
 * parity checks; `valid` == the line was pushed into data_lines_. */
typedef struct m3t_data_line {
  float center_f_body[3];
  float center_u, center_v;
  float normal_u, normal_v;
  float delta_r;
  float normal_component_to_scale;
  float continuous_distance;
  float mean;
  float measured_variance;
  float distribution[M3T_MAX_DISTRIBUTION_LENGTH];
  int valid;
  int model_point_index;
} m3t_data_line;

/* DepthModality::DataPoint (depth_modality.h:96-108) */
typedef struct m3t_data_point {
  float center_f_body[3];
  float normal_f_body[3];
  float center_u, center_v;
  float depth;
  float correspondence_center_f_camera[3];
  int valid;
  int model_point_index;
} m3t_data_point;

/* fills the reference's header defaults */
static inline void m3t_region_modality_params_default(m3t_region_modality_params* p) {
  static const int s[4] = {6, 4, 2, 1};
  static const float d[4] = {15.0f, 5.0f, 3.5f, 1.5f};
  int i;
  p->n_lines_max = 200;
  p->use_adaptive_coverage = 0;
  p->reference_contour_length = 0.0f;
  p->min_continuous_distance = 3.0f;
  p->function_length = 8;
  p->distribution_length = 12;
  p->function_amplitude = 0.43f;
  p->function_slope = 0.5f;
  p->learning_rate = 1.3f;
  p->n_global_iterations = 1;
  p->n_scales = 4;
  p->n_standard_deviations = 4;
  for (i = 0; i < M3T_MAX_SCALES; ++i) {
    p->scales[i] = i < 4 ? s[i] : 0;
    p->standard_deviations[i] = i < 4 ? d[i] : 0.0f;
  }
  p->n_histogram_bins = 16;
  p->learning_rate_f = 0.2f;
  p->learning_rate_b = 0.2f;
  p->unconsidered_line_length = 0.5f;
  p->max_considered_line_length = 20.0f;
  p->use_region_checking = 0;
  p->measure_occlusions = 0;
  p->measured_depth_offset_radius = 0.01f;
  p->measured_occlusion_radius = 0.01f;
  p->measured_occlusion_threshold = 0.03f;
  p->model_occlusions = 0;
  p->n_unoccluded_iterations = 10;
  p->min_n_unoccluded_lines = 0;
  p->modeled_depth_offset_radius = 0.01f;
  p->modeled_occlusion_radius = 0.01f;
  p->modeled_occlusion_threshold = 0.03f;
}

static inline void m3t_model_generation_params_default(m3t_model_generation_params* p) {
  p->sphere_radius = 0.8f;
  p->n_divides = 4;
  p->n_points = 200;
  p->max_radius_depth_offset = 0.05f;
  p->stride_depth_offset = 0.002f;
  p->image_size = 2000;
}

static inline void m3t_depth_modality_params_default(m3t_depth_modality_params* p) {
  static const float c[3] = {0.05f, 0.02f, 0.01f};
  static const float d[3] = {0.05f, 0.03f, 0.02f};
  int i;
  p->n_points_max = 200;
  p->use_adaptive_coverage = 0;
  p->use_depth_scaling = 0;
  p->reference_surface_area = 0.0f;
  p->stride_length = 0.005f;
  p->n_considered_distances = 3;
  p->n_standard_deviations = 3;
  for (i = 0; i < M3T_MAX_SCALES; ++i) {
    p->considered_distances[i] = i < 3 ? c[i] : 0.0f;
    p->standard_deviations[i] = i < 3 ? d[i] : 0.0f;
  }
  p->use_silhouette_checking = 0;
  p->measure_occlusions = 0;
  p->measured_depth_offset_radius = 0.01f;
  p->measured_occlusion_radius = 0.01f;
  p->measured_occlusion_threshold = 0.03f;
  p->model_occlusions = 0;
  p->n_unoccluded_iterations = 10;
  p->min_n_unoccluded_points = 0;
  p->modeled_depth_offset_radius = 0.01f;
  p->modeled_occlusion_radius = 0.01f;
  p->modeled_occlusion_threshold = 0.03f;
}

#ifdef __cplusplus
}
#endif
#endif /* M3T_TYPES_H_ */
