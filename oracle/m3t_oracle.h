/* m3t_oracle.h — CPU restatement of M3T's per-frame pose-optimisation hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is the parity checker and the CPU
 * baseline ("port") for the HIP product in 3dobjecttracking_amd/.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (libm3t_hip.so) never links, loads or calls anything in oracle/.
 *
 * It is a from-scratch single-threaded scalar restatement (no Eigen, no
 * OpenCV) of the reference functions listed in SURVEY.md §8(a); every function
 * in m3t_oracle.cpp cites the reference file:line it follows.  The reference
 * itself cannot be compiled in this image (needs Eigen/OpenCV/GLEW/GLFW), so
 * there is no oracle/_ref binary.  Parity pinning status:
 *   pinned   : ColorHistograms (closed-form KAT of color_histograms_test.cpp),
 *              .bin model loader + GetClosestView (data/model_test goldens),
 *              RegionModality / DepthModality correspondences, gradients, Hessians
 *              (modality_test goldens, the reference's own 1e-3 criterion) and all
 *              their visualisation goldens pixel for pixel, incl. measured and modelled
 *              occlusions, region and silhouette checking,
 *              Optimizer/Link solve + pose update (optimizer_test golden pose),
 *              a whole tracking step (tracker_test golden pose, 1e-5 relative).
 *   residual : the refiner_test pose (a sequence that is chaotic on this fixture).
 * The models those goldens need are regenerated without OpenGL by
 * tests/golden/gl_model.py, itself checked against the .bin files of data/model_test.
 *
 * The function set mirrors include/m3t_hip.h one to one (prefix m3t_oracle_
 * instead of m3t_hip_) so the parity tests drive both through the same code.
 */
#ifndef M3T_ORACLE_H_
#define M3T_ORACLE_H_

#include "../include/m3t_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct m3t_oracle_context m3t_oracle_context;

int m3t_oracle_create(m3t_oracle_context** out, int device_id /* ignored */);
void m3t_oracle_destroy(m3t_oracle_context* ctx);
const char* m3t_oracle_last_error(m3t_oracle_context* ctx);

/* models (region_model.cpp:259-307, depth_model.cpp:215-283, model.cpp:218-284) */
int m3t_oracle_region_model_create(m3t_oracle_context*, const m3t_region_model_desc*);
int m3t_oracle_region_model_load(m3t_oracle_context*, const char* path);
int m3t_oracle_depth_model_create(m3t_oracle_context*, const m3t_depth_model_desc*);
int m3t_oracle_depth_model_load(m3t_oracle_context*, const char* path);
int m3t_oracle_region_model_info(m3t_oracle_context*, int model_id, int* n_views, int* n_points,
                                 float* max_contour_length);
int m3t_oracle_depth_model_info(m3t_oracle_context*, int model_id, int* n_views, int* n_points,
                                float* max_surface_area);
/* GetClosestView (region_model.cpp:105-130 / depth_model.cpp:81-106) */
int m3t_oracle_region_model_closest_view(m3t_oracle_context*, int model_id,
                                         const float body2camera[16], int* view_index);
int m3t_oracle_depth_model_closest_view(m3t_oracle_context*, int model_id,
                                        const float body2camera[16], int* view_index);

/* cameras (camera.h:32-88): image = BGR8 (color) or u16 (depth), row_step in bytes */
int m3t_oracle_color_camera_create(m3t_oracle_context*, const m3t_intrinsics*, const float world2camera[16]);
int m3t_oracle_depth_camera_create(m3t_oracle_context*, const m3t_intrinsics*, const float world2camera[16],
                                   float depth_scale);
int m3t_oracle_camera_upload(m3t_oracle_context*, int camera_id, const void* pixels, size_t row_step);
int m3t_oracle_camera_set_world2camera_pose(m3t_oracle_context*, int camera_id, const float world2camera[16]);

/* bodies (body.h: body2world_pose) */
int m3t_oracle_body_create(m3t_oracle_context*, const float body2world[16]);
int m3t_oracle_body_set_body2world_pose(m3t_oracle_context*, int body_id, const float body2world[16]);
int m3t_oracle_body_get_body2world_pose(m3t_oracle_context*, int body_id, float body2world[16]);

/* modalities (modality.h:56-155) */
int m3t_oracle_region_modality_create(m3t_oracle_context*, const m3t_region_modality_params*, int body_id,
                                      int color_camera_id, int region_model_id, int depth_camera_id);
int m3t_oracle_depth_modality_create(m3t_oracle_context*, const m3t_depth_modality_params*, int body_id,
                                     int depth_camera_id, int depth_model_id);

/* ---- renderer-fed branches: body meshes, focused depth / silhouette renderings ----
 * A software restatement of FocusedBasicDepthRenderer / FocusedSilhouetteRenderer (renderer.cpp:348-405,
 * basic_depth_renderer.cpp:45-84, silhouette_renderer.cpp:54-100): the square image_size x image_size crop
 * around the referenced bodies, rasterised with OpenGL's rules (pixel centres, 1/256 sub-pixel snapping,
 * top-left fill rule, 16-bit depth, GL_LESS in draw order).  Renderers a modality references are rendered
 * where the reference's Tracker does it: at start_modalities / calculate_results for region modalities and
 * before every calculate_correspondences (tracker.cpp:430-517). */
int m3t_oracle_body_set_geometry(m3t_oracle_context*, int body_id, const m3t_body_geometry*);
int m3t_oracle_renderer_geometry_create(m3t_oracle_context*);                                   /* RendererGeometry */
int m3t_oracle_renderer_geometry_add_body(m3t_oracle_context*, int geometry_id, int body_id);  /* draw order = add order */
int m3t_oracle_focused_depth_renderer_create(m3t_oracle_context*, int geometry_id, int camera_id, int image_size, float z_min,
                                          float z_max);                      /* basic_depth_renderer.h:120-128 */
int m3t_oracle_focused_silhouette_renderer_create(m3t_oracle_context*, int geometry_id, int camera_id, int id_type, int image_size,
                                               float z_min, float z_max);    /* silhouette_renderer.h:150-155 */
int m3t_oracle_renderer_add_referenced_body(m3t_oracle_context*, int renderer_id, int body_id);
int m3t_oracle_renderer_start_rendering(m3t_oracle_context*, int renderer_id);
/* depth: image_size^2 u16 (65535 = nothing); silhouette: image_size^2 u8 ids (NULL for depth renderers);
 * info: corner_u, corner_v, scale; n_visible: referenced bodies that passed FocusedRenderer's visibility test */
int m3t_oracle_renderer_get_images(m3t_oracle_context*, int renderer_id, uint16_t* depth, uint8_t* silhouette, float info[3],
                                int* n_visible);
/* RegionModality::ModelOcclusions / UseRegionChecking, DepthModality::ModelOcclusions / UseSilhouetteChecking */
int m3t_oracle_region_modality_model_occlusions(m3t_oracle_context*, int modality_id, int depth_renderer_id);
int m3t_oracle_region_modality_use_region_checking(m3t_oracle_context*, int modality_id, int silhouette_renderer_id);
int m3t_oracle_depth_modality_model_occlusions(m3t_oracle_context*, int modality_id, int depth_renderer_id);
int m3t_oracle_depth_modality_use_silhouette_checking(m3t_oracle_context*, int modality_id, int silhouette_renderer_id);

/* ColorHistograms shared by several RegionModalities (color_histograms.h:36-40, RegionModality::UseSharedColorHistograms
 * region_modality.cpp:168-173; cleared / initialised / updated once per step around all modalities, tracker.cpp:435-443,
 * 507-515).  The modality's own n_histogram_bins and learning rates are ignored once it shares. */
int m3t_oracle_color_histograms_create(m3t_oracle_context*, int n_bins, float learning_rate_f, float learning_rate_b);
int m3t_oracle_region_modality_use_shared_color_histograms(m3t_oracle_context*, int modality_id, int histograms_id);

/* links / optimizers (link.h:67, optimizer.h:48) */
int m3t_oracle_link_create(m3t_oracle_context*, int body_id, int parent_link_id, const float body2joint[16],
                           const float joint2parent[16], const int free_directions[6],
                           int fixed_body2joint_pose);
int m3t_oracle_link_add_modality(m3t_oracle_context*, int link_id, int modality_id);
int m3t_oracle_optimizer_create(m3t_oracle_context*, int root_link_id, float tikhonov_parameter_rotation,
                                float tikhonov_parameter_translation);
/* convenience: one free 6-dof root link holding `n` modalities of one body */
int m3t_oracle_optimizer_create_rigid(m3t_oracle_context*, int body_id, int n_modalities,
                                      const int* modality_ids, float tikhonov_parameter_rotation,
                                      float tikhonov_parameter_translation);
/* hard constraint between two links (constraint.h) */
int m3t_oracle_constraint_create(m3t_oracle_context*, int optimizer_id, int link1_id, int link2_id,
                                 const float body12joint1[16], const float body22joint2[16],
                                 const int constraint_directions[6]);
/* soft constraint between two links (soft_constraint.h; added to the links' g/H, soft_constraint.cpp:113-131) */
int m3t_oracle_soft_constraint_create(m3t_oracle_context*, int optimizer_id, int link1_id, int link2_id,
                                      const float body12joint1[16], const float body22joint2[16],
                                      const int constraint_directions[6], float max_distance_rotation,
                                      float max_distance_translation, float standard_deviation_rotation,
                                      float standard_deviation_translation);
int m3t_oracle_link_get_link2world_pose(m3t_oracle_context*, int link_id, float pose[16]);
/* Link::set_link2world_pose (link.cpp:138-140; what Detector::UpdatePoses writes, detector.cpp:42-53):
 * the body's pose for a link with a body, the link's own frame for a body-less root */
int m3t_oracle_link_set_link2world_pose(m3t_oracle_context*, int link_id, const float pose[16]);
int m3t_oracle_link_set_joint_poses(m3t_oracle_context*, int link_id, const float body2joint[16],
                                    const float joint2parent[16]); /* either may be NULL */
int m3t_oracle_link_get_joint_poses(m3t_oracle_context*, int link_id, float body2joint[16], float joint2parent[16]);
int m3t_oracle_calculate_consistent_poses(m3t_oracle_context*); /* tracker.cpp:423, optimizer.cpp:135 */

/* tracker (tracker.h:131-160, tracker.cpp:344-517) */
int m3t_oracle_tracker_set_iterations(m3t_oracle_context*, int n_corr_iterations, int n_update_iterations);
int m3t_oracle_start_modalities(m3t_oracle_context*, int iteration);
int m3t_oracle_calculate_correspondences(m3t_oracle_context*, int iteration, int corr_iteration);
int m3t_oracle_calculate_gradient_and_hessian(m3t_oracle_context*, int iteration, int corr_iteration,
                                              int opt_iteration);
int m3t_oracle_calculate_optimization(m3t_oracle_context*, int iteration, int corr_iteration,
                                      int opt_iteration);
/* the same optimisation split where a kinematic structure spread over GPUs all-reduces its
 * stacked [dof*dof | dof] sums: begin -> (sum `partial` over ranks) -> end */
int m3t_oracle_calculate_optimization_begin(m3t_oracle_context*, float** partial, size_t* count);
int m3t_oracle_calculate_optimization_end(m3t_oracle_context*);
int m3t_oracle_calculate_results(m3t_oracle_context*, int iteration);
int m3t_oracle_execute_tracking_step(m3t_oracle_context*, int iteration);
/* oracle only (no m3t_hip twin): the same step with an OpenMP parallel-for over independent rigid objects, for the
 * CPU baseline at nproc threads; seconds[4] accumulates the evaluators' four time buckets summed over threads */
int m3t_oracle_execute_tracking_step_parallel(m3t_oracle_context*, int iteration, int n_threads, double* seconds);
int m3t_oracle_execute_tracking_cycle(m3t_oracle_context*, int iteration); /* ICG name */
/* Refiner::RefinePoses (refiner.cpp:76-117): CalculateConsistentPoses, then n_corr_iterations x
 * (StartModalities + CalculateCorrespondences + n_update_iterations x (g/H + optimisation)), iteration index 0 */
int m3t_oracle_refine_poses(m3t_oracle_context*, int n_corr_iterations, int n_update_iterations);
int m3t_oracle_sync(m3t_oracle_context*);

/* accessors */
int m3t_oracle_modality_get_gradient_hessian(m3t_oracle_context*, int modality_id, float gradient[6],
                                             float hessian[36]);
int m3t_oracle_modalities_get_gradient_hessian(m3t_oracle_context*, float* out, int capacity_modalities);
int m3t_oracle_modality_set_gradient_hessian(m3t_oracle_context*, int modality_id, const float gradient[6],
                                             const float hessian[36]);
int m3t_oracle_region_modality_get_lines(m3t_oracle_context*, int modality_id, m3t_data_line* out,
                                         int capacity, int* n_lines);
int m3t_oracle_depth_modality_get_points(m3t_oracle_context*, int modality_id, m3t_data_point* out,
                                         int capacity, int* n_points);
int m3t_oracle_region_modality_get_histograms(m3t_oracle_context*, int modality_id, float* histogram_f,
                                              float* histogram_b);
int m3t_oracle_region_modality_set_histograms(m3t_oracle_context*, int modality_id, const float* histogram_f,
                                              const float* histogram_b);

/* stand-alone ColorHistograms object for the closed-form KAT
 * (color_histograms.cpp:50-102,174-214; color_histograms_test.cpp:71-103) */
typedef struct m3t_oracle_histograms m3t_oracle_histograms;
m3t_oracle_histograms* m3t_oracle_histograms_create(int n_bins, float learning_rate_f, float learning_rate_b);
void m3t_oracle_histograms_destroy(m3t_oracle_histograms*);
void m3t_oracle_histograms_clear_memory(m3t_oracle_histograms*);
void m3t_oracle_histograms_add_foreground(m3t_oracle_histograms*, const uint8_t bgr[3]);
void m3t_oracle_histograms_add_background(m3t_oracle_histograms*, const uint8_t bgr[3]);
void m3t_oracle_histograms_initialize(m3t_oracle_histograms*);
void m3t_oracle_histograms_update(m3t_oracle_histograms*);
void m3t_oracle_histograms_get_probabilities(m3t_oracle_histograms*, const uint8_t bgr[3], float* pf, float* pb);

#ifdef __cplusplus
}
#endif
#endif /* M3T_ORACLE_H_ */
