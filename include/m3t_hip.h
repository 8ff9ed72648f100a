/* m3t_hip.h — C-ABI of libm3t_hip.so: the MI355X (gfx950) implementation of M3T's
 * per-frame pose-optimisation hot path.
 *
 * This is the drop-in boundary.  The reference keeps
 * m3t::Tracker / m3t::Optimizer / m3t::Link concrete and m3t::Modality abstract
 * (M3T/include/m3t/modality.h:56-155); a C++ host binds these entry points from
 * Modality subclasses or directly from its tracking loop (INTEGRATION.md shows
 * the adapter).  Plain C: opaque context, int handles, borrowed host pointers
 * that only have to stay valid for the duration of a call, no exceptions, no
 * torch types.
 *
 * Conventions
 *  - every function returns M3T_OK (0) / a non-negative handle on success and a
 *    negative M3T_ERR_* code on failure; m3t_hip_last_error() gives the message
 *    (reference: `bool` + std::cerr, e.g. region_modality.cpp:1813-1819).
 *  - poses: float[16], column-major 4x4 (Eigen::Transform<float,3,Affine>::data()).
 *  - a context is bound to one GPU and one host thread (the reference holds
 *    tracking_mutex_ for a whole step, tracker.cpp:251-255); calls are
 *    asynchronous on the context's HIP stream unless they return data.
 *  - batching: one call of a tracking sub-step processes ALL modalities /
 *    optimizers registered in the context in one launch, the way
 *    Tracker::CalculateCorrespondences loops over tracking_modality_ptrs_
 *    (tracker.cpp:452-455), only in parallel.
 *  - there is no CPU fallback: without a usable gfx950 device m3t_hip_create
 *    fails with M3T_ERR_DEVICE.
 */
#ifndef M3T_HIP_H_
#define M3T_HIP_H_

#include "m3t_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The entry points below are the ONLY dynamic symbols of libm3t_hip.so: the library is built with -fvisibility=hidden
 * and everything declared between this push and the pop at the end of the header has default visibility (the
 * library's own C++ -- std:: instantiations, kernel launch stubs -- stays private, so that two libraries carrying the
 * same inline C++ in one process cannot interpose each other; tests/test_abi_symbols.py checks `nm -D`). */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#define M3T_HIP_VISIBILITY_PUSHED 1
#endif

typedef struct m3t_hip_context m3t_hip_context;

/* ---- context ------------------------------------------------------------- */
int m3t_hip_create(m3t_hip_context** out, int device_id);
void m3t_hip_destroy(m3t_hip_context* ctx);
const char* m3t_hip_last_error(m3t_hip_context* ctx); /* ctx may be NULL: last create() error */
/* device name / CU count / total HBM bytes of the bound GPU */
int m3t_hip_device_info(m3t_hip_context*, char* name, size_t name_capacity, int* compute_units,
                        size_t* total_memory_bytes);
/* the hipStream_t all work of this context is enqueued on (for events / interop) */
int m3t_hip_get_stream(m3t_hip_context*, void** hip_stream);

/* ---- Sparse Viewpoint Models ---------------------------------------------
 * replaces RegionModel::LoadModel (region_model.cpp:259-307) / DepthModel::LoadModel
 * (depth_model.cpp:215-283): parses the reference's .bin (model.cpp:218-284) or takes
 * the view table from memory, and keeps it resident in HBM.  Returns a model id. */
int m3t_hip_region_model_create(m3t_hip_context*, const m3t_region_model_desc*);
int m3t_hip_region_model_load(m3t_hip_context*, const char* path);
int m3t_hip_depth_model_create(m3t_hip_context*, const m3t_depth_model_desc*);
int m3t_hip_depth_model_load(m3t_hip_context*, const char* path);
int m3t_hip_region_model_info(m3t_hip_context*, int model_id, int* n_views, int* n_points,
                              float* max_contour_length);
int m3t_hip_depth_model_info(m3t_hip_context*, int model_id, int* n_views, int* n_points,
                             float* max_surface_area);
/* Model generation without OpenGL (RegionModel::GenerateModel region_model.cpp:187-258,
 * DepthModel::GenerateModel depth_model.cpp:144-212) for one body whose mesh was given with
 * m3t_hip_body_set_geometry: the template views are rasterised on the device, sampled on the host
 * (use_random_seed = false: std::mt19937{7}).  Returns the new model id.  *_get_views copies a model out
 * ([n_views][n_points][38 | 36] points, [n_views][3] orientations, [n_views] contour lengths / surface areas). */
int m3t_hip_region_model_generate(m3t_hip_context*, int body_id, const m3t_model_generation_params*);
int m3t_hip_depth_model_generate(m3t_hip_context*, int body_id, const m3t_model_generation_params*);
/* ... with associated bodies (RegionModel::AddAssociatedBody region_model.cpp:365-388: bodies that are drawn with the
 * main body when its views are rendered -- fixed or movable relative to it, of the same colour region or not;
 * region_model.cpp:417-463, 598-640, 695-770) and with occlusion bodies (DepthModel::AddOcclusionBody
 * depth_model.cpp:61-70, 165-177).  The bodies share the main body's frame; all need m3t_hip_body_set_geometry. */
int m3t_hip_region_model_generate_associated(m3t_hip_context*, int body_id, const m3t_model_generation_params*,
                                             int n_associated, const int* body_ids, const int* movable,
                                             const int* same_region);
int m3t_hip_depth_model_generate_occluded(m3t_hip_context*, int body_id, const m3t_model_generation_params*,
                                          int n_occlusion_bodies, const int* body_ids);
int m3t_hip_region_model_get_views(m3t_hip_context*, int model_id, float* points, float* orientations, float* extents);
int m3t_hip_depth_model_get_views(m3t_hip_context*, int model_id, float* points, float* orientations, float* extents);
/* RegionModel::GetClosestView (region_model.cpp:105-130) / DepthModel (depth_model.cpp:81-106),
 * evaluated on the device; returns the view index */
int m3t_hip_region_model_closest_view(m3t_hip_context*, int model_id, const float body2camera[16],
                                      int* view_index);
int m3t_hip_depth_model_closest_view(m3t_hip_context*, int model_id, const float body2camera[16],
                                     int* view_index);

/* ---- Cameras (camera.h:32-88) ----------------------------------------------
 * camera_upload == Camera::UpdateImage: copies the host frame (BGR8, as cv::imread
 * delivers it, loader_camera.cpp:76-98; or u16 depth) into HBM. */
int m3t_hip_color_camera_create(m3t_hip_context*, const m3t_intrinsics*, const float world2camera[16]);
int m3t_hip_depth_camera_create(m3t_hip_context*, const m3t_intrinsics*, const float world2camera[16],
                                float depth_scale);
int m3t_hip_camera_upload(m3t_hip_context*, int camera_id, const void* pixels, size_t row_step);
int m3t_hip_camera_set_world2camera_pose(m3t_hip_context*, int camera_id, const float world2camera[16]);
/* device-side frame ring: pre-stage n_slots frames, then switch the current frame
 * without a host copy (double-buffered ingest; slot 0 is what camera_upload writes) */
int m3t_hip_camera_set_ring(m3t_hip_context*, int camera_id, int n_slots);
int m3t_hip_camera_upload_slot(m3t_hip_context*, int camera_id, int slot, const void* pixels, size_t row_step);
int m3t_hip_camera_select_slot(m3t_hip_context*, int camera_id, int slot);
int m3t_hip_cameras_select_slot(m3t_hip_context*, int slot); /* all cameras that have a ring (one without keeps its frame) */
/* asynchronous ingest (replaces the blocking cv::Mat hand-over of Camera::UpdateImage,
 * camera.h:32-88, for callers that keep their frames in page-locked memory).  upload_slot_async enqueues
 * the copy on the context's copy stream and returns; `pixels` must stay valid until ingest_sync() or
 * the next sync() after a step that used the slot.  Copies already enqueued are visible to every step
 * launched after them; a slot is overwritten only after the last execute_tracking_step that read it.
 * The natural double-buffer loop is: select_slot(k%2); execute_tracking_step(k); upload_slot_async((k+1)%2).
 * Buffers that are not page-locked still work but the copy is then staged synchronously by the runtime;
 * host_register page-locks a caller-owned buffer once (e.g. the capture ring). */
int m3t_hip_host_register(m3t_hip_context*, void* ptr, size_t bytes);
int m3t_hip_host_unregister(m3t_hip_context*, void* ptr);
int m3t_hip_camera_upload_slot_async(m3t_hip_context*, int camera_id, int slot, const void* pixels, size_t row_step);
/* Batch ingest: a group of cameras of equal geometry fed by one capture process.  cameras_set_ring puts their frame
 * rings into one allocation ([slot][camera][frame]); cameras_upload_batch_async then moves the n frames of one
 * batch-frame (frame i at base + i * camera_stride, rows row_step apart) in ONE asynchronous transfer when the ids are
 * listed in the ring's order and the host block has the ring's layout (row_step = width * bytes per pixel rounded up
 * to 64, camera_stride = height * row_step); otherwise it falls back to one transfer per camera.  Same lifetime and
 * ordering rules as camera_upload_slot_async. */
int m3t_hip_cameras_set_ring(m3t_hip_context*, const int* camera_ids, int n_cameras, int n_slots);
int m3t_hip_cameras_upload_batch_async(m3t_hip_context*, const int* camera_ids, int n_cameras, int slot, const void* base,
                                       size_t camera_stride, size_t row_step);
int m3t_hip_ingest_sync(m3t_hip_context*); /* wait until all enqueued frame copies have landed */
/* wait until the last upload into (camera, slot) -- camera_upload_slot_async, a batch upload this camera was part of,
 * a rectangle pull -- has left its host buffer, and for nothing else: a loader that recycles one camera's page-locked
 * buffers does not hold up the copies of the other cameras.  A slot that holds a rectangle also waits for the step
 * that read it: should a body have outrun its rectangle, that step's repair reads the host block once more. */
int m3t_hip_camera_slot_sync(m3t_hip_context*, int camera_id, int slot);
/* ROI ingest: only the part of a frame the trackers can read crosses PCIe.  set_roi_ingest(enable, margin_px) switches
 * it on for the fused step of rigid objects; cameras_upload_batch_roi_async is cameras_upload_batch_async for
 * rectangles: one kernel on the copy stream computes every camera's rectangle -- the projected box around the model
 * points of the bodies tracked through it, widened by the modalities' reach (region_modality.cpp:1433-1508, 1640-1780,
 * 1343-1389; depth_modality.cpp:736-776, 826-884) and by a margin for the motion until the frame has been used (two
 * steps: the rectangle comes from the poses the step enqueued last starts from) -- and pulls its rows out of the host
 * block, which has to be page-locked and mapped (host_register).  enable = 1: margin_px for every body; enable = 2:
 * per body from the motion of its rectangle -- max(twice the last step, 1.25 x the largest step of the recent past) + 3
 * pixels, at least 4, at most margin_px: fewer bytes where bodies move little, and nothing lost where one does not
 * (next paragraph).  Where
 * rectangles are not possible (ROI ingest off, no fused step yet, cameras not in one ring in this order, strides that are
 * not multiples of 16 bytes) the whole frames are uploaded.
 * A step that reads rectangles checks on the device, at every pose its searches and histogram lines run at, that what
 * it can touch lies inside them.  A body whose step does not is NOT committed (pose, histograms, modality state stay
 * those before the step); the library then fetches the whole frames of that body's cameras from the same host block
 * -- which therefore has to stay what it is until the step is done, the lifetime rule of every asynchronous upload --
 * and repeats the step for those bodies, inside the same execute_tracking_step call.  Results equal the whole-frame
 * results bit for bit either way.  roi_get_status returns (and clears) the bodies that were repeated since the last
 * call -- a count of how often the margin was too small; roi_get_unrecovered the ones whose repeat could not see a
 * whole frame either (the host block of the slot was not known any more, e.g. tables rebuilt in between): their step
 * is still uncommitted -- upload the frame in full and execute the step again. */
int m3t_hip_set_roi_ingest(m3t_hip_context*, int enable, float margin_px);
int m3t_hip_cameras_upload_batch_roi_async(m3t_hip_context*, const int* camera_ids, int n_cameras, int slot, const void* base,
                                           size_t camera_stride, size_t row_step);
int m3t_hip_roi_get_status(m3t_hip_context*, int* body_ids, int capacity, int* n_misses,
                            long long* n_rectangle_uploads /* batch-frames that went as rectangles so far; may be NULL */);
int m3t_hip_roi_get_unrecovered(m3t_hip_context*, int* body_ids, int capacity, int* n_bodies);
/* Keep n_cus compute units free of the tracking kernels and run the ROI pull kernel on them (CU-masked streams), so
 * that the rectangles of frame k + 1 cross PCIe WHILE step k runs: on shared CUs the pull's PCIe reads hold the CUs'
 * memory pipelines and the step takes 2.3-3 x longer, on CUs of its own the pull costs the step little.  n_cus has to
 * take the same number of CUs from every shader engine: a multiple of 32 on MI355X (8 XCDs x 4 engines; anything else
 * is refused); 32 and 64 were measured (tools/ubench_cumask.hip, bench.py's ROI leg).  The tracking launches plan with the remaining CUs (fewer workgroups per object
 * where the batch no longer fits).  0 = off (default).  The call synchronises and REPLACES the context's streams:
 * fetch m3t_hip_get_stream again afterwards.  Results do not depend on it.  The layout is MI355X's (gfx950, 256 CUs in
 * one partition): other devices are refused (M3T_ERR_UNSUPPORTED).  CU-masked streams are BLOCKING streams (the runtime
 * offers no other kind): while CUs are reserved, work on the legacy NULL stream -- a plain hipMemcpy, torch's default
 * stream -- synchronises with the context's compute stream and its first copy stream; keep such work off the device
 * meanwhile or the pull / step overlap is lost. */
int m3t_hip_reserve_ingest_cus(m3t_hip_context*, int n_cus);

/* ---- Bodies (body.h:46: only body2world_pose crosses the boundary) ------------ */
int m3t_hip_body_create(m3t_hip_context*, const float body2world[16]);
int m3t_hip_body_set_body2world_pose(m3t_hip_context*, int body_id, const float body2world[16]);
int m3t_hip_body_get_body2world_pose(m3t_hip_context*, int body_id, float body2world[16]);
/* bulk variants: poses[n][16] for body ids 0..n-1 */
int m3t_hip_bodies_set_poses(m3t_hip_context*, const float* poses, int n);
int m3t_hip_bodies_get_poses(m3t_hip_context*, float* poses, int n);

/* ---- Modalities (modality.h:56-155) -------------------------------------------
 * RegionModality ctor + SetUp (region_modality.h:169-176, region_modality.cpp:25-99);
 * depth_camera_id = -1 unless measure_occlusions.  The renderer-fed options (use_region_checking /
 * model_occlusions / use_silhouette_checking) are switched on afterwards with the *_model_occlusions / *_use_*
 * calls below, which name the renderer; set in the parameter struct they are rejected (M3T_ERR_INVALID_ARGUMENT). */
int m3t_hip_region_modality_create(m3t_hip_context*, const m3t_region_modality_params*, int body_id,
                                   int color_camera_id, int region_model_id, int depth_camera_id);
int m3t_hip_depth_modality_create(m3t_hip_context*, const m3t_depth_modality_params*, int body_id,
                                  int depth_camera_id, int depth_model_id);
/* Modality::gradient() / hessian() (modality.h:89-90): 6 floats + column-major 6x6 */
int m3t_hip_modality_get_gradient_hessian(m3t_hip_context*, int modality_id, float gradient[6],
                                          float hessian[36]);
int m3t_hip_modality_set_gradient_hessian(m3t_hip_context*, int modality_id, const float gradient[6],
                                          const float hessian[36]);
/* the same for ALL modalities at once, in creation order: out[modality][6 + 36]; one device-to-host copy per round
 * instead of one per modality (what an adapter host calls after calculate_gradient_and_hessian) */
int m3t_hip_modalities_get_gradient_hessian(m3t_hip_context*, float* out, int capacity_modalities);
/* data_lines_ / data_points_ of the last CalculateCorrespondences (for visualisation / parity) */
int m3t_hip_region_modality_get_lines(m3t_hip_context*, int modality_id, m3t_data_line* out, int capacity,
                                      int* n_lines);
int m3t_hip_depth_modality_get_points(m3t_hip_context*, int modality_id, m3t_data_point* out, int capacity,
                                      int* n_points);
/* ColorHistograms state (color_histograms.h): n_bins^3 floats each */
int m3t_hip_region_modality_get_histograms(m3t_hip_context*, int modality_id, float* histogram_f,
                                           float* histogram_b);
int m3t_hip_region_modality_set_histograms(m3t_hip_context*, int modality_id, const float* histogram_f,
                                           const float* histogram_b);

/* ---- Links / Optimizers / Constraints (link.h:67, optimizer.h:48, constraint.h) -----
 * A free 6-dof root link per body (every RBOT / YCB configuration) takes the fused
 * rigid path; kinematic trees (child links, partial joints, links without body) and hard
 * constraints run through the general device path (m3t_links.hip).  body_id / parent may be -1. */
/* ---- renderer-fed branches: body meshes, focused depth / silhouette renderings ----
 * A software restatement of FocusedBasicDepthRenderer / FocusedSilhouetteRenderer (renderer.cpp:348-405,
 * basic_depth_renderer.cpp:45-84, silhouette_renderer.cpp:54-100): the square image_size x image_size crop
 * around the referenced bodies, rasterised with OpenGL's rules (pixel centres, 1/256 sub-pixel snapping,
 * top-left fill rule, 16-bit depth, GL_LESS in draw order).  Renderers a modality references are rendered
 * where the reference's Tracker does it: at start_modalities / calculate_results for region modalities and
 * before every calculate_correspondences (tracker.cpp:430-517). */
int m3t_hip_body_set_geometry(m3t_hip_context*, int body_id, const m3t_body_geometry*);
int m3t_hip_renderer_geometry_create(m3t_hip_context*);                                   /* RendererGeometry */
int m3t_hip_renderer_geometry_add_body(m3t_hip_context*, int geometry_id, int body_id);  /* draw order = add order */
int m3t_hip_focused_depth_renderer_create(m3t_hip_context*, int geometry_id, int camera_id, int image_size, float z_min,
                                          float z_max);                      /* basic_depth_renderer.h:120-128 */
int m3t_hip_focused_silhouette_renderer_create(m3t_hip_context*, int geometry_id, int camera_id, int id_type, int image_size,
                                               float z_min, float z_max);    /* silhouette_renderer.h:150-155 */
int m3t_hip_renderer_add_referenced_body(m3t_hip_context*, int renderer_id, int body_id);
int m3t_hip_renderer_start_rendering(m3t_hip_context*, int renderer_id);
/* depth: image_size^2 u16 (65535 = nothing); silhouette: image_size^2 u8 ids (NULL for depth renderers);
 * info: corner_u, corner_v, scale; n_visible: referenced bodies that passed FocusedRenderer's visibility test */
int m3t_hip_renderer_get_images(m3t_hip_context*, int renderer_id, uint16_t* depth, uint8_t* silhouette, float info[3],
                                int* n_visible);
/* RegionModality::ModelOcclusions / UseRegionChecking, DepthModality::ModelOcclusions / UseSilhouetteChecking */
int m3t_hip_region_modality_model_occlusions(m3t_hip_context*, int modality_id, int depth_renderer_id);
int m3t_hip_region_modality_use_region_checking(m3t_hip_context*, int modality_id, int silhouette_renderer_id);
int m3t_hip_depth_modality_model_occlusions(m3t_hip_context*, int modality_id, int depth_renderer_id);
int m3t_hip_depth_modality_use_silhouette_checking(m3t_hip_context*, int modality_id, int silhouette_renderer_id);

/* ColorHistograms shared by several RegionModalities (color_histograms.h:36-40, RegionModality::UseSharedColorHistograms
 * region_modality.cpp:168-173; cleared / initialised / updated once per step around all modalities, tracker.cpp:435-443,
 * 507-515).  The modality's own n_histogram_bins and learning rates are ignored once it shares. */
int m3t_hip_color_histograms_create(m3t_hip_context*, int n_bins, float learning_rate_f, float learning_rate_b);
int m3t_hip_region_modality_use_shared_color_histograms(m3t_hip_context*, int modality_id, int histograms_id);

int m3t_hip_link_create(m3t_hip_context*, int body_id, int parent_link_id, const float body2joint[16],
                        const float joint2parent[16], const int free_directions[6],
                        int fixed_body2joint_pose);
int m3t_hip_link_add_modality(m3t_hip_context*, int link_id, int modality_id);
int m3t_hip_optimizer_create(m3t_hip_context*, int root_link_id, float tikhonov_parameter_rotation,
                             float tikhonov_parameter_translation);
int m3t_hip_optimizer_create_rigid(m3t_hip_context*, int body_id, int n_modalities, const int* modality_ids,
                                   float tikhonov_parameter_rotation, float tikhonov_parameter_translation);
int m3t_hip_constraint_create(m3t_hip_context*, int optimizer_id, int link1_id, int link2_id,
                              const float body12joint1[16], const float body22joint2[16],
                              const int constraint_directions[6]);
/* SoftConstraint (include/m3t/soft_constraint.h:52-62; soft_constraint.cpp:113-131,220-272): a joint that only
 * pulls once its rotation / translation error exceeds max_distance_*, weighted with 1/standard_deviation^2;
 * it adds to the g/H of both links before the projection.  In a structure spread over several processes
 * (begin -> all-reduce -> end) every process adds them after the all-reduce, as one process does. */
int m3t_hip_soft_constraint_create(m3t_hip_context*, int optimizer_id, int link1_id, int link2_id,
                                   const float body12joint1[16], const float body22joint2[16],
                                   const int constraint_directions[6], float max_distance_rotation,
                                   float max_distance_translation, float standard_deviation_rotation,
                                   float standard_deviation_translation);
int m3t_hip_link_get_link2world_pose(m3t_hip_context*, int link_id, float pose[16]);
/* Link::set_link2world_pose (link.cpp:138-140; what Detector::UpdatePoses writes, detector.cpp:42-53):
 * the body's pose for a link with a body, the link's own frame for a body-less root */
int m3t_hip_link_set_link2world_pose(m3t_hip_context*, int link_id, const float pose[16]);
/* Link::set_body2joint_pose / set_joint2parent_pose (either may be NULL) and the getters */
int m3t_hip_link_set_joint_poses(m3t_hip_context*, int link_id, const float body2joint[16],
                                 const float joint2parent[16]);
int m3t_hip_link_get_joint_poses(m3t_hip_context*, int link_id, float body2joint[16], float joint2parent[16]);

/* ---- Tracker sub-steps (tracker.h:131-160; tracker.cpp:344-364, 430-517) -----------
 * Same names, arguments and order as the reference's public Tracker methods. */
int m3t_hip_tracker_set_iterations(m3t_hip_context*, int n_corr_iterations, int n_update_iterations);
int m3t_hip_start_modalities(m3t_hip_context*, int iteration);                     /* tracker.cpp:430 */
int m3t_hip_calculate_correspondences(m3t_hip_context*, int iteration, int corr_iteration); /* :447 */
int m3t_hip_calculate_gradient_and_hessian(m3t_hip_context*, int iteration, int corr_iteration,
                                           int opt_iteration);                    /* :471 */
int m3t_hip_calculate_optimization(m3t_hip_context*, int iteration, int corr_iteration,
                                   int opt_iteration);                            /* :481 */
/* Optimizer::CalculateOptimization split where a kinematic structure spread over several GPUs
 * exchanges data: begin() leaves the link sums of this process's modalities -- gradient (6) and Hessian (36)
 * of every link of every structure, Link::CalculateGradientAndHessian link.cpp:184-193, zero for links whose
 * modalities live in another process -- in ONE contiguous device buffer (*partial, `count` = 42 x links floats);
 * the host sums that buffer over the participating ranks with a single all-reduce (ncclAllReduce /
 * torch.distributed on RCCL, on the stream of m3t_hip_get_stream) and calls end(), which adds the soft
 * constraints, projects with the Jacobians (optimizer.cpp:309-321), adds constraint rows and the Tikhonov
 * diagonal, solves and updates the poses identically on every rank.  Keep all modalities of a link in one process
 * (3dobjecttracking_amd/sharding.py place_bodies does): every other process then adds +0.0 to that link's 42
 * numbers, the all-reduce is exact whatever its order, and the poses of N processes are the poses of one process
 * bit for bit.  (The projected [dof x dof | dof] blocks would be a smaller message for long chains, but their sum
 * over the ranks is a reassociation of the sum over the links, which the tracker's discrete decisions amplify.) */
int m3t_hip_calculate_optimization_begin(m3t_hip_context*, float** partial, size_t* count);
int m3t_hip_calculate_optimization_end(m3t_hip_context*);
/* The collective itself, inside the library: one ncclAllReduce(sum, float, count) on the context's stream, in place on
 * the buffer of begin().  The communicator is either the library's own (comm_get_unique_id on one rank, the 128 bytes
 * carried to the others by whatever the host uses -- MPI, a file, torch.distributed -- then comm_init_rank on every
 * rank, one rank per GPU) or one the host already has (comm_set(ctx, ncclComm_t)).  While a communicator is set,
 * m3t_hip_calculate_optimization (and with it execute_tracking_step / refine_poses) runs begin -> allreduce -> end
 * by itself, so the host's tracking loop is the single-GPU one.  librccl.so.1 is opened on first use only. */
int m3t_hip_comm_get_unique_id(m3t_hip_context*, void* id, size_t id_bytes /* >= 128 */);
int m3t_hip_comm_init_rank(m3t_hip_context*, const void* id, size_t id_bytes, int n_ranks, int rank);
int m3t_hip_comm_set(m3t_hip_context*, void* nccl_comm /* ncclComm_t or NULL */);
int m3t_hip_comm_destroy(m3t_hip_context*);
int m3t_hip_calculate_optimization_allreduce(m3t_hip_context*);
/* The host's own transport in the collective's place (MPI, gloo, shared memory between threads, a test harness that
 * plays several ranks on one GPU): fn is called wherever the library would call ncclAllReduce -- once per Newton step,
 * with the DEVICE buffer of begin() / of the fused distributed step, its length in floats and the context's
 * hipStream_t -- and has to leave the sum over the host's ranks in that buffer, ordered on that stream (it may
 * synchronise the stream and add on the host).  Non-zero return = failure: the step returns M3T_ERR_DEVICE.  While a
 * callback is set the context takes the distributed paths exactly as with a communicator (comm_get_allreduce_count
 * counts the calls; comm_get_rank_count stays 0: the library does not know the host's world).  NULL removes it.
 * Replaces nothing in the reference (optimizer.cpp:309-321 sums in one process); the seam next to comm_set. */
typedef int (*m3t_hip_reduce_fn)(void* user, float* device_buffer, size_t count, void* hip_stream);
int m3t_hip_comm_set_reduce_callback(m3t_hip_context*, m3t_hip_reduce_fn fn, void* user);
/* number of ncclAllReduce calls the context has issued so far (one per Newton step of a tracking step while a
 * communicator is set: the observable a host or a test checks the distributed path with) */
int m3t_hip_comm_get_allreduce_count(m3t_hip_context*, long long* count);
/* ranks of the communicator the context holds, asked of RCCL itself (ncclCommCount); 0 without a communicator: what a
 * benchmark line reports as the collective's width -- an observation, not an echo of the launcher's arguments */
int m3t_hip_comm_get_rank_count(m3t_hip_context*, int* n_ranks);
int m3t_hip_calculate_consistent_poses(m3t_hip_context*); /* tracker.cpp:423, optimizer.cpp:135 */
int m3t_hip_calculate_results(m3t_hip_context*, int iteration);                    /* :503 */
/* Tracker::ExecuteTrackingStep (M3T tracker.cpp:344) == Tracker::ExecuteTrackingCycle
 * (ICG tracker.cpp:247): the whole loop nest on the device, one launch per frame (two for large batches). */
int m3t_hip_execute_tracking_step(m3t_hip_context*, int iteration);
int m3t_hip_execute_tracking_cycle(m3t_hip_context*, int iteration);
/* 0: execute_tracking_step issues the sub-step kernels one by one (line state and
 *    g/H of every iteration observable); 1 (default): fused device loop;
 * 2: fused + line/point state and g/H of the last iteration written back. */
int m3t_hip_set_fused_step(m3t_hip_context*, int mode);
/* Batches that leave CUs idle run with several (4, 8 or 16) workgroups per object, which hand each other their
 * share of the line / point results inside the launch; that needs all of them resident at once, which holds while
 * this context has the GPU to itself (the launch shape is chosen against the runtime's occupancy query for an
 * otherwise idle device).  AUTOMATIC SPLIT THEREFORE ASSUMES EXCLUSIVE USE OF THE GPU: a second context or stream of
 * the same process, or another process, that occupies CUs can keep a partner workgroup from starting.  If a workgroup
 * waits in vain (about 5 ms), the step of that object is abandoned part-way -- workgroups that were already done may
 * have written the new pose and their share of the histogram bins, so pose and histograms of that object are
 * undefined -- and the NEXT call of execute_tracking_step / sync / a pose getter returns M3T_ERR_DEVICE: set the
 * poses again and call start_modalities.  A process that shares its GPU switches the split off.  The same holds for
 * kinematic structures: their one-launch step (tracking_step_tree_kernel, a workgroup per tracked link, the link sums
 * handed over inside the launch) is chosen under the same assumption and fails the same way; enable = 0 switches it
 * off as well (per-sub-step launches instead).  enable: 0 = off, 1 = automatic (default), 2..16 = at most that many
 * workgroups per object.  Results are bit-identical in every shape. */
int m3t_hip_set_object_split(m3t_hip_context*, int enable);
/* Refiner::RefinePoses (refiner.cpp:76-117): CalculateConsistentPoses, then n_corr_iterations x
 * (StartModalities + CalculateCorrespondences + n_update_iterations x (g/H + optimisation)), iteration index 0 */
int m3t_hip_refine_poses(m3t_hip_context*, int n_corr_iterations, int n_update_iterations);
int m3t_hip_sync(m3t_hip_context*);
/* (The gradient / Hessian sums over lines / points are always taken in the reference's sequential f32 order,
 * region_modality.cpp:550-554, depth_modality.cpp:361-377: whole tracking sequences reproduce the CPU path bit for
 * bit in every launch shape; there is no summation-mode switch.) */
/* Test hook: the logarithm of RegionModality::CalculateGradientAndHessian (region_modality.cpp:520-523) exactly as
 * the kernels take it (csrc/m3t_log.h, with the general double logarithm where the table path does not vouch for its
 * rounding), evaluated on the device for every float with a bit pattern in [first_bits, last_bits]:
 * out[0] = sum of result_bits * (input_bits | 1) mod 2^64, out[1] = evaluations that took the general logarithm,
 * out[2] = the sum over those evaluations alone.  tests/test_gpu_log.py compares all three with
 * float(std::log(double(x))) on the host over all of [FLT_MIN, 1]. */
int m3t_hip_debug_log_checksum(m3t_hip_context*, unsigned first_bits, unsigned last_bits, unsigned long long out[3]);
/* measurement aid (bench.py roofline leg): HIP events on the context stream.  enable = 1: a pair around every launch
 * of [0] the fused tracking kernel and [1] the histogram kernel, totals since enable (each pair also times the
 * launch gap in front of its kernel and the markers slow the stream a little: the sum exceeds the device time of the
 * undisturbed loop by a few per cent).  enable = 2: ONE pair -- the first event where timing is switched on, the second
 * when get_kernel_timing asks -- and launch counts in between: total_ms[0] = device time of the whole region, nothing
 * is inserted between the launches.  0: off */
int m3t_hip_set_kernel_timing(m3t_hip_context*, int enable);
/* name of the kernel the last execute_tracking_step launched for the tracking loop ("" = one launch per sub-step) */
int m3t_hip_get_step_kernel(m3t_hip_context*, char* name, size_t capacity);
int m3t_hip_get_kernel_timing(m3t_hip_context*, float total_ms[2], int launches[2]);
/* launch shape of the last fused tracking step: [0] objects, [1] workgroups per object (1, or 4 / 8 / 16 =
 * tracking_step_split_kernel when the batch leaves CUs idle), [2] threads per workgroup,
 * [3] 1 if the histogram update ran inside the same launch; zeros before the first fused step */
int m3t_hip_get_step_shape(m3t_hip_context*, int shape[4]);

#ifdef M3T_HIP_VISIBILITY_PUSHED
#pragma GCC visibility pop
#undef M3T_HIP_VISIBILITY_PUSHED
#endif

#ifdef __cplusplus
}
#endif
#endif /* M3T_HIP_H_ */
