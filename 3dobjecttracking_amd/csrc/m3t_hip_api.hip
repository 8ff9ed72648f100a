// m3t_hip_api.hip — host side of libm3t_hip.so: context, device-resident tables,
// .bin model parser and kernel launches behind the C-ABI of include/m3t_hip.h.
// Single translation unit together with the kernels (one hipcc invocation).

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is opened with dlopen when a structure spans GPUs

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../include/m3t_hip.h"
#include "m3t_device.h"
#include "m3t_view_rows.h"
#include "m3t_kernels.hip"
#include "m3t_compact.hip"
#include "m3t_render.hip"
#include "m3t_modelgen.hip"
#include "m3t_links.hip"
#include "m3t_ingest.hip"

namespace {

thread_local std::string g_create_error;

struct DevMem {
  void* p = nullptr;
  size_t bytes = 0;
  DevMem() = default;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
  ~DevMem() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  hipError_t alloc(size_t n) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
};

struct Model {
  bool region = true;
  int n_views = 0, n_points = 0, point_floats = 0;
  float stride_depth_offset = 0.002f, max_radius_depth_offset = 0.05f, max_extent = 0.0f;
  DevMem points, orientations, extents;
  DevMem points8, orientations4;  // device-only compact copies of the hot fields
  DevMem view_neighbors;          // [n_views][M3T_VIEW_ROW] float4 (closest_view_local), empty for tiny view sets
  std::vector<float> h_orientations;  // host copy: modalities whose models share their view table share the view search
  float box_min[3] = {0, 0, 0}, box_max[3] = {0, 0, 0};  // around the centres of all data points (ROI ingest, m3t_roi.h)
  float box_rho = 0.0f;  // ... which also lie in the ellipsoid of box_rho x the box's half extents (0: a flat box, not used)
};

struct Camera {
  bool is_depth = false;
  m3t_intrinsics intr{};
  float world2camera[16];
  float depth_scale = 0.0f;
  uint32_t pitch = 0;
  size_t frame_bytes = 0;
  int n_slots = 1, current = 0;
  std::vector<bool> has_image;
  std::vector<long> last_read_step;  // per slot: the last streaming step that read it (-1: none)
  std::vector<bool> slot_is_roi;     // per slot: only the trackers' rectangle of the frame was uploaded (m3t_ingest.hip)
  std::vector<hipEvent_t> slot_copied;  // per slot: behind its last camera_upload_slot_async (m3t_hip_camera_slot_sync); lazily made
  DevMem ring;          // this camera's own frame ring, or empty when it lives in a shared slab
  int slab = -1, slab_index = 0;  // shared slab (m3t_hip_cameras_set_ring): [slot][camera of the group][frame]
  uint8_t* frames = nullptr;      // slot 0 of this camera
  size_t slot_stride = 0;         // bytes from one slot of this camera to the next
  uint8_t* frame(int slot) const { return frames + size_t(slot) * slot_stride; }
};
struct FrameSlab {  // the frame rings of a group of cameras of equal geometry in one allocation
  DevMem mem;
  std::vector<int> cameras;
  // per ring slot: behind the last batch upload / rectangle pull into it (m3t_hip_camera_slot_sync of any of its
  // cameras).  ONE event per batch-frame: an event per camera cost a 64-camera upload 0.3 ms of host time.
  std::vector<hipEvent_t> slot_copied;
  ~FrameSlab() {
    for (auto& e : slot_copied)
      if (e) (void)hipEventDestroy(e);
  }
};

struct BodyGeometryH {  // body.h:46-60 on the device
  bool set = false;
  DevMem vertices, triangles;
  std::vector<float> h_vertices;  // host copies for model generation
  std::vector<int> h_triangles;
  int n_triangles = 0;
  float geometry2body[16];
  int culling = 1, body_id = 0, region_id = 0;
  float maximum_body_diameter = 0.0f;
};
struct RendererH {  // FocusedBasicDepthRenderer / FocusedSilhouetteRenderer
  bool silhouette = false;
  int geometry = -1, camera = -1, id_type = M3T_ID_TYPE_BODY, image_size = 200;
  float z_min = 0.02f, z_max = 10.0f;
  std::vector<int> referenced;
  DevMem depth, sil, packed, state;
  DevMem survivors, n_survivors;  // focused_setup_kernel -> focused_resolve_kernel (m3t_render.hip)
  int survivor_capacity = 0;
  bool rendered = false;
};
struct SharedHistogramsH {  // a ColorHistograms object used by several RegionModalities
  int n_bins = 16, bitshift = 4;
  float learning_rate_f = 0.2f, learning_rate_b = 0.2f;
  DevMem hist_f, hist_b, hist_norm, counts;
};
struct RegionMod {
  m3t_region_modality_params p{};
  int depth_renderer = -1, silhouette_renderer = -1;
  int shared_histograms = -1;
  int body, camera, depth_camera, model;
  DevMem hist_f, hist_b, hist_norm, occupancy, count_scratch, line_state, gh;
  RegionModDev dev{};
};
struct DepthMod {
  m3t_depth_modality_params p{};
  int depth_renderer = -1, silhouette_renderer = -1;
  int body, camera, model;
  DevMem point_state, gh;
  DepthModDev dev{};
};
struct ModalityRef {
  bool region;
  int index;
};
struct Link {
  int body = -1, parent = -1;
  std::vector<int> children, modalities;
  float body2joint[16], joint2parent[16], link2world[16];
  int free_directions[6] = {1, 1, 1, 1, 1, 1};
  int fixed_body2joint_pose = 1;
  bool simple = true;  // free 6-dof root link with identity joint poses and a body
  int optimizer = -1, local_index = -1;
};
struct ConstraintH {
  int link1, link2;
  float body12joint1[16], body22joint2[16];
  int directions[6];
};
struct SoftConstraintH {
  ConstraintH joint;
  float max_distance_rotation, max_distance_translation, standard_deviation_rotation, standard_deviation_translation;
};
struct Optimizer {
  int link;
  float tr, tt;
  std::vector<int> constraints, soft_constraints;
  std::vector<int> order;  // global link ids, depth first (parents before children)
  int dof = 0, n_rows = 0;
  size_t partial_offset = 0;
};

}  // namespace

struct m3t_hip_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipDeviceProp_t prop{};
  std::string error;
  std::vector<std::unique_ptr<Model>> region_models, depth_models;
  std::vector<std::unique_ptr<Camera>> cameras;
  std::vector<std::unique_ptr<FrameSlab>> slabs;
  std::vector<float> body_poses;  // host mirror [n][16] (valid when !poses_on_device_newer)
  std::vector<std::unique_ptr<RegionMod>> region_mods;
  std::vector<std::unique_ptr<DepthMod>> depth_mods;
  std::vector<ModalityRef> modalities;
  std::vector<std::unique_ptr<SharedHistogramsH>> shared_histograms;
  DevMem d_shared_histograms;
  std::vector<std::unique_ptr<BodyGeometryH>> body_geometries;  // parallel to the bodies (may be shorter)
  std::vector<std::vector<int>> renderer_geometries;
  std::vector<std::unique_ptr<RendererH>> renderers;
  DevMem d_renderers, d_render_region, d_render_all;  // RendererDev table; which renderers to run when
  DevMem d_render_region_pairs, d_render_all_pairs;   // the same lists as {renderer, twin or -1} (LaunchRenderers)
  int n_render_region_pairs = 0, n_render_all_pairs = 0;
  int n_render_region = 0, n_render_all = 0;
  int lds_raster = -1;  // focused_resolve_kernel (z-buffer in LDS) usable on this device: -1 not tried yet
  std::vector<Link> links;
  std::vector<ConstraintH> constraints;
  std::vector<SoftConstraintH> soft_constraints;
  std::vector<Optimizer> optimizers;
  bool tree_mode = false;          // any kinematic tree / constraint -> links_* kernels
  bool links_device_newer = false;  // joint poses on the device are ahead of the host mirror
  DevMem d_links, d_constraints, d_soft, d_treeopts, d_work, d_partial;
  DevMem d_link_sums, d_link_first;  // what a structure spread over processes exchanges (links_gather_kernel): [links][42], first link per structure
  DevMem d_link_sums_alt, d_links_alt;  // tracking_step_tree_segment_kernel: the second copies (sums, link table) that take turns
  DevMem d_poses_first;  // ... and the bodies' start poses for a frame of one Newton step
  bool tree_untracked_structure = false;  // some structure has no modality in this context (its bodies live on other ranks)
  size_t link_sums_count = 0;
  // tracking_step_tree_kernel: one workgroup per link that carries modalities
  DevMem d_treesteps, d_tracked_links, d_tree_exchange;
  int n_treesteps = 0;
  bool tree_fused_possible = false;
  bool tree_constrained = false;  // some structure has Constraint / SoftConstraint objects: tracking_step_tree_constrained_kernel
  size_t tree_block_floats = 0;  // LDS of the structure copy (link table, link sums, system, work arrays), largest structure
  unsigned tree_seq = 0;
  size_t partial_count = 0;
  size_t tree_lds = 0;  // bytes of LDS per structure for the link kernels, 0: work arrays in global memory
  bool partial_ready = false;
  // a kinematic structure spread over GPUs: this rank's RCCL communicator (m3t_hip_comm_init_rank) or the host's
  ncclComm_t comm = nullptr;
  long long allreduce_calls = 0;  // ncclAllReduce calls issued since the context was created (m3t_hip_comm_get_allreduce_count)
  bool comm_owned = false;
  // ... or the host's own transport in the collective's place (m3t_hip_comm_set_reduce_callback): MPI, gloo, threads
  m3t_hip_reduce_fn reduce_fn = nullptr;
  void* reduce_user = nullptr;
  bool Distributed() const { return comm != nullptr || reduce_fn != nullptr; }
  int n_corr_iterations = 5, n_update_iterations = 2;
  int fused_mode = 1;
  // device tables
  DevMem d_cams, d_region, d_depth, d_opts, d_poses, d_scratch_view;
  DevMem d_gh_sources, d_gh_all;  // m3t_hip_modalities_get_gradient_hessian
  int gh_sources_count = -1;
  bool tables_dirty = true, cams_dirty = true, slots_dirty = false, poses_dirty_host = true;
  const CameraDev* cams_active = nullptr;  // the camera table version the kernels read (UploadTables)
  size_t pose_capacity = 0;
  std::vector<RigidOptDev> opt_table;
  bool fused_possible = false;
  bool fused_per_search_possible = false;  // ... if it were not for renderer-fed branches: one launch per correspondence search
  bool fuse_histogram_possible = false;  // ... and the histogram update can ride in the same launch
  // several workgroups per object (tracking_step_split_kernel) for batches that leave most CUs idle
  bool split_possible = false;
  size_t tree_split_lds_attribute = 0;  // the dynamic LDS limit tracking_step_tree_split_kernel was given last
  bool split_enabled = true;  // m3t_hip_set_object_split
  std::map<std::tuple<const void*, int, size_t>, int> occupancy_cache;  // ResidentBlocks
  int ingest_cus = 0;   // m3t_hip_reserve_ingest_cus: CUs kept free of the tracking kernels for the ROI pull kernel
  int compute_cus = 0;  // what the tracking launches may count on (set with the device properties)
  size_t tree_lds_attribute = 0;  // dynamic LDS limit last set on the one-launch tree kernel ...
  const void* tree_lds_kernel = nullptr;  // ... and which of the two it was
  size_t tree_segment_lds_attribute = 0;  // the same for tracking_step_tree_segment_kernel
  const void* tree_segment_lds_kernel = nullptr;
  DevMem d_split;            // [objects][2 slots][parts][32 fields][256 / parts] granules, then one abort word per object
  size_t split_objects = 0;  // capacity of d_split
  unsigned split_seq = 0;    // launch counter inside the granule tags
  int split_parts_override = 0;  // m3t_hip_set_object_split(ctx, n > 1): workgroups per object (0: automatic)
  unsigned* split_abort_host = nullptr;  // mapped host word: sequence number of a split step that gave up
  unsigned* split_abort_dev = nullptr;   // the same word as the device sees it
  unsigned split_abort_seen = 0;
  unsigned split_launches = 0;  // never reset (unlike split_seq): the value an aborting launch leaves in the host word
  int last_step_shape[4] = {0, 0, 0, 0};  // m3t_hip_get_step_shape
  bool state_valid = false;  // line/point state + g/H on the device reflect the last step
  TrackLdsLayout layout{};
  // tracking_step_compact_kernel (m3t_compact.hip): the launch shape of batches with two objects per CU and more
  CompactLayout compact{};
  size_t lds_compact = 0;
  CompactLayout compact_table{};  // ... with the pair table compacted in LDS (tracking_step_compact_table_kernel)
  size_t lds_compact_table = 0;   // 0: not available for this context's modalities
  unsigned* table_overflow_host = nullptr;  // mapped: mixed bins that did not fit the LDS table, largest over the workgroups
  unsigned* table_overflow_dev = nullptr;
  bool compact_possible = false;       // every modality fits the kernel's assumptions (UploadTables)
  bool compact_fuses_histogram = false;  // ... and the count table fits next to the scratch block (<= 16 bins)
  const char* last_step_kernel = "";   // m3t_hip_get_step_kernel
  // ROI ingest (m3t_ingest.hip): rectangles instead of whole frames
  bool roi_enabled = false;        // m3t_hip_set_roi_ingest
  bool roi_adaptive = false;       // ... with enable = 2: per-body margins from the motion over the last step
  float roi_margin_px = 0.0f;
  bool roi_recorded = false;       // the last step was a fused rigid launch with the ROI tables in place: the next pull may go as rectangles
  int roi_n_poses = 0;             // n_corr_iterations + 2
  int n_roi_items = 0;
  long long roi_pulls = 0;         // batch-frames uploaded as rectangles so far
  long long roi_repeated = 0;      // bodies whose step was repeated on whole frames so far (read with roi_get_status)
  DevMem d_search_poses, d_roi_items, d_roi_item_first, d_roi_cam_ids, d_roi_all_cam_ids, d_roi_rects, d_roi_pose_snapshot,
         d_roi_motion_peak,  // [reader]: adaptive margins (roi_rect_kernel)
         d_roi_opt_of_region;  // [region modality]: its optimizer's row (region_histogram_kernel behind a guarded step)
  std::vector<int> roi_cam_ids;    // what d_roi_cam_ids holds (a batch whose ids are not consecutive)
  int roi_rect_slots = 0;          // d_roi_rects: [slot][camera id]
  // Where the whole frames behind the rectangles of a slot are (the repair after a guarded step reads them): one entry
  // per batch upload into the slot, replaced by the next upload of the same cameras.  The host block has to stay what
  // it is until the step that reads the slot is done -- the lifetime rule of every asynchronous upload.
  struct RoiSource {
    std::vector<int> ids;          // the batch's cameras (ring order)
    const int* d_ids = nullptr;    // the same on the device
    const uint8_t* src = nullptr;  // device address of the mapped host block
    size_t camera_stride = 0;
    uint32_t row_step = 0;
  };
  std::vector<std::vector<RoiSource>> roi_sources;  // [slot]
  // three snapshots of the bodies' poses, in turn: taken at the END of every fused rigid step (= the poses the next
  // step starts from, available a whole pull earlier than a snapshot at that step's start), or at a step's start when
  // the previous step's end does not vouch for them (first step, poses set by the host, other launches in between).
  // A rectangle upload reads the newest one and, for adaptive margins, the one before it.
  static constexpr int kRoiSnapshots = 3;
  hipEvent_t roi_snapshot_done[kRoiSnapshots] = {nullptr, nullptr, nullptr};
  bool roi_snapshot_valid = false;
  int roi_use = 0;              // the snapshot the next rectangle upload reads: the poses at the start of the step enqueued last
  int roi_prev = -1;            // the snapshot one step before roi_use (-1: none)
  int roi_end_index = 0;        // where the last step's end-of-step snapshot went
  bool roi_end_valid = false;   // ... and whether it still describes the device poses
  int* roi_miss_host = nullptr;             // mapped: [0] count, [1 ..] body ids: steps that left their rectangles (and were repeated)
  int* roi_miss_dev = nullptr;
  int* roi_unrecovered_host = nullptr;      // the same for bodies whose repeat missed again (no whole frame within reach)
  int* roi_unrecovered_dev = nullptr;
  static constexpr int kRoiMissCapacity = 255;
  int np_max = 0, off_points = 0;
  size_t lds_track = 0, lds_corr = 0, lds_hist = 0, lds_depth = 0;
  bool hist_counts_in_lds = true;
  // optional HIP-event timing of the two per-frame kernels (bench.py roofline leg)
  // pinned staging ring for the camera table: a frame switch is an async 1-2 KB copy, no host sync
  static constexpr int kStage = 8;
  void* cam_stage[kStage] = {nullptr};
  size_t cam_stage_bytes = 0;
  hipEvent_t cam_stage_done[kStage] = {nullptr};
  int cam_stage_next = 0;
  // asynchronous ingest: frame copies run on their own stream beside the tracking kernels
  static constexpr int kStepEvents = 16;
  static constexpr int kCopyStreams = 4;  // cameras are spread round-robin: per-copy DMA latencies overlap
  hipStream_t copy_stream[kCopyStreams] = {nullptr};
  hipEvent_t copies_done[kCopyStreams] = {nullptr};
  hipEvent_t step_done[kStepEvents] = {nullptr};
  bool async_ingest = false, untracked_launches = false;
  unsigned copies_pending = 0;  // bit s: copy stream s has frames not yet ordered before the compute stream
  long step_counter = 0, copy_waited_step[kCopyStreams] = {-1, -1, -1, -1};
  std::vector<void*> registered;
  bool timing = false;        // m3t_hip_set_kernel_timing(1): an event pair around every launch
  bool timing_region = false; // m3t_hip_set_kernel_timing(2): ONE pair around all launches until the query
  hipEvent_t region_a = nullptr, region_b = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  struct Pending { hipEvent_t a, b; int which; };
  std::vector<Pending> pending;
  double kernel_ms[2] = {0.0, 0.0};
  int kernel_launches[2] = {0, 0};
};

namespace {

using Ctx = m3t_hip_context;

// RCCL entry points, resolved on first use from the process's librccl.so.1 (the one torch.distributed loaded, if any:
// a communicator and the all-reduce that uses it must come from the same library)
struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  bool Load() {
    if (AllReduce) return true;
    handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!handle) { error = std::string("librccl.so.1: ") + dlerror(); return false; }
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(handle, "ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(handle, "ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
    CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(handle, "ncclCommCount"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(handle, "ncclAllReduce"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce) { error = "librccl.so.1 lacks the nccl* entry points"; AllReduce = nullptr; return false; }
    return true;
  }
};
Rccl g_rccl;

int Fail(Ctx* c, int code, const std::string& msg) {
  if (c) c->error = msg;
  return code;
}
#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return Fail(ctx, M3T_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));    \
  } while (0)
#define REQUIRE(cond, code, msg) \
  do {                           \
    if (!(cond)) return Fail(ctx, code, msg); \
  } while (0)

// ---- .bin parser: model.cpp:218-284 + region_model.cpp:259-307,346-363 + depth_model.cpp:215-283
bool SkipBodyData(std::ifstream& ifs) {
  uint64_t len = 0;
  ifs.read(reinterpret_cast<char*>(&len), 8);
  if (!ifs || len > (1u << 20)) return false;
  ifs.seekg(std::streamoff(len + 4 + 1 + 1 + 4 + 64), std::ios::cur);
  return bool(ifs);
}
struct HostModel {
  int n_views = 0, n_points = 0;
  float stride = 0, max_radius = 0;
  std::vector<float> points, orientations, extents;
};
bool ParseModelFile(const char* path, bool region, HostModel* m, std::string* err) {
  std::ifstream ifs{path, std::ios::in | std::ios::binary};
  if (!ifs.is_open()) { *err = std::string("Could not open model file ") + path; return false; }
  char type = 0;
  int32_t version = 0, n_divides = 0, n_points = 0, image_size = 0;
  float sphere_radius = 0, max_radius = 0, stride = 0;
  uint8_t use_random_seed = 0;
  ifs.read(&type, 1);
  ifs.read(reinterpret_cast<char*>(&version), 4);
  ifs.read(reinterpret_cast<char*>(&sphere_radius), 4);
  ifs.read(reinterpret_cast<char*>(&n_divides), 4);
  ifs.read(reinterpret_cast<char*>(&n_points), 4);
  ifs.read(reinterpret_cast<char*>(&max_radius), 4);
  ifs.read(reinterpret_cast<char*>(&stride), 4);
  ifs.read(reinterpret_cast<char*>(&use_random_seed), 1);
  ifs.read(reinterpret_cast<char*>(&image_size), 4);
  if (!ifs) { *err = "truncated model header"; return false; }
  if (type != (region ? 'r' : 'd')) { *err = "Wrong model type"; return false; }
  if (version != (region ? 10 : 9)) { *err = "Wrong version id"; return false; }
  if (!SkipBodyData(ifs)) { *err = "bad body data"; return false; }
  uint64_t n_assoc = 0;
  ifs.read(reinterpret_cast<char*>(&n_assoc), 8);
  if (region) {
    for (int g = 0; g < 4; ++g) {
      uint64_t n = 0;
      ifs.read(reinterpret_cast<char*>(&n), 8);
      for (uint64_t i = 0; i < n && ifs; ++i)
        if (!SkipBodyData(ifs)) { *err = "bad associated body data"; return false; }
    }
  } else {
    for (uint64_t i = 0; i < n_assoc && ifs; ++i)
      if (!SkipBodyData(ifs)) { *err = "bad occlusion body data"; return false; }
  }
  uint64_t n_views = 0;
  ifs.read(reinterpret_cast<char*>(&n_views), 8);
  if (!ifs || n_views == 0 || n_views > (1u << 24) || n_points <= 0 || n_points > (1 << 20)) {
    *err = "bad view table";
    return false;
  }
  const int pf = region ? M3T_REGION_POINT_FLOATS : M3T_DEPTH_POINT_FLOATS;
  m->n_views = int(n_views);
  m->n_points = n_points;
  m->stride = stride;
  m->max_radius = max_radius;
  m->points.resize(size_t(n_views) * n_points * pf);
  m->orientations.resize(size_t(n_views) * 3);
  m->extents.resize(n_views);
  for (uint64_t v = 0; v < n_views; ++v) {
    ifs.read(reinterpret_cast<char*>(&m->points[v * n_points * pf]), std::streamsize(size_t(n_points) * pf * 4));
    ifs.read(reinterpret_cast<char*>(&m->orientations[v * 3]), 12);
    ifs.read(reinterpret_cast<char*>(&m->extents[v]), 4);
  }
  if (!ifs) { *err = "truncated view data"; return false; }
  return true;
}

int CreateModel(Ctx* ctx, bool region, int n_views, int n_points, const float* pts, const float* ori,
                const float* ext, float stride, float max_radius) {
  REQUIRE(n_views > 0 && n_points > 0 && pts && ori && ext, M3T_ERR_INVALID_ARGUMENT, "bad model description");
  auto m = std::make_unique<Model>();
  m->region = region;
  m->n_views = n_views;
  m->n_points = n_points;
  m->point_floats = region ? M3T_REGION_POINT_FLOATS : M3T_DEPTH_POINT_FLOATS;
  m->stride_depth_offset = stride;
  m->max_radius_depth_offset = max_radius;
  for (int v = 0; v < n_views; ++v) m->max_extent = std::max(m->max_extent, ext[v]);
  m->h_orientations.assign(ori, ori + size_t(n_views) * 3);
  for (int k = 0; k < 3; ++k) {
    m->box_min[k] = 3.0e38f;
    m->box_max[k] = -3.0e38f;
  }
  for (size_t i = 0; i < size_t(n_views) * n_points; ++i)
    for (int k = 0; k < 3; ++k) {
      m->box_min[k] = std::min(m->box_min[k], pts[i * m->point_floats + k]);
      m->box_max[k] = std::max(m->box_max[k], pts[i * m->point_floats + k]);
    }
  {  // the largest normalised radius of a point about the box's centre (m3t_roi_ellipsoid)
    double mid[3], half[3], worst = 0.0;
    bool flat = false;
    for (int k = 0; k < 3; ++k) {
      mid[k] = 0.5 * (double(m->box_min[k]) + double(m->box_max[k]));
      half[k] = 0.5 * (double(m->box_max[k]) - double(m->box_min[k]));
      flat = flat || !(half[k] > 1e-9);
    }
    for (size_t i = 0; !flat && i < size_t(n_views) * n_points; ++i) {
      double r2 = 0.0;
      for (int k = 0; k < 3; ++k) {
        const double q = (double(pts[i * m->point_floats + k]) - mid[k]) / half[k];
        r2 += q * q;
      }
      worst = std::max(worst, r2);
    }
    m->box_rho = flat ? 0.0f : float(std::sqrt(worst) * (1.0 + 1e-4));
  }
  size_t pb = size_t(n_views) * n_points * m->point_floats * 4;
  HIPCHK(m->points.alloc(pb));
  HIPCHK(m->orientations.alloc(size_t(n_views) * 12));
  HIPCHK(m->extents.alloc(size_t(n_views) * 4));
  HIPCHK(hipMemcpy(m->points.p, pts, pb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m->orientations.p, ori, size_t(n_views) * 12, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m->extents.p, ext, size_t(n_views) * 4, hipMemcpyHostToDevice));
  {  // compact, 16-byte aligned copies: 32 B per point instead of 152/144 B, float4 per orientation
    const int pf = m->point_floats;
    std::vector<float> p8(size_t(n_views) * n_points * 8, 0.0f), o4(size_t(n_views) * 4, 0.0f);
    for (size_t i = 0; i < size_t(n_views) * n_points; ++i)
      for (int k = 0; k < (region ? 8 : 6); ++k) p8[i * 8 + k] = pts[i * pf + k];
    for (int v = 0; v < n_views; ++v)
      for (int k = 0; k < 3; ++k) o4[size_t(v) * 4 + k] = ori[size_t(v) * 3 + k];
    HIPCHK(m->points8.alloc(p8.size() * 4));
    HIPCHK(m->orientations4.alloc(o4.size() * 4));
    HIPCHK(hipMemcpy(m->points8.p, p8.data(), p8.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m->orientations4.p, o4.data(), o4.size() * 4, hipMemcpyHostToDevice));
  }
  if (n_views > M3T_VIEW_ROW) {
    const std::vector<float> rows = m3t_view_rows(ori, n_views);  // closest_view_local's table (m3t_view_rows.h)
    HIPCHK(m->view_neighbors.alloc(rows.size() * 4));
    HIPCHK(hipMemcpy(m->view_neighbors.p, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  }
  auto& vec = region ? ctx->region_models : ctx->depth_models;
  vec.push_back(std::move(m));
  return int(vec.size()) - 1;
}

int LoadModel(Ctx* ctx, const char* path, bool region) {
  REQUIRE(path, M3T_ERR_INVALID_ARGUMENT, "null path");
  HostModel h;
  std::string err;
  if (!ParseModelFile(path, region, &h, &err)) return Fail(ctx, M3T_ERR_IO, err);
  return CreateModel(ctx, region, h.n_views, h.n_points, h.points.data(), h.orientations.data(), h.extents.data(),
                     h.stride, h.max_radius);
}

int CreateCamera(Ctx* ctx, const m3t_intrinsics* intr, const float* w2c, bool depth, float depth_scale) {
  REQUIRE(intr && w2c && intr->width > 0 && intr->height > 0, M3T_ERR_INVALID_ARGUMENT, "bad camera description");
  auto c = std::make_unique<Camera>();
  c->is_depth = depth;
  c->intr = *intr;
  std::memcpy(c->world2camera, w2c, 64);
  c->depth_scale = depth_scale;
  size_t row = size_t(intr->width) * (depth ? 2 : 3);
  c->pitch = uint32_t((row + 63) / 64 * 64);
  c->frame_bytes = size_t(c->pitch) * intr->height;
  c->n_slots = 1;
  c->has_image.assign(1, false);
  c->last_read_step.assign(1, -1);
  c->slot_is_roi.assign(1, false);
  HIPCHK(c->ring.alloc(c->frame_bytes + 64));  // +64: pixels are fetched as one 4-byte load (B,G,R,+1)
  c->frames = c->ring.as<uint8_t>();
  c->slot_stride = c->frame_bytes;
  ctx->cameras.push_back(std::move(c));
  ctx->cams_dirty = true;
  return int(ctx->cameras.size()) - 1;
}

void RoiMarkWholeFrame(Ctx* ctx, int camera, int slot, hipStream_t stream);  // (ROI ingest, below)
void RoiMarkWholeFrames(Ctx* ctx, const int* ids, int n, int slot, hipStream_t stream);

int UploadFrame(Ctx* ctx, int id, int slot, const void* pixels, size_t row_step) {
  REQUIRE(id >= 0 && id < int(ctx->cameras.size()) && pixels, M3T_ERR_INVALID_ARGUMENT, "bad camera id");
  Camera& c = *ctx->cameras[id];
  REQUIRE(slot >= 0 && slot < c.n_slots, M3T_ERR_INVALID_ARGUMENT, "bad frame slot");
  size_t row = size_t(c.intr.width) * (c.is_depth ? 2 : 3);
  REQUIRE(row_step >= row, M3T_ERR_INVALID_ARGUMENT, "row_step smaller than one image row");
  uint8_t* dst = c.frame(slot);
  HIPCHK(hipMemcpy2DAsync(dst, c.pitch, pixels, row_step, row, c.intr.height, hipMemcpyHostToDevice, ctx->stream));
  // the host buffer is only borrowed for the duration of the call
  RoiMarkWholeFrame(ctx, id, slot, ctx->stream);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  c.has_image[slot] = true;
  return M3T_OK;
}

// ---- device tables -----------------------------------------------------------
// Workgroups of `kernel` (threads, dynamic LDS bytes) a CU keeps resident: asked once per (kernel, shape) and kept --
// the per-frame launch decision must not call into the driver (round-3 advisor).  0 = the query failed.
template <typename K>
int ResidentBlocks(Ctx* ctx, K kernel, int threads, size_t lds) {
  const auto key = std::make_tuple(reinterpret_cast<const void*>(kernel), threads, lds);
  auto it = ctx->occupancy_cache.find(key);
  if (it != ctx->occupancy_cache.end()) return it->second;
  int resident = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kernel, threads, lds) != hipSuccess) {
    (void)hipGetLastError();
    resident = 0;
  }
  ctx->occupancy_cache[key] = resident;
  return resident;
}

// After a synchronisation: did the workgroups of a split object ever give up waiting for each other?  (They
// are all resident by construction; the bounded wait exists so that a surprise cannot hang the device.)
int CheckSplitExchange(Ctx* ctx) {
  // the word lives in mapped host memory: no copy, no synchronisation; the kernel writes it with system scope
  if (!ctx->split_abort_host) return M3T_OK;
  const unsigned value = __atomic_load_n(ctx->split_abort_host, __ATOMIC_ACQUIRE);
  if (value != ctx->split_abort_seen) {
    ctx->split_abort_seen = value;
    const bool tree = std::strncmp(ctx->last_step_kernel, "tracking_step_tree_", 19) == 0;
    const bool render = std::strcmp(ctx->last_step_kernel, "tracking_step_split_render_kernel") == 0;
    const std::string msg =
        std::string(tree ? ctx->last_step_kernel : (render ? "tracking_step_split_render_kernel" : "tracking_step_split_kernel")) +
        ": a workgroup waited in vain for the other workgroups of its " + (tree ? "kinematic structure" : "object") +
        " (is another process or stream using this GPU?); the step was abandoned part-way: workgroups that had "
        "already finished may have written the new pose" + (tree ? "s and joints" : "") +
        " and blended their share of the histogram bins, so the poses and histograms involved are undefined; set the "
        "poses again and call start_modalities.  m3t_hip_set_object_split(ctx, 0) keeps both kernels out: one "
        "workgroup per rigid object, per-sub-step launches for kinematic structures";
    return Fail(ctx, M3T_ERR_DEVICE, msg.c_str());
  }
  return M3T_OK;
}
int SyncPosesToHost(Ctx* ctx) {
  // device is authoritative after any device-side optimisation
  size_t n = ctx->body_poses.size() / 16;
  if (n == 0 || ctx->poses_dirty_host) return M3T_OK;
  HIPCHK(hipMemcpyAsync(ctx->body_poses.data(), ctx->d_poses.p, n * 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return CheckSplitExchange(ctx);
}

void ComputeLayout(Ctx* ctx) {
  int nl = 1, ns = 1, bins3 = 0, np = 1;
  bool hist_lds = !ctx->region_mods.empty();
  for (auto& m : ctx->region_mods) {
    nl = std::max(nl, m->p.n_lines_max);
    ns = std::max(ns, std::max(9, m->dev.n_seg));
    int b = m->p.n_histogram_bins;
    bins3 = std::max(bins3, b * b * b);
  }
  for (auto& m : ctx->depth_mods) np = std::max(np, m->p.n_points_max);
  // the LDS-staged pair table must fit next to the line buffers (<= 16 bins: 32 KB)
  if (bins3 * 8 > 32 * 1024) hist_lds = false;
  TrackLdsLayout L{};
  L.nl = nl;
  L.ns = ns;
  int off = 0;
  L.off_misc = off; off += M3T_MISC_FLOATS;
  L.off_state = off; off += LS_FIELDS * nl;
  off = (off + 3) / 4 * 4;
  // chain | seg_f | seg_b live during a correspondence search; the product rows of the g/H sums during the Newton
  // steps between two searches: the same LDS
  const int block = off;
  L.off_chain = off; off += nl * ns;
  L.off_seg_f = off; off += nl * ns;
  L.off_seg_b = off; off += nl * ns;
  L.pitch_r = chain_pitch(nl);
  L.pitch_d = chain_pitch(np);
  L.off_rows_r = block;
  L.off_rows_d = block + (ctx->region_mods.empty() ? 0 : 27 * L.pitch_r);
  const int rows_end = L.off_rows_d + (ctx->depth_mods.empty() ? 0 : 27 * L.pitch_d);
  off = (std::max(off, rows_end) + 3) / 4 * 4;
  ctx->off_points = off;
  ctx->np_max = np;
  int off_with_points = off + (ctx->depth_mods.empty() ? 0 : PS_FIELDS * np);
  if (hist_lds) {
    L.off_hist = (off_with_points + 3) / 4 * 4;
    L.total_floats = L.off_hist + bins3 * 2;
  } else {
    L.off_hist = -1;
    L.total_floats = off_with_points;
  }
  ctx->layout = L;
  ctx->lds_track = size_t(L.total_floats) * 4;
  // the correspondence-only kernel does not need the depth point block
  ctx->lds_corr = ctx->lds_track;
  ctx->lds_depth = size_t(M3T_MISC_FLOATS + (PS_FIELDS * np + 3) / 4 * 4 + 27 * L.pitch_d) * 4;
  size_t counts = size_t(bins3) * 4;
  ctx->hist_counts_in_lds = (M3T_MISC_FLOATS * 4 + counts) <= 160 * 1024;
  ctx->lds_hist = M3T_MISC_FLOATS * 4 + (ctx->hist_counts_in_lds ? counts : 0);
  // the compact carve-up: scratch | line state | factor rows (+ point state | factor rows)
  CompactLayout C{};
  C.nl = nl;
  C.np = np;
  int o = M3T_COMPACT_MISC_FLOATS;
  C.off_state = o; o += ctx->region_mods.empty() ? 0 : CS_FIELDS * nl;
  o = (o + 3) / 4 * 4;
  C.pitch_r = chain_pitch(nl);
  C.off_rows_r = o; o += ctx->region_mods.empty() ? 0 : M3T_COMPACT_ROWS * C.pitch_r;
  o = (o + 3) / 4 * 4;
  C.off_points = o; o += ctx->depth_mods.empty() ? 0 : PS_FIELDS * np;
  o = (o + 3) / 4 * 4;
  C.pitch_d = chain_pitch(np);
  C.off_rows_d = o; o += ctx->depth_mods.empty() ? 0 : M3T_COMPACT_ROWS * C.pitch_d;
  C.total_floats = (o + 3) / 4 * 4;
  // The histogram update rides along in the tail of the launch.  Up to 16 bins per channel the packed count table
  // (one word per bin) fits behind the 1024-float scratch block; with 32 bins the samples' bins go to a 16-bit list
  // and the bins are counted and blended in passes (region_histogram_update<.., LIST>).  Budget: 40 KB per
  // workgroup, so that four stay resident per CU.  (Counting by L2 atomics into a table in HBM instead was measured:
  // 0.77 ms per step for 4096 objects x 8000 samples -- more than the whole separate histogram kernel.)
  ctx->compact_fuses_histogram = false;
  C.tail_list_row = C.tail_pass_bins = C.off_tail_list = C.off_tail_counts = 0;
  const int budget = 40 * 1024 / 4;
  if (bins3 > 0 && M3T_MISC_FLOATS + bins3 <= budget) {
    ctx->compact_fuses_histogram = true;
    C.off_tail_counts = M3T_MISC_FLOATS;
    C.total_floats = std::max(C.total_floats, M3T_MISC_FLOATS + bins3);
  } else if (bins3 > 0 && bins3 <= 32768) {
    float longest = 0.0f;
    for (auto& m : ctx->region_mods) longest = std::max(longest, m->p.max_considered_line_length);
    const int row = int(longest + 0.5f) + 1;
    const int list_words = nl * row;  // two walkers per line, two entries per word
    int pass = bins3;
    while (pass > 256 && M3T_MISC_FLOATS + list_words + pass > budget) pass /= 2;
    if (M3T_MISC_FLOATS + list_words + pass <= budget) {
      ctx->compact_fuses_histogram = true;
      C.tail_list_row = row;
      C.tail_pass_bins = pass;
      C.off_tail_list = M3T_MISC_FLOATS;
      C.off_tail_counts = M3T_MISC_FLOATS + list_words;
      C.total_floats = std::max(C.total_floats, M3T_MISC_FLOATS + list_words + pass);
    }
  }
  ctx->compact = C;
  ctx->lds_compact = size_t(C.total_floats) * 4;
  // tracking_step_compact_table_kernel: the same carve-up with the compacted pair table behind the search's part (the
  // tail's lists and counts lie over it: the table is dead by then).  Region-only batches with histograms of at least
  // 1024 bins; the budget is a third of the CU's LDS (three workgroups per CU instead of four).
  ctx->compact_table = C;
  ctx->compact_table.off_table = 0;
  ctx->lds_compact_table = 0;
  if (bins3 >= 1024 && bins3 % 32 == 0 && bins3 <= 32768 && !ctx->region_mods.empty()) {
    CompactLayout T = C;
    T.off_table = (o + 3) / 4 * 4;  // (o: the end of the search's carve-up)
    T.table_words = bins3 / 32;
    // 52 KB: three workgroups per CU with room for the allocation granule (a third of 160 KB, 54 356 B, does not fit three
    // times: measured, the kernel ran at two workgroups per CU)
    int budget_floats = 52 * 256;
    if (const char* e = std::getenv("M3T_HIP_COMPACT_TABLE_KB")) budget_floats = std::atoi(e) * 256;  // developer override
    const int index_floats = 2 * T.table_words + (T.table_words + 1) / 2;  // bit words, 16-bit ranks
    // (a stored pair takes five bytes: m3t_compact.hip, CompactTable)
    int cap = int((long(budget_floats - T.off_table - index_floats - 12) * 4) / 5) - 3;
    if (const char* e = std::getenv("M3T_HIP_COMPACT_TABLE_CAP")) cap = std::min(cap, std::atoi(e));  // (tests: force the overflow path)
    if (cap >= 64) {
      T.table_cap = std::min(std::min(cap, bins3), 65000);
      T.total_floats = std::max(C.total_floats,
                                (T.off_table + index_floats + (3 + T.table_cap) + (3 + T.table_cap + 3) / 4 + 3) / 4 * 4);
      T.table_overflow = nullptr;  // (set where the kernel is launched: a mapped host word)
      ctx->compact_table = T;
      ctx->lds_compact_table = size_t(T.total_floats) * 4;
    }
  }
}

void DfsOrder(Ctx* ctx, int link, std::vector<int>* order) {
  order->push_back(link);
  for (int c : ctx->links[link].children) DfsOrder(ctx, c, order);
}

// device -> host mirror of the joint poses (after device-side UpdatePoses)
int PullLinks(Ctx* ctx) {
  if (!ctx->links_device_newer || !ctx->tree_mode) return M3T_OK;
  size_t n = 0;
  for (auto& o : ctx->optimizers) n += o.order.size();
  std::vector<LinkDev> dev(n);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (n) HIPCHK(hipMemcpy(dev.data(), ctx->d_links.p, n * sizeof(LinkDev), hipMemcpyDeviceToHost));
  size_t off = 0;
  for (auto& o : ctx->optimizers)
    for (int lid : o.order) {
      Link& l = ctx->links[lid];
      std::memcpy(l.body2joint, dev[off].body2joint, 64);
      std::memcpy(l.joint2parent, dev[off].joint2parent, 64);
      std::memcpy(l.link2world, dev[off].link2world, 64);
      ++off;
    }
  ctx->links_device_newer = false;
  return M3T_OK;
}

int UploadTreeTables(Ctx* ctx) {
  int r = PullLinks(ctx);
  if (r) return r;
  std::vector<LinkDev> links;
  std::vector<ConstraintDev> cons;
  std::vector<SoftConstraintDev> soft;
  std::vector<TreeOptDev> opts(ctx->optimizers.size());
  std::vector<size_t> link_off(ctx->optimizers.size()), con_off(ctx->optimizers.size()), work_off(ctx->optimizers.size()),
      soft_off(ctx->optimizers.size());
  size_t work_total = 0, partial_total = 0, tree_work_max = 0;
  for (auto& l : ctx->links) { l.optimizer = -1; l.local_index = -1; }
  for (size_t oi = 0; oi < ctx->optimizers.size(); ++oi) {
    Optimizer& o = ctx->optimizers[oi];
    o.order.clear();
    DfsOrder(ctx, o.link, &o.order);
    int dof = 0;
    for (size_t k = 0; k < o.order.size(); ++k) {
      Link& l = ctx->links[o.order[k]];
      REQUIRE(l.optimizer < 0, M3T_ERR_INVALID_ARGUMENT, "a link belongs to two optimizers");
      l.optimizer = int(oi);
      l.local_index = int(k);
    }
    link_off[oi] = links.size();
    for (int lid : o.order) {
      const Link& l = ctx->links[lid];
      LinkDev d{};
      d.body = l.body;
      d.parent = l.parent >= 0 ? ctx->links[l.parent].local_index : -1;
      std::memcpy(d.body2joint, l.body2joint, 64);
      std::memcpy(d.joint2parent, l.joint2parent, 64);
      std::memcpy(d.link2world, l.link2world, 64);
      for (int i = 0; i < 6; ++i) d.free_directions[i] = l.free_directions[i];
      d.fixed_body2joint_pose = l.fixed_body2joint_pose;
      d.first_jacobian_index = dof;  // optimizer.cpp:237-250
      for (int i = 0; i < 6; ++i) dof += l.free_directions[i] ? 1 : 0;
      REQUIRE(int(l.modalities.size()) <= M3T_MAX_LINK_MODALITIES, M3T_ERR_UNSUPPORTED, "too many modalities per link");
      d.n_gh = int(l.modalities.size());
      for (int k = 0; k < d.n_gh; ++k) {
        const ModalityRef& ref = ctx->modalities[l.modalities[k]];
        d.gh[k] = ref.region ? ctx->region_mods[ref.index]->gh.as<float>() : ctx->depth_mods[ref.index]->gh.as<float>();
      }
      links.push_back(d);
    }
    o.dof = dof;
    con_off[oi] = cons.size();
    o.n_rows = 0;
    for (int cid : o.constraints) {
      const ConstraintH& c = ctx->constraints[cid];
      REQUIRE(ctx->links[c.link1].optimizer == int(oi) && ctx->links[c.link2].optimizer == int(oi),
              M3T_ERR_INVALID_ARGUMENT, "constraint links must belong to the optimizer's structure");
      ConstraintDev d{};
      d.link1 = ctx->links[c.link1].local_index;
      d.link2 = ctx->links[c.link2].local_index;
      std::memcpy(d.body12joint1, c.body12joint1, 64);
      std::memcpy(d.body22joint2, c.body22joint2, 64);
      d.n = 0;
      for (int i = 0; i < 6; ++i) { d.directions[i] = c.directions[i]; d.n += c.directions[i] ? 1 : 0; }
      o.n_rows += d.n;
      cons.push_back(d);
    }
    soft_off[oi] = soft.size();
    for (int sid : o.soft_constraints) {
      const SoftConstraintH& c = ctx->soft_constraints[sid];
      REQUIRE(ctx->links[c.joint.link1].optimizer == int(oi) && ctx->links[c.joint.link2].optimizer == int(oi),
              M3T_ERR_INVALID_ARGUMENT, "soft constraint links must belong to the optimizer's structure");
      SoftConstraintDev d{};
      d.joint.link1 = ctx->links[c.joint.link1].local_index;
      d.joint.link2 = ctx->links[c.joint.link2].local_index;
      std::memcpy(d.joint.body12joint1, c.joint.body12joint1, 64);
      std::memcpy(d.joint.body22joint2, c.joint.body22joint2, 64);
      for (int i = 0; i < 6; ++i) { d.joint.directions[i] = c.joint.directions[i]; d.joint.n += c.joint.directions[i] ? 1 : 0; }
      d.max_distance_rotation = c.max_distance_rotation;
      d.max_distance_translation = c.max_distance_translation;
      d.standard_deviation_rotation = c.standard_deviation_rotation;
      d.standard_deviation_translation = c.standard_deviation_translation;
      soft.push_back(d);
    }
    work_off[oi] = work_total;
    work_total += tree_work_floats(int(o.order.size()), dof, o.n_rows);
    tree_work_max = std::max(tree_work_max, tree_work_floats(int(o.order.size()), dof, o.n_rows));
    o.partial_offset = partial_total;
    partial_total += size_t(dof) * dof + dof;
  }
  HIPCHK(ctx->d_links.alloc(std::max<size_t>(1, links.size()) * sizeof(LinkDev)));
  HIPCHK(ctx->d_links_alt.alloc(std::max<size_t>(1, links.size()) * sizeof(LinkDev)));
  HIPCHK(ctx->d_constraints.alloc(std::max<size_t>(1, cons.size()) * sizeof(ConstraintDev)));
  HIPCHK(ctx->d_soft.alloc(std::max<size_t>(1, soft.size()) * sizeof(SoftConstraintDev)));
  HIPCHK(ctx->d_treeopts.alloc(std::max<size_t>(1, opts.size()) * sizeof(TreeOptDev)));
  HIPCHK(ctx->d_work.alloc(std::max<size_t>(1, work_total) * 4));
  // the structures' work arrays live in LDS when the largest fits (one wave per structure, m3t_links.hip)
  ctx->tree_lds = tree_work_max * 4 <= size_t(152) * 1024 ? tree_work_max * 4 : 0;
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(links_project_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->tree_lds)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(links_solve_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->tree_lds)));
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(links_solve_sums_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->tree_lds)));
  {
    std::vector<int> first(std::max<size_t>(1, opts.size()), 0);
    for (size_t oi = 0; oi < opts.size(); ++oi) first[oi] = int(link_off[oi]);
    HIPCHK(ctx->d_link_first.alloc(first.size() * sizeof(int)));
    HIPCHK(hipMemcpy(ctx->d_link_first.p, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
    ctx->link_sums_count = links.size() * 42;
    HIPCHK(ctx->d_link_sums.alloc(std::max<size_t>(1, ctx->link_sums_count) * 4));
    HIPCHK(hipMemset(ctx->d_link_sums.p, 0, ctx->d_link_sums.bytes));
    HIPCHK(ctx->d_link_sums_alt.alloc(std::max<size_t>(1, ctx->link_sums_count) * 4));
    HIPCHK(hipMemset(ctx->d_link_sums_alt.p, 0, ctx->d_link_sums_alt.bytes));
  }
  HIPCHK(ctx->d_partial.alloc(std::max<size_t>(1, partial_total) * 4));
  HIPCHK(hipMemset(ctx->d_partial.p, 0, ctx->d_partial.bytes));
  ctx->partial_count = partial_total;
  for (size_t oi = 0; oi < opts.size(); ++oi) {
    const Optimizer& o = ctx->optimizers[oi];
    TreeOptDev& d = opts[oi];
    d.n_links = int(o.order.size());
    d.links = ctx->d_links.as<LinkDev>() + link_off[oi];
    d.dof = o.dof;
    d.n_constraints = int(o.constraints.size());
    d.constraints = ctx->d_constraints.as<ConstraintDev>() + con_off[oi];
    d.n_rows = o.n_rows;
    d.n_soft = int(o.soft_constraints.size());
    d.soft = ctx->d_soft.as<SoftConstraintDev>() + soft_off[oi];
    d.tikhonov_rotation = o.tr;
    d.tikhonov_translation = o.tt;
    d.work = ctx->d_work.as<float>() + work_off[oi];
    d.partial = ctx->d_partial.as<float>() + o.partial_offset;
    d.links_alt = ctx->d_links_alt.as<LinkDev>() + link_off[oi];
    d.first_link = int(link_off[oi]);
  }
  if (!links.empty()) {
    HIPCHK(hipMemcpy(ctx->d_links.p, links.data(), links.size() * sizeof(LinkDev), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->d_links_alt.p, links.data(), links.size() * sizeof(LinkDev), hipMemcpyHostToDevice));
  }
  if (!cons.empty()) HIPCHK(hipMemcpy(ctx->d_constraints.p, cons.data(), cons.size() * sizeof(ConstraintDev), hipMemcpyHostToDevice));
  if (!soft.empty()) HIPCHK(hipMemcpy(ctx->d_soft.p, soft.data(), soft.size() * sizeof(SoftConstraintDev), hipMemcpyHostToDevice));
  // tracking_step_tree_kernel: the links that carry modalities, one workgroup each (at most one region and one
  // depth modality per link, every modality on a link), their exchange buffers, the LDS of a structure's copy
  {
    std::vector<TreeStepDev> steps;
    std::vector<int> tracked;
    std::vector<size_t> tracked_off(opts.size()), exchange_off(opts.size());
    size_t exchange_total = 0, attached = 0;
    bool possible = true;
    ctx->tree_block_floats = 0;
    ctx->tree_untracked_structure = false;
    for (size_t oi = 0; oi < opts.size(); ++oi) {
      const Optimizer& o = ctx->optimizers[oi];
      tracked_off[oi] = tracked.size();
      int n_tracked = 0;
      for (size_t k = 0; k < o.order.size(); ++k) {
        const Link& l = ctx->links[o.order[k]];
        if (l.modalities.empty()) continue;
        TreeStepDev st{};
        st.opt = int(oi);
        st.link = int(k);
        st.tracked = n_tracked++;
        st.region_modality = st.depth_modality = -1;
        st.region_first = 1;
        for (size_t m = 0; m < l.modalities.size(); ++m) {
          const ModalityRef& ref = ctx->modalities[l.modalities[m]];
          int& slot = ref.region ? st.region_modality : st.depth_modality;
          if (slot >= 0 || l.body < 0) possible = false;  // two of a kind on one link, or a modality without a body
          slot = ref.index;
          if (m == 0) st.region_first = ref.region ? 1 : 0;
          ++attached;
        }
        steps.push_back(st);
        tracked.push_back(int(k));
      }
      opts[oi].n_tracked = n_tracked;
      if (n_tracked == 0) ctx->tree_untracked_structure = true;
      exchange_off[oi] = exchange_total;
      exchange_total += size_t(2) * n_tracked * M3T_TREE_GRANULES + 1;  // + the structure's abort word
      const size_t block = (o.order.size() * sizeof(LinkDev) + 3) / 4 + o.order.size() * 42 + size_t(o.dof) * o.dof + o.dof +
                           tree_work_floats(int(o.order.size()), o.dof, o.n_rows);
      ctx->tree_block_floats = std::max(ctx->tree_block_floats, (block + 3) / 4 * 4);
    }
    if (attached != ctx->modalities.size() || steps.empty()) possible = false;
    // what the one-launch step's structure code is laid out for (m3t_links.hip, round 5); larger structures take the
    // per-sub-step launches.  Structures with Constraint / SoftConstraint objects run the kernel that has their code.
    ctx->tree_constrained = false;
    for (size_t oi = 0; oi < opts.size(); ++oi) {
      const Optimizer& o = ctx->optimizers[oi];
      if (int(o.order.size()) > M3T_TREE_FUSED_MAX_LINKS || o.dof + o.n_rows > M3T_TREE_FUSED_MAX_SIZE) possible = false;
      if (!o.constraints.empty() || !o.soft_constraints.empty()) ctx->tree_constrained = true;
    }
    HIPCHK(ctx->d_treesteps.alloc(std::max<size_t>(1, steps.size()) * sizeof(TreeStepDev)));
    HIPCHK(ctx->d_tracked_links.alloc(std::max<size_t>(1, tracked.size()) * sizeof(int)));
    HIPCHK(ctx->d_tree_exchange.alloc(std::max<size_t>(1, exchange_total) * 8));
    HIPCHK(hipMemset(ctx->d_tree_exchange.p, 0, ctx->d_tree_exchange.bytes));
    ctx->tree_seq = 0;
    if (!steps.empty()) HIPCHK(hipMemcpy(ctx->d_treesteps.p, steps.data(), steps.size() * sizeof(TreeStepDev), hipMemcpyHostToDevice));
    if (!tracked.empty()) HIPCHK(hipMemcpy(ctx->d_tracked_links.p, tracked.data(), tracked.size() * sizeof(int), hipMemcpyHostToDevice));
    for (size_t oi = 0; oi < opts.size(); ++oi) {
      opts[oi].tracked_links = ctx->d_tracked_links.as<int>() + tracked_off[oi];
      opts[oi].exchange = ctx->d_tree_exchange.as<unsigned long long>() + exchange_off[oi];
    }
    ctx->n_treesteps = int(steps.size());
    ctx->tree_fused_possible = possible;
  }
  if (!opts.empty()) HIPCHK(hipMemcpy(ctx->d_treeopts.p, opts.data(), opts.size() * sizeof(TreeOptDev), hipMemcpyHostToDevice));
  return M3T_OK;
}

// RendererDev table + the renderer fields of the modality tables + the two "which renderers" lists
// (region modalities' renderers for start / results, all referenced ones for correspondences)
int UploadRendererTables(Ctx* ctx) {
  {  // shared ColorHistograms objects
    std::vector<SharedHistogramsDev> sh(ctx->shared_histograms.size());
    for (size_t i = 0; i < sh.size(); ++i) {
      SharedHistogramsH& h = *ctx->shared_histograms[i];
      sh[i].n_bins = h.n_bins;
      sh[i].learning_rate_f = h.learning_rate_f;
      sh[i].learning_rate_b = h.learning_rate_b;
      sh[i].histogram_f = h.hist_f.as<float>();
      sh[i].histogram_b = h.hist_b.as<float>();
      sh[i].histogram_norm = h.hist_norm.as<float2>();
      sh[i].counts = h.counts.as<unsigned long long>();
    }
    HIPCHK(ctx->d_shared_histograms.alloc(std::max<size_t>(1, sh.size()) * sizeof(SharedHistogramsDev)));
    if (!sh.empty())
      HIPCHK(hipMemcpy(ctx->d_shared_histograms.p, sh.data(), sh.size() * sizeof(SharedHistogramsDev), hipMemcpyHostToDevice));
  }
  const size_t n = ctx->renderers.size();
  std::vector<RendererDev> table(n);
  for (size_t i = 0; i < n; ++i) {
    RendererH& h = *ctx->renderers[i];
    RendererDev& d = table[i];
    std::memset(&d, 0, sizeof(d));
    d.camera = h.camera;
    d.silhouette = h.silhouette ? 1 : 0;
    d.id_type = h.id_type;
    d.image_size = h.image_size;
    d.z_min = h.z_min;
    d.z_max = h.z_max;
    const std::vector<int>& bodies = ctx->renderer_geometries[h.geometry];
    d.n_bodies = int(bodies.size());
    for (int k = 0; k < d.n_bodies; ++k) {
      const BodyGeometryH& g = *ctx->body_geometries[bodies[k]];
      d.body[k] = bodies[k];
      d.vertices[k] = g.vertices.as<float>();
      d.triangles[k] = g.triangles.as<int>();
      d.n_triangles[k] = g.n_triangles;
      std::memcpy(d.geometry2body[k], g.geometry2body, 64);
      d.culling[k] = g.culling;
      d.id[k] = h.id_type == M3T_ID_TYPE_REGION ? g.region_id : g.body_id;
    }
    d.n_referenced = int(h.referenced.size());
    for (int k = 0; k < d.n_referenced; ++k) {
      d.referenced[k] = h.referenced[k];
      d.referenced_diameter[k] = ctx->body_geometries[h.referenced[k]]->maximum_body_diameter;
    }
    d.depth_image = h.depth.as<uint16_t>();
    d.silhouette_image = h.sil.as<uint8_t>();
    d.packed = h.packed.as<uint32_t>();
    d.state = h.state.as<float>();
    int n_triangles = 0;
    for (int k = 0; k < d.n_bodies; ++k) n_triangles += d.n_triangles[k];
    if (h.survivor_capacity < n_triangles || !h.n_survivors.p) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(h.survivors.alloc(std::max<size_t>(1, size_t(n_triangles)) * M3T_SURVIVOR_BYTES));
      HIPCHK(h.n_survivors.alloc(64));
      HIPCHK(hipMemset(h.n_survivors.p, 0, 64));
      h.survivor_capacity = n_triangles;
    }
    d.survivors = h.survivors.p;
    d.n_survivors = h.n_survivors.as<int>();
    d.survivor_capacity = h.survivor_capacity;
  }
  HIPCHK(ctx->d_renderers.alloc(std::max<size_t>(1, n) * sizeof(RendererDev)));
  if (n) HIPCHK(hipMemcpy(ctx->d_renderers.p, table.data(), n * sizeof(RendererDev), hipMemcpyHostToDevice));
  auto slot_of = [&](int renderer, int body) {
    const std::vector<int>& ref = ctx->renderers[renderer]->referenced;
    for (size_t k = 0; k < ref.size(); ++k)
      if (ref[k] == body) return int(k);
    return -1;
  };
  std::vector<char> for_region(n, 0), for_all(n, 0);
  for (auto& m : ctx->region_mods) {
    RegionModDev& d = m->dev;
    d.depth_renderer = d.model_occlusions ? ctx->d_renderers.as<RendererDev>() + m->depth_renderer : nullptr;
    d.silhouette_renderer = d.use_region_checking ? ctx->d_renderers.as<RendererDev>() + m->silhouette_renderer : nullptr;
    d.depth_renderer_slot = d.model_occlusions ? slot_of(m->depth_renderer, m->body) : -1;
    d.silhouette_renderer_slot = d.use_region_checking ? slot_of(m->silhouette_renderer, m->body) : -1;
    if (d.model_occlusions) for_region[m->depth_renderer] = for_all[m->depth_renderer] = 1;
    if (d.use_region_checking) for_region[m->silhouette_renderer] = for_all[m->silhouette_renderer] = 1;
  }
  for (auto& m : ctx->depth_mods) {
    DepthModDev& d = m->dev;
    d.depth_renderer = d.model_occlusions ? ctx->d_renderers.as<RendererDev>() + m->depth_renderer : nullptr;
    d.silhouette_renderer =
        d.use_silhouette_checking ? ctx->d_renderers.as<RendererDev>() + m->silhouette_renderer : nullptr;
    d.depth_renderer_slot = d.model_occlusions ? slot_of(m->depth_renderer, m->body) : -1;
    d.silhouette_renderer_slot = d.use_silhouette_checking ? slot_of(m->silhouette_renderer, m->body) : -1;
    if (d.model_occlusions) for_all[m->depth_renderer] = 1;
    if (d.use_silhouette_checking) for_all[m->silhouette_renderer] = 1;
  }
  std::vector<int> list_region, list_all;
  for (size_t i = 0; i < n; ++i) {
    if (for_region[i]) list_region.push_back(int(i));
    if (for_all[i]) list_all.push_back(int(i));
  }
  // twins: two renderers of a list that differ in nothing but the id written (focused_setup_kernel) are drawn once
  auto same_rendering = [&](int a, int b) {
    const RendererH& x = *ctx->renderers[a];
    const RendererH& y = *ctx->renderers[b];
    return x.camera == y.camera && x.geometry == y.geometry && x.image_size == y.image_size && x.z_min == y.z_min &&
           x.z_max == y.z_max && x.referenced == y.referenced;
  };
  auto pairs_of = [&](const std::vector<int>& list) {
    std::vector<int> pairs;
    std::vector<char> taken(list.size(), 0);
    for (size_t i = 0; i < list.size(); ++i) {
      if (taken[i]) continue;
      int twin = -1;
      for (size_t j = i + 1; j < list.size() && twin < 0; ++j)
        if (!taken[j] && same_rendering(list[i], list[j])) {
          twin = list[j];
          taken[j] = 1;
        }
      pairs.push_back(list[i]);
      pairs.push_back(twin);
    }
    return pairs;
  };
  const std::vector<int> pairs_region = pairs_of(list_region), pairs_all = pairs_of(list_all);
  ctx->n_render_region_pairs = int(pairs_region.size() / 2);
  ctx->n_render_all_pairs = int(pairs_all.size() / 2);
  HIPCHK(ctx->d_render_region_pairs.alloc(std::max<size_t>(1, pairs_region.size()) * 4));
  HIPCHK(ctx->d_render_all_pairs.alloc(std::max<size_t>(1, pairs_all.size()) * 4));
  if (!pairs_region.empty())
    HIPCHK(hipMemcpy(ctx->d_render_region_pairs.p, pairs_region.data(), pairs_region.size() * 4, hipMemcpyHostToDevice));
  if (!pairs_all.empty())
    HIPCHK(hipMemcpy(ctx->d_render_all_pairs.p, pairs_all.data(), pairs_all.size() * 4, hipMemcpyHostToDevice));
  ctx->n_render_region = int(list_region.size());
  ctx->n_render_all = int(list_all.size());
  HIPCHK(ctx->d_render_region.alloc(std::max<size_t>(1, list_region.size()) * 4));
  HIPCHK(ctx->d_render_all.alloc(std::max<size_t>(1, list_all.size()) * 4));
  if (!list_region.empty())
    HIPCHK(hipMemcpy(ctx->d_render_region.p, list_region.data(), list_region.size() * 4, hipMemcpyHostToDevice));
  if (!list_all.empty())
    HIPCHK(hipMemcpy(ctx->d_render_all.p, list_all.data(), list_all.size() * 4, hipMemcpyHostToDevice));
  return M3T_OK;
}

// Renderings whose z-buffer fits the LDS of a CU: set-up + survivor list (32 slices of the triangle lists per renderer),
// then one workgroup per renderer that rasterises the survivors in LDS and writes the images.  Larger ones: clear + crop,
// rasterise into a z-buffer in memory, unpack.
int LaunchRenderers(Ctx* ctx, const int* which, int n_which, const int* pairs, int n_pairs, int largest_image_size) {
  if (n_which == 0) return M3T_OK;
  // work spread: ~128 set-up workgroups (slices of the triangle lists) and ~256 resolve workgroups (bands of image
  // rows, at least two rows each) per launch, whatever the number of renderings (measured on the reference's test
  // scene, two pairs of twins, renderer-fed step: 32 bands 0.633 ms, 64: 0.624, 100: 0.611; 64 -> 128 slices: no
  // change, 256: slower)
  // (round 5: at 128 pairs -- the renderer-fed step of 64 objects -- 32 slices per pair were 4096 workgroups of 149
  // VGPRs, one per CU at a time, each paying the projection and the matrix chain for two trips over its triangles:
  // 131 us per set-up launch; 2 slices per pair = one workgroup per CU: 2.33 -> 1.64 ms per step,
  // profiles/r05_render64_knobs.txt)
  int slices = std::min(128, std::max(2, 128 / std::max(1, n_pairs)));
  int bands = std::min(std::max(8, largest_image_size / 2), std::max(8, 256 / std::max(1, n_pairs)));
  if (const char* e = std::getenv("M3T_HIP_RASTER_BANDS")) bands = std::max(1, std::atoi(e));    // developer overrides
  if (const char* e = std::getenv("M3T_HIP_RASTER_SLICES")) slices = std::max(1, std::atoi(e));
  const size_t band_rows = (size_t(largest_image_size) + bands - 1) / bands;
  const size_t lds = band_rows * largest_image_size * 4 + (M3T_BLOCK_THREADS + 1 + 16 + 4 * M3T_BLOCK_THREADS) * 4;  // z-buffer band | prefix sums | wave totals | per-thread counts
  if (ctx->lds_raster < 0) {  // once per context (= per device)
    ctx->lds_raster = 1;
    if (std::getenv("M3T_HIP_NO_LDS_RASTER")) ctx->lds_raster = 0;
    else if (hipFuncSetAttribute(reinterpret_cast<const void*>(focused_resolve_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void)hipGetLastError();
      ctx->lds_raster = 0;
    }
  }
  if (ctx->lds_raster == 1 && largest_image_size > 0 && lds <= size_t(160) * 1024) {
    hipLaunchKernelGGL(focused_setup_kernel, dim3(slices, n_pairs), dim3(M3T_BLOCK_THREADS), 0, ctx->stream,
                       ctx->d_renderers.as<RendererDev>(), pairs, ctx->cams_active, ctx->d_poses.as<float>());
    hipLaunchKernelGGL(focused_resolve_kernel, dim3(n_pairs, bands), dim3(M3T_BLOCK_THREADS), lds, ctx->stream,
                       ctx->d_renderers.as<RendererDev>(), pairs);
    HIPCHK(hipGetLastError());
    return M3T_OK;
  }
#ifndef M3T_RASTER_SLICES
#define M3T_RASTER_SLICES 32
#endif
  hipLaunchKernelGGL(focused_clear_kernel, dim3(16, n_which), dim3(M3T_BLOCK_THREADS), 0, ctx->stream,
                     ctx->d_renderers.as<RendererDev>(), which, ctx->cams_active, ctx->d_poses.as<float>());
  hipLaunchKernelGGL(focused_raster_kernel, dim3(M3T_RASTER_SLICES, n_which), dim3(M3T_BLOCK_THREADS), 0, ctx->stream,
                     ctx->d_renderers.as<RendererDev>(), which, ctx->cams_active, ctx->d_poses.as<float>());
  hipLaunchKernelGGL(focused_unpack_kernel, dim3(16, n_which), dim3(M3T_BLOCK_THREADS), 0, ctx->stream,
                     ctx->d_renderers.as<RendererDev>(), which);
  HIPCHK(hipGetLastError());
  return M3T_OK;
}
int LargestRendererImage(Ctx* ctx) {
  int s = 0;
  for (auto& r : ctx->renderers) s = std::max(s, r->image_size);
  return s;
}
int RenderForModalities(Ctx* ctx, bool region_only) {
  for (auto& r : ctx->renderers) r->rendered = true;
  return LaunchRenderers(ctx, region_only ? ctx->d_render_region.as<int>() : ctx->d_render_all.as<int>(),
                         region_only ? ctx->n_render_region : ctx->n_render_all,
                         region_only ? ctx->d_render_region_pairs.as<int>() : ctx->d_render_all_pairs.as<int>(),
                         region_only ? ctx->n_render_region_pairs : ctx->n_render_all_pairs, LargestRendererImage(ctx));
}

// ROI ingest: the readers of every camera (one per modality of a rigid optimizer: which body, which box of model
// points, which reach -- m3t_roi.h) sorted by camera, and where the fused kernels leave their search poses.
// Called while the optimizer table is being rebuilt.
int BuildRoiTables(Ctx* ctx) {
  ctx->n_roi_items = 0;
  ctx->roi_recorded = false;
  ctx->roi_snapshot_valid = false;
  ctx->roi_end_valid = false;
  for (auto& od : ctx->opt_table) od.search_poses = nullptr;
  if (!ctx->roi_enabled || ctx->tree_mode || ctx->opt_table.empty()) return M3T_OK;
  ctx->roi_n_poses = ctx->n_corr_iterations + 2;
  // per object: the poses, then the guard's header (RoiGuardDev: which rows of the reader table are this object's)
  static_assert(sizeof(RoiGuardDev) % 4 == 0, "RoiGuardDev is addressed in floats");
  const size_t per_object = size_t(ctx->roi_n_poses) * 16 + sizeof(RoiGuardDev) / 4;
  HIPCHK(ctx->d_search_poses.alloc(ctx->opt_table.size() * per_object * 4));
  HIPCHK(hipMemset(ctx->d_search_poses.p, 0, ctx->opt_table.size() * per_object * 4));
  ctx->roi_prev = -1;
  std::vector<RoiItemDev> items;
  auto add = [&](int camera, int body, int opt, const Model& model, float reach_px, float reach_m) {
    RoiItemDev it{};
    it.camera = camera;
    it.body = body;
    it.opt = opt;
    for (int k = 0; k < 3; ++k) {
      it.box_min[k] = model.box_min[k];
      it.box_max[k] = model.box_max[k];
    }
    it.reach_px = reach_px;
    it.reach_m = reach_m;
    it.rho = model.box_rho;
    items.push_back(it);
  };
  for (size_t oi = 0; oi < ctx->opt_table.size(); ++oi) {
    RigidOptDev& od = ctx->opt_table[oi];
    od.search_poses = ctx->d_search_poses.as<float>() + oi * per_object;
    if (od.region_modality >= 0) {
      const RegionMod& m = *ctx->region_mods[od.region_modality];
      const Model& model = *ctx->region_models[m.model];
      float reach = m3t_roi_region_histogram_reach(&m.p);
      for (int c = 0; c < ctx->n_corr_iterations; ++c) reach = std::max(reach, m3t_roi_region_line_reach(&m.p, c));
      add(m.camera, od.body, int(oi), model, reach, 0.0f);
      if (m.p.measure_occlusions) {
        float reach_m = 0.0f, reach_px = 0.0f;
        m3t_roi_region_depth_reach(&m.p, &reach_m, &reach_px);
        add(m.depth_camera, od.body, int(oi), model, reach_px, reach_m);
      }
    }
    if (od.depth_modality >= 0) {
      const DepthMod& m = *ctx->depth_mods[od.depth_modality];
      float reach_m = 0.0f, reach_px = 0.0f;
      m3t_roi_depth_reach(&m.p, ctx->cameras[m.camera]->intr.fu, &reach_m, &reach_px);
      add(m.camera, od.body, int(oi), *ctx->depth_models[m.model], reach_px, reach_m);
    }
  }
  std::stable_sort(items.begin(), items.end(), [](const RoiItemDev& a, const RoiItemDev& b) { return a.camera < b.camera; });
  const size_t n_cams = ctx->cameras.size();
  std::vector<int> first(n_cams + 1, 0);
  for (auto& it : items) ++first[size_t(it.camera) + 1];
  for (size_t c = 0; c < n_cams; ++c) first[c + 1] += first[c];
  HIPCHK(ctx->d_roi_items.alloc(std::max<size_t>(1, items.size()) * sizeof(RoiItemDev)));
  {  // region modality -> its optimizer (region_histogram_kernel behind a guarded step)
    std::vector<int> opt_of(std::max<size_t>(1, ctx->region_mods.size()), -1);
    for (size_t oi = 0; oi < ctx->opt_table.size(); ++oi)
      if (ctx->opt_table[oi].region_modality >= 0) opt_of[size_t(ctx->opt_table[oi].region_modality)] = int(oi);
    HIPCHK(ctx->d_roi_opt_of_region.alloc(opt_of.size() * sizeof(int)));
    HIPCHK(hipMemcpy(ctx->d_roi_opt_of_region.p, opt_of.data(), opt_of.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  HIPCHK(ctx->d_roi_item_first.alloc(first.size() * sizeof(int)));
  if (!items.empty())
    HIPCHK(hipMemcpy(ctx->d_roi_items.p, items.data(), items.size() * sizeof(RoiItemDev), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_roi_item_first.p, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
  ctx->n_roi_items = int(items.size());
  HIPCHK(ctx->d_roi_motion_peak.alloc(std::max<size_t>(1, items.size()) * sizeof(float)));
  {  // (until a body's motion is known its margin is the caller's)
    std::vector<float> peaks(std::max<size_t>(1, items.size()), -1.0f);
    HIPCHK(hipMemcpy(ctx->d_roi_motion_peak.p, peaks.data(), peaks.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  {  // the guard's headers: every object's readers by their rows in the sorted table
    std::vector<float> blocks(ctx->opt_table.size() * per_object, 0.0f);
    for (size_t j = 0; j < items.size(); ++j) {
      RoiGuardDev* g = reinterpret_cast<RoiGuardDev*>(blocks.data() + size_t(items[j].opt) * per_object + size_t(ctx->roi_n_poses) * 16);
      // (a reader beyond the header's capacity would shrink its camera's upload without any kernel checking it)
      REQUIRE(g->n_items < M3T_ROI_GUARD_ITEMS, M3T_ERR_UNSUPPORTED,
              "ROI ingest: an optimizer has more frame readers than the guard header holds (M3T_ROI_GUARD_ITEMS)");
      g->item[g->n_items++] = int(j);
    }
    HIPCHK(hipMemcpy(ctx->d_search_poses.p, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice));
  }
  {  // the camera ids in order: a batch of consecutive ids needs no list of its own
    std::vector<int> all(std::max<size_t>(1, n_cams));
    for (size_t c = 0; c < n_cams; ++c) all[c] = int(c);
    HIPCHK(ctx->d_roi_all_cam_ids.alloc(all.size() * sizeof(int)));
    HIPCHK(hipMemcpy(ctx->d_roi_all_cam_ids.p, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
  }
  int slots = 1;
  for (auto& c : ctx->cameras) slots = std::max(slots, c->n_slots);
  ctx->roi_rect_slots = slots;
  HIPCHK(ctx->d_roi_rects.alloc(size_t(slots) * n_cams * sizeof(m3t_roi_rect)));
  ctx->roi_sources.assign(size_t(slots), {});
  HIPCHK(ctx->d_roi_pose_snapshot.alloc(Ctx::kRoiSnapshots * std::max<size_t>(64, ctx->body_poses.size() * 4)));
  for (auto& e : ctx->roi_snapshot_done)
    if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  {  // what every slot holds now: a whole frame, or (a rectangle from before the rebuild) nothing that can be vouched for
    std::vector<m3t_roi_rect> rects(size_t(slots) * n_cams, m3t_roi_empty());
    for (size_t c = 0; c < n_cams; ++c)
      for (int sl = 0; sl < ctx->cameras[c]->n_slots; ++sl)
        if (!ctx->cameras[c]->slot_is_roi[sl])
          rects[size_t(sl) * n_cams + c] = m3t_roi_rect{0, 0, ctx->cameras[c]->intr.width - 1, ctx->cameras[c]->intr.height - 1};
    HIPCHK(hipMemcpy(ctx->d_roi_rects.p, rects.data(), rects.size() * sizeof(m3t_roi_rect), hipMemcpyHostToDevice));
  }
  if (!ctx->roi_miss_host) {  // two lists in one mapped block: misses (repeated), misses of the repeat
    void *host = nullptr, *dev = nullptr;
    const size_t list = Ctx::kRoiMissCapacity + 1;
    HIPCHK(hipHostMalloc(&host, 2 * list * sizeof(int), hipHostMallocMapped));
    std::memset(host, 0, 2 * list * sizeof(int));
    HIPCHK(hipHostGetDevicePointer(&dev, host, 0));
    ctx->roi_miss_host = static_cast<int*>(host);
    ctx->roi_miss_dev = static_cast<int*>(dev);
    ctx->roi_unrecovered_host = ctx->roi_miss_host + list;
    ctx->roi_unrecovered_dev = ctx->roi_miss_dev + list;
  }
  return M3T_OK;
}

// whole frames were (or are being, on `stream`) written into the slot of the cameras ids[0 .. n): ONE launch notes
// that in the rectangle table (round 4 launched one single-thread kernel per camera: 64 per batch-frame)
void RoiMarkWholeFrames(Ctx* ctx, const int* ids, int n, int slot, hipStream_t stream) {
  bool consecutive = true;
  for (int i = 0; i < n; ++i) {
    ctx->cameras[ids[i]]->slot_is_roi[slot] = false;
    consecutive = consecutive && ids[i] == ids[0] + i;
  }
  if (slot < int(ctx->roi_sources.size())) {  // the whole frames replace what the rectangles of these cameras came from
    auto& sources = ctx->roi_sources[size_t(slot)];
    for (size_t k = 0; k < sources.size();) {
      bool overlaps = false;
      for (int i = 0; i < n && !overlaps; ++i)
        overlaps = std::find(sources[k].ids.begin(), sources[k].ids.end(), ids[i]) != sources[k].ids.end();
      if (overlaps) sources.erase(sources.begin() + long(k)); else ++k;
    }
  }
  if (!ctx->roi_enabled || ctx->n_roi_items == 0 || slot >= ctx->roi_rect_slots || ctx->tables_dirty) return;
  m3t_roi_rect* rects = ctx->d_roi_rects.as<m3t_roi_rect>() + size_t(slot) * ctx->cameras.size();
  // (a batch is a group of cameras of equal geometry: m3t_hip_cameras_set_ring)
  for (int i = 0; i < n; i += consecutive ? n : 1) {
    const Camera& c = *ctx->cameras[ids[i]];
    hipLaunchKernelGGL(roi_set_rects_kernel, dim3(((consecutive ? n : 1) + 63) / 64), dim3(64), 0, stream, rects, ids[i],
                       consecutive ? n : 1, c.intr.width, c.intr.height);
  }
}
void RoiMarkWholeFrame(Ctx* ctx, int camera, int slot, hipStream_t stream) { RoiMarkWholeFrames(ctx, &camera, 1, slot, stream); }

int UploadTables(Ctx* ctx) {
  if (ctx->copies_pending) {
    // frames enqueued on the copy stream so far become visible to everything launched from here on
    for (int k = 0; k < Ctx::kCopyStreams; ++k) {
      if (!(ctx->copies_pending >> k & 1u)) continue;
      HIPCHK(hipEventRecord(ctx->copies_done[k], ctx->copy_stream[k]));
      HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->copies_done[k], 0));
    }
    ctx->copies_pending = 0;
  }
  if (ctx->cams_dirty || ctx->tables_dirty || ctx->slots_dirty) {
    // Camera table versions: when every camera has the same number of ring slots, version s holds all cameras
    // looking at slot s, and a frame switch of the whole batch (m3t_hip_cameras_select_slot) only moves the
    // table pointer the kernels receive: nothing is copied between two tracking steps.  Cameras on different
    // slots use the extra version behind them, rebuilt and staged per switch.
    const size_t n_cams = ctx->cameras.size();
    int n_versions = n_cams ? ctx->cameras[0]->n_slots : 0;
    int uniform_slot = n_cams ? ctx->cameras[0]->current : 0;
    for (auto& c : ctx->cameras) {
      if (c->n_slots != n_versions) n_versions = 0;
      if (c->current != uniform_slot) uniform_slot = -1;
    }
    const size_t version_bytes = n_cams * sizeof(CameraDev);
    if (n_versions > 4096 || version_bytes * size_t(n_versions + 1) > (size_t(32) << 20)) n_versions = 0;
    auto fill = [&](CameraDev* out, int slot /* -1: every camera's own current slot */) {
      for (size_t i = 0; i < n_cams; ++i) {
        const Camera& c = *ctx->cameras[i];
        CameraDev& d = out[i];
        d.image = c.frame(slot < 0 ? c.current : slot);
        d.pitch = c.pitch;
        d.width = c.intr.width;
        d.height = c.intr.height;
        d.fu = c.intr.fu; d.fv = c.intr.fv; d.ppu = c.intr.ppu; d.ppv = c.intr.ppv;
        d.depth_scale = c.depth_scale;
        d.slot = slot < 0 ? c.current : slot;
        std::memcpy(d.world2camera, c.world2camera, 64);
      }
    };
    const size_t bytes = version_bytes * size_t(n_versions + 1);
    if (ctx->d_cams.bytes < bytes) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(ctx->d_cams.alloc(bytes * 2));
      ctx->cams_dirty = true;
    }
    const bool rebuild = ctx->cams_dirty || ctx->tables_dirty;
    const bool custom = uniform_slot < 0 || n_versions == 0;
    if (n_cams && rebuild && n_versions > 0) {
      // all slot versions at once: rare (set-up, a camera pose change), so a plain blocking copy
      std::vector<CameraDev> all(n_cams * size_t(n_versions));
      for (int v = 0; v < n_versions; ++v) fill(all.data() + size_t(v) * n_cams, v);
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipMemcpy(ctx->d_cams.p, all.data(), all.size() * sizeof(CameraDev), hipMemcpyHostToDevice));
    }
    if (n_cams && custom) {
      // one table for cameras on different slots, staged through a small pinned ring: no host sync per switch
      if (ctx->cam_stage_bytes < version_bytes) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < Ctx::kStage; ++i) {
          if (ctx->cam_stage[i]) (void)hipHostFree(ctx->cam_stage[i]);
          HIPCHK(hipHostMalloc(&ctx->cam_stage[i], version_bytes * 2, hipHostMallocDefault));
          if (!ctx->cam_stage_done[i]) HIPCHK(hipEventCreateWithFlags(&ctx->cam_stage_done[i], hipEventDisableTiming));
        }
        ctx->cam_stage_bytes = version_bytes * 2;
      }
      const int stage = ctx->cam_stage_next;
      ctx->cam_stage_next = (stage + 1) % Ctx::kStage;
      HIPCHK(hipEventSynchronize(ctx->cam_stage_done[stage]));  // normally long complete
      fill(static_cast<CameraDev*>(ctx->cam_stage[stage]), -1);
      HIPCHK(hipMemcpyAsync(ctx->d_cams.as<uint8_t>() + size_t(n_versions) * version_bytes, ctx->cam_stage[stage],
                            version_bytes, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipEventRecord(ctx->cam_stage_done[stage], ctx->stream));
    }
    ctx->cams_active = ctx->d_cams.as<CameraDev>() + (custom ? size_t(n_versions) : size_t(uniform_slot)) * n_cams;
    ctx->cams_dirty = false;
    ctx->slots_dirty = false;
  }
  if (ctx->tables_dirty) {
    int rr = UploadRendererTables(ctx);
    if (rr) return rr;
    std::vector<RegionModDev> r(ctx->region_mods.size());
    for (size_t i = 0; i < r.size(); ++i) r[i] = ctx->region_mods[i]->dev;
    std::vector<DepthModDev> d(ctx->depth_mods.size());
    for (size_t i = 0; i < d.size(); ++i) d[i] = ctx->depth_mods[i]->dev;
    HIPCHK(ctx->d_region.alloc(std::max<size_t>(1, r.size()) * sizeof(RegionModDev)));
    HIPCHK(ctx->d_depth.alloc(std::max<size_t>(1, d.size()) * sizeof(DepthModDev)));
    if (!r.empty()) HIPCHK(hipMemcpy(ctx->d_region.p, r.data(), r.size() * sizeof(RegionModDev), hipMemcpyHostToDevice));
    if (!d.empty()) HIPCHK(hipMemcpy(ctx->d_depth.p, d.data(), d.size() * sizeof(DepthModDev), hipMemcpyHostToDevice));
    // kinematic structures (m3t_links.hip) as soon as one optimizer is more than a free rigid body
    ctx->tree_mode = false;
    for (auto& o : ctx->optimizers) {
      if (!ctx->links[o.link].simple || !ctx->links[o.link].children.empty() || !o.constraints.empty() ||
          !o.soft_constraints.empty())
        ctx->tree_mode = true;
      // a body seen by several cameras carries several modalities of one kind; the rigid table holds one region
      // and one depth modality per body, the link kernels sum any number (Link::CalculateGradientAndHessian,
      // link.cpp:184-193, in the order they were added)
      int n_region = 0, n_depth = 0;
      for (int mid : ctx->links[o.link].modalities) (ctx->modalities[mid].region ? n_region : n_depth) += 1;
      if (n_region > 1 || n_depth > 1) ctx->tree_mode = true;
    }
    if (ctx->tree_mode) {
      int r = UploadTreeTables(ctx);
      if (r) return r;
    }
    // optimizer table
    ctx->opt_table.clear();
    bool depth_table_changed = false;
    ctx->fused_possible = !ctx->optimizers.empty() && !ctx->tree_mode;
    std::vector<char> body_used(ctx->body_poses.size() / 16, 0);
    for (auto& o : ctx->optimizers) {
      if (ctx->tree_mode) break;
      const Link& l = ctx->links[o.link];
      RigidOptDev od{};
      od.body = l.body;
      od.region_modality = -1;
      od.depth_modality = -1;
      od.tikhonov_rotation = o.tr;
      od.tikhonov_translation = o.tt;
      for (int mid : l.modalities) {
        const ModalityRef& ref = ctx->modalities[mid];
        if (ref.region) od.region_modality = ref.index;
        else od.depth_modality = ref.index;
      }
      if (od.region_modality >= 0 && od.depth_modality >= 0) {
        const RegionMod& rmod = *ctx->region_mods[od.region_modality];
        DepthMod& dmod = *ctx->depth_mods[od.depth_modality];
        const Model& a = *ctx->region_models[rmod.model];
        const Model& b = *ctx->depth_models[dmod.model];
        const bool shared = a.h_orientations == b.h_orientations &&
                            std::memcmp(ctx->cameras[rmod.camera]->world2camera, ctx->cameras[dmod.camera]->world2camera, 64) == 0;
        if (dmod.dev.view_search_shared != (shared ? 1 : 0)) {
          dmod.dev.view_search_shared = shared ? 1 : 0;
          depth_table_changed = true;
        }
      }
      if (body_used[l.body]) ctx->fused_possible = false;
      body_used[l.body] = 1;
      ctx->opt_table.push_back(od);
    }
    // the fused kernel iterates optimizers; every modality must hang on exactly one of them
    size_t attached = 0;
    for (auto& o : ctx->optimizers) attached += ctx->links[o.link].modalities.size();
    if (attached != ctx->modalities.size()) ctx->fused_possible = false;
    // renderer-fed branches read other bodies' poses between the sub-steps: one launch per sub-step
    ctx->fused_per_search_possible = ctx->fused_possible && ctx->n_render_all > 0;
    if (ctx->n_render_all > 0) ctx->fused_possible = false;
    // the histogram update can ride in the tracking launch when every region modality sits alone on its
    // optimizer, owns its histograms, and the count table fits the LDS
    ctx->fuse_histogram_possible = ctx->fused_possible && !ctx->region_mods.empty();
    for (auto& m : ctx->region_mods)
      if (m->shared_histograms >= 0) ctx->fuse_histogram_possible = false;
    if (depth_table_changed) {
      for (size_t i = 0; i < d.size(); ++i) d[i] = ctx->depth_mods[i]->dev;
      HIPCHK(hipMemcpy(ctx->d_depth.p, d.data(), d.size() * sizeof(DepthModDev), hipMemcpyHostToDevice));
    }
    {
      int r = BuildRoiTables(ctx);  // (sets RigidOptDev::search_poses)
      if (r) return r;
    }
    HIPCHK(ctx->d_opts.alloc(std::max<size_t>(1, ctx->opt_table.size()) * sizeof(RigidOptDev)));
    if (!ctx->opt_table.empty())
      HIPCHK(hipMemcpy(ctx->d_opts.p, ctx->opt_table.data(), ctx->opt_table.size() * sizeof(RigidOptDev),
                       hipMemcpyHostToDevice));
    ComputeLayout(ctx);
    size_t max_lds = std::max(std::max(ctx->lds_track, ctx->lds_hist), ctx->lds_depth);
    REQUIRE(max_lds <= 160 * 1024, M3T_ERR_UNSUPPORTED,
            "per-object working set exceeds the 160 KB LDS of a CU");
    if (!ctx->hist_counts_in_lds) ctx->fuse_histogram_possible = false;
    // the compact kernel: the reference's default function / distribution lengths, unrolled scales, a thread per line
    ctx->compact_possible = ctx->fused_possible && ctx->lds_compact <= 80 * 1024;
    for (auto& m : ctx->region_mods) {
      if (m->p.function_length != 8 || m->p.distribution_length != 12 || m->p.n_lines_max > M3T_COMPACT_THREADS)
        ctx->compact_possible = false;
      for (int i = 0; i < m->p.n_scales; ++i)
        if (m->p.scales[i] < 1 || m->p.scales[i] > 9) ctx->compact_possible = false;
      // (the LDS pair table is laid out for ONE histogram size, each modality's own table, no shared ColorHistograms)
      if (m->p.n_histogram_bins != ctx->region_mods[0]->p.n_histogram_bins || m->shared_histograms >= 0)
        ctx->lds_compact_table = 0;
    }
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(tracking_step_compact_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->lds_compact)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(tracking_step_compact_wide_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->lds_compact)));
    if (ctx->lds_compact_table)
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(tracking_step_compact_table_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->lds_compact_table)));
    // the split kernel: free rigid bodies with their own histograms (the pair table is read from L2)
    ctx->split_possible = ctx->fused_possible;
    for (auto& m : ctx->region_mods)
      if (m->shared_histograms >= 0 || m->p.n_histogram_bins < 4) ctx->split_possible = false;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(tracking_step_split_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(std::max(ctx->lds_track, ctx->lds_hist))));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(tracking_step_split_render_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(std::max(ctx->lds_track, ctx->lds_hist))));
    for (const void* pair_kernel : {reinterpret_cast<const void*>(tracking_step_split_pair_kernel),
                                    reinterpret_cast<const void*>(tracking_step_pair_kernel),
                                    reinterpret_cast<const void*>(tracking_step_lds_pair_kernel)})
      HIPCHK(hipFuncSetAttribute(pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 int(std::max(ctx->lds_track, ctx->lds_hist))));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(tracking_step_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(std::max(ctx->lds_track, ctx->lds_hist))));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(tracking_step_lds_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(std::max(ctx->lds_track, ctx->lds_hist))));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(region_correspondence_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->lds_corr)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(region_correspondence_lds_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->lds_corr)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(region_gradient_hessian_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->lds_corr)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(region_histogram_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->lds_hist)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(depth_correspondence_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->lds_depth)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(depth_gradient_hessian_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, int(ctx->lds_depth)));
    ctx->tables_dirty = false;
  }
  size_t n_bodies = ctx->body_poses.size() / 16;
  if (ctx->pose_capacity < n_bodies) {
    // keep device poses if they are newer than the host mirror
    if (!ctx->poses_dirty_host && ctx->pose_capacity > 0) {
      std::vector<float> tmp(ctx->pose_capacity * 16);
      HIPCHK(hipMemcpy(tmp.data(), ctx->d_poses.p, tmp.size() * 4, hipMemcpyDeviceToHost));
      std::memcpy(ctx->body_poses.data(), tmp.data(), tmp.size() * 4);
    }
    HIPCHK(ctx->d_poses.alloc(std::max<size_t>(1, n_bodies) * 64 * 2));
    ctx->pose_capacity = ctx->d_poses.bytes / 64;
    ctx->poses_dirty_host = true;
  }
  if (ctx->poses_dirty_host && n_bodies) {
    HIPCHK(hipMemcpyAsync(ctx->d_poses.p, ctx->body_poses.data(), n_bodies * 64, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->poses_dirty_host = false;
  }
  return M3T_OK;
}

int CheckImages(Ctx* ctx) {
  for (auto& m : ctx->region_mods) {
    const Camera& c = *ctx->cameras[m->camera];
    REQUIRE(c.has_image[c.current], M3T_ERR_NOT_SET_UP, "Set up color camera first (no image uploaded)");
    if (m->p.measure_occlusions) {
      const Camera& d = *ctx->cameras[m->depth_camera];
      REQUIRE(d.has_image[d.current], M3T_ERR_NOT_SET_UP, "Set up depth camera first (no image uploaded)");
    }
  }
  for (auto& m : ctx->depth_mods) {
    const Camera& d = *ctx->cameras[m->camera];
    REQUIRE(d.has_image[d.current], M3T_ERR_NOT_SET_UP, "Set up depth camera first (no image uploaded)");
  }
  return M3T_OK;
}

struct ScopedKernelTimer {
  Ctx* ctx;
  int which;
  hipEvent_t a = nullptr, b = nullptr;
  ScopedKernelTimer(Ctx* c, int w) : ctx(c), which(w) {
    if (ctx->timing_region) ctx->kernel_launches[w] += 1;  // (no event between the launches)
    if (!ctx->timing) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
    (void)hipEventRecord(a, ctx->stream);
  }
  ~ScopedKernelTimer() {
    if (!a) return;
    (void)hipEventRecord(b, ctx->stream);
    ctx->pending.push_back({a, b, which});
  }
};

int LaunchHistogram(Ctx* ctx, int iteration, bool initialize, bool behind_guarded_step = false) {
  int n = int(ctx->region_mods.size());
  if (n == 0) return M3T_OK;
  ScopedKernelTimer timer(ctx, 1);
  const bool guarded = behind_guarded_step && ctx->d_roi_opt_of_region.p != nullptr && !ctx->opt_table.empty();
  hipLaunchKernelGGL(region_histogram_kernel, dim3(n), dim3(M3T_BLOCK_THREADS), ctx->lds_hist, ctx->stream,
                     ctx->d_region.as<RegionModDev>(), ctx->cams_active, ctx->d_poses.as<float>(), iteration,
                     initialize ? 1 : 0, ctx->hist_counts_in_lds ? 1 : 0,
                     guarded ? ctx->d_opts.as<RigidOptDev>() : (const RigidOptDev*)nullptr,
                     guarded ? ctx->d_roi_opt_of_region.as<int>() : (const int*)nullptr, ctx->roi_n_poses);
  if (!ctx->shared_histograms.empty())  // Initialize / UpdateHistograms of the shared objects, tracker.cpp:441-443,513-515
    hipLaunchKernelGGL(shared_histogram_finish_kernel, dim3(unsigned(ctx->shared_histograms.size())),
                       dim3(M3T_BLOCK_THREADS), 0, ctx->stream, ctx->d_shared_histograms.as<SharedHistogramsDev>(),
                       initialize ? 1 : 0);
  HIPCHK(hipGetLastError());
  return M3T_OK;
}

int LaunchCorrespondences(Ctx* ctx, int iteration, int corr_iteration) {
  int nr = int(ctx->region_mods.size()), nd = int(ctx->depth_mods.size());
  if (nr) {
    hipLaunchKernelGGL(ctx->layout.off_hist >= 0 ? region_correspondence_lds_kernel : region_correspondence_kernel,
                       dim3(nr), dim3(M3T_BLOCK_THREADS), ctx->lds_corr, ctx->stream,
                       ctx->d_region.as<RegionModDev>(), ctx->cams_active, ctx->d_poses.as<float>(),
                       ctx->layout, iteration, corr_iteration);
    HIPCHK(hipGetLastError());
  }
  if (nd) {
    hipLaunchKernelGGL(depth_correspondence_kernel, dim3(nd), dim3(M3T_BLOCK_THREADS), ctx->lds_depth, ctx->stream,
                       ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(),
                       ctx->np_max, iteration, corr_iteration);
    HIPCHK(hipGetLastError());
  }
  return M3T_OK;
}

int LaunchGradientHessian(Ctx* ctx, int corr_iteration, int opt_iteration) {
  int nr = int(ctx->region_mods.size()), nd = int(ctx->depth_mods.size());
  if (nr) {
    hipLaunchKernelGGL(region_gradient_hessian_kernel, dim3(nr), dim3(M3T_BLOCK_THREADS), ctx->lds_corr, ctx->stream,
                       ctx->d_region.as<RegionModDev>(), ctx->cams_active, ctx->d_poses.as<float>(),
                       ctx->layout, corr_iteration, opt_iteration);
    HIPCHK(hipGetLastError());
  }
  if (nd) {
    hipLaunchKernelGGL(depth_gradient_hessian_kernel, dim3(nd), dim3(M3T_BLOCK_THREADS), ctx->lds_depth, ctx->stream,
                       ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(),
                       ctx->np_max, corr_iteration);
    HIPCHK(hipGetLastError());
  }
  return M3T_OK;
}

int LaunchProject(Ctx* ctx) {
  int n = int(ctx->optimizers.size());
  if (n == 0) return M3T_OK;
  hipLaunchKernelGGL(links_project_kernel, dim3(n), dim3(64), ctx->tree_lds, ctx->stream,
                     ctx->d_treeopts.as<TreeOptDev>(), n, ctx->d_poses.as<float>(), ctx->tree_lds ? 1 : 0);
  HIPCHK(hipGetLastError());
  ctx->partial_ready = true;
  return M3T_OK;
}
int LaunchSolve(Ctx* ctx, bool zero_theta) {
  int n = int(ctx->optimizers.size());
  if (n == 0) return M3T_OK;
  hipLaunchKernelGGL(links_solve_kernel, dim3(n), dim3(64), ctx->tree_lds, ctx->stream,
                     ctx->d_treeopts.as<TreeOptDev>(), n, ctx->d_poses.as<float>(), zero_theta ? 1 : 0,
                     ctx->tree_lds ? 1 : 0);
  HIPCHK(hipGetLastError());
  ctx->links_device_newer = true;
  ctx->partial_ready = false;
  return M3T_OK;
}

// The two halves around the exchange of a structure spread over processes (m3t_links.hip): the link sums of this
// process's modalities, and everything else of Optimizer::CalculateOptimization from the summed link sums
int LaunchGather(Ctx* ctx) {
  int n = int(ctx->optimizers.size());
  if (n == 0) return M3T_OK;
  hipLaunchKernelGGL(links_gather_kernel, dim3(n), dim3(64), 0, ctx->stream, ctx->d_treeopts.as<TreeOptDev>(), n,
                     ctx->d_link_sums.as<float>(), ctx->d_link_first.as<int>());
  HIPCHK(hipGetLastError());
  ctx->partial_ready = true;
  return M3T_OK;
}
int LaunchSolveSums(Ctx* ctx) {
  int n = int(ctx->optimizers.size());
  if (n == 0) return M3T_OK;
  hipLaunchKernelGGL(links_solve_sums_kernel, dim3(n), dim3(64), ctx->tree_lds, ctx->stream,
                     ctx->d_treeopts.as<TreeOptDev>(), n, ctx->d_poses.as<float>(), 0, ctx->tree_lds ? 1 : 0,
                     ctx->d_link_sums.as<float>(), ctx->d_link_first.as<int>());
  HIPCHK(hipGetLastError());
  ctx->links_device_newer = true;
  ctx->partial_ready = false;
  return M3T_OK;
}

// sum of the stacked link sums ([links of all structures][6 + 36]) over the ranks of the communicator: ONE
// ncclAllReduce on the context's stream, in place.  Exact: every link's modalities live on one rank, the others add
// +0.0 (m3t_links.hip, links_gather_kernel)
int AllReducePartial(Ctx* ctx, float* buffer = nullptr) {
  REQUIRE(ctx->partial_ready, M3T_ERR_NOT_SET_UP, "calculate_optimization_begin has to be called first");
  REQUIRE(ctx->Distributed(), M3T_ERR_NOT_SET_UP,
          "no communicator: m3t_hip_comm_init_rank / m3t_hip_comm_set / m3t_hip_comm_set_reduce_callback first");
  if (!buffer) buffer = ctx->d_link_sums.as<float>();
  if (ctx->reduce_fn) {  // the host's transport: leaves the sum over its ranks in the buffer, in stream order
    const int rc = ctx->reduce_fn(ctx->reduce_user, buffer, ctx->link_sums_count, static_cast<void*>(ctx->stream));
    if (rc != 0) return Fail(ctx, M3T_ERR_DEVICE, "the reduce callback failed with code " + std::to_string(rc));
    ++ctx->allreduce_calls;
    return M3T_OK;
  }
  const ncclResult_t rc =
      g_rccl.AllReduce(buffer, buffer, ctx->link_sums_count, ncclFloat, ncclSum, ctx->comm, ctx->stream);
  if (rc != ncclSuccess)
    return Fail(ctx, M3T_ERR_DEVICE,
                std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
  ++ctx->allreduce_calls;
  return M3T_OK;
}

// Optimizer::CalculateOptimization for every optimizer.  With a communicator set the structures span GPUs
// (SURVEY 8e): link sums -> ONE all-reduce of them -> project + solve, on every path that reaches this function
// (m3t_hip_calculate_optimization and the sub-step loop of m3t_hip_execute_tracking_step alike).
int LaunchOptimization(Ctx* ctx) {
  if (ctx->Distributed()) {
    if (!ctx->tree_mode) {  // rigid bodies only: force the general path so that the sums exist
      ctx->tree_mode = true;
      ctx->fused_possible = false;
      int r = UploadTreeTables(ctx);
      if (r) return r;
    }
    int r = LaunchGather(ctx);
    if (r) return r;
    if ((r = AllReducePartial(ctx))) return r;
    return LaunchSolveSums(ctx);
  }
  if (ctx->tree_mode) {
    int r = LaunchProject(ctx);
    if (r) return r;
    return LaunchSolve(ctx, false);
  }
  int n = int(ctx->opt_table.size());
  if (n == 0) return M3T_OK;
  hipLaunchKernelGGL(rigid_optimize_kernel, dim3(n), dim3(64), 0, ctx->stream,
                     ctx->d_opts.as<RigidOptDev>(), n, ctx->d_region.as<RegionModDev>(),
                     ctx->d_depth.as<DepthModDev>(), ctx->d_poses.as<float>());
  HIPCHK(hipGetLastError());
  return M3T_OK;
}

// LDS of one workgroup of tracking_step_tree_kernel: the tracking carve-up (or the histogram update's count table,
// whichever is larger) with the structure block behind it
size_t TreeStepLds(Ctx* ctx, bool fused_histogram) {
  const size_t front = fused_histogram ? std::max(ctx->lds_track, ctx->lds_hist) : ctx->lds_track;
  return (front + 15) / 16 * 16 + ctx->tree_block_floats * 4;
}
// Can this step run as ONE launch of tracking_step_tree_kernel?  Needs: kinematic structures whose modalities sit on
// links (at most one region and one depth modality per link), no renderer-fed branches, no communicator, no shared
// histogram objects, line / point state not requested (fused mode 1), fewer than 63 Newton steps per frame, the
// structure copy next to the tracking carve-up in LDS, and every workgroup of the grid resident at once.
bool TreeStepFused(Ctx* ctx) {
  // (m3t_hip_set_object_split(ctx, 0) = "this context shares its GPU": no launch whose workgroups wait for each other)
  if (!(ctx->tree_mode && ctx->tree_fused_possible && ctx->fused_mode == 1 && !ctx->Distributed() && ctx->n_render_all == 0 &&
        ctx->shared_histograms.empty() && ctx->n_treesteps > 0 && ctx->split_enabled &&
        !std::getenv("M3T_HIP_NO_TREE_FUSION")))
    return false;
  if (ctx->layout.off_hist >= 0) return false;  // (the LDS-staged pair table belongs to tracking_step_lds_kernel)
  if (ctx->n_corr_iterations * ctx->n_update_iterations >= 63) return false;
  const bool fused_histogram = !ctx->region_mods.empty() && ctx->hist_counts_in_lds;
  const size_t lds = TreeStepLds(ctx, fused_histogram);
  if (lds > size_t(160) * 1024) return false;
  auto kernel = ctx->tree_constrained ? tracking_step_tree_constrained_kernel : tracking_step_tree_kernel;
  if (ctx->tree_lds_attribute != lds || ctx->tree_lds_kernel != reinterpret_cast<const void*>(kernel)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            int(lds)) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    ctx->tree_lds_attribute = lds;
    ctx->tree_lds_kernel = reinterpret_cast<const void*>(kernel);
  }
  int resident = ResidentBlocks(ctx, kernel, M3T_BLOCK_THREADS, lds);
  resident = std::min(resident, int(size_t(160) * 1024 / lds));
  return resident >= 1 && ctx->n_treesteps <= ctx->compute_cus * resident;
}

// A structure spread over processes (a communicator is set): can the step run as ONE launch + ONE all-reduce per
// Newton step (tracking_step_tree_segment_kernel)?  The conditions of the one-launch step, except that no workgroup
// waits for another one inside a launch: no co-residency needed, any grid.
bool TreeStepSegmented(Ctx* ctx) {
  if (!(ctx->tree_mode && ctx->tree_fused_possible && ctx->fused_mode == 1 && ctx->Distributed() && ctx->n_render_all == 0 &&
        ctx->shared_histograms.empty() && ctx->n_treesteps > 0 && !std::getenv("M3T_HIP_NO_TREE_SEGMENTS")))
    return false;
  // a structure's solve is done by the workgroups of its tracked links: one whose modalities all live on other ranks
  // has none here, and still has to be solved from the summed link sums -> the per-sub-step launches (grid = structures)
  if (ctx->tree_untracked_structure) return false;
  if (ctx->layout.off_hist >= 0) return false;
  const bool fused_histogram = !ctx->region_mods.empty() && ctx->hist_counts_in_lds;
  const size_t lds = TreeStepLds(ctx, fused_histogram);
  if (lds > size_t(160) * 1024) return false;
  auto kernel = ctx->tree_constrained ? tracking_step_tree_segment_constrained_kernel : tracking_step_tree_segment_kernel;
  if (ctx->tree_segment_lds_attribute != lds || ctx->tree_segment_lds_kernel != reinterpret_cast<const void*>(kernel)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            int(lds)) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    ctx->tree_segment_lds_attribute = lds;
    ctx->tree_segment_lds_kernel = reinterpret_cast<const void*>(kernel);
  }
  return true;
}

int Prepare(Ctx* ctx, bool need_images) {
  if (ctx->async_ingest) ctx->untracked_launches = true;  // execute_tracking_step undoes this for itself
  if (need_images) {
    int r = CheckImages(ctx);
    if (r) return r;
  }
  return UploadTables(ctx);
}

RegionMod* GetRegion(Ctx* ctx, int id) {
  if (id < 0 || id >= int(ctx->modalities.size()) || !ctx->modalities[id].region) return nullptr;
  return ctx->region_mods[ctx->modalities[id].index].get();
}
DepthMod* GetDepth(Ctx* ctx, int id) {
  if (id < 0 || id >= int(ctx->modalities.size()) || ctx->modalities[id].region) return nullptr;
  return ctx->depth_mods[ctx->modalities[id].index].get();
}

// The context's streams.  With CUs reserved for ingest (m3t_hip_reserve_ingest_cus) the compute stream runs on the
// CUs of mask bits [0, CUs - n) and copy stream 0 -- the one the ROI pull kernel is launched on -- on bits
// [CUs - n, CUs).  Bit i of a mask belongs to XCD i mod 8 (amdkfd deals the bits round-robin; tools/ubench_cumask.hip
// counts 30 / 2 CUs per XCD for 240 / 16 bits), inside an XCD the highest bits are the last CU of each shader engine.
hipError_t CreateMaskedStream(Ctx* ctx, hipStream_t* stream, bool ingest_side) {
  if (ctx->ingest_cus <= 0) return hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
  const int cus = ctx->prop.multiProcessorCount, split = cus - ctx->ingest_cus;
  std::vector<uint32_t> mask(size_t(cus + 31) / 32, 0u);
  for (int i = ingest_side ? split : 0; i < (ingest_side ? cus : split); ++i) mask[size_t(i) / 32] |= 1u << (i % 32);
  return hipExtStreamCreateWithCUMask(stream, uint32_t(mask.size()), mask.data());
}
hipError_t CreateCopyStreams(Ctx* ctx) {
  for (int i = 0; i < Ctx::kCopyStreams; ++i) {
    const hipError_t e = i == 0 ? CreateMaskedStream(ctx, &ctx->copy_stream[i], true)
                                : hipStreamCreateWithFlags(&ctx->copy_stream[i], hipStreamNonBlocking);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace

#define CHECK_CTX() \
  if (!ctx) return M3T_ERR_INVALID_ARGUMENT

extern "C" {

int m3t_hip_create(m3t_hip_context** out, int device_id) {
  if (!out) return M3T_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = std::string("no usable HIP device: ") + (e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return M3T_ERR_DEVICE;
  }
  if (device_id < 0 || device_id >= n) {
    g_create_error = "device id out of range";
    return M3T_ERR_INVALID_ARGUMENT;
  }
  auto ctx = std::make_unique<m3t_hip_context>();
  ctx->device = device_id;
  if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&ctx->prop, device_id)) != hipSuccess ||
      (e = CreateMaskedStream(ctx.get(), &ctx->stream, false)) != hipSuccess) {
    g_create_error = std::string("device initialisation failed: ") + hipGetErrorString(e);
    return M3T_ERR_DEVICE;
  }
  ctx->compute_cus = ctx->prop.multiProcessorCount;
  *out = ctx.release();
  return M3T_OK;
}

void m3t_hip_destroy(m3t_hip_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
  }
  for (auto& cs : ctx->copy_stream) {
    if (!cs) continue;
    (void)hipStreamSynchronize(cs);
    (void)hipStreamDestroy(cs);
  }
  for (void* p : ctx->registered) (void)hipHostUnregister(p);
  for (auto& e : ctx->copies_done)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : ctx->step_done)
    if (e) (void)hipEventDestroy(e);
  for (int i = 0; i < m3t_hip_context::kStage; ++i) {
    if (ctx->cam_stage[i]) (void)hipHostFree(ctx->cam_stage[i]);
    if (ctx->cam_stage_done[i]) (void)hipEventDestroy(ctx->cam_stage_done[i]);
  }
  for (auto& c : ctx->cameras)
    for (auto& e : c->slot_copied)
      if (e) (void)hipEventDestroy(e);
  if (ctx->split_abort_host) (void)hipHostFree(ctx->split_abort_host);
  if (ctx->table_overflow_host) (void)hipHostFree(ctx->table_overflow_host);
  if (ctx->roi_miss_host) (void)hipHostFree(ctx->roi_miss_host);
  for (auto& e : ctx->roi_snapshot_done)
    if (e) (void)hipEventDestroy(e);
  if (ctx->region_a) (void)hipEventDestroy(ctx->region_a);
  if (ctx->region_b) (void)hipEventDestroy(ctx->region_b);
  if (ctx->comm && ctx->comm_owned && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(ctx->comm);
  delete ctx;
}

const char* m3t_hip_last_error(m3t_hip_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int m3t_hip_device_info(m3t_hip_context* ctx, char* name, size_t cap, int* cus, size_t* mem) {
  CHECK_CTX();
  if (name && cap) {
    std::snprintf(name, cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
  }
  if (cus) *cus = ctx->prop.multiProcessorCount;
  if (mem) *mem = ctx->prop.totalGlobalMem;
  return M3T_OK;
}
int m3t_hip_get_stream(m3t_hip_context* ctx, void** s) {
  CHECK_CTX();
  REQUIRE(s, M3T_ERR_INVALID_ARGUMENT, "null out pointer");
  *s = ctx->stream;
  return M3T_OK;
}

// ---- models -------------------------------------------------------------------
int m3t_hip_region_model_create(m3t_hip_context* ctx, const m3t_region_model_desc* d) {
  CHECK_CTX();
  REQUIRE(d, M3T_ERR_INVALID_ARGUMENT, "null desc");
  HIPCHK(hipSetDevice(ctx->device));
  return CreateModel(ctx, true, d->n_views, d->n_points, d->data_points, d->orientations, d->contour_lengths,
                     d->stride_depth_offset, d->max_radius_depth_offset);
}
int m3t_hip_depth_model_create(m3t_hip_context* ctx, const m3t_depth_model_desc* d) {
  CHECK_CTX();
  REQUIRE(d, M3T_ERR_INVALID_ARGUMENT, "null desc");
  HIPCHK(hipSetDevice(ctx->device));
  return CreateModel(ctx, false, d->n_views, d->n_points, d->data_points, d->orientations, d->surface_areas,
                     d->stride_depth_offset, d->max_radius_depth_offset);
}
int m3t_hip_region_model_load(m3t_hip_context* ctx, const char* path) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  return LoadModel(ctx, path, true);
}
int m3t_hip_depth_model_load(m3t_hip_context* ctx, const char* path) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  return LoadModel(ctx, path, false);
}
static int ModelInfo(m3t_hip_context* ctx, bool region, int id, int* nv, int* np, float* me) {
  auto& vec = region ? ctx->region_models : ctx->depth_models;
  REQUIRE(id >= 0 && id < int(vec.size()), M3T_ERR_INVALID_ARGUMENT, "bad model id");
  if (nv) *nv = vec[id]->n_views;
  if (np) *np = vec[id]->n_points;
  if (me) *me = vec[id]->max_extent;
  return M3T_OK;
}
int m3t_hip_region_model_info(m3t_hip_context* ctx, int id, int* nv, int* np, float* me) {
  CHECK_CTX();
  return ModelInfo(ctx, true, id, nv, np, me);
}
int m3t_hip_depth_model_info(m3t_hip_context* ctx, int id, int* nv, int* np, float* me) {
  CHECK_CTX();
  return ModelInfo(ctx, false, id, nv, np, me);
}

}  // extern "C"

// single-block helper kernel for the stand-alone GetClosestView entry point
extern "C" __global__ void __launch_bounds__(M3T_BLOCK_THREADS)
closest_view_kernel(const float4* orientations, int n_views, const float* body2camera, int* out) {
  __shared__ float misc[256];
  const Affine b2c = load_pose(body2camera);
  int v = closest_view((G<v4f>)orientations, n_views, b2c, misc);
  if (threadIdx.x == 0) *out = v;
}

extern "C" {

static int ClosestView(m3t_hip_context* ctx, bool region, int id, const float* pose, int* view) {
  auto& vec = region ? ctx->region_models : ctx->depth_models;
  REQUIRE(id >= 0 && id < int(vec.size()) && pose && view, M3T_ERR_INVALID_ARGUMENT, "bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->d_scratch_view.bytes < 128) HIPCHK(ctx->d_scratch_view.alloc(128));
  float* d_pose = ctx->d_scratch_view.as<float>();
  int* d_out = reinterpret_cast<int*>(d_pose + 16);
  HIPCHK(hipMemcpyAsync(d_pose, pose, 64, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(closest_view_kernel, dim3(1), dim3(M3T_BLOCK_THREADS), 0, ctx->stream,
                     vec[id]->orientations4.as<float4>(), vec[id]->n_views, d_pose, d_out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(view, d_out, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return M3T_OK;
}
int m3t_hip_region_model_closest_view(m3t_hip_context* ctx, int id, const float pose[16], int* view) {
  CHECK_CTX();
  return ClosestView(ctx, true, id, pose, view);
}
int m3t_hip_depth_model_closest_view(m3t_hip_context* ctx, int id, const float pose[16], int* view) {
  CHECK_CTX();
  return ClosestView(ctx, false, id, pose, view);
}

// ---- cameras --------------------------------------------------------------------
int m3t_hip_color_camera_create(m3t_hip_context* ctx, const m3t_intrinsics* i, const float w2c[16]) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  return CreateCamera(ctx, i, w2c, false, 0.0f);
}
int m3t_hip_depth_camera_create(m3t_hip_context* ctx, const m3t_intrinsics* i, const float w2c[16], float depth_scale) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  return CreateCamera(ctx, i, w2c, true, depth_scale);
}
int m3t_hip_camera_upload(m3t_hip_context* ctx, int id, const void* pixels, size_t row_step) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  REQUIRE(id >= 0 && id < int(ctx->cameras.size()), M3T_ERR_INVALID_ARGUMENT, "bad camera id");
  return UploadFrame(ctx, id, ctx->cameras[id]->current, pixels, row_step);
}
int m3t_hip_camera_set_world2camera_pose(m3t_hip_context* ctx, int id, const float w2c[16]) {
  CHECK_CTX();
  REQUIRE(id >= 0 && id < int(ctx->cameras.size()) && w2c, M3T_ERR_INVALID_ARGUMENT, "bad camera id");
  std::memcpy(ctx->cameras[id]->world2camera, w2c, 64);
  ctx->cams_dirty = true;
  ctx->tables_dirty = true;  // (whether a body's two modalities share their view search depends on the camera poses)
  return M3T_OK;
}
int m3t_hip_camera_set_ring(m3t_hip_context* ctx, int id, int n_slots) {
  CHECK_CTX();
  REQUIRE(id >= 0 && id < int(ctx->cameras.size()) && n_slots >= 1, M3T_ERR_INVALID_ARGUMENT, "bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  Camera& c = *ctx->cameras[id];
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(c.ring.alloc(c.frame_bytes * size_t(n_slots) + 64));
  c.frames = c.ring.as<uint8_t>();
  c.slot_stride = c.frame_bytes;
  c.slab = -1;
  c.n_slots = n_slots;
  c.current = 0;
  c.has_image.assign(n_slots, false);
  c.last_read_step.assign(n_slots, -1);
  c.slot_is_roi.assign(n_slots, false);
  if (ctx->roi_enabled) ctx->tables_dirty = true;  // (the rectangle table has a row per ring slot)
  ctx->cams_dirty = true;
  return M3T_OK;
}
int m3t_hip_camera_upload_slot(m3t_hip_context* ctx, int id, int slot, const void* pixels, size_t row_step) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  return UploadFrame(ctx, id, slot, pixels, row_step);
}
int m3t_hip_host_register(m3t_hip_context* ctx, void* ptr, size_t bytes) {
  CHECK_CTX();
  REQUIRE(ptr && bytes, M3T_ERR_INVALID_ARGUMENT, "null buffer");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  ctx->registered.push_back(ptr);
  return M3T_OK;
}
int m3t_hip_host_unregister(m3t_hip_context* ctx, void* ptr) {
  CHECK_CTX();
  auto it = std::find(ctx->registered.begin(), ctx->registered.end(), ptr);
  REQUIRE(it != ctx->registered.end(), M3T_ERR_INVALID_ARGUMENT, "buffer was not registered with this context");
  HIPCHK(hipSetDevice(ctx->device));
  for (auto& cs : ctx->copy_stream)
    if (cs) HIPCHK(hipStreamSynchronize(cs));
  HIPCHK(hipHostUnregister(ptr));
  ctx->registered.erase(it);
  return M3T_OK;
}
int m3t_hip_camera_upload_slot_async(m3t_hip_context* ctx, int id, int slot, const void* pixels, size_t row_step) {
  CHECK_CTX();
  REQUIRE(id >= 0 && id < int(ctx->cameras.size()) && pixels, M3T_ERR_INVALID_ARGUMENT, "bad camera id");
  Camera& c = *ctx->cameras[id];
  REQUIRE(slot >= 0 && slot < c.n_slots, M3T_ERR_INVALID_ARGUMENT, "bad frame slot");
  size_t row = size_t(c.intr.width) * (c.is_depth ? 2 : 3);
  REQUIRE(row_step >= row, M3T_ERR_INVALID_ARGUMENT, "row_step smaller than one image row");
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->async_ingest) {
    HIPCHK(CreateCopyStreams(ctx));
    for (auto& e : ctx->copies_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : ctx->step_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->async_ingest = true;
    ctx->untracked_launches = true;  // whatever ran before was not tracked by step events
  }
  const int cs = id % Ctx::kCopyStreams;
  if (ctx->untracked_launches) {
    // sub-step launches (or anything before the first asynchronous upload) carry no step event
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->untracked_launches = false;
  } else if (c.last_read_step[slot] > ctx->copy_waited_step[cs]) {
    // overwrite only after the last step that read this slot (a recycled event is a later step: still
    // safe); a copy stream is in order, so one wait per step covers all of its cameras
    HIPCHK(hipStreamWaitEvent(ctx->copy_stream[cs], ctx->step_done[c.last_read_step[slot] % Ctx::kStepEvents], 0));
    ctx->copy_waited_step[cs] = c.last_read_step[slot];
  }
  uint8_t* dst = c.frame(slot);
  if (row_step == c.pitch)  // contiguous on both sides: one linear DMA transfer
    HIPCHK(hipMemcpyAsync(dst, pixels, c.frame_bytes - (c.pitch - row), hipMemcpyHostToDevice, ctx->copy_stream[cs]));
  else
    HIPCHK(hipMemcpy2DAsync(dst, c.pitch, pixels, row_step, row, c.intr.height, hipMemcpyHostToDevice,
                            ctx->copy_stream[cs]));
  c.has_image[slot] = true;
  RoiMarkWholeFrame(ctx, id, slot, ctx->copy_stream[cs]);
  if (c.slot_copied.size() < size_t(c.n_slots)) c.slot_copied.resize(size_t(c.n_slots), nullptr);
  if (!c.slot_copied[slot]) HIPCHK(hipEventCreateWithFlags(&c.slot_copied[slot], hipEventDisableTiming));
  HIPCHK(hipEventRecord(c.slot_copied[slot], ctx->copy_stream[cs]));
  ctx->copies_pending |= 1u << cs;
  return M3T_OK;
}
// Wait until the last upload into (camera, slot) -- camera_upload_slot_async, a batch upload, a rectangle pull -- has
// left its host buffer, and for nothing else: the copies of other cameras and of this camera's other slots stay in
// flight (m3t_hip_ingest_sync waits for all).  A rectangle slot also waits for the step that read it (its repair
// reads the host block once more).
int m3t_hip_camera_slot_sync(m3t_hip_context* ctx, int id, int slot) {
  CHECK_CTX();
  REQUIRE(id >= 0 && id < int(ctx->cameras.size()), M3T_ERR_INVALID_ARGUMENT, "bad camera id");
  Camera& c = *ctx->cameras[id];
  REQUIRE(slot >= 0 && slot < c.n_slots, M3T_ERR_INVALID_ARGUMENT, "bad frame slot");
  HIPCHK(hipSetDevice(ctx->device));
  if (c.slot_is_roi[slot]) {
    // a rectangle went into the slot (m3t_hip_cameras_upload_batch_roi_async): the host block is read by the pull and,
    // should a body outrun its rectangle, again by the repair of the step that reads the slot -- both must be over
    hipEvent_t pulled = nullptr;
    if (c.slab >= 0 && size_t(slot) < ctx->slabs[size_t(c.slab)]->slot_copied.size())
      pulled = ctx->slabs[size_t(c.slab)]->slot_copied[slot];
    if (pulled) HIPCHK(hipEventSynchronize(pulled));
    else if (ctx->async_ingest && ctx->copy_stream[0]) HIPCHK(hipStreamSynchronize(ctx->copy_stream[0]));
    if (c.last_read_step[slot] >= 0 && c.last_read_step[slot] + Ctx::kStepEvents > ctx->step_counter)
      HIPCHK(hipEventSynchronize(ctx->step_done[c.last_read_step[slot] % Ctx::kStepEvents]));
    else if (c.last_read_step[slot] >= 0)
      HIPCHK(hipStreamSynchronize(ctx->stream));
    return M3T_OK;
  }
  // (a camera may have been fed both ways: its own uploads and batch uploads of its slab; an event that has long
  // completed costs nothing to wait for)
  if (c.slab >= 0 && size_t(slot) < ctx->slabs[size_t(c.slab)]->slot_copied.size() &&
      ctx->slabs[size_t(c.slab)]->slot_copied[slot])
    HIPCHK(hipEventSynchronize(ctx->slabs[size_t(c.slab)]->slot_copied[slot]));
  if (size_t(slot) >= c.slot_copied.size() || !c.slot_copied[slot]) return M3T_OK;  // nothing (else) was enqueued
  HIPCHK(hipEventSynchronize(c.slot_copied[slot]));
  return M3T_OK;
}
// One frame ring for a group of cameras of equal geometry: [slot][camera][frame], so that slot s of the whole group is
// one contiguous block of device memory and a batch-frame arrives as ONE transfer (m3t_hip_cameras_upload_batch_async).
int m3t_hip_cameras_set_ring(m3t_hip_context* ctx, const int* ids, int n, int n_slots) {
  CHECK_CTX();
  REQUIRE(ids && n >= 1 && n_slots >= 1, M3T_ERR_INVALID_ARGUMENT, "bad arguments");
  for (int i = 0; i < n; ++i) {
    REQUIRE(ids[i] >= 0 && ids[i] < int(ctx->cameras.size()), M3T_ERR_INVALID_ARGUMENT, "bad camera id");
    const Camera& a = *ctx->cameras[ids[0]];
    const Camera& b = *ctx->cameras[ids[i]];
    // (the row pitch is rounded up to 64 bytes, so equal pitch does not imply equal width: the one-block upload
    // copies every camera's rows with camera 0's row length)
    REQUIRE(a.frame_bytes == b.frame_bytes && a.pitch == b.pitch && a.is_depth == b.is_depth &&
                a.intr.width == b.intr.width && a.intr.height == b.intr.height,
            M3T_ERR_INVALID_ARGUMENT, "the cameras of a shared ring must have the same image geometry (width, height, type)");
    for (int j = 0; j < i; ++j) REQUIRE(ids[j] != ids[i], M3T_ERR_INVALID_ARGUMENT, "camera listed twice");
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  auto slab = std::make_unique<FrameSlab>();
  const size_t frame_bytes = ctx->cameras[ids[0]]->frame_bytes;
  HIPCHK(slab->mem.alloc(frame_bytes * size_t(n) * size_t(n_slots) + 64));
  slab->cameras.assign(ids, ids + n);
  const int slab_id = int(ctx->slabs.size());
  for (int i = 0; i < n; ++i) {
    Camera& c = *ctx->cameras[ids[i]];
    c.ring.release();
    c.slab = slab_id;
    c.slab_index = i;
    c.frames = slab->mem.as<uint8_t>() + size_t(i) * frame_bytes;
    c.slot_stride = frame_bytes * size_t(n);
    c.n_slots = n_slots;
    c.current = 0;
    c.has_image.assign(n_slots, false);
    c.last_read_step.assign(n_slots, -1);
    c.slot_is_roi.assign(n_slots, false);
    if (ctx->roi_enabled) ctx->tables_dirty = true;  // (the rectangle table has a row per ring slot)
  }
  ctx->slabs.push_back(std::move(slab));
  ctx->cams_dirty = true;
  return M3T_OK;
}
// The frames of n cameras for one ring slot, frame i at base + i * camera_stride, enqueued from here (what a capture
// process hands over per batch-frame; replaces n blocking cv::Mat hand-overs, loader_camera.cpp:76-98).  Cameras
// that share a ring (m3t_hip_cameras_set_ring, listed in its order) and a host block in the ring's layout
// (camera_stride = height * row_step, row_step = the device pitch) make it ONE linear DMA transfer.
int m3t_hip_cameras_upload_batch_async(m3t_hip_context* ctx, const int* ids, int n, int slot, const void* base,
                                       size_t camera_stride, size_t row_step) {
  CHECK_CTX();
  REQUIRE(ids && n >= 1 && base, M3T_ERR_INVALID_ARGUMENT, "bad arguments");
  bool one_block = true;
  for (int i = 0; i < n; ++i) {
    REQUIRE(ids[i] >= 0 && ids[i] < int(ctx->cameras.size()), M3T_ERR_INVALID_ARGUMENT, "bad camera id");
    const Camera& c = *ctx->cameras[ids[i]];
    REQUIRE(slot >= 0 && slot < c.n_slots, M3T_ERR_INVALID_ARGUMENT, "bad frame slot");
    const Camera& c0 = *ctx->cameras[ids[0]];
    if (c.slab < 0 || c.slab != c0.slab || c.slab_index != c0.slab_index + i) one_block = false;
  }
  const Camera& c0 = *ctx->cameras[ids[0]];
  const size_t row = size_t(c0.intr.width) * (c0.is_depth ? 2 : 3);
  REQUIRE(row_step >= row, M3T_ERR_INVALID_ARGUMENT, "row_step smaller than one image row");
  if (!(one_block && camera_stride == size_t(c0.intr.height) * row_step)) {
    for (int i = 0; i < n; ++i) {  // scattered on one side: one transfer per camera, still without a host round trip each
      int r = m3t_hip_camera_upload_slot_async(ctx, ids[i], slot, static_cast<const uint8_t*>(base) + size_t(i) * camera_stride,
                                               row_step);
      if (r) return r;
    }
    return M3T_OK;
  }
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->async_ingest) {
    HIPCHK(CreateCopyStreams(ctx));
    for (auto& e : ctx->copies_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : ctx->step_done) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->async_ingest = true;
    ctx->untracked_launches = true;
  }
  const int cs = 0;
  long last_read = -1;
  for (int i = 0; i < n; ++i) last_read = std::max(last_read, ctx->cameras[ids[i]]->last_read_step[slot]);
  if (ctx->untracked_launches) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->untracked_launches = false;
  } else if (last_read > ctx->copy_waited_step[cs]) {  // overwrite only after the last step that read this slot
    HIPCHK(hipStreamWaitEvent(ctx->copy_stream[cs], ctx->step_done[last_read % Ctx::kStepEvents], 0));
    ctx->copy_waited_step[cs] = last_read;
  }
  uint8_t* dst = c0.frame(slot);
  if (row_step == c0.pitch)
    HIPCHK(hipMemcpyAsync(dst, base, c0.frame_bytes * size_t(n) - (c0.pitch - row), hipMemcpyHostToDevice,
                          ctx->copy_stream[cs]));
  else
    HIPCHK(hipMemcpy2DAsync(dst, c0.pitch, base, row_step, row, size_t(c0.intr.height) * size_t(n), hipMemcpyHostToDevice,
                            ctx->copy_stream[cs]));
  for (int i = 0; i < n; ++i) ctx->cameras[ids[i]]->has_image[slot] = true;
  RoiMarkWholeFrames(ctx, ids, n, slot, ctx->copy_stream[cs]);
  // m3t_hip_camera_slot_sync: every camera of the batch waits for this transfer before its host block is written again
  // (also reached from cameras_upload_batch_roi_async whenever rectangles are not possible)
  {
    FrameSlab& slab = *ctx->slabs[size_t(c0.slab)];
    if (slab.slot_copied.size() < size_t(c0.n_slots)) slab.slot_copied.resize(size_t(c0.n_slots), nullptr);
    if (!slab.slot_copied[slot]) HIPCHK(hipEventCreateWithFlags(&slab.slot_copied[slot], hipEventDisableTiming));
    HIPCHK(hipEventRecord(slab.slot_copied[slot], ctx->copy_stream[cs]));
  }
  ctx->copies_pending |= 1u << cs;
  return M3T_OK;
}
// ROI ingest (m3t_ingest.hip).  enable: the fused rigid launches record the poses their searches run at,
// and m3t_hip_cameras_upload_batch_roi_async may upload rectangles instead of frames; margin_px: how far (in pixels,
// in every direction) the trackers' rectangle may move between the pose it is computed from and the last pose of
// the step that reads the frame, i.e. two frames of motion.
int m3t_hip_set_roi_ingest(m3t_hip_context* ctx, int enable, float margin_px) {
  CHECK_CTX();
  REQUIRE(margin_px >= 0.0f && margin_px < 1.0e6f, M3T_ERR_INVALID_ARGUMENT, "bad margin");
  REQUIRE(enable >= 0 && enable <= 2, M3T_ERR_INVALID_ARGUMENT, "enable: 0 (off), 1 (one margin for all bodies) or 2 (adaptive margins)");
  if ((enable != 0) != ctx->roi_enabled) ctx->tables_dirty = true;
  ctx->roi_enabled = enable != 0;
  ctx->roi_adaptive = enable == 2;
  ctx->roi_margin_px = margin_px;
  return M3T_OK;
}
// Keep n_cus CUs free of the tracking kernels and run the ROI pull kernel there (CU masks on the context's streams).
// Why: the pull kernel's PCIe reads (~2 us each) sit in the memory pipelines of the CUs it runs on; a tracking
// workgroup that shares its CU takes 3-4 x longer, and since an object's workgroups wait for each other so does the
// step.  On CUs of its own the pull costs the step ~14 % (tools/ubench_cumask.hip) and frame k + 1 crosses PCIe while
// step k runs.  n_cus should take the same number of CUs from every shader engine -- on MI355X a multiple of 32
// (8 XCDs x 4 shader engines): 16 leaves the engines unequal and a one-workgroup-per-CU launch 1.5 x slower, 64 gives
// the pull more lanes than PCIe can feed and costs the step more.  0 = off.  The call synchronises and REPLACES the
// context's streams: fetch m3t_hip_get_stream again afterwards.
int m3t_hip_reserve_ingest_cus(m3t_hip_context* ctx, int n_cus) {
  CHECK_CTX();
  REQUIRE(n_cus >= 0 && n_cus <= ctx->prop.multiProcessorCount / 2, M3T_ERR_INVALID_ARGUMENT,
          "n_cus must lie in [0, CUs / 2]");
  // unequal shader engines are not only slow: the hardware deals workgroups to engines, so a launch planned with one
  // workgroup per remaining CU may find two of its workgroups queued behind each other -- and the split / tree launches
  // need all of theirs resident at once
  REQUIRE(n_cus % 32 == 0, M3T_ERR_INVALID_ARGUMENT,
          "n_cus must be a multiple of 32 (one CU from each of the 4 shader engines of each of the 8 XCDs)");
  // the mask layout above (bit i -> XCD i mod 8, the highest bits of an XCD = the last CU of each shader engine) and the
  // multiple-of-32 rule were probed on MI355X in SPX mode (tools/ubench_cumask.hip): on another part or partition mode
  // the split / tree launches, which need all their workgroups resident, could be planned wrong
  REQUIRE(n_cus == 0 || (ctx->prop.multiProcessorCount == 256 && std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) == 0),
          M3T_ERR_UNSUPPORTED, "CU reservation is laid out for MI355X (gfx950, 256 CUs in one partition)");
  if (n_cus == ctx->ingest_cus) return M3T_OK;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (auto& cs : ctx->copy_stream)
    if (cs) HIPCHK(hipStreamSynchronize(cs));
  // all new streams first, then the swap: a failure leaves the context as it was.  (hipExtStreamCreateWithCUMask makes
  // BLOCKING streams: while CUs are reserved, work on the legacy NULL stream -- a plain hipMemcpy, torch's default
  // stream -- synchronises with the compute stream and copy stream 0; keep such work off the device meanwhile, or
  // the pull / step overlap is lost.)
  const int before = ctx->ingest_cus;
  ctx->ingest_cus = n_cus;
  hipStream_t compute = nullptr;
  hipStream_t copies_old[Ctx::kCopyStreams];
  for (int i = 0; i < Ctx::kCopyStreams; ++i) {
    copies_old[i] = ctx->copy_stream[i];
    ctx->copy_stream[i] = nullptr;
  }
  hipError_t e = CreateMaskedStream(ctx, &compute, false);
  if (e == hipSuccess && ctx->async_ingest) e = CreateCopyStreams(ctx);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (compute) (void)hipStreamDestroy(compute);
    for (int i = 0; i < Ctx::kCopyStreams; ++i) {
      if (ctx->copy_stream[i]) (void)hipStreamDestroy(ctx->copy_stream[i]);
      ctx->copy_stream[i] = copies_old[i];
    }
    ctx->ingest_cus = before;
    return Fail(ctx, M3T_ERR_UNSUPPORTED, "this device / runtime does not create CU-masked streams");
  }
  (void)hipStreamDestroy(ctx->stream);
  ctx->stream = compute;
  for (auto& cs : copies_old)
    if (cs) (void)hipStreamDestroy(cs);
  ctx->compute_cus = ctx->prop.multiProcessorCount - n_cus;
  ctx->copies_pending = 0;
  return M3T_OK;
}
// m3t_hip_cameras_upload_batch_async for the trackers' rectangles only: ONE kernel on the copy stream computes every
// camera's rectangle from the bodies' poses as of the start of the step enqueued last, and pulls its rows out of the
// mapped, page-locked host block.  Whole frames are uploaded instead (same result, more bytes) whenever the
// conditions are not met: ROI ingest off, no recorded step yet, cameras not in one slab in this order, a host block
// that is not registered, or strides that are not multiples of 16 bytes.
int m3t_hip_cameras_upload_batch_roi_async(m3t_hip_context* ctx, const int* ids, int n, int slot, const void* base,
                                           size_t camera_stride, size_t row_step) {
  CHECK_CTX();
  REQUIRE(ids && n >= 1 && base, M3T_ERR_INVALID_ARGUMENT, "bad arguments");
  bool pull = ctx->roi_enabled && ctx->roi_recorded && ctx->roi_snapshot_valid && ctx->n_roi_items > 0 &&
              !ctx->tables_dirty && !ctx->cams_dirty && ctx->async_ingest;
  for (int i = 0; i < n && pull; ++i) {
    if (ids[i] < 0 || ids[i] >= int(ctx->cameras.size())) { pull = false; break; }
    const Camera& c = *ctx->cameras[ids[i]];
    const Camera& c0 = *ctx->cameras[ids[0]];
    if (slot < 0 || slot >= c.n_slots || slot >= ctx->roi_rect_slots) pull = false;
    if (c.slab < 0 || c.slab != c0.slab || c.slab_index != c0.slab_index + i) pull = false;
  }
  const uint8_t* src = nullptr;
  if (pull) {
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, const_cast<void*>(base), 0) != hipSuccess) {
      (void)hipGetLastError();
      pull = false;
    }
    src = static_cast<const uint8_t*>(dev);
  }
  if (pull) {
    const Camera& c0 = *ctx->cameras[ids[0]];
    const size_t row = size_t(c0.intr.width) * (c0.is_depth ? 2 : 3), row16 = (row + 15) / 16 * 16;
    if (reinterpret_cast<uintptr_t>(src) % 16 || camera_stride % 16 || row_step % 16 || c0.pitch % 16 ||
        c0.frame_bytes % 16 || reinterpret_cast<uintptr_t>(c0.frame(slot)) % 16 || row16 > row_step || row16 > c0.pitch)
      pull = false;
  }
  if (!pull) return m3t_hip_cameras_upload_batch_async(ctx, ids, n, slot, base, camera_stride, row_step);
  HIPCHK(hipSetDevice(ctx->device));
  const int cs = 0;
  bool consecutive = true;
  for (int i = 0; i < n; ++i) consecutive = consecutive && ids[i] == ids[0] + i;
  const int* d_ids = ctx->d_roi_all_cam_ids.as<int>() + ids[0];
  if (!consecutive) {
    if (ctx->roi_cam_ids != std::vector<int>(ids, ids + n)) {  // (rare: a scattered batch whose camera list changed)
      HIPCHK(hipStreamSynchronize(ctx->copy_stream[cs]));
      HIPCHK(hipStreamSynchronize(ctx->stream));  // (a repair may still read the old list)
      for (auto& sources : ctx->roi_sources) sources.clear();
      ctx->roi_cam_ids.assign(ids, ids + n);
      HIPCHK(ctx->d_roi_cam_ids.alloc(size_t(n) * sizeof(int)));
      HIPCHK(hipMemcpy(ctx->d_roi_cam_ids.p, ids, size_t(n) * sizeof(int), hipMemcpyHostToDevice));
    }
    d_ids = ctx->d_roi_cam_ids.as<int>();
  }
  long last_read = -1;
  for (int i = 0; i < n; ++i) last_read = std::max(last_read, ctx->cameras[ids[i]]->last_read_step[slot]);
  if (ctx->untracked_launches) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->untracked_launches = false;
  } else if (last_read > ctx->copy_waited_step[cs]) {  // overwrite only after the last step that read this slot
    HIPCHK(hipStreamWaitEvent(ctx->copy_stream[cs], ctx->step_done[last_read % Ctx::kStepEvents], 0));
    ctx->copy_waited_step[cs] = last_read;
  }
  HIPCHK(hipStreamWaitEvent(ctx->copy_stream[cs], ctx->roi_snapshot_done[ctx->roi_use], 0));  // the poses the rectangles come from
  const bool adaptive = ctx->roi_adaptive && ctx->roi_prev >= 0;
  if (adaptive) HIPCHK(hipStreamWaitEvent(ctx->copy_stream[cs], ctx->roi_snapshot_done[ctx->roi_prev], 0));
  const Camera& c0 = *ctx->cameras[ids[0]];
  // the camera table is only read for intrinsics and world2camera here: any slot version will do (the first one)
  m3t_roi_rect* rects = ctx->d_roi_rects.as<m3t_roi_rect>() + size_t(slot) * ctx->cameras.size();
  const size_t snapshot_floats = ctx->d_roi_pose_snapshot.bytes / (4 * Ctx::kRoiSnapshots);
  const float* snapshots = ctx->d_roi_pose_snapshot.as<float>();
  // (without a second snapshot an adaptive upload takes the whole margin)
  hipLaunchKernelGGL(roi_rect_kernel, dim3((n + 63) / 64), dim3(64), 0, ctx->copy_stream[cs],
                     ctx->d_roi_items.as<RoiItemDev>(), ctx->d_roi_item_first.as<int>(), d_ids, n,
                     ctx->d_cams.as<CameraDev>(), snapshots + size_t(ctx->roi_use) * snapshot_floats,
                     adaptive ? snapshots + size_t(ctx->roi_prev) * snapshot_floats : (const float*)nullptr,
                     adaptive ? ctx->d_roi_motion_peak.as<float>() : (float*)nullptr, ctx->roi_margin_px,
                     std::min(ctx->roi_margin_px, 4.0f), rects);
  hipLaunchKernelGGL(roi_pull_kernel, dim3((c0.intr.height + 7) / 8, n), dim3(256), 0, ctx->copy_stream[cs],
                     d_ids, rects, src, camera_stride, uint32_t(row_step), c0.frame(slot),
                     c0.frame_bytes, c0.pitch, c0.is_depth ? 2 : 3);
  HIPCHK(hipGetLastError());
  ++ctx->roi_pulls;
  for (int i = 0; i < n; ++i) {
    Camera& c = *ctx->cameras[ids[i]];
    c.has_image[slot] = true;
    c.slot_is_roi[slot] = true;
  }
  {  // m3t_hip_camera_slot_sync waits for THIS pull, not for whatever else the copy stream carries by then
    FrameSlab& slab = *ctx->slabs[size_t(c0.slab)];
    if (slab.slot_copied.size() < size_t(c0.n_slots)) slab.slot_copied.resize(size_t(c0.n_slots), nullptr);
    if (!slab.slot_copied[slot]) HIPCHK(hipEventCreateWithFlags(&slab.slot_copied[slot], hipEventDisableTiming));
    HIPCHK(hipEventRecord(slab.slot_copied[slot], ctx->copy_stream[cs]));
  }
  {  // where the whole frames are, should a body outrun its rectangle (the repair of the step that reads this slot)
    auto& sources = ctx->roi_sources[size_t(slot)];
    for (size_t k = 0; k < sources.size();) {
      bool overlaps = false;
      for (int i = 0; i < n && !overlaps; ++i)
        overlaps = std::find(sources[k].ids.begin(), sources[k].ids.end(), ids[i]) != sources[k].ids.end();
      if (overlaps) sources.erase(sources.begin() + long(k)); else ++k;
    }
    Ctx::RoiSource source;
    source.ids.assign(ids, ids + n);
    source.d_ids = d_ids;
    source.src = src;
    source.camera_stride = camera_stride;
    source.row_step = uint32_t(row_step);
    sources.push_back(std::move(source));
  }
  ctx->copies_pending |= 1u << cs;
  return M3T_OK;
}
// Bodies whose steps since the last call needed pixels outside the rectangle that had been uploaded: up to `capacity`
// body ids, *n = how many there were; the list is cleared.  Such a step is not committed; the library fetched the whole
// frames and repeated it (m3t_hip_execute_tracking_step), so the poses ARE the whole-frame poses -- the list says how
// often the margin was too small.  m3t_hip_roi_get_unrecovered names the ones for which that was not possible.
int m3t_hip_roi_get_status(m3t_hip_context* ctx, int* bodies, int capacity, int* n, long long* n_rectangle_uploads) {
  CHECK_CTX();
  REQUIRE(n && capacity >= 0 && (bodies || capacity == 0), M3T_ERR_INVALID_ARGUMENT, "null output");
  *n = 0;
  if (n_rectangle_uploads) *n_rectangle_uploads = ctx->roi_pulls;
  if (!ctx->roi_miss_host) return M3T_OK;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int count = __atomic_load_n(&ctx->roi_miss_host[0], __ATOMIC_ACQUIRE);
  *n = count;
  for (int i = 0; i < std::min(std::min(count, capacity), int(Ctx::kRoiMissCapacity)); ++i) bodies[i] = ctx->roi_miss_host[1 + i];
  std::memset(ctx->roi_miss_host, 0, (Ctx::kRoiMissCapacity + 1) * sizeof(int));
  return M3T_OK;
}
// Bodies whose step left its rectangle AND whose repeat did too (the whole frames were not within reach: the host
// block of the upload was not recorded for the slot, e.g. after the tables were rebuilt).  Their step was not
// committed: pose, histograms and modality state are those before it.  Re-upload the frame in full
// (m3t_hip_cameras_upload_batch_async) and call m3t_hip_execute_tracking_step again -- the bodies that were committed
// are then stepped a second time, so set their poses back first or accept the extra step.  Cleared by the call.
int m3t_hip_roi_get_unrecovered(m3t_hip_context* ctx, int* bodies, int capacity, int* n) {
  CHECK_CTX();
  REQUIRE(n && capacity >= 0 && (bodies || capacity == 0), M3T_ERR_INVALID_ARGUMENT, "null output");
  *n = 0;
  if (!ctx->roi_unrecovered_host) return M3T_OK;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int count = __atomic_load_n(&ctx->roi_unrecovered_host[0], __ATOMIC_ACQUIRE);
  *n = count;
  for (int i = 0; i < std::min(std::min(count, capacity), int(Ctx::kRoiMissCapacity)); ++i) bodies[i] = ctx->roi_unrecovered_host[1 + i];
  std::memset(ctx->roi_unrecovered_host, 0, (Ctx::kRoiMissCapacity + 1) * sizeof(int));
  return M3T_OK;
}
int m3t_hip_ingest_sync(m3t_hip_context* ctx) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  for (auto& cs : ctx->copy_stream)
    if (cs) HIPCHK(hipStreamSynchronize(cs));
  return M3T_OK;
}
int m3t_hip_camera_select_slot(m3t_hip_context* ctx, int id, int slot) {
  CHECK_CTX();
  REQUIRE(id >= 0 && id < int(ctx->cameras.size()), M3T_ERR_INVALID_ARGUMENT, "bad camera id");
  Camera& c = *ctx->cameras[id];
  REQUIRE(slot >= 0 && slot < c.n_slots, M3T_ERR_INVALID_ARGUMENT, "bad frame slot");
  if (c.current != slot) {
    c.current = slot;
    ctx->slots_dirty = true;
  }
  return M3T_OK;
}
int m3t_hip_cameras_select_slot(m3t_hip_context* ctx, int slot) {
  CHECK_CTX();
  // every camera that has a ring: a camera without one keeps its only frame (the cameras of bodies whose modalities
  // live on another rank -- a structure spread over processes -- hold no frames at all; round 5: before, the call
  // failed on them with "bad frame slot", i.e. the N > 1 chain leg of bench.py could not have run)
  for (size_t i = 0; i < ctx->cameras.size(); ++i) {
    if (ctx->cameras[i]->n_slots <= 1 && slot != 0) continue;
    int r = m3t_hip_camera_select_slot(ctx, int(i), slot);
    if (r) return r;
  }
  return M3T_OK;
}

// ---- bodies ---------------------------------------------------------------------
int m3t_hip_body_create(m3t_hip_context* ctx, const float pose[16]) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  int r = SyncPosesToHost(ctx);
  if (r) return r;
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  const float* src = pose ? pose : ident;
  ctx->body_poses.insert(ctx->body_poses.end(), src, src + 16);
  ctx->poses_dirty_host = true;
  ctx->tables_dirty = true;
  return int(ctx->body_poses.size() / 16) - 1;
}
int m3t_hip_body_set_body2world_pose(m3t_hip_context* ctx, int id, const float pose[16]) {
  CHECK_CTX();
  REQUIRE(id >= 0 && id < int(ctx->body_poses.size() / 16) && pose, M3T_ERR_INVALID_ARGUMENT, "bad body id");
  HIPCHK(hipSetDevice(ctx->device));
  int r = SyncPosesToHost(ctx);
  if (r) return r;
  std::memcpy(&ctx->body_poses[size_t(id) * 16], pose, 64);
  ctx->poses_dirty_host = true;
  return M3T_OK;
}
int m3t_hip_body_get_body2world_pose(m3t_hip_context* ctx, int id, float pose[16]) {
  CHECK_CTX();
  REQUIRE(id >= 0 && id < int(ctx->body_poses.size() / 16) && pose, M3T_ERR_INVALID_ARGUMENT, "bad body id");
  HIPCHK(hipSetDevice(ctx->device));
  int r = SyncPosesToHost(ctx);
  if (r) return r;
  std::memcpy(pose, &ctx->body_poses[size_t(id) * 16], 64);
  return M3T_OK;
}
int m3t_hip_bodies_set_poses(m3t_hip_context* ctx, const float* poses, int n) {
  CHECK_CTX();
  REQUIRE(poses && n >= 0 && n <= int(ctx->body_poses.size() / 16), M3T_ERR_INVALID_ARGUMENT, "bad pose count");
  HIPCHK(hipSetDevice(ctx->device));
  int r = SyncPosesToHost(ctx);
  if (r) return r;
  std::memcpy(ctx->body_poses.data(), poses, size_t(n) * 64);
  ctx->poses_dirty_host = true;
  return M3T_OK;
}
int m3t_hip_bodies_get_poses(m3t_hip_context* ctx, float* poses, int n) {
  CHECK_CTX();
  REQUIRE(poses && n >= 0 && n <= int(ctx->body_poses.size() / 16), M3T_ERR_INVALID_ARGUMENT, "bad pose count");
  HIPCHK(hipSetDevice(ctx->device));
  int r = SyncPosesToHost(ctx);
  if (r) return r;
  std::memcpy(poses, ctx->body_poses.data(), size_t(n) * 64);
  return M3T_OK;
}

// ---- modalities -------------------------------------------------------------------
int m3t_hip_region_modality_create(m3t_hip_context* ctx, const m3t_region_modality_params* p, int body,
                                   int color_camera, int model, int depth_camera) {
  CHECK_CTX();
  REQUIRE(p, M3T_ERR_INVALID_ARGUMENT, "null params");
  REQUIRE(body >= 0 && body < int(ctx->body_poses.size() / 16), M3T_ERR_INVALID_ARGUMENT, "bad body id");
  REQUIRE(color_camera >= 0 && color_camera < int(ctx->cameras.size()) && !ctx->cameras[color_camera]->is_depth,
          M3T_ERR_INVALID_ARGUMENT, "bad color camera id");
  REQUIRE(model >= 0 && model < int(ctx->region_models.size()), M3T_ERR_INVALID_ARGUMENT, "bad region model id");
  REQUIRE(!p->use_region_checking && !p->model_occlusions, M3T_ERR_INVALID_ARGUMENT,
          "switch region checking / modelled occlusions on with their renderer afterwards");
  if (p->measure_occlusions)
    REQUIRE(depth_camera >= 0 && depth_camera < int(ctx->cameras.size()) && ctx->cameras[depth_camera]->is_depth,
            M3T_ERR_INVALID_ARGUMENT, "measure_occlusions needs a depth camera");
  REQUIRE(p->function_length >= 1 && p->function_length <= M3T_MAX_FUNCTION_LENGTH && p->distribution_length >= 2 &&
              p->distribution_length <= M3T_MAX_DISTRIBUTION_LENGTH && p->n_scales >= 1 &&
              p->n_scales <= M3T_MAX_SCALES && p->n_standard_deviations >= 1 &&
              p->n_standard_deviations <= M3T_MAX_SCALES && p->n_lines_max >= 1 && p->n_lines_max <= 1024,
          M3T_ERR_INVALID_ARGUMENT, "bad region modality parameters");
  int bitshift;
  switch (p->n_histogram_bins) {  // color_histograms.cpp:131-158
    case 2: bitshift = 7; break;
    case 4: bitshift = 6; break;
    case 8: bitshift = 5; break;
    case 16: bitshift = 4; break;
    case 32: bitshift = 3; break;
    case 64: bitshift = 2; break;
    default:
      return Fail(ctx, M3T_ERR_INVALID_ARGUMENT, "n_bins has to be of value 2, 4, 8, 16, 32, or 64");
  }
  HIPCHK(hipSetDevice(ctx->device));
  const Model& mdl = *ctx->region_models[model];
  auto m = std::make_unique<RegionMod>();
  m->p = *p;
  m->body = body;
  m->camera = color_camera;
  m->depth_camera = p->measure_occlusions ? depth_camera : -1;
  m->model = model;
  RegionModDev& d = m->dev;
  d.body = body;
  d.camera = color_camera;
  d.depth_camera = m->depth_camera;
  d.points = mdl.points.as<float>();
  d.points8 = mdl.points8.as<float4>();
  d.orientations4 = mdl.orientations4.as<float4>();
  d.extents = mdl.extents.as<float>();
  d.n_views = mdl.n_views;
  d.n_points = mdl.n_points;
  d.max_extent = mdl.max_extent;
  d.n_lines_max = p->n_lines_max;
  d.use_adaptive_coverage = p->use_adaptive_coverage;
  d.reference_contour_length = p->reference_contour_length;
  d.min_continuous_distance = p->min_continuous_distance;
  d.function_length = p->function_length;
  d.distribution_length = p->distribution_length;
  d.n_seg = p->function_length + p->distribution_length - 1;
  // PrecalculateFunctionLookup region_modality.cpp:910-923
  for (int i = 0; i < p->function_length; ++i) {
    float x = float(i) - float(p->function_length - 1) / 2.0f;
    if (p->function_slope == 0.0f)
      d.function_lookup_f[i] = 0.5f - p->function_amplitude * ((0.0f < x) - (x < 0.0f));
    else
      d.function_lookup_f[i] = 0.5f - p->function_amplitude * std::tanh(x / (2.0f * p->function_slope));
    d.function_lookup_b[i] = 1.0f - d.function_lookup_f[i];
  }
  // PrecalculateDistributionVariables region_modality.cpp:925-936
  d.distribution_length_minus_1_half = (float(p->distribution_length) - 1.0f) / 2.0f;
  d.distribution_length_plus_1_half = (float(p->distribution_length) + 1.0f) / 2.0f;
  float mev_laplace = 1.0f / (2.0f * powf(atanhf(2.0f * p->function_amplitude), 2.0f));
  d.min_expected_variance = std::max(mev_laplace, p->function_slope);
  d.learning_rate = p->learning_rate;
  d.n_global_iterations = p->n_global_iterations;
  d.n_scales = p->n_scales;
  d.n_standard_deviations = p->n_standard_deviations;
  for (int i = 0; i < M3T_MAX_SCALES; ++i) {
    d.scales[i] = p->scales[i];
    d.standard_deviations[i] = p->standard_deviations[i];
  }
  for (int i = 0; i < p->n_scales; ++i)
    REQUIRE(p->scales[i] >= 1, M3T_ERR_INVALID_ARGUMENT, "scales must be >= 1");
  d.n_bins = p->n_histogram_bins;
  d.bitshift = bitshift;
  d.learning_rate_f = p->learning_rate_f;
  d.learning_rate_b = p->learning_rate_b;
  d.unconsidered_line_length = p->unconsidered_line_length;
  d.max_considered_line_length = p->max_considered_line_length;
  d.measure_occlusions = p->measure_occlusions;
  d.measured_depth_offset_id = 0;
  if (p->measure_occlusions) {  // PrecalculateModelVariables region_modality.cpp:966-991
    REQUIRE(p->measured_depth_offset_radius <= mdl.max_radius_depth_offset, M3T_ERR_INVALID_ARGUMENT,
            "Measured depth offset radius too large");
    d.measured_depth_offset_id = int(p->measured_depth_offset_radius / mdl.stride_depth_offset + 0.5f);
    REQUIRE(d.measured_depth_offset_id < M3T_N_DEPTH_OFFSETS, M3T_ERR_INVALID_ARGUMENT, "depth offset id out of range");
  }
  d.measured_occlusion_radius = p->measured_occlusion_radius;
  d.measured_occlusion_threshold = p->measured_occlusion_threshold;
  d.n_unoccluded_iterations = p->n_unoccluded_iterations;
  d.min_n_unoccluded_lines = p->min_n_unoccluded_lines;
  d.first_iteration = 0;
  size_t bins3 = size_t(d.n_bins) * d.n_bins * d.n_bins;
  HIPCHK(m->hist_f.alloc(bins3 * 4));
  HIPCHK(m->hist_b.alloc(bins3 * 4));
  HIPCHK(m->hist_norm.alloc(bins3 * 8));
  HIPCHK(m->occupancy.alloc(bins3 / 4));
  HIPCHK(hipMemset(m->occupancy.p, 1, bins3 / 4));  // (the uniform start histograms below: every group is non-zero)
  if (bins3 * 4 + M3T_MISC_FLOATS * 4 > 160 * 1024) HIPCHK(m->count_scratch.alloc(bins3 * 4));
  HIPCHK(m->line_state.alloc(size_t(LS_FIELDS) * d.n_lines_max * 4));
  HIPCHK(m->gh.alloc(48 * 4));  // 42 floats g / H, then (at [44]) the view of the last search (int, -1: none yet)
  HIPCHK(hipMemset(m->line_state.p, 0, m->line_state.bytes));
  HIPCHK(hipMemset(m->gh.p, 0, m->gh.bytes));
  HIPCHK(hipMemset(m->gh.as<int>() + 44, 0xff, 4));
  {  // SetUpHistograms color_histograms.cpp:160-172: uniform 1/n^3
    std::vector<float> u(bins3, 1.0f / float(bins3));
    std::vector<float> nrm(bins3 * 2, 0.5f);
    HIPCHK(hipMemcpy(m->hist_f.p, u.data(), bins3 * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m->hist_b.p, u.data(), bins3 * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m->hist_norm.p, nrm.data(), bins3 * 8, hipMemcpyHostToDevice));
  }
  d.histogram_f = m->hist_f.as<float>();
  d.histogram_b = m->hist_b.as<float>();
  d.histogram_norm = m->hist_norm.as<float2>();
  d.occupancy = m->occupancy.as<uint8_t>();
  d.count_scratch = m->count_scratch.as<uint32_t>();
  d.line_state = m->line_state.as<float>();
  d.gradient_hessian = m->gh.as<float>();
  d.view_neighbors = mdl.view_neighbors.as<float4>();
  d.last_view = m->gh.as<int>() + 44;
  ctx->region_mods.push_back(std::move(m));
  ctx->modalities.push_back({true, int(ctx->region_mods.size()) - 1});
  ctx->tables_dirty = true;
  return int(ctx->modalities.size()) - 1;
}

int m3t_hip_depth_modality_create(m3t_hip_context* ctx, const m3t_depth_modality_params* p, int body, int depth_camera,
                                  int model) {
  CHECK_CTX();
  REQUIRE(p, M3T_ERR_INVALID_ARGUMENT, "null params");
  REQUIRE(body >= 0 && body < int(ctx->body_poses.size() / 16), M3T_ERR_INVALID_ARGUMENT, "bad body id");
  REQUIRE(depth_camera >= 0 && depth_camera < int(ctx->cameras.size()) && ctx->cameras[depth_camera]->is_depth,
          M3T_ERR_INVALID_ARGUMENT, "bad depth camera id");
  REQUIRE(model >= 0 && model < int(ctx->depth_models.size()), M3T_ERR_INVALID_ARGUMENT, "bad depth model id");
  REQUIRE(!p->use_silhouette_checking && !p->model_occlusions, M3T_ERR_INVALID_ARGUMENT,
          "switch silhouette checking / modelled occlusions on with their renderer afterwards");
  REQUIRE(p->n_considered_distances >= 1 && p->n_considered_distances <= M3T_MAX_SCALES &&
              p->n_standard_deviations >= 1 && p->n_standard_deviations <= M3T_MAX_SCALES && p->n_points_max >= 1,
          M3T_ERR_INVALID_ARGUMENT, "bad depth modality parameters");
  HIPCHK(hipSetDevice(ctx->device));
  const Model& mdl = *ctx->depth_models[model];
  auto m = std::make_unique<DepthMod>();
  m->p = *p;
  m->body = body;
  m->camera = depth_camera;
  m->model = model;
  DepthModDev& d = m->dev;
  d.body = body;
  d.camera = depth_camera;
  d.points = mdl.points.as<float>();
  d.points8 = mdl.points8.as<float4>();
  d.orientations4 = mdl.orientations4.as<float4>();
  d.extents = mdl.extents.as<float>();
  d.n_views = mdl.n_views;
  d.n_points = mdl.n_points;
  d.max_extent = mdl.max_extent;
  d.stride_depth_offset = mdl.stride_depth_offset;
  d.n_points_max = p->n_points_max;
  d.use_adaptive_coverage = p->use_adaptive_coverage;
  d.use_depth_scaling = p->use_depth_scaling;
  d.reference_surface_area = p->reference_surface_area;
  d.stride_length = p->stride_length;
  d.n_considered_distances = p->n_considered_distances;
  d.n_standard_deviations = p->n_standard_deviations;
  for (int i = 0; i < M3T_MAX_SCALES; ++i) {
    d.considered_distances[i] = p->considered_distances[i];
    d.standard_deviations[i] = p->standard_deviations[i];
  }
  d.measure_occlusions = p->measure_occlusions;
  d.measured_depth_offset_radius = p->measured_depth_offset_radius;
  d.measured_occlusion_radius = p->measured_occlusion_radius;
  d.measured_occlusion_threshold = p->measured_occlusion_threshold;
  d.n_unoccluded_iterations = p->n_unoccluded_iterations;
  d.min_n_unoccluded_points = p->min_n_unoccluded_points;
  d.first_iteration = 0;  // DepthModality never sets first_iteration_ (depth_modality.h:358)
  HIPCHK(m->point_state.alloc(size_t(PS_FIELDS) * d.n_points_max * 4));
  HIPCHK(m->gh.alloc(42 * 4));
  HIPCHK(hipMemset(m->point_state.p, 0, m->point_state.bytes));
  HIPCHK(hipMemset(m->gh.p, 0, m->gh.bytes));
  d.point_state = m->point_state.as<float>();
  d.gradient_hessian = m->gh.as<float>();
  ctx->depth_mods.push_back(std::move(m));
  ctx->modalities.push_back({false, int(ctx->depth_mods.size()) - 1});
  ctx->tables_dirty = true;
  return int(ctx->modalities.size()) - 1;
}

static float* ModalityGh(m3t_hip_context* ctx, int id) {
  if (id < 0 || id >= int(ctx->modalities.size())) return nullptr;
  const ModalityRef& r = ctx->modalities[id];
  return r.region ? ctx->region_mods[r.index]->gh.as<float>() : ctx->depth_mods[r.index]->gh.as<float>();
}
int m3t_hip_modality_get_gradient_hessian(m3t_hip_context* ctx, int id, float g[6], float h[36]) {
  CHECK_CTX();
  float* d = ModalityGh(ctx, id);
  REQUIRE(d, M3T_ERR_INVALID_ARGUMENT, "bad modality id");
  REQUIRE(ctx->state_valid, M3T_ERR_NOT_SET_UP,
          "gradient/hessian are not written back in fused mode 1 (use m3t_hip_set_fused_step(ctx, 0 or 2))");
  HIPCHK(hipSetDevice(ctx->device));
  float buf[42];
  HIPCHK(hipMemcpyAsync(buf, d, sizeof(buf), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (g) std::memcpy(g, buf, 24);
  if (h) std::memcpy(h, buf + 6, 144);
  return M3T_OK;
}
}  // extern "C"
// gradient | Hessian of every modality into one contiguous buffer (one read-back per round for an adapter host)
extern "C" __global__ void gather_gradient_hessian_kernel(const float* const* sources, float* out) {
  out[blockIdx.x * 42 + threadIdx.x] = sources[blockIdx.x][threadIdx.x];
}
extern "C" {
int m3t_hip_modalities_get_gradient_hessian(m3t_hip_context* ctx, float* out, int capacity) {
  CHECK_CTX();
  const int n = int(ctx->modalities.size());
  REQUIRE(out && capacity >= n, M3T_ERR_INVALID_ARGUMENT, "the buffer must hold 42 floats per modality");
  REQUIRE(ctx->state_valid, M3T_ERR_NOT_SET_UP,
          "gradient/hessian are not written back in fused mode 1 (use m3t_hip_set_fused_step(ctx, 0 or 2))");
  if (n == 0) return M3T_OK;
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->gh_sources_count != n) {
    std::vector<const float*> src(size_t(n), nullptr);
    for (int i = 0; i < n; ++i) src[size_t(i)] = ModalityGh(ctx, i);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx->d_gh_sources.alloc(size_t(n) * sizeof(float*)));
    HIPCHK(ctx->d_gh_all.alloc(size_t(n) * 42 * sizeof(float)));
    HIPCHK(hipMemcpy(ctx->d_gh_sources.p, src.data(), size_t(n) * sizeof(float*), hipMemcpyHostToDevice));
    ctx->gh_sources_count = n;
  }
  hipLaunchKernelGGL(gather_gradient_hessian_kernel, dim3(n), dim3(42), 0, ctx->stream,
                     ctx->d_gh_sources.as<const float*>(), ctx->d_gh_all.as<float>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, ctx->d_gh_all.p, size_t(n) * 42 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return M3T_OK;
}
int m3t_hip_modality_set_gradient_hessian(m3t_hip_context* ctx, int id, const float g[6], const float h[36]) {
  CHECK_CTX();
  float* d = ModalityGh(ctx, id);
  REQUIRE(d && g && h, M3T_ERR_INVALID_ARGUMENT, "bad modality id");
  HIPCHK(hipSetDevice(ctx->device));
  float buf[42];
  std::memcpy(buf, g, 24);
  std::memcpy(buf + 6, h, 144);
  HIPCHK(hipMemcpyAsync(d, buf, sizeof(buf), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->state_valid = true;
  return M3T_OK;
}

int m3t_hip_region_modality_get_lines(m3t_hip_context* ctx, int id, m3t_data_line* out, int capacity, int* n) {
  CHECK_CTX();
  RegionMod* m = GetRegion(ctx, id);
  REQUIRE(m, M3T_ERR_INVALID_ARGUMENT, "bad region modality id");
  REQUIRE(ctx->state_valid, M3T_ERR_NOT_SET_UP,
          "line state is not written back in fused mode 1 (use m3t_hip_set_fused_step(ctx, 0 or 2))");
  HIPCHK(hipSetDevice(ctx->device));
  const int nl = m->p.n_lines_max;
  std::vector<float> st(size_t(LS_FIELDS) * nl);
  HIPCHK(hipMemcpyAsync(st.data(), m->line_state.p, st.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int count = 0;
  for (int l = 0; l < nl; ++l) {
    int flags;
    std::memcpy(&flags, &st[size_t(LS_VALID) * nl + l], 4);
    if (!(flags & 1)) continue;
    if (out && count < capacity) {
      m3t_data_line& o = out[count];
      std::memset(&o, 0, sizeof(o));
      o.center_f_body[0] = st[size_t(LS_CX) * nl + l];
      o.center_f_body[1] = st[size_t(LS_CY) * nl + l];
      o.center_f_body[2] = st[size_t(LS_CZ) * nl + l];
      o.center_u = st[size_t(LS_CENTER_U) * nl + l];
      o.center_v = st[size_t(LS_CENTER_V) * nl + l];
      o.normal_u = st[size_t(LS_NORMAL_U) * nl + l];
      o.normal_v = st[size_t(LS_NORMAL_V) * nl + l];
      o.delta_r = st[size_t(LS_DELTA_R) * nl + l];
      o.normal_component_to_scale = st[size_t(LS_NCTS) * nl + l];
      o.continuous_distance = st[size_t(LS_CONT) * nl + l];
      o.mean = st[size_t(LS_MEAN) * nl + l];
      o.measured_variance = st[size_t(LS_VAR) * nl + l];
      for (int d = 0; d < m->p.distribution_length; ++d) o.distribution[d] = st[size_t(LS_DIST0 + d) * nl + l];
      o.valid = 1;
      o.model_point_index = l;
    }
    ++count;
  }
  if (n) *n = count;
  return M3T_OK;
}

int m3t_hip_depth_modality_get_points(m3t_hip_context* ctx, int id, m3t_data_point* out, int capacity, int* n) {
  CHECK_CTX();
  DepthMod* m = GetDepth(ctx, id);
  REQUIRE(m, M3T_ERR_INVALID_ARGUMENT, "bad depth modality id");
  REQUIRE(ctx->state_valid, M3T_ERR_NOT_SET_UP,
          "point state is not written back in fused mode 1 (use m3t_hip_set_fused_step(ctx, 0 or 2))");
  HIPCHK(hipSetDevice(ctx->device));
  const int np = m->p.n_points_max;
  std::vector<float> st(size_t(PS_FIELDS) * np);
  HIPCHK(hipMemcpyAsync(st.data(), m->point_state.p, st.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int count = 0;
  for (int i = 0; i < np; ++i) {
    int flags;
    std::memcpy(&flags, &st[size_t(PS_VALID) * np + i], 4);
    if (!(flags & 1)) continue;
    if (out && count < capacity) {
      m3t_data_point& o = out[count];
      std::memset(&o, 0, sizeof(o));
      for (int k = 0; k < 3; ++k) {
        o.center_f_body[k] = st[size_t(PS_CX + k) * np + i];
        o.normal_f_body[k] = st[size_t(PS_NX + k) * np + i];
        o.correspondence_center_f_camera[k] = st[size_t(PS_CORR_X + k) * np + i];
      }
      o.center_u = st[size_t(PS_CENTER_U) * np + i];
      o.center_v = st[size_t(PS_CENTER_V) * np + i];
      o.depth = st[size_t(PS_DEPTH) * np + i];
      o.valid = 1;
      o.model_point_index = i;
    }
    ++count;
  }
  if (n) *n = count;
  return M3T_OK;
}

int m3t_hip_region_modality_get_histograms(m3t_hip_context* ctx, int id, float* f, float* b) {
  CHECK_CTX();
  RegionMod* m = GetRegion(ctx, id);
  REQUIRE(m, M3T_ERR_INVALID_ARGUMENT, "bad region modality id");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const size_t bytes = size_t(m->dev.n_bins) * m->dev.n_bins * m->dev.n_bins * 4;  // private or shared tables
  if (f) HIPCHK(hipMemcpy(f, m->dev.histogram_f, bytes, hipMemcpyDeviceToHost));
  if (b) HIPCHK(hipMemcpy(b, m->dev.histogram_b, bytes, hipMemcpyDeviceToHost));
  return M3T_OK;
}
int m3t_hip_region_modality_set_histograms(m3t_hip_context* ctx, int id, const float* f, const float* b) {
  CHECK_CTX();
  RegionMod* m = GetRegion(ctx, id);
  REQUIRE(m && f && b, M3T_ERR_INVALID_ARGUMENT, "bad region modality id");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  size_t bins3 = size_t(m->dev.n_bins) * m->dev.n_bins * m->dev.n_bins;
  std::vector<float> nrm(bins3 * 2);
  for (size_t i = 0; i < bins3; ++i) {  // MultiplyPixelColorProbability :1585-1593 per bin
    float pf = f[i], pb = b[i];
    if (pf || pb) {
      float sum = pf;
      sum += pb;
      nrm[2 * i] = pf / sum;
      nrm[2 * i + 1] = pb / sum;
    } else {
      nrm[2 * i] = 0.5f;
      nrm[2 * i + 1] = 0.5f;
    }
  }
  HIPCHK(hipMemcpy(m->dev.histogram_f, f, bins3 * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m->dev.histogram_b, b, bins3 * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m->dev.histogram_norm, nrm.data(), bins3 * 8, hipMemcpyHostToDevice));
  std::vector<uint8_t> occupied(bins3 / 4);
  for (size_t g = 0; g < bins3 / 4; ++g) {
    bool any = false;
    for (size_t k = 4 * g; k < 4 * g + 4; ++k) any = any || f[k] != 0.0f || b[k] != 0.0f;
    occupied[g] = any ? 1 : 0;
  }
  HIPCHK(hipMemcpy(m->dev.occupancy, occupied.data(), occupied.size(), hipMemcpyHostToDevice));
  return M3T_OK;
}

// ---- links / optimizers ---------------------------------------------------------------
static bool IsIdentity(const float* p) {
  if (!p) return true;
  static const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  return std::memcmp(p, ident, 64) == 0;
}
static const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};

// m3t::Link (link.h:67): body may be -1 (pure joint), parent -1 = root
// ---- renderer-fed branches (a14 / f-3) ---------------------------------------------------------------
int m3t_hip_body_set_geometry(m3t_hip_context* ctx, int body, const m3t_body_geometry* g) {
  CHECK_CTX();
  REQUIRE(body >= 0 && body < int(ctx->body_poses.size() / 16) && g && g->vertices && g->triangles &&
              g->n_vertices >= 3 && g->n_triangles >= 1 && g->body_id >= 0 && g->body_id <= 255 &&
              g->region_id >= 0 && g->region_id <= 255,
          M3T_ERR_INVALID_ARGUMENT, "bad body geometry");
  for (int i = 0; i < g->n_triangles * 3; ++i)
    REQUIRE(g->triangles[i] >= 0 && g->triangles[i] < g->n_vertices, M3T_ERR_INVALID_ARGUMENT,
            "triangle index out of range");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->body_geometries.size() <= size_t(body)) ctx->body_geometries.resize(size_t(body) + 1);
  auto bg = std::make_unique<BodyGeometryH>();
  std::vector<int> tri(size_t(g->n_triangles) * 3);
  for (int t = 0; t < g->n_triangles; ++t)  // body.cpp:227-236
    for (int k = 0; k < 3; ++k)
      tri[size_t(t) * 3 + k] = g->triangles[size_t(t) * 3 + (g->geometry_counterclockwise ? k : 2 - k)];
  HIPCHK(bg->vertices.alloc(size_t(g->n_vertices) * 12));
  HIPCHK(bg->triangles.alloc(tri.size() * 4));
  HIPCHK(hipMemcpy(bg->vertices.p, g->vertices, size_t(g->n_vertices) * 12, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(bg->triangles.p, tri.data(), tri.size() * 4, hipMemcpyHostToDevice));
  bg->n_triangles = g->n_triangles;
  bg->h_vertices.assign(g->vertices, g->vertices + size_t(g->n_vertices) * 3);
  bg->h_triangles = tri;
  std::memcpy(bg->geometry2body, g->geometry2body, 64);
  bg->culling = g->geometry_enable_culling ? 1 : 0;
  bg->body_id = g->body_id;
  bg->region_id = g->region_id;
  float max_radius = 0.0f;  // Body::CalculateMaximumBodyDiameter body.cpp:244-250
  const float* m = g->geometry2body;
  for (int i = 0; i < g->n_vertices; ++i) {
    const float* v = g->vertices + size_t(i) * 3;
    float q[3];
    for (int k = 0; k < 3; ++k) q[k] = m[12 + k] + ((m[k] * v[0] + m[4 + k] * v[1]) + m[8 + k] * v[2]);
    max_radius = std::max(max_radius, std::sqrt(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2])));
  }
  bg->maximum_body_diameter = 2.0f * max_radius;
  bg->set = true;
  ctx->body_geometries[body] = std::move(bg);
  ctx->tables_dirty = true;
  return M3T_OK;
}
static bool HasGeometry(m3t_hip_context* ctx, int body) {
  return body >= 0 && size_t(body) < ctx->body_geometries.size() && ctx->body_geometries[body] &&
         ctx->body_geometries[body]->set;
}
int m3t_hip_renderer_geometry_create(m3t_hip_context* ctx) {
  CHECK_CTX();
  ctx->renderer_geometries.emplace_back();
  return int(ctx->renderer_geometries.size()) - 1;
}
int m3t_hip_renderer_geometry_add_body(m3t_hip_context* ctx, int geometry, int body) {
  CHECK_CTX();
  REQUIRE(geometry >= 0 && geometry < int(ctx->renderer_geometries.size()) && body >= 0 &&
              body < int(ctx->body_poses.size() / 16),
          M3T_ERR_INVALID_ARGUMENT, "bad renderer geometry / body id");
  REQUIRE(HasGeometry(ctx, body), M3T_ERR_NOT_SET_UP, "body has no geometry");
  REQUIRE(ctx->renderer_geometries[geometry].size() < M3T_MAX_RENDERER_BODIES, M3T_ERR_UNSUPPORTED, "too many bodies");
  ctx->renderer_geometries[geometry].push_back(body);
  ctx->tables_dirty = true;
  return M3T_OK;
}
static int CreateRenderer(m3t_hip_context* ctx, bool silhouette, int geometry, int camera, int id_type, int image_size,
                          float z_min, float z_max) {
  CHECK_CTX();
  REQUIRE(geometry >= 0 && geometry < int(ctx->renderer_geometries.size()) && camera >= 0 &&
              camera < int(ctx->cameras.size()) && image_size >= 8 && image_size <= 1024 && z_min > 0.0f &&
              z_max > z_min && (id_type == M3T_ID_TYPE_BODY || id_type == M3T_ID_TYPE_REGION),
          M3T_ERR_INVALID_ARGUMENT, "bad renderer arguments");
  HIPCHK(hipSetDevice(ctx->device));
  auto r = std::make_unique<RendererH>();
  r->silhouette = silhouette;
  r->geometry = geometry;
  r->camera = camera;
  r->id_type = id_type;
  r->image_size = image_size;
  r->z_min = z_min;
  r->z_max = z_max;
  const size_t px = size_t(image_size) * image_size;
  HIPCHK(r->depth.alloc(px * 2));
  HIPCHK(r->sil.alloc(px));
  HIPCHK(r->packed.alloc(px * 4));
  HIPCHK(r->state.alloc(RS_FLOATS * 4));
  HIPCHK(hipMemset(r->state.p, 0, RS_FLOATS * 4));
  ctx->renderers.push_back(std::move(r));
  ctx->tables_dirty = true;
  return int(ctx->renderers.size()) - 1;
}
int m3t_hip_focused_depth_renderer_create(m3t_hip_context* ctx, int geometry, int camera, int image_size, float z_min,
                                          float z_max) {
  return CreateRenderer(ctx, false, geometry, camera, M3T_ID_TYPE_BODY, image_size, z_min, z_max);
}
int m3t_hip_focused_silhouette_renderer_create(m3t_hip_context* ctx, int geometry, int camera, int id_type,
                                               int image_size, float z_min, float z_max) {
  return CreateRenderer(ctx, true, geometry, camera, id_type, image_size, z_min, z_max);
}
int m3t_hip_renderer_add_referenced_body(m3t_hip_context* ctx, int renderer, int body) {
  CHECK_CTX();
  REQUIRE(renderer >= 0 && renderer < int(ctx->renderers.size()) && body >= 0 && body < int(ctx->body_poses.size() / 16),
          M3T_ERR_INVALID_ARGUMENT, "bad renderer / body id");
  REQUIRE(HasGeometry(ctx, body), M3T_ERR_NOT_SET_UP, "body has no geometry");
  RendererH& r = *ctx->renderers[renderer];
  REQUIRE(r.referenced.size() < M3T_MAX_RENDERER_BODIES, M3T_ERR_UNSUPPORTED, "too many referenced bodies");
  r.referenced.push_back(body);
  r.rendered = false;
  ctx->tables_dirty = true;
  return M3T_OK;
}
int m3t_hip_renderer_start_rendering(m3t_hip_context* ctx, int renderer) {
  CHECK_CTX();
  REQUIRE(renderer >= 0 && renderer < int(ctx->renderers.size()), M3T_ERR_INVALID_ARGUMENT, "bad renderer id");
  REQUIRE(!ctx->renderers[renderer]->referenced.empty(), M3T_ERR_NOT_SET_UP, "no referenced body");
  HIPCHK(hipSetDevice(ctx->device));
  int r = Prepare(ctx, false);
  if (r) return r;
  DevMem which;  // freed after the synchronisation below
  const int pair[2] = {renderer, -1};
  HIPCHK(which.alloc(8));
  HIPCHK(hipMemcpy(which.p, pair, 8, hipMemcpyHostToDevice));
  if ((r = LaunchRenderers(ctx, which.as<int>(), 1, which.as<int>(), 1, ctx->renderers[renderer]->image_size))) return r;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->renderers[renderer]->rendered = true;
  return M3T_OK;
}
int m3t_hip_renderer_get_images(m3t_hip_context* ctx, int renderer, uint16_t* depth, uint8_t* silhouette,
                                float info[3], int* n_visible) {
  CHECK_CTX();
  REQUIRE(renderer >= 0 && renderer < int(ctx->renderers.size()), M3T_ERR_INVALID_ARGUMENT, "bad renderer id");
  RendererH& r = *ctx->renderers[renderer];
  REQUIRE(r.rendered, M3T_ERR_NOT_SET_UP, "renderer has not rendered yet");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const size_t px = size_t(r.image_size) * r.image_size;
  if (depth) HIPCHK(hipMemcpy(depth, r.depth.p, px * 2, hipMemcpyDeviceToHost));
  if (silhouette) HIPCHK(hipMemcpy(silhouette, r.sil.p, px, hipMemcpyDeviceToHost));
  float state[RS_FLOATS];
  HIPCHK(hipMemcpy(state, r.state.p, sizeof(state), hipMemcpyDeviceToHost));
  if (info) { info[0] = state[RS_CORNER_U]; info[1] = state[RS_CORNER_V]; info[2] = state[RS_SCALE]; }
  if (n_visible) *n_visible = int(state[RS_N_VISIBLE]);
  return M3T_OK;
}
static int CheckAttach(m3t_hip_context* ctx, int modality, int renderer, bool want_region, bool want_silhouette,
                       int body) {
  REQUIRE(renderer >= 0 && renderer < int(ctx->renderers.size()) &&
              ctx->renderers[renderer]->silhouette == want_silhouette,
          M3T_ERR_INVALID_ARGUMENT, "bad renderer id / kind");
  bool referenced = false;
  for (int b : ctx->renderers[renderer]->referenced) referenced |= b == body;
  REQUIRE(referenced, M3T_ERR_INVALID_ARGUMENT, "the modality's body is not referenced by the renderer");
  return M3T_OK;
}
int m3t_hip_region_modality_model_occlusions(m3t_hip_context* ctx, int modality, int renderer) {
  CHECK_CTX();
  RegionMod* m = GetRegion(ctx, modality);
  REQUIRE(m, M3T_ERR_INVALID_ARGUMENT, "bad region modality id");
  int r = CheckAttach(ctx, modality, renderer, true, false, m->body);
  if (r) return r;
  const Model& mdl = *ctx->region_models[m->model];
  REQUIRE(m->p.modeled_depth_offset_radius <= mdl.max_radius_depth_offset, M3T_ERR_INVALID_ARGUMENT,
          "Modeled depth offset radius too large");
  m->depth_renderer = renderer;
  m->p.model_occlusions = 1;
  m->dev.model_occlusions = 1;
  m->dev.modeled_depth_offset_id = int(m->p.modeled_depth_offset_radius / mdl.stride_depth_offset + 0.5f);
  m->dev.modeled_occlusion_radius = m->p.modeled_occlusion_radius;
  m->dev.modeled_occlusion_threshold = m->p.modeled_occlusion_threshold;
  ctx->tables_dirty = true;
  return M3T_OK;
}
int m3t_hip_region_modality_use_region_checking(m3t_hip_context* ctx, int modality, int renderer) {
  CHECK_CTX();
  RegionMod* m = GetRegion(ctx, modality);
  REQUIRE(m, M3T_ERR_INVALID_ARGUMENT, "bad region modality id");
  int r = CheckAttach(ctx, modality, renderer, true, true, m->body);
  if (r) return r;
  m->silhouette_renderer = renderer;
  m->p.use_region_checking = 1;
  m->dev.use_region_checking = 1;
  m->dev.region_id = ctx->body_geometries[m->body]->region_id;
  ctx->tables_dirty = true;
  return M3T_OK;
}
int m3t_hip_depth_modality_model_occlusions(m3t_hip_context* ctx, int modality, int renderer) {
  CHECK_CTX();
  REQUIRE(modality >= 0 && modality < int(ctx->modalities.size()) && !ctx->modalities[modality].region,
          M3T_ERR_INVALID_ARGUMENT, "bad depth modality id");
  DepthMod* m = ctx->depth_mods[ctx->modalities[modality].index].get();
  int r = CheckAttach(ctx, modality, renderer, false, false, m->body);
  if (r) return r;
  m->depth_renderer = renderer;
  m->p.model_occlusions = 1;
  m->dev.model_occlusions = 1;
  m->dev.modeled_depth_offset_radius = m->p.modeled_depth_offset_radius;
  m->dev.modeled_occlusion_radius = m->p.modeled_occlusion_radius;
  m->dev.modeled_occlusion_threshold = m->p.modeled_occlusion_threshold;
  ctx->tables_dirty = true;
  return M3T_OK;
}
int m3t_hip_depth_modality_use_silhouette_checking(m3t_hip_context* ctx, int modality, int renderer) {
  CHECK_CTX();
  REQUIRE(modality >= 0 && modality < int(ctx->modalities.size()) && !ctx->modalities[modality].region,
          M3T_ERR_INVALID_ARGUMENT, "bad depth modality id");
  DepthMod* m = ctx->depth_mods[ctx->modalities[modality].index].get();
  int r = CheckAttach(ctx, modality, renderer, false, true, m->body);
  if (r) return r;
  m->silhouette_renderer = renderer;
  m->p.use_silhouette_checking = 1;
  m->dev.use_silhouette_checking = 1;
  m->dev.body_id = ctx->body_geometries[m->body]->body_id;
  ctx->tables_dirty = true;
  return M3T_OK;
}

// ---- model generation (f-1) ---------------------------------------------------------------------------
// groups (RegionModel::AddAssociatedBody region_model.cpp:365-388): 0 fixed, 1 movable, 2 fixed same-region,
// 3 movable same-region; for the depth model 0 = occlusion bodies (DepthModel::AddOcclusionBody depth_model.cpp:61-70)
static int GenerateModel(m3t_hip_context* ctx, bool region, int body, const m3t_model_generation_params* p,
                         const std::vector<int> (&groups)[4] = {}) {
  CHECK_CTX();
  REQUIRE(p && p->sphere_radius > 0.0f && p->n_divides >= 0 && p->n_divides <= 6 && p->n_points >= 1 &&
              p->image_size >= 64 && p->image_size <= 4096 && p->stride_depth_offset > 0.0f &&
              p->max_radius_depth_offset >= 0.0f &&
              int(p->max_radius_depth_offset / p->stride_depth_offset + 1.0f) <= M3T_N_DEPTH_OFFSETS,
          M3T_ERR_INVALID_ARGUMENT, "bad model generation parameters");
  REQUIRE(HasGeometry(ctx, body), M3T_ERR_NOT_SET_UP, "body has no geometry");
  for (auto& grp : groups)
    for (int b : grp) REQUIRE(HasGeometry(ctx, b) && b != body, M3T_ERR_NOT_SET_UP, "associated body has no geometry");
  HIPCHK(hipSetDevice(ctx->device));
  using namespace modelgen;
  const BodyGeometryH& g = *ctx->body_geometries[body];
  const int S = p->image_size;
  const float d = g.maximum_body_diameter, radius = p->sphere_radius;
  REQUIRE(0.5f * d < radius, M3T_ERR_INVALID_ARGUMENT, "sphere radius smaller than the body");
  // Model::SetUpRenderer model.cpp:120-153: intrinsics from the main body
  const float fu = 0.5f * float(S - 20) / tanf(asinf(0.5f * d / radius));
  const float pp = float(S) / 2.0f;
  // the renderers of one view: which bodies are drawn, in which order, with which id.  Renderers that draw the same
  // bodies share one rasterisation; their silhouette images differ only in the ids.
  constexpr uint8_t M = 255, B = 0, D = 120;  // kMainBodyID, kBackgroundID, kDifferentBodyID
  struct Draw { int body; uint8_t id; };
  struct Renderer { std::vector<Draw> draws; };
  auto draws = [&](std::initializer_list<std::pair<const std::vector<int>*, uint8_t>> parts, uint8_t main_id) {
    Renderer r;
    r.draws.push_back({body, main_id});
    for (auto& part : parts)
      for (int b : *part.first) r.draws.push_back({b, part.second});
    return r;
  };
  const std::vector<int>&fixed = groups[0], &movable = groups[1], &fixed_same = groups[2], &movable_same = groups[3];
  const bool any_associated = !fixed.empty() || !movable.empty() || !fixed_same.empty() || !movable_same.empty();
  enum { R_MAIN = 0, R_OCCLUSION, R_SAME_REGION, R_FOREGROUND, R_BACKGROUND, N_RENDERERS };
  Renderer renderers[N_RENDERERS];
  bool used[N_RENDERERS] = {true, false, false, false, false};
  if (region) {  // region_model.cpp:207-213, 417-463
    renderers[R_MAIN] = draws({{&fixed, D}}, M);
    if (!movable.empty()) {
      renderers[R_OCCLUSION] = draws({{&fixed, B}, {&movable, M}}, B);
      used[R_OCCLUSION] = true;
    }
    if (!fixed_same.empty() || !movable_same.empty()) {
      renderers[R_SAME_REGION] = draws({{&fixed, B}, {&fixed_same, M}, {&movable_same, M}}, B);
      used[R_SAME_REGION] = true;
    }
    if (!movable.empty() || !fixed_same.empty() || !movable_same.empty()) {
      renderers[R_FOREGROUND] = draws({{&fixed, B}, {&movable, B}, {&fixed_same, M}}, M);
      renderers[R_BACKGROUND] = draws({{&fixed, B}, {&fixed_same, M}, {&movable_same, M}}, M);
      used[R_FOREGROUND] = used[R_BACKGROUND] = true;
    }
  } else {  // depth_model.cpp:165-177: the main (normal) renderer draws the body alone
    renderers[R_MAIN] = draws({}, M);
    if (!fixed.empty()) {
      renderers[R_OCCLUSION] = draws({{&fixed, B}}, M);
      used[R_OCCLUSION] = true;
    }
  }
  for (auto& r : renderers) REQUIRE(r.draws.size() <= 64, M3T_ERR_UNSUPPORTED, "more than 63 associated bodies");
  // clip range of a renderer: widened to every body it draws (Model::AddBodiesToRenderer model.cpp:164-192)
  auto clip_range = [&](const Renderer& r, float* z_min, float* z_max) {
    *z_min = radius - d * 0.5f;
    *z_max = radius + d * 0.5f;
    for (const Draw& dr : r.draws) {
      const float db = ctx->body_geometries[dr.body]->maximum_body_diameter;
      *z_min = std::min(*z_min, radius - db * 0.5f);
      *z_max = std::max(*z_max, radius + db * 0.5f);
    }
  };
  auto projection = [&](float z_min, float z_max) {
    P4 P;
    for (float& f : P.m) f = 0.0f;  // FullRenderer::CalculateProjectionMatrix renderer.cpp:257-264
    P(0, 0) = 2.0f * fu / float(S);
    P(0, 2) = 2.0f * (pp + 0.5f) / float(S) - 1.0f;
    P(1, 1) = 2.0f * fu / float(S);
    P(1, 2) = 2.0f * (pp + 0.5f) / float(S) - 1.0f;
    P(2, 2) = (z_max + z_min) / (z_max - z_min);
    P(2, 3) = -2.0f * z_max * z_min / (z_max - z_min);
    P(3, 2) = 1.0f;
    return P;
  };
  {
    float z_min, z_max;
    for (int r = 0; r < N_RENDERERS; ++r)
      if (used[r]) {
        clip_range(renderers[r], &z_min, &z_max);
        REQUIRE(z_min >= radius * 0.2f, M3T_ERR_INVALID_ARGUMENT,  // kMinimumClipSpaceRatio model.h:59
                "z_min too small for the model's sphere radius");
      }
  }
  P4 g2b;
  std::memcpy(g2b.m, g.geometry2body, 64);
  const std::vector<P4> poses = GeodesicPoses(p->n_divides, radius);
  const int n_views = int(poses.size());
  const int pf = region ? M3T_REGION_POINT_FLOATS : M3T_DEPTH_POINT_FLOATS;
  std::vector<float> points(size_t(n_views) * p->n_points * pf), orientations(size_t(n_views) * 3), extents(n_views);
  const size_t px = size_t(S) * S;
  const int batch = int(std::max<size_t>(1, std::min<size_t>(16, (size_t(768) << 20) / (px * 8))));
  DevMem d_z, d_trans, d_depth, d_tri, d_body;
  HIPCHK(d_z.alloc(size_t(batch) * px * 8));
  HIPCHK(d_trans.alloc(size_t(batch) * 64));
  HIPCHK(d_depth.alloc(size_t(batch) * px * 2));
  if (!region) HIPCHK(d_tri.alloc(size_t(batch) * px * 4));
  if (any_associated) HIPCHK(d_body.alloc(size_t(batch) * px));
  std::vector<uint16_t> h_depth(size_t(batch) * px);
  std::vector<int> h_tri(region ? 0 : size_t(batch) * px);
  std::vector<uint8_t> h_body(any_associated ? size_t(batch) * px : 0);
  std::vector<uint8_t> h_ids[N_RENDERERS];  // silhouette images of the renderers in use (only with associated bodies)
  if (any_associated)
    for (int r = 0; r < N_RENDERERS; ++r)
      if (used[r]) h_ids[r].resize(size_t(batch) * px);
  std::vector<P4> trans(batch), geometry2camera(batch);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float main_z_min, main_z_max;
  clip_range(renderers[R_MAIN], &main_z_min, &main_z_max);
  for (int first = 0; first < n_views; first += batch) {
    const int n = std::min(batch, n_views - first);
    for (int k = 0; k < n; ++k) geometry2camera[k] = MulAffine(InverseAffine(poses[first + k]), g2b);  // body at the identity pose
    for (int r = 0; r < N_RENDERERS; ++r) {
      if (!used[r]) continue;
      if (r == R_BACKGROUND && used[R_SAME_REGION]) continue;  // same bodies as the same-region renderer: ids only
      if (!any_associated && r != R_MAIN) continue;
      float z_min, z_max;
      clip_range(renderers[r], &z_min, &z_max);
      const P4 P = projection(z_min, z_max);
      HIPCHK(hipMemsetAsync(d_z.p, 0xff, size_t(n) * px * 8, ctx->stream));
      for (size_t order = 0; order < renderers[r].draws.size(); ++order) {
        const BodyGeometryH& bg = *ctx->body_geometries[renderers[r].draws[order].body];
        P4 bg2b;
        std::memcpy(bg2b.m, bg.geometry2body, 64);
        for (int k = 0; k < n; ++k) trans[k] = MulGeneral(P, MulAffine(InverseAffine(poses[first + k]), bg2b));
        HIPCHK(hipMemcpyAsync(d_trans.p, trans.data(), size_t(n) * 64, hipMemcpyHostToDevice, ctx->stream));
        ModelRenderDev job{};
        job.vertices = bg.vertices.as<float>();
        job.triangles = bg.triangles.as<int>();
        job.n_triangles = bg.n_triangles;
        job.culling = bg.culling;
        job.image_size = S;
        job.order = int(order);
        job.trans = d_trans.as<float>();
        job.z_buffer = d_z.as<unsigned long long>();
        const int slices = std::max(1, std::min(64, (bg.n_triangles + M3T_BLOCK_THREADS - 1) / M3T_BLOCK_THREADS));
        hipLaunchKernelGGL(model_render_kernel, dim3(slices, n), dim3(M3T_BLOCK_THREADS), 0, ctx->stream, job);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(ctx->stream));  // `trans` is reused by the next body
      }
      const bool is_main = r == R_MAIN;
      hipLaunchKernelGGL(model_unpack_kernel, dim3(1024), dim3(M3T_BLOCK_THREADS), 0, ctx->stream,
                         d_z.as<unsigned long long>(), size_t(n) * px, is_main ? d_depth.as<uint16_t>() : nullptr,
                         (is_main && !region) ? d_tri.as<int>() : static_cast<int*>(nullptr),
                         any_associated ? d_body.as<uint8_t>() : static_cast<uint8_t*>(nullptr));
      HIPCHK(hipGetLastError());
      if (is_main) {
        HIPCHK(hipMemcpyAsync(h_depth.data(), d_depth.p, size_t(n) * px * 2, hipMemcpyDeviceToHost, ctx->stream));
        if (!region) HIPCHK(hipMemcpyAsync(h_tri.data(), d_tri.p, size_t(n) * px * 4, hipMemcpyDeviceToHost, ctx->stream));
      }
      if (any_associated) HIPCHK(hipMemcpyAsync(h_body.data(), d_body.p, size_t(n) * px, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      if (any_associated) {  // draw order -> id, for this renderer and the one that shares its bodies
        auto to_ids = [&](int which) {
          uint8_t lut[256];
          for (int i = 0; i < 256; ++i) lut[i] = 0;
          for (size_t o = 0; o < renderers[which].draws.size(); ++o) lut[o] = renderers[which].draws[o].id;
          for (size_t i = 0; i < size_t(n) * px; ++i) h_ids[which][i] = lut[h_body[i]];
        };
        to_ids(r);
        if (r == R_SAME_REGION && used[R_BACKGROUND]) to_ids(R_BACKGROUND);
      }
    }
    std::vector<std::thread> workers;
    for (int k = 0; k < n; ++k)
      workers.emplace_back([&, k]() {
        View v;
        v.S = S;
        v.fu = fu;
        v.pp = pp;
        v.term_a = main_z_max * main_z_min * 65535.0f / (main_z_max - main_z_min);  // renderer.cpp:475-478
        v.term_b = main_z_max * 65535.0f / (main_z_max - main_z_min);
        v.depth = h_depth.data() + size_t(k) * px;
        v.triangle = region ? nullptr : h_tri.data() + size_t(k) * px;
        if (any_associated) {
          auto image = [&](int which) { return used[which] ? h_ids[which].data() + size_t(k) * px : nullptr; };
          if (region) v.main_id = image(R_MAIN);
          v.occlusion = image(R_OCCLUSION);
          v.same_region = image(R_SAME_REGION);
          v.foreground = image(R_FOREGROUND);
          v.background = image(R_BACKGROUND);
        }
        const int view = first + k;
        const P4& c2b = poses[view];
        float* out = points.data() + size_t(view) * p->n_points * pf;
        if (region)
          RegionViewData(v, c2b, radius, p->n_points, p->max_radius_depth_offset, p->stride_depth_offset, out,
                         &extents[view]);
        else
          DepthViewData(v, c2b, geometry2camera[k], g.h_vertices, g.h_triangles, radius, p->n_points,
                        p->max_radius_depth_offset, p->stride_depth_offset, out, &extents[view]);
        for (int c = 0; c < 3; ++c) orientations[size_t(view) * 3 + c] = c2b(c, 2);
      });
    for (auto& w : workers) w.join();
  }
  return CreateModel(ctx, region, n_views, p->n_points, points.data(), orientations.data(), extents.data(),
                     p->stride_depth_offset, p->max_radius_depth_offset);
}
int m3t_hip_region_model_generate(m3t_hip_context* ctx, int body, const m3t_model_generation_params* p) {
  return GenerateModel(ctx, true, body, p);
}
int m3t_hip_depth_model_generate(m3t_hip_context* ctx, int body, const m3t_model_generation_params* p) {
  return GenerateModel(ctx, false, body, p);
}
int m3t_hip_region_model_generate_associated(m3t_hip_context* ctx, int body, const m3t_model_generation_params* p, int n,
                                             const int* body_ids, const int* movable, const int* same_region) {
  CHECK_CTX();
  REQUIRE(n >= 0 && (n == 0 || (body_ids && movable && same_region)), M3T_ERR_INVALID_ARGUMENT, "bad associated bodies");
  std::vector<int> groups[4];
  for (int i = 0; i < n; ++i) groups[(movable[i] ? 1 : 0) + (same_region[i] ? 2 : 0)].push_back(body_ids[i]);
  return GenerateModel(ctx, true, body, p, groups);
}
int m3t_hip_depth_model_generate_occluded(m3t_hip_context* ctx, int body, const m3t_model_generation_params* p, int n,
                                          const int* body_ids) {
  CHECK_CTX();
  REQUIRE(n >= 0 && (n == 0 || body_ids), M3T_ERR_INVALID_ARGUMENT, "bad occlusion bodies");
  std::vector<int> groups[4];
  groups[0].assign(body_ids, body_ids + n);
  return GenerateModel(ctx, false, body, p, groups);
}
static int GetViews(m3t_hip_context* ctx, bool region, int id, float* points, float* orientations, float* extents) {
  CHECK_CTX();
  auto& models = region ? ctx->region_models : ctx->depth_models;
  REQUIRE(id >= 0 && id < int(models.size()), M3T_ERR_INVALID_ARGUMENT, "bad model id");
  HIPCHK(hipSetDevice(ctx->device));
  const Model& m = *models[id];
  if (points) HIPCHK(hipMemcpy(points, m.points.p, size_t(m.n_views) * m.n_points * m.point_floats * 4, hipMemcpyDeviceToHost));
  if (orientations) HIPCHK(hipMemcpy(orientations, m.orientations.p, size_t(m.n_views) * 12, hipMemcpyDeviceToHost));
  if (extents) HIPCHK(hipMemcpy(extents, m.extents.p, size_t(m.n_views) * 4, hipMemcpyDeviceToHost));
  return M3T_OK;
}
int m3t_hip_region_model_get_views(m3t_hip_context* ctx, int id, float* points, float* orientations, float* extents) {
  return GetViews(ctx, true, id, points, orientations, extents);
}
int m3t_hip_depth_model_get_views(m3t_hip_context* ctx, int id, float* points, float* orientations, float* extents) {
  return GetViews(ctx, false, id, points, orientations, extents);
}

int m3t_hip_link_create(m3t_hip_context* ctx, int body, int parent, const float body2joint[16],
                        const float joint2parent[16], const int free_directions[6], int fixed_body2joint_pose) {
  CHECK_CTX();
  REQUIRE(body >= -1 && body < int(ctx->body_poses.size() / 16), M3T_ERR_INVALID_ARGUMENT, "bad body id");
  REQUIRE(parent >= -1 && parent < int(ctx->links.size()), M3T_ERR_INVALID_ARGUMENT, "bad parent link id");
  HIPCHK(hipSetDevice(ctx->device));
  int r = PullLinks(ctx);
  if (r) return r;
  Link l;
  l.body = body;
  l.parent = parent;
  std::memcpy(l.body2joint, body2joint ? body2joint : kIdentity, 64);
  std::memcpy(l.joint2parent, joint2parent ? joint2parent : kIdentity, 64);
  if (body >= 0) {
    r = SyncPosesToHost(ctx);
    if (r) return r;
    std::memcpy(l.link2world, &ctx->body_poses[size_t(body) * 16], 64);
  } else {
    std::memcpy(l.link2world, kIdentity, 64);
  }
  bool all_free = true;
  if (free_directions)
    for (int i = 0; i < 6; ++i) {
      l.free_directions[i] = free_directions[i] != 0;
      all_free &= free_directions[i] != 0;
    }
  l.fixed_body2joint_pose = fixed_body2joint_pose != 0;
  l.simple = body >= 0 && parent < 0 && IsIdentity(body2joint) && IsIdentity(joint2parent) && all_free;
  ctx->links.push_back(l);
  int id = int(ctx->links.size()) - 1;
  if (parent >= 0) ctx->links[parent].children.push_back(id);
  ctx->tables_dirty = true;
  return id;
}
int m3t_hip_link_add_modality(m3t_hip_context* ctx, int link, int modality) {
  CHECK_CTX();
  REQUIRE(link >= 0 && link < int(ctx->links.size()) && modality >= 0 && modality < int(ctx->modalities.size()),
          M3T_ERR_INVALID_ARGUMENT, "bad ids");
  const ModalityRef& ref = ctx->modalities[modality];
  int mbody = ref.region ? ctx->region_mods[ref.index]->body : ctx->depth_mods[ref.index]->body;
  REQUIRE(mbody == ctx->links[link].body, M3T_ERR_INVALID_ARGUMENT, "modality and link refer to different bodies");
  ctx->links[link].modalities.push_back(modality);
  ctx->tables_dirty = true;
  return M3T_OK;
}
int m3t_hip_optimizer_create(m3t_hip_context* ctx, int root_link, float tr, float tt) {
  CHECK_CTX();
  REQUIRE(root_link >= 0 && root_link < int(ctx->links.size()), M3T_ERR_INVALID_ARGUMENT, "bad root link");
  Optimizer o;
  o.link = root_link;
  o.tr = tr;
  o.tt = tt;
  ctx->optimizers.push_back(o);
  ctx->tables_dirty = true;
  return int(ctx->optimizers.size()) - 1;
}
int m3t_hip_optimizer_create_rigid(m3t_hip_context* ctx, int body, int n, const int* mids, float tr, float tt) {
  CHECK_CTX();
  REQUIRE(body >= 0, M3T_ERR_INVALID_ARGUMENT, "bad body id");
  int link = m3t_hip_link_create(ctx, body, -1, nullptr, nullptr, nullptr, 1);
  if (link < 0) return link;
  for (int i = 0; i < n; ++i) {
    int r = m3t_hip_link_add_modality(ctx, link, mids[i]);
    if (r < 0) return r;
  }
  return m3t_hip_optimizer_create(ctx, link, tr, tt);
}
// m3t::Constraint (constraint.h): hard constraint between two links of one structure
int m3t_hip_constraint_create(m3t_hip_context* ctx, int optimizer, int link1, int link2, const float b1[16],
                              const float b2[16], const int dirs[6]) {
  CHECK_CTX();
  REQUIRE(optimizer >= 0 && optimizer < int(ctx->optimizers.size()) && link1 >= 0 && link2 >= 0 &&
              link1 < int(ctx->links.size()) && link2 < int(ctx->links.size()) && dirs,
          M3T_ERR_INVALID_ARGUMENT, "bad constraint arguments");
  ConstraintH c;
  c.link1 = link1;
  c.link2 = link2;
  std::memcpy(c.body12joint1, b1 ? b1 : kIdentity, 64);
  std::memcpy(c.body22joint2, b2 ? b2 : kIdentity, 64);
  for (int i = 0; i < 6; ++i) c.directions[i] = dirs[i] != 0;
  ctx->constraints.push_back(c);
  ctx->optimizers[optimizer].constraints.push_back(int(ctx->constraints.size()) - 1);
  ctx->tables_dirty = true;
  return int(ctx->constraints.size()) - 1;
}
int m3t_hip_color_histograms_create(m3t_hip_context* ctx, int n_bins, float learning_rate_f, float learning_rate_b) {
  CHECK_CTX();
  int bitshift;
  switch (n_bins) {  // color_histograms.cpp:131-158
    case 2: bitshift = 7; break;
    case 4: bitshift = 6; break;
    case 8: bitshift = 5; break;
    case 16: bitshift = 4; break;
    case 32: bitshift = 3; break;
    case 64: bitshift = 2; break;
    default: return Fail(ctx, M3T_ERR_INVALID_ARGUMENT, "n_bins has to be of value 2, 4, 8, 16, 32, or 64");
  }
  HIPCHK(hipSetDevice(ctx->device));
  auto h = std::make_unique<SharedHistogramsH>();
  h->n_bins = n_bins;
  h->bitshift = bitshift;
  h->learning_rate_f = learning_rate_f;
  h->learning_rate_b = learning_rate_b;
  const size_t bins3 = size_t(n_bins) * n_bins * n_bins;
  HIPCHK(h->hist_f.alloc(bins3 * 4));
  HIPCHK(h->hist_b.alloc(bins3 * 4));
  HIPCHK(h->hist_norm.alloc(bins3 * 8));
  HIPCHK(h->counts.alloc(bins3 * 8));
  HIPCHK(hipMemset(h->counts.p, 0, bins3 * 8));
  std::vector<float> u(bins3, 1.0f / float(bins3)), nrm(bins3 * 2, 0.5f);  // SetUpHistograms color_histograms.cpp:160-172
  HIPCHK(hipMemcpy(h->hist_f.p, u.data(), bins3 * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->hist_b.p, u.data(), bins3 * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->hist_norm.p, nrm.data(), bins3 * 8, hipMemcpyHostToDevice));
  ctx->shared_histograms.push_back(std::move(h));
  ctx->tables_dirty = true;
  return int(ctx->shared_histograms.size()) - 1;
}
int m3t_hip_region_modality_use_shared_color_histograms(m3t_hip_context* ctx, int modality, int histograms) {
  CHECK_CTX();
  RegionMod* m = GetRegion(ctx, modality);
  REQUIRE(m && histograms >= 0 && histograms < int(ctx->shared_histograms.size()), M3T_ERR_INVALID_ARGUMENT,
          "bad modality / histograms id");
  const SharedHistogramsH& h = *ctx->shared_histograms[histograms];
  m->shared_histograms = histograms;
  m->dev.n_bins = h.n_bins;
  m->dev.bitshift = h.bitshift;
  m->dev.learning_rate_f = h.learning_rate_f;
  m->dev.learning_rate_b = h.learning_rate_b;
  m->dev.histogram_f = h.hist_f.as<float>();
  m->dev.histogram_b = h.hist_b.as<float>();
  m->dev.histogram_norm = h.hist_norm.as<float2>();
  m->dev.shared_counts = h.counts.as<unsigned long long>();
  ctx->tables_dirty = true;
  return M3T_OK;
}
int m3t_hip_soft_constraint_create(m3t_hip_context* ctx, int optimizer, int link1, int link2, const float b1[16],
                                   const float b2[16], const int dirs[6], float max_distance_rotation,
                                   float max_distance_translation, float standard_deviation_rotation,
                                   float standard_deviation_translation) {
  CHECK_CTX();
  REQUIRE(optimizer >= 0 && optimizer < int(ctx->optimizers.size()) && link1 >= 0 && link2 >= 0 &&
              link1 < int(ctx->links.size()) && link2 < int(ctx->links.size()) && dirs,
          M3T_ERR_INVALID_ARGUMENT, "bad soft constraint arguments");
  REQUIRE(standard_deviation_rotation > 0.0f && standard_deviation_translation > 0.0f, M3T_ERR_INVALID_ARGUMENT,
          "standard deviations must be positive");
  SoftConstraintH c{};
  c.joint.link1 = link1;
  c.joint.link2 = link2;
  std::memcpy(c.joint.body12joint1, b1 ? b1 : kIdentity, 64);
  std::memcpy(c.joint.body22joint2, b2 ? b2 : kIdentity, 64);
  for (int i = 0; i < 6; ++i) c.joint.directions[i] = dirs[i] != 0;
  c.max_distance_rotation = max_distance_rotation;
  c.max_distance_translation = max_distance_translation;
  c.standard_deviation_rotation = standard_deviation_rotation;
  c.standard_deviation_translation = standard_deviation_translation;
  ctx->soft_constraints.push_back(c);
  ctx->optimizers[optimizer].soft_constraints.push_back(int(ctx->soft_constraints.size()) - 1);
  ctx->tables_dirty = true;
  return int(ctx->soft_constraints.size()) - 1;
}
int m3t_hip_link_get_link2world_pose(m3t_hip_context* ctx, int link, float pose[16]) {
  CHECK_CTX();
  REQUIRE(link >= 0 && link < int(ctx->links.size()) && pose, M3T_ERR_INVALID_ARGUMENT, "bad link id");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->links[link].body >= 0) return m3t_hip_body_get_body2world_pose(ctx, ctx->links[link].body, pose);
  int r = PullLinks(ctx);
  if (r) return r;
  std::memcpy(pose, ctx->links[link].link2world, 64);
  return M3T_OK;
}
int m3t_hip_link_set_link2world_pose(m3t_hip_context* ctx, int link, const float pose[16]) {
  CHECK_CTX();
  REQUIRE(link >= 0 && link < int(ctx->links.size()) && pose, M3T_ERR_INVALID_ARGUMENT, "bad link id");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->links[link].body >= 0) return m3t_hip_body_set_body2world_pose(ctx, ctx->links[link].body, pose);
  int r = PullLinks(ctx);
  if (r) return r;
  std::memcpy(ctx->links[link].link2world, pose, 64);
  ctx->tables_dirty = true;
  return M3T_OK;
}
int m3t_hip_link_set_joint_poses(m3t_hip_context* ctx, int link, const float body2joint[16],
                                 const float joint2parent[16]) {
  CHECK_CTX();
  REQUIRE(link >= 0 && link < int(ctx->links.size()), M3T_ERR_INVALID_ARGUMENT, "bad link id");
  HIPCHK(hipSetDevice(ctx->device));
  int r = PullLinks(ctx);
  if (r) return r;
  Link& l = ctx->links[link];
  if (body2joint) std::memcpy(l.body2joint, body2joint, 64);
  if (joint2parent) std::memcpy(l.joint2parent, joint2parent, 64);
  if (!IsIdentity(l.body2joint) || !IsIdentity(l.joint2parent)) l.simple = false;
  ctx->tables_dirty = true;
  return M3T_OK;
}
int m3t_hip_link_get_joint_poses(m3t_hip_context* ctx, int link, float body2joint[16], float joint2parent[16]) {
  CHECK_CTX();
  REQUIRE(link >= 0 && link < int(ctx->links.size()), M3T_ERR_INVALID_ARGUMENT, "bad link id");
  HIPCHK(hipSetDevice(ctx->device));
  int r = PullLinks(ctx);
  if (r) return r;
  if (body2joint) std::memcpy(body2joint, ctx->links[link].body2joint, 64);
  if (joint2parent) std::memcpy(joint2parent, ctx->links[link].joint2parent, 64);
  return M3T_OK;
}

// ---- tracker sub-steps -------------------------------------------------------------------
int m3t_hip_tracker_set_iterations(m3t_hip_context* ctx, int n_corr, int n_update) {
  CHECK_CTX();
  REQUIRE(n_corr >= 0 && n_update >= 0, M3T_ERR_INVALID_ARGUMENT, "bad iteration counts");
  if (ctx->roi_enabled && n_corr != ctx->n_corr_iterations) ctx->tables_dirty = true;  // (the search-pose slots per object)
  ctx->n_corr_iterations = n_corr;
  ctx->n_update_iterations = n_update;
  return M3T_OK;
}
int m3t_hip_set_fused_step(m3t_hip_context* ctx, int mode) {
  CHECK_CTX();
  REQUIRE(mode >= 0 && mode <= 2, M3T_ERR_INVALID_ARGUMENT, "mode must be 0, 1 or 2");
  ctx->fused_mode = mode;
  return M3T_OK;
}

int m3t_hip_set_object_split(m3t_hip_context* ctx, int enable) {
  CHECK_CTX();
  REQUIRE(enable >= 0 && enable <= M3T_SPLIT_MAX_PARTS, M3T_ERR_INVALID_ARGUMENT, "0 (off), 1 (automatic) or the largest number of workgroups per object (2..16)");
  ctx->split_enabled = enable != 0;
  ctx->split_parts_override = enable > 1 ? enable : 0;
  return M3T_OK;
}

int m3t_hip_start_modalities(m3t_hip_context* ctx, int iteration) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  for (auto& m : ctx->region_mods) {  // RegionModality::StartModality :378
    if (m->dev.first_iteration != iteration) {
      m->dev.first_iteration = iteration;
      ctx->tables_dirty = true;
    }
  }
  int r = Prepare(ctx, true);
  if (r) return r;
  if (ctx->table_overflow_host) {  // new histograms: the LDS pair table gets its chance again
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *static_cast<volatile unsigned*>(ctx->table_overflow_host) = 0u;
  }
  if ((r = RenderForModalities(ctx, true))) return r;  // start_modality_renderer_ptrs tracker.cpp:430-436
  return LaunchHistogram(ctx, iteration, true);
}
int m3t_hip_calculate_correspondences(m3t_hip_context* ctx, int iteration, int corr_iteration) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  int r = Prepare(ctx, true);
  if (r) return r;
  ctx->state_valid = true;
  if ((r = RenderForModalities(ctx, false))) return r;  // correspondence_renderer_ptrs tracker.cpp:447-452
  return LaunchCorrespondences(ctx, iteration, corr_iteration);
}
int m3t_hip_calculate_gradient_and_hessian(m3t_hip_context* ctx, int, int corr_iteration, int opt_iteration) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  int r = Prepare(ctx, false);
  if (r) return r;
  REQUIRE(ctx->state_valid, M3T_ERR_NOT_SET_UP, "CalculateCorrespondences has to be called first");
  return LaunchGradientHessian(ctx, corr_iteration, opt_iteration);
}
int m3t_hip_calculate_optimization(m3t_hip_context* ctx, int, int, int) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  int r = Prepare(ctx, false);
  if (r) return r;
  return LaunchOptimization(ctx);  // (with a communicator: link sums, ONE all-reduce, project + solve)
}
// Optimizer::CalculateOptimization split at the multi-GPU exchange point: *partial is a
// DEVICE pointer to `count` floats (the 6 + 36 gradient / Hessian sums of every link of every structure, zero for
// links whose modalities live in another process); sum it over the ranks that hold bodies of the structures (one
// RCCL all-reduce on the context stream), then call _end.
int m3t_hip_calculate_optimization_begin(m3t_hip_context* ctx, float** partial, size_t* count) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  int r = Prepare(ctx, false);
  if (r) return r;
  if (!ctx->tree_mode) {  // rigid bodies only: force the general path so that the sums exist
    ctx->tree_mode = true;
    ctx->fused_possible = false;
    r = UploadTreeTables(ctx);
    if (r) return r;
  }
  r = LaunchGather(ctx);
  if (r) return r;
  if (partial) *partial = ctx->d_link_sums.as<float>();
  if (count) *count = ctx->link_sums_count;
  return M3T_OK;
}
int m3t_hip_calculate_optimization_end(m3t_hip_context* ctx) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  REQUIRE(ctx->partial_ready, M3T_ERR_NOT_SET_UP, "calculate_optimization_begin has to be called first");
  return LaunchSolveSums(ctx);
}
// ---- RCCL: the one collective of the path (a kinematic structure spread over GPUs) ----
#define RCCLCHK(expr)                                                                                      \
  do {                                                                                                     \
    ncclResult_t _r = (expr);                                                                              \
    if (_r != ncclSuccess)                                                                                 \
      return Fail(ctx, M3T_ERR_DEVICE, std::string(#expr) + ": " +                                         \
                                           (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "RCCL error")); \
  } while (0)
int m3t_hip_comm_get_unique_id(m3t_hip_context* ctx, void* id, size_t bytes) {
  CHECK_CTX();
  REQUIRE(id && bytes >= sizeof(ncclUniqueId), M3T_ERR_INVALID_ARGUMENT, "the id buffer must hold 128 bytes");
  REQUIRE(g_rccl.Load(), M3T_ERR_DEVICE, g_rccl.error);
  ncclUniqueId u;
  RCCLCHK(g_rccl.GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return M3T_OK;
}
int m3t_hip_comm_init_rank(m3t_hip_context* ctx, const void* id, size_t bytes, int n_ranks, int rank) {
  CHECK_CTX();
  REQUIRE(id && bytes >= sizeof(ncclUniqueId) && n_ranks >= 1 && rank >= 0 && rank < n_ranks, M3T_ERR_INVALID_ARGUMENT,
          "bad communicator description");
  REQUIRE(g_rccl.Load(), M3T_ERR_DEVICE, g_rccl.error);
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->comm && ctx->comm_owned) RCCLCHK(g_rccl.CommDestroy(ctx->comm));
  ctx->comm = nullptr;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclComm_t comm = nullptr;
  RCCLCHK(g_rccl.CommInitRank(&comm, n_ranks, u, rank));
  ctx->comm = comm;
  ctx->comm_owned = true;
  // (soft constraints need no care here: every rank adds them AFTER the all-reduce of the link sums, as one process
  // does -- links_solve_sums_kernel)
  return M3T_OK;
}
int m3t_hip_comm_set(m3t_hip_context* ctx, void* nccl_comm) {
  CHECK_CTX();
  REQUIRE(nccl_comm == nullptr || g_rccl.Load(), M3T_ERR_DEVICE, g_rccl.error);
  if (ctx->comm && ctx->comm_owned) RCCLCHK(g_rccl.CommDestroy(ctx->comm));
  ctx->comm = static_cast<ncclComm_t>(nccl_comm);
  ctx->comm_owned = false;
  return M3T_OK;
}
int m3t_hip_comm_destroy(m3t_hip_context* ctx) {
  CHECK_CTX();
  if (ctx->comm && ctx->comm_owned) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    RCCLCHK(g_rccl.CommDestroy(ctx->comm));
  }
  ctx->comm = nullptr;
  ctx->comm_owned = false;
  return M3T_OK;
}
// The host's own transport in the collective's place: fn(user, device buffer, count, hipStream_t) is called wherever
// ncclAllReduce would be, and has to leave the sum over the host's ranks in the buffer, ordered on that stream (it may
// synchronise the stream and do the sum on the host).  While it is set the context takes the distributed paths exactly
// as with a communicator.  NULL takes it out again.
int m3t_hip_comm_set_reduce_callback(m3t_hip_context* ctx, m3t_hip_reduce_fn fn, void* user) {
  CHECK_CTX();
  ctx->reduce_fn = fn;
  ctx->reduce_user = fn ? user : nullptr;
  return M3T_OK;
}
// sum of the stacked link sums of all structures over the ranks of the communicator: ONE ncclAllReduce on the
// context's stream, in place (Link::CalculateGradientAndHessian link.cpp:184-193 is the sum being distributed)
int m3t_hip_calculate_optimization_allreduce(m3t_hip_context* ctx) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  return AllReducePartial(ctx);
}
// Tracker::CalculateConsistentPoses tracker.cpp:423 -> Optimizer::CalculateConsistentPoses optimizer.cpp:135
int m3t_hip_calculate_consistent_poses(m3t_hip_context* ctx) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  int r = Prepare(ctx, false);
  if (r) return r;
  if (!ctx->tree_mode) return M3T_OK;  // T * [I | 0] == T exactly for a free rigid body
  return LaunchSolve(ctx, true);
}
int m3t_hip_calculate_results(m3t_hip_context* ctx, int iteration) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  int r = Prepare(ctx, true);
  if (r) return r;
  if ((r = RenderForModalities(ctx, true))) return r;  // results_renderer_ptrs tracker.cpp:503-509
  return LaunchHistogram(ctx, iteration, false);
}

// tracking_step_split_kernel / _split_render_kernel: how many workgroups per object (0: none) for a batch of n.
// parts x padded elements per part = 256 (the collecting threads of split_exchange_state); the grid must fit the GPU
// with every workgroup resident at once (their in-kernel exchange needs that).
extern "C++" {
template <typename K>
static int ChooseSplitParts(Ctx* ctx, K kernel, int n, int threads, bool want_fused_histogram, size_t* lds_out,
                     int default_limit = 8) {
  const size_t lds_tracking = size_t(ctx->layout.off_hist >= 0 ? ctx->layout.off_hist : ctx->layout.total_floats) * 4;
  // (each workgroup counts its share of the histogram bins: that share of the count table; the pair table is
  // read from L2, never staged)
  auto lds_split_for = [&](int p) {
    return want_fused_histogram ? std::max(lds_tracking, M3T_MISC_FLOATS * 4 + (ctx->lds_hist - M3T_MISC_FLOATS * 4) / p)
                                : lds_tracking;
  };
  // (default_limit: 8 for the one-launch step -- 16 was not faster there in round 2; 16 for the per-search launches of
  // renderer-fed steps, measured 0.643 -> 0.635 ms for one object)
  int limit = ctx->split_parts_override > 1 ? ctx->split_parts_override : default_limit;
  if (const char* e = std::getenv("M3T_HIP_SPLIT_PARTS")) limit = std::atoi(e);  // developer override
  const int elements = std::max(ctx->layout.nl, ctx->depth_mods.empty() ? 1 : ctx->np_max);
  for (int p = M3T_SPLIT_MAX_PARTS; p >= 2; p >>= 1) {
    // 256-thread workgroups (developer override): two are resident per CU if their LDS fits twice
    const int per_cu = (threads == M3T_SPLIT_LANES && lds_split_for(p) * 2 <= size_t(160) * 1024) ? 2 : 1;
    const int padded = (n + 7) / 8 * 8;  // grid blocks / p: every XCD gets the blocks of the fullest one
    if (p > limit || padded * p > ctx->compute_cus * per_cu) continue;
    if ((elements + p - 1) / p > M3T_SPLIT_LANES / p) continue;  // a part's elements fit its share of the lanes
    // the exchange needs every workgroup of the grid resident at once: ask the runtime how many of these
    // workgroups (registers, LDS) a CU takes, instead of assuming the LDS arithmetic above is the only limit
    int resident = ResidentBlocks(ctx, kernel, threads, lds_split_for(p));
    if (resident > per_cu) resident = per_cu;  // (the query is known to over-report by one block for SGPR-heavy kernels)
    if (resident < 1 || padded * p > ctx->compute_cus * resident) continue;
    *lds_out = lds_split_for(p);
    return p;
  }
  return 0;
}
}  // extern "C++"
// the exchange buffers and the per-launch descriptor of a split launch
static int PrepareSplit(Ctx* ctx, int n, int parts, SplitParams* out) {
  const size_t per_object = size_t(2) * M3T_SPLIT_LANES * 32;  // granules
  if (!ctx->split_abort_host) {
    void* host = nullptr;
    HIPCHK(hipHostMalloc(&host, 64, hipHostMallocMapped));
    std::memset(host, 0, 64);
    void* dev = nullptr;
    HIPCHK(hipHostGetDevicePointer(&dev, host, 0));
    ctx->split_abort_host = static_cast<unsigned*>(host);
    ctx->split_abort_dev = static_cast<unsigned*>(dev);
  }
  if (ctx->split_objects < size_t(n) || ctx->split_seq >= (1u << 26) - 1) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->split_objects < size_t(n)) {
      HIPCHK(ctx->d_split.alloc(size_t(n) * per_object * sizeof(unsigned long long) + size_t(n) * sizeof(unsigned)));
      ctx->split_objects = size_t(n);
    }
    HIPCHK(hipMemset(ctx->d_split.p, 0, ctx->d_split.bytes));
    ctx->split_seq = 0;
  }
  ++ctx->split_seq;
  SplitParams sp{};
  sp.granules = ctx->d_split.as<unsigned long long>();
  sp.object_abort = reinterpret_cast<unsigned*>(sp.granules + ctx->split_objects * per_object);
  sp.host_abort = ctx->split_abort_dev;
  sp.seq = ctx->split_seq;
  if (++ctx->split_launches == 0) ctx->split_launches = 1;  // (0 = the word's initial value)
  sp.abort_id = ctx->split_launches;
  sp.n_objects = n;
  sp.n_parts = parts;
  sp.lshift = 0;
  while ((parts << sp.lshift) < M3T_SPLIT_LANES) ++sp.lshift;
  sp.per_part_lines = (ctx->layout.nl + parts - 1) / parts;
  sp.per_part_points = (ctx->np_max + parts - 1) / parts;
  *out = sp;
  return M3T_OK;
}

int m3t_hip_execute_tracking_step(m3t_hip_context* ctx, int iteration) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  const bool untracked_before = ctx->untracked_launches;
  const bool host_poses_before = ctx->poses_dirty_host;  // (Prepare uploads them)
  int r = Prepare(ctx, true);
  if (r) return r;
  ctx->untracked_launches = untracked_before;  // a whole step is tracked by its step_done event below
  if ((r = CheckSplitExchange(ctx))) return r;  // an earlier step that was abandoned on the device
  bool histogram_fused = false;
  const bool rigid_fused = ctx->fused_mode >= 1 && ctx->fused_possible && !ctx->Distributed();
  bool roi_frames = false;  // does this step read a slot that holds a rectangle only
  for (auto& cam : ctx->cameras) roi_frames = roi_frames || cam->slot_is_roi[cam->current];
  REQUIRE(!roi_frames || (rigid_fused && ctx->n_roi_items > 0), M3T_ERR_UNSUPPORTED,
          "a frame slot holds the trackers' rectangle only (ROI ingest): that needs the fused step of rigid objects");
  const bool roi_active = ctx->roi_enabled && ctx->n_roi_items > 0 && rigid_fused;
  float* const roi_snapshots = ctx->d_roi_pose_snapshot.as<float>();
  const size_t roi_snapshot_floats = ctx->d_roi_pose_snapshot.bytes / (4 * Ctx::kRoiSnapshots);
  if (roi_active) {
    // the poses the NEXT frame's rectangles are computed from (m3t_hip_cameras_upload_batch_roi_async, copy stream):
    // the poses this step starts from.  The previous step's end-of-step snapshot holds exactly those -- and was ready
    // before this step had to wait for its own frame, so the next pull can follow the current one at once -- unless
    // something else touched the poses in between.  The snapshot before that one (adaptive margins) is the start of
    // the previous step.
    if (ctx->roi_end_valid && !host_poses_before && !untracked_before) {
      ctx->roi_prev = ctx->roi_use;
      ctx->roi_use = ctx->roi_end_index;
    } else {
      ctx->roi_prev = -1;
      ctx->roi_use = 0;
      HIPCHK(hipMemcpyAsync(roi_snapshots, ctx->d_poses.p, ctx->body_poses.size() * 4, hipMemcpyDeviceToDevice, ctx->stream));
      HIPCHK(hipEventRecord(ctx->roi_snapshot_done[0], ctx->stream));
    }
    ctx->roi_snapshot_valid = true;
  }
  ctx->roi_end_valid = false;
  ctx->roi_recorded = roi_active;
  // a step that reads rectangles runs the guarded kernels (m3t_kernels.hip, roi_guard_outside): an object whose poses
  // leave what was uploaded is not committed but flagged; roi_repair_kernel then fetches the whole frames of the
  // flagged objects' cameras and the step is launched again for the flagged objects alone
  RoiGuardArgs guard{};
  if (roi_frames) {
    guard.items = ctx->d_roi_items.as<RoiItemDev>();
    guard.rects = ctx->d_roi_rects.as<m3t_roi_rect>();
    guard.n_cams = int(ctx->cameras.size());
    guard.n_rect_slots = ctx->roi_rect_slots;
    guard.n_poses = ctx->roi_n_poses;
    guard.mode = 1;
    guard.misses = ctx->roi_miss_dev;
    guard.miss_capacity = int(Ctx::kRoiMissCapacity);
    guard.unrecovered = ctx->roi_unrecovered_dev;
  }
  if (rigid_fused) {  // (a communicator: the structures span GPUs)
    int n = int(ctx->opt_table.size());
    ScopedKernelTimer timer(ctx, 0);
    // From two objects per CU on (and if two working sets fit the CU's LDS) the kernel runs with 256-thread
    // workgroups, two per CU: one object's serial solve overlaps the other's parallel phases and no register is
    // spilled (measured, pose-updates/s: 512 objects 1.11 M vs 0.87 M with 512 threads, 4096: 1.24 M vs 0.91 M;
    // 128-VGPR variants of the 512-thread kernel reached 1.03 M / 1.11 M)
    int threads = M3T_BLOCK_THREADS;
    if (n >= 2 * ctx->compute_cus && ctx->lds_track * 2 <= 160 * 1024) threads = M3T_BLOCK_THREADS / 2;
    if (const char* e = std::getenv("M3T_HIP_THREADS")) threads = std::atoi(e);  // developer override
    // Batches with region AND depth modalities: the _pair_ kernels (m3t_kernels.hip, PAIR: the two modalities' products
    // side by side, their sums on two waves).  M3T_HIP_NO_PAIR: developer override.
    const bool pair = !roi_frames && !ctx->region_mods.empty() && !ctx->depth_mods.empty() && !std::getenv("M3T_HIP_NO_PAIR");
    auto split_kernel = pair ? tracking_step_split_pair_kernel : tracking_step_split_kernel;
    auto kernel = pair ? (ctx->layout.off_hist >= 0 ? tracking_step_lds_pair_kernel : tracking_step_pair_kernel)
                       : (ctx->layout.off_hist >= 0 ? tracking_step_lds_kernel : tracking_step_kernel);
    // One workgroup per CU: the histogram update (CalculateResults) runs at the end of the same launch, its
    // count table taking over the line buffers' LDS.  With two workgroups per CU that table (128 KB at 32 bins)
    // would not fit twice, so large batches keep the separate region_histogram_kernel.
    const bool want_fused_histogram = ctx->fuse_histogram_possible && !std::getenv("M3T_HIP_NO_FUSED_HISTOGRAM");
    histogram_fused = want_fused_histogram && threads == M3T_BLOCK_THREADS;
    const size_t lds = histogram_fused ? std::max(ctx->lds_track, ctx->lds_hist) : ctx->lds_track;
    // Batches that leave CUs idle: several workgroups per object, each on its own CU (all resident at once, which
    // their in-kernel exchange needs; a wait that runs out abandons the object's step, see CheckSplitExchange).
    // parts x padded elements per part = 256 (the collecting threads of split_exchange_state).
    int parts = 0;
    size_t lds_split = 0;
    if (ctx->split_possible && ctx->split_enabled && threads % M3T_SPLIT_LANES == 0 &&
        ctx->n_corr_iterations < 64 && !std::getenv("M3T_HIP_NO_SPLIT"))
      parts = roi_frames ? ChooseSplitParts(ctx, tracking_step_split_guard_kernel, n, threads, want_fused_histogram, &lds_split)
                         : ChooseSplitParts(ctx, split_kernel, n, threads,
                                            want_fused_histogram, &lds_split);
    const bool split = parts >= 2;
    // More objects than CUs: the compact kernel (<= 47 KB of LDS, <= 128 VGPRs per object: 3-4 workgroups per CU;
    // measured crossover on 256 CUs: 256 objects 0.249 vs 0.225 ms with one 512-thread workgroup per CU, 384 objects
    // 0.309 vs 0.426 ms).  M3T_HIP_COMPACT=0 / 1: developer override (never / whenever possible).
    bool compact = !split && ctx->compact_possible && ctx->fused_mode == 1 && n > ctx->compute_cus;
    if (const char* e = std::getenv("M3T_HIP_COMPACT")) compact = !split && ctx->compact_possible && ctx->fused_mode == 1 && std::atoi(e) != 0;
    if (std::getenv("M3T_HIP_THREADS")) compact = false;
    // Region-only batches with >= 1024-bin histograms: the pair table compacted in LDS (round 6; 4096 objects 1.96 ->
    // 1.78 ms).  While the mixed bins of every object fit the table, that is; histograms that outgrow it by more than
    // half its size (the kernels report it through a mapped word) go back to the kernel that gathers from L2.
    // M3T_HIP_COMPACT_TABLE=0 / 1: developer override.
    // (Region + Depth batches keep the plain kernel: measured with the table, synth512 0.769 / 0.784 vs 0.778 ms -- their
    // 16-bin pair table is 32 KB and sits in the L1 / L2 anyway, and the depth scan is most of their step)
    bool compact_table = compact && !roi_frames && ctx->lds_compact_table > 0 && ctx->depth_mods.empty();
    // (three workgroups per CU instead of four: a batch that is ONE round of the plain kernel but not of this one keeps
    // the plain kernel -- 1024 objects on 256 CUs: 0.556 vs 0.583 ms; 384: 0.314 / 0.290, 512: 0.325 / 0.298, 2048:
    // 1.041 / 0.935, 4096: 1.964 / 1.769)
    if (n > 3 * ctx->compute_cus && n <= 4 * ctx->compute_cus) compact_table = false;
    if (compact_table && !ctx->table_overflow_host) {
      void *host = nullptr, *dev = nullptr;
      HIPCHK(hipHostMalloc(&host, 64, hipHostMallocMapped));
      std::memset(host, 0, 64);
      HIPCHK(hipHostGetDevicePointer(&dev, host, 0));
      ctx->table_overflow_host = static_cast<unsigned*>(host);
      ctx->table_overflow_dev = static_cast<unsigned*>(dev);
    }
    if (compact_table && *static_cast<volatile unsigned*>(ctx->table_overflow_host) > unsigned(ctx->compact_table.table_cap) / 2)
      compact_table = false;
    if (const char* e = std::getenv("M3T_HIP_COMPACT_TABLE")) compact_table = compact_table && std::atoi(e) != 0;
    if (compact_table) ctx->compact_table.table_overflow = ctx->table_overflow_dev;
    // Batches with depth modalities: 512-thread workgroups, two per CU (round 6).  Their step is the depth scan -- sixteen
    // lanes per point, 200 points: 12.5 rounds of a 256-thread workgroup, half of that here --, the registers and the LDS
    // per object stay, 16 waves per CU instead of 12.  Measured (Region + Depth, YCB parameters, ms per step, 256 / 512
    // threads): 257 objects 0.732 / 0.577, 512: 0.790 / 0.635, 640: 0.970 / 0.949, 700: 0.986 / 1.058, 768: 0.996 / 1.083,
    // 900: 1.473 / 1.168, 1024: 1.567 / 1.216, 2048: 2.640 / 2.410, 4096: 5.097 / 4.736 -- the 256-thread kernel keeps the
    // batches that are ONE round of its three workgroups per CU but not of two.  Region-only batches: 384 / 512 objects
    // 0.314 / 0.295 and 0.324 / 0.306 ms, behind the LDS pair table's 0.286 / 0.292 -- not taken.
    // M3T_HIP_COMPACT_WIDE=0 / 1: developer override.
    bool compact_wide = compact && !roi_frames && !compact_table && !ctx->depth_mods.empty() &&
                        !(2 * n > 5 * ctx->compute_cus && n <= 3 * ctx->compute_cus);
    if (const char* e = std::getenv("M3T_HIP_COMPACT_WIDE"))
      compact_wide = compact && !roi_frames && !compact_table && std::atoi(e) != 0;
    if (roi_frames)
      ctx->last_step_kernel = split ? "tracking_step_split_guard_kernel"
                                    : (compact ? "tracking_step_compact_guard_kernel"
                                               : (ctx->layout.off_hist >= 0 ? "tracking_step_lds_guard_kernel" : "tracking_step_guard_kernel"));
    else
    ctx->last_step_kernel = split ? (pair ? "tracking_step_split_pair_kernel" : "tracking_step_split_kernel")
                                  : (compact ? (compact_table ? "tracking_step_compact_table_kernel"
                                                              : (compact_wide ? "tracking_step_compact_wide_kernel" : "tracking_step_compact_kernel"))
                                             : (ctx->layout.off_hist >= 0 ? (pair ? "tracking_step_lds_pair_kernel" : "tracking_step_lds_kernel")
                                                                          : (pair ? "tracking_step_pair_kernel" : "tracking_step_kernel")));
    if (compact) {
      threads = compact_wide ? 2 * M3T_COMPACT_THREADS : M3T_COMPACT_THREADS;
      histogram_fused = want_fused_histogram && ctx->compact_fuses_histogram;
    } else if (split) {
      histogram_fused = want_fused_histogram;
    }
    // pass 0: the step; passes 1 (repair) and 2 (the flagged objects again, on whole frames) only behind rectangles
    for (int pass = 0; pass < (roi_frames ? 2 : 1); ++pass) {
      if (pass == 1) {
        for (size_t sl = 0; sl < ctx->roi_sources.size(); ++sl)
          for (const Ctx::RoiSource& source : ctx->roi_sources[sl]) {
            const Camera& c0 = *ctx->cameras[source.ids[0]];
            if (c0.current != int(sl) || !c0.slot_is_roi[sl]) continue;  // (this step reads another slot of these cameras)
            m3t_roi_rect* rects = ctx->d_roi_rects.as<m3t_roi_rect>() + sl * ctx->cameras.size();
            hipLaunchKernelGGL(roi_repair_kernel, dim3((c0.intr.height + 7) / 8, unsigned(source.ids.size())), dim3(256), 0,
                               ctx->stream, source.d_ids, ctx->d_roi_items.as<RoiItemDev>(), ctx->d_roi_item_first.as<int>(),
                               ctx->d_opts.as<RigidOptDev>(), ctx->roi_n_poses, rects, source.src, source.camera_stride,
                               source.row_step, c0.frame(int(sl)), c0.frame_bytes, c0.pitch, c0.intr.width, c0.intr.height,
                               c0.is_depth ? 2 : 3);
          }
        guard.mode = 2;
      }
      if (compact) {
        if (roi_frames)
          hipLaunchKernelGGL(tracking_step_compact_guard_kernel, dim3(n), dim3(threads), ctx->lds_compact, ctx->stream,
                             ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                             ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(), ctx->compact,
                             iteration, ctx->n_corr_iterations, ctx->n_update_iterations, histogram_fused ? 1 : 0, guard);
        else if (compact_table)
          hipLaunchKernelGGL(tracking_step_compact_table_kernel, dim3(n), dim3(threads), ctx->lds_compact_table, ctx->stream,
                             ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                             ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(), ctx->compact_table,
                             iteration, ctx->n_corr_iterations, ctx->n_update_iterations, histogram_fused ? 1 : 0);
        else if (compact_wide)
          hipLaunchKernelGGL(tracking_step_compact_wide_kernel, dim3(n), dim3(threads), ctx->lds_compact, ctx->stream,
                             ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                             ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(), ctx->compact,
                             iteration, ctx->n_corr_iterations, ctx->n_update_iterations, histogram_fused ? 1 : 0);
        else
          hipLaunchKernelGGL(tracking_step_compact_kernel, dim3(n), dim3(threads), ctx->lds_compact, ctx->stream,
                             ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                             ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(), ctx->compact,
                             iteration, ctx->n_corr_iterations, ctx->n_update_iterations, histogram_fused ? 1 : 0);
      } else if (split) {
        SplitParams sp{};
        if ((r = PrepareSplit(ctx, n, parts, &sp))) return r;
        if (roi_frames)
          hipLaunchKernelGGL(tracking_step_split_guard_kernel, dim3((n + 7) / 8 * 8 * parts), dim3(threads), lds_split,
                             ctx->stream, ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                             ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(),
                             ctx->layout, ctx->off_points, ctx->np_max, iteration, ctx->n_corr_iterations,
                             ctx->n_update_iterations, ctx->fused_mode == 2 ? 1 : 0, histogram_fused ? 1 : 0, sp, guard);
        else
          hipLaunchKernelGGL(split_kernel, dim3((n + 7) / 8 * 8 * parts), dim3(threads), lds_split, ctx->stream,
                             ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                             ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(),
                             ctx->layout, ctx->off_points, ctx->np_max, iteration, ctx->n_corr_iterations,
                             ctx->n_update_iterations, ctx->fused_mode == 2 ? 1 : 0, histogram_fused ? 1 : 0, sp);
      } else if (roi_frames) {
        hipLaunchKernelGGL(ctx->layout.off_hist >= 0 ? tracking_step_lds_guard_kernel : tracking_step_guard_kernel, dim3(n),
                           dim3(threads), lds, ctx->stream, ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                           ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(),
                           ctx->layout, ctx->off_points, ctx->np_max, iteration, ctx->n_corr_iterations,
                           ctx->n_update_iterations, ctx->fused_mode == 2 ? 1 : 0, histogram_fused ? 1 : 0, guard);
      } else {
        hipLaunchKernelGGL(kernel, dim3(n), dim3(threads), lds, ctx->stream,
                           ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                           ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(),
                           ctx->layout, ctx->off_points, ctx->np_max, iteration, ctx->n_corr_iterations,
                           ctx->n_update_iterations, ctx->fused_mode == 2 ? 1 : 0, histogram_fused ? 1 : 0, 0);
      }
      HIPCHK(hipGetLastError());
    }
    // (where the repair fetched a whole frame only the device's rectangle table knows it: for the host the slot keeps
    // counting as a rectangle slot -- conservative -- until the next upload into it)
    ctx->last_step_shape[0] = n;
    ctx->last_step_shape[1] = split ? parts : 1;
    ctx->last_step_shape[2] = threads;
    ctx->last_step_shape[3] = histogram_fused ? 1 : 0;
    ctx->state_valid = ctx->fused_mode == 2;
  } else if (ctx->fused_mode >= 1 && ctx->fused_per_search_possible && !ctx->Distributed() && !ctx->tree_mode &&
             !std::getenv("M3T_HIP_NO_SEARCH_FUSION")) {
    // Renderer-fed branches (modelled occlusions, region / silhouette checking): the focused renderings are redrawn
    // before every correspondence search from the bodies' current poses (correspondence_renderer_ptrs,
    // tracker.cpp:447-452), so the loop nest runs one search per launch -- the renderers, then ONE launch for the
    // search and its Newton steps of all objects -- instead of one launch per sub-step: 4 instead of 11 per search.
    ScopedKernelTimer timer(ctx, 0);
    const int n = int(ctx->opt_table.size());
    auto kernel = ctx->layout.off_hist >= 0 ? tracking_step_lds_kernel : tracking_step_kernel;
    // a batch that leaves CUs idle: several workgroups per object here too (tracking_step_split_render_kernel: the
    // split kernel with the renderer-fed branches compiled in, one search per launch)
    int parts = 0;
    size_t lds_split = 0;
    bool shared = false;
    for (auto& m : ctx->region_mods) shared = shared || m->shared_histograms >= 0 || m->p.n_histogram_bins < 4;
    if (!shared && ctx->split_enabled && ctx->n_corr_iterations < 64 && !std::getenv("M3T_HIP_NO_SPLIT"))
      parts = ChooseSplitParts(ctx, tracking_step_split_render_kernel, n, M3T_BLOCK_THREADS, false, &lds_split, 16);
    for (int c = 0; c < ctx->n_corr_iterations; ++c) {
      if ((r = RenderForModalities(ctx, false))) return r;
      if (parts >= 2) {
        SplitParams sp{};
        if ((r = PrepareSplit(ctx, n, parts, &sp))) return r;
        hipLaunchKernelGGL(tracking_step_split_render_kernel, dim3((n + 7) / 8 * 8 * parts), dim3(M3T_BLOCK_THREADS),
                           lds_split, ctx->stream, ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                           ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(), ctx->layout,
                           ctx->off_points, ctx->np_max, iteration, ctx->n_update_iterations,
                           ctx->fused_mode == 2 ? 1 : 0, c, sp);
      } else {
        hipLaunchKernelGGL(kernel, dim3(n), dim3(M3T_BLOCK_THREADS), ctx->lds_track, ctx->stream,
                           ctx->d_opts.as<RigidOptDev>(), ctx->d_region.as<RegionModDev>(),
                           ctx->d_depth.as<DepthModDev>(), ctx->cams_active, ctx->d_poses.as<float>(), ctx->layout,
                           ctx->off_points, ctx->np_max, iteration, 1, ctx->n_update_iterations,
                           ctx->fused_mode == 2 ? 1 : 0, 0, c);
      }
    }
    HIPCHK(hipGetLastError());
    ctx->last_step_kernel = parts >= 2 ? "tracking_step_split_render_kernel"
                                       : (ctx->layout.off_hist >= 0 ? "tracking_step_lds_kernel" : "tracking_step_kernel");
    ctx->last_step_shape[0] = n;
    ctx->last_step_shape[1] = parts >= 2 ? parts : 1;
    ctx->last_step_shape[2] = M3T_BLOCK_THREADS;
    ctx->last_step_shape[3] = 0;
    ctx->state_valid = ctx->fused_mode == 2;
  } else if (TreeStepFused(ctx)) {
    // kinematic structures: the whole loop nest in one launch, one workgroup per link that carries modalities
    ScopedKernelTimer timer(ctx, 0);
    const bool want_fused_histogram = !ctx->region_mods.empty() && ctx->hist_counts_in_lds && ctx->shared_histograms.empty() &&
                                      !std::getenv("M3T_HIP_NO_FUSED_HISTOGRAM");
    const size_t lds = TreeStepLds(ctx, want_fused_histogram);
    if (!ctx->split_abort_host) {
      void* host = nullptr;
      HIPCHK(hipHostMalloc(&host, 64, hipHostMallocMapped));
      std::memset(host, 0, 64);
      void* dev = nullptr;
      HIPCHK(hipHostGetDevicePointer(&dev, host, 0));
      ctx->split_abort_host = static_cast<unsigned*>(host);
      ctx->split_abort_dev = static_cast<unsigned*>(dev);
    }
    if (ctx->tree_seq >= (1u << 26) - 1) {  // the tags restart: clear them first
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipMemset(ctx->d_tree_exchange.p, 0, ctx->d_tree_exchange.bytes));
      ctx->tree_seq = 0;
    }
    TreeStepParams xp{};
    xp.seq = ++ctx->tree_seq;
    if (++ctx->split_launches == 0) ctx->split_launches = 1;
    xp.abort_id = ctx->split_launches;
    xp.host_abort = ctx->split_abort_dev;
    const int off_tree = int((lds / 4 - ctx->tree_block_floats));
    // Structures that leave CUs idle (the 8-body chain: 8 workgroups on 256 CUs): several workgroups per tracked link
    // (tracking_step_tree_split_kernel; open structures only -- the constrained kernel keeps one).  The parts of a link
    // exchange their lines' results, so all workgroups must be resident: checked like the split kernel's launch.
    int tree_parts = 0;
    if (!ctx->tree_constrained && ctx->n_corr_iterations < 64 && !std::getenv("M3T_HIP_NO_TREE_SPLIT")) {
      bool can = true;
      for (auto& m : ctx->region_mods) can = can && m->shared_histograms < 0 && m->p.n_histogram_bins >= 4;
      const int elements = std::max(ctx->layout.nl, ctx->depth_mods.empty() ? 1 : ctx->np_max);
      int limit = 8;  // (the 8-body chain, ms per step: 1 part 0.3046, 2: 0.2939, 4: 0.2756, 8: 0.2649; profiles/r05_chain8_parts.txt)
      if (const char* e = std::getenv("M3T_HIP_TREE_PARTS")) limit = std::atoi(e);  // developer override
      for (int p = M3T_SPLIT_MAX_PARTS; can && p >= 2; p >>= 1) {
        if (p > limit || (elements + p - 1) / p > M3T_SPLIT_LANES / p) continue;
        if (ctx->tree_split_lds_attribute != lds) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(tracking_step_tree_split_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)) != hipSuccess) {
            (void)hipGetLastError();
            break;
          }
          ctx->tree_split_lds_attribute = lds;
        }
        int resident = ResidentBlocks(ctx, tracking_step_tree_split_kernel, M3T_BLOCK_THREADS, lds);
        resident = std::min(resident, int(size_t(160) * 1024 / lds));
        if (resident >= 1 && ctx->n_treesteps * p <= ctx->compute_cus * resident) { tree_parts = p; break; }
      }
    }
    if (tree_parts >= 2) {
      SplitParams sp{};
      if ((r = PrepareSplit(ctx, ctx->n_treesteps, tree_parts, &sp))) return r;
      hipLaunchKernelGGL(tracking_step_tree_split_kernel, dim3(ctx->n_treesteps * tree_parts), dim3(M3T_BLOCK_THREADS), lds,
                         ctx->stream, ctx->d_treesteps.as<TreeStepDev>(), ctx->d_treeopts.as<TreeOptDev>(),
                         ctx->d_region.as<RegionModDev>(), ctx->d_depth.as<DepthModDev>(), ctx->cams_active,
                         ctx->d_poses.as<float>(), ctx->layout, ctx->off_points, ctx->np_max, off_tree, iteration,
                         ctx->n_corr_iterations, ctx->n_update_iterations, want_fused_histogram ? 1 : 0, xp, sp);
    } else
    hipLaunchKernelGGL(ctx->tree_constrained ? tracking_step_tree_constrained_kernel : tracking_step_tree_kernel,
                       dim3(ctx->n_treesteps), dim3(M3T_BLOCK_THREADS), lds, ctx->stream,
                       ctx->d_treesteps.as<TreeStepDev>(), ctx->d_treeopts.as<TreeOptDev>(),
                       ctx->d_region.as<RegionModDev>(), ctx->d_depth.as<DepthModDev>(), ctx->cams_active,
                       ctx->d_poses.as<float>(), ctx->layout, ctx->off_points, ctx->np_max, off_tree, iteration,
                       ctx->n_corr_iterations, ctx->n_update_iterations, want_fused_histogram ? 1 : 0, xp);
    HIPCHK(hipGetLastError());
    histogram_fused = want_fused_histogram;
    ctx->links_device_newer = true;
    ctx->last_step_kernel = tree_parts >= 2 ? "tracking_step_tree_split_kernel"
                                            : (ctx->tree_constrained ? "tracking_step_tree_constrained_kernel" : "tracking_step_tree_kernel");
    ctx->last_step_shape[0] = ctx->n_treesteps;
    ctx->last_step_shape[1] = tree_parts >= 2 ? tree_parts : 1;
    ctx->last_step_shape[2] = M3T_BLOCK_THREADS;
    ctx->last_step_shape[3] = histogram_fused ? 1 : 0;
    ctx->state_valid = false;
  } else if (TreeStepSegmented(ctx)) {
    // kinematic structures spread over processes: one launch and one all-reduce (of the link sums) per Newton step,
    // a last launch for the last solve, the bodies and the histogram update (m3t_links.hip, tree_segment_body)
    ScopedKernelTimer timer(ctx, 0);
    const bool want_fused_histogram = !ctx->region_mods.empty() && ctx->hist_counts_in_lds && ctx->shared_histograms.empty() &&
                                      !std::getenv("M3T_HIP_NO_FUSED_HISTOGRAM");
    const size_t lds = TreeStepLds(ctx, want_fused_histogram);
    const int off_tree = int((lds / 4 - ctx->tree_block_floats));
    auto kernel = ctx->tree_constrained ? tracking_step_tree_segment_constrained_kernel : tracking_step_tree_segment_kernel;
    float* sums[2] = {ctx->d_link_sums.as<float>(), ctx->d_link_sums_alt.as<float>()};
    const int n_newton = ctx->n_corr_iterations * ctx->n_update_iterations;
    // launch k reads the link table launch k - 1 wrote: the primary table at k <= 1 (launch 0 solves nothing and writes
    // nothing), then alternately; launch k >= 1 writes the other one.  The last launch (k = n_newton) must leave the
    // joints in the primary table, the one the rest of the library reads: an odd n_newton ends in the second one and
    // is copied over.
    // FIRST (the links take their bodies' poses) and FINAL (the bodies take their links') meet in one launch only when
    // the frame has a single Newton step: a workgroup that starts late would then seed from poses a faster one has
    // already written back, so that launch seeds from a snapshot taken in front of the loop
    const float* first_poses = ctx->d_poses.as<float>();
    if (n_newton == 1) {
      if (ctx->d_poses_first.bytes < ctx->d_poses.bytes) HIPCHK(ctx->d_poses_first.alloc(ctx->d_poses.bytes));
      HIPCHK(hipMemcpyAsync(ctx->d_poses_first.p, ctx->d_poses.p, ctx->body_poses.size() * 4, hipMemcpyDeviceToDevice,
                            ctx->stream));
      first_poses = ctx->d_poses_first.as<float>();
    }
    for (int k = 0; k <= n_newton; ++k) {
      TreeSegmentParams sp{};
      const int c = k / std::max(ctx->n_update_iterations, 1), u = k % std::max(ctx->n_update_iterations, 1);
      const bool last = k == n_newton;
      int flags = 0;
      if (k <= 1) flags |= TSEG_FIRST;  // (nothing has written the bodies' link2world yet: launch 0 solves nothing)
      if (k > 0) flags |= TSEG_SOLVE;
      if (!last) {
        flags |= TSEG_SUMS;
        if (u == 0) flags |= TSEG_SEARCH | (ctx->n_update_iterations > 1 ? TSEG_STORE_STATE : 0);
        else flags |= TSEG_LOAD_STATE;
      } else {
        flags |= TSEG_FINAL;
      }
      const bool read_alt = k >= 2 && (k % 2 == 0);
      const bool write_alt = k >= 1 && (k % 2 == 1);
      if (read_alt) flags |= TSEG_LINKS_FROM_ALT;
      if (write_alt) flags |= TSEG_LINKS_TO_ALT;
      sp.flags = flags;
      sp.corr_iteration = last ? 0 : c;
      sp.opt_iteration = last ? 0 : u;
      sp.sums_in = sums[(k + 1) & 1];  // what launch k - 1 wrote and the all-reduce summed
      sp.sums_out = sums[k & 1];
      sp.first_poses = first_poses;
      hipLaunchKernelGGL(kernel, dim3(ctx->n_treesteps), dim3(M3T_BLOCK_THREADS), lds, ctx->stream,
                         ctx->d_treesteps.as<TreeStepDev>(), ctx->d_treeopts.as<TreeOptDev>(),
                         ctx->d_region.as<RegionModDev>(), ctx->d_depth.as<DepthModDev>(), ctx->cams_active,
                         ctx->d_poses.as<float>(), ctx->layout, ctx->off_points, ctx->np_max, off_tree, iteration,
                         (last && want_fused_histogram) ? 1 : 0, sp);
      HIPCHK(hipGetLastError());
      if (!last) {
        ctx->partial_ready = true;
        if ((r = AllReducePartial(ctx, sums[k & 1]))) return r;
        ctx->partial_ready = false;
      }
    }
    if (n_newton >= 1 && (n_newton % 2 == 1))  // the last launch wrote the second table
      HIPCHK(hipMemcpyAsync(ctx->d_links.p, ctx->d_links_alt.p, ctx->d_links.bytes, hipMemcpyDeviceToDevice, ctx->stream));
    histogram_fused = want_fused_histogram;
    ctx->links_device_newer = true;
    ctx->last_step_kernel = ctx->tree_constrained ? "tracking_step_tree_segment_constrained_kernel" : "tracking_step_tree_segment_kernel";
    ctx->last_step_shape[0] = ctx->n_treesteps;
    ctx->last_step_shape[1] = 1;
    ctx->last_step_shape[2] = M3T_BLOCK_THREADS;
    ctx->last_step_shape[3] = histogram_fused ? 1 : 0;
    ctx->state_valid = false;
  } else {
    ctx->last_step_kernel = "";
    // Tracker::ExecuteTrackingStep tracker.cpp:344-364, one launch per sub-step
    for (int c = 0; c < ctx->n_corr_iterations; ++c) {
      if ((r = RenderForModalities(ctx, false))) return r;
      if ((r = LaunchCorrespondences(ctx, iteration, c))) return r;
      for (int u = 0; u < ctx->n_update_iterations; ++u) {
        if ((r = LaunchGradientHessian(ctx, c, u))) return r;
        if ((r = LaunchOptimization(ctx))) return r;
      }
    }
    ctx->state_valid = true;
  }
  if (!histogram_fused) {
    if ((r = RenderForModalities(ctx, true))) return r;
    if ((r = LaunchHistogram(ctx, iteration, false, roi_frames))) return r;
  }
  if (roi_active) {  // the poses the step ends on = the poses the next one starts from
    // (the third buffer: a rectangle upload that is still to run on the copy stream reads roi_use and roi_prev)
    ctx->roi_end_index = 0;
    while (ctx->roi_end_index == ctx->roi_use || ctx->roi_end_index == ctx->roi_prev) ++ctx->roi_end_index;
    HIPCHK(hipMemcpyAsync(roi_snapshots + size_t(ctx->roi_end_index) * roi_snapshot_floats, ctx->d_poses.p,
                          ctx->body_poses.size() * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->roi_snapshot_done[ctx->roi_end_index], ctx->stream));
    ctx->roi_end_valid = true;
  }
  if (ctx->async_ingest) {
    // remember which frame slots this step reads, so that a later asynchronous upload into one of
    // them waits for exactly this step and not for the ones enqueued after it
    HIPCHK(hipEventRecord(ctx->step_done[ctx->step_counter % Ctx::kStepEvents], ctx->stream));
    for (auto& cam : ctx->cameras) cam->last_read_step[cam->current] = ctx->step_counter;
    ++ctx->step_counter;
  }
  return M3T_OK;
}
// Refiner::RefinePoses src/refiner.cpp:76-117 (one launch per sub-step)
int m3t_hip_refine_poses(m3t_hip_context* ctx, int n_corr_iterations, int n_update_iterations) {
  CHECK_CTX();
  REQUIRE(n_corr_iterations >= 0 && n_update_iterations >= 0, M3T_ERR_INVALID_ARGUMENT, "bad iteration counts");
  int r = m3t_hip_calculate_consistent_poses(ctx);
  if (r) return r;
  for (int c = 0; c < n_corr_iterations; ++c) {
    if ((r = m3t_hip_start_modalities(ctx, 0))) return r;  // StartModality(0, corr_iteration)
    if ((r = m3t_hip_calculate_correspondences(ctx, 0, c))) return r;
    for (int u = 0; u < n_update_iterations; ++u) {
      if ((r = m3t_hip_calculate_gradient_and_hessian(ctx, 0, c, u))) return r;
      if ((r = m3t_hip_calculate_optimization(ctx, 0, c, u))) return r;
    }
  }
  return M3T_OK;
}
int m3t_hip_execute_tracking_cycle(m3t_hip_context* ctx, int iteration) {
  return m3t_hip_execute_tracking_step(ctx, iteration);
}
int m3t_hip_set_kernel_timing(m3t_hip_context* ctx, int enable) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (auto& p : ctx->pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  ctx->pending.clear();
  ctx->timing = enable == 1;
  ctx->timing_region = enable == 2;
  ctx->kernel_ms[0] = ctx->kernel_ms[1] = 0.0;
  ctx->kernel_launches[0] = ctx->kernel_launches[1] = 0;
  if (ctx->timing_region) {  // the stream is idle (synchronised above): the region starts here
    if (!ctx->region_a) {
      HIPCHK(hipEventCreate(&ctx->region_a));
      HIPCHK(hipEventCreate(&ctx->region_b));
    }
    HIPCHK(hipEventRecord(ctx->region_a, ctx->stream));
  }
  return M3T_OK;
}
int m3t_hip_get_step_shape(m3t_hip_context* ctx, int shape[4]) {
  CHECK_CTX();
  REQUIRE(shape, M3T_ERR_INVALID_ARGUMENT, "null output");
  std::memcpy(shape, ctx->last_step_shape, sizeof(ctx->last_step_shape));
  return M3T_OK;
}
int m3t_hip_debug_log_checksum(m3t_hip_context* ctx, unsigned first_bits, unsigned last_bits,
                               unsigned long long out[3]) {
  CHECK_CTX();
  REQUIRE(out && first_bits <= last_bits, M3T_ERR_INVALID_ARGUMENT, "bad range");
  HIPCHK(hipSetDevice(ctx->device));
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&d), 24));
  hipError_t e = hipMemsetAsync(d, 0, 24, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(log_checksum_kernel, dim3(ctx->prop.multiProcessorCount * 8), dim3(256), 0, ctx->stream,
                       first_bits, last_bits, d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d, 24, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  HIPCHK(e);
  return M3T_OK;
}
int m3t_hip_comm_get_allreduce_count(m3t_hip_context* ctx, long long* count) {
  CHECK_CTX();
  REQUIRE(count, M3T_ERR_INVALID_ARGUMENT, "null output");
  *count = ctx->allreduce_calls;
  return M3T_OK;
}
int m3t_hip_comm_get_rank_count(m3t_hip_context* ctx, int* n_ranks) {
  CHECK_CTX();
  REQUIRE(n_ranks, M3T_ERR_INVALID_ARGUMENT, "null output");
  *n_ranks = 0;
  if (!ctx->comm) return M3T_OK;
  REQUIRE(g_rccl.Load() && g_rccl.CommCount, M3T_ERR_UNSUPPORTED, "ncclCommCount is not available in the loaded librccl");
  const ncclResult_t rc = g_rccl.CommCount(ctx->comm, n_ranks);
  if (rc != ncclSuccess)
    return Fail(ctx, M3T_ERR_DEVICE,
                std::string("ncclCommCount: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
  return M3T_OK;
}
int m3t_hip_get_step_kernel(m3t_hip_context* ctx, char* name, size_t capacity) {
  CHECK_CTX();
  REQUIRE(name && capacity > 0, M3T_ERR_INVALID_ARGUMENT, "null output");
  std::snprintf(name, capacity, "%s", ctx->last_step_kernel);
  return M3T_OK;
}
int m3t_hip_get_kernel_timing(m3t_hip_context* ctx, float total_ms[2], int launches[2]) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->timing_region) {  // device time from the enabling call to here, all of it booked on the tracking launches
    HIPCHK(hipEventRecord(ctx->region_b, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0.0f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->region_a, ctx->region_b));
    if (total_ms) { total_ms[0] = ms; total_ms[1] = 0.0f; }
    if (launches) { launches[0] = ctx->kernel_launches[0]; launches[1] = ctx->kernel_launches[1]; }
    return M3T_OK;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (auto& p : ctx->pending) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->kernel_ms[p.which] += ms;
      ctx->kernel_launches[p.which] += 1;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  ctx->pending.clear();
  if (total_ms) { total_ms[0] = float(ctx->kernel_ms[0]); total_ms[1] = float(ctx->kernel_ms[1]); }
  if (launches) { launches[0] = ctx->kernel_launches[0]; launches[1] = ctx->kernel_launches[1]; }
  return M3T_OK;
}
#ifdef M3T_PHASE_TIMING
int m3t_hip_debug_phase_cycles(m3t_hip_context* ctx, unsigned long long* out24, int reset) {
  CHECK_CTX();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_phase_cycles), 32 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[32] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)));
  }
  return M3T_OK;
}
int m3t_hip_debug_exchange_times(m3t_hip_context* ctx, unsigned long long* out768) {
  CHECK_CTX();
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpyFromSymbol(out768, HIP_SYMBOL(g_exchange_times), 768 * sizeof(unsigned long long)));
  return M3T_OK;
}
#endif
int m3t_hip_sync(m3t_hip_context* ctx) {
  CHECK_CTX();
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return CheckSplitExchange(ctx);
}

}  // extern "C"
