// m3t_kernels.hip — gfx950 (CDNA4, wave64) kernels for M3T's per-frame pose
// optimisation hot path.  Hand-written HIP, no MFMA (gather + reduce work),
// one workgroup per tracked object, all per-object intermediates in LDS.
//
// Kernel inventory:
//   region_histogram_kernel   StartModality / CalculateResults: contour-line colour
//                             sampling into an LDS count table + histogram blend
//   region_correspondence_kernel (+ _lds_: pair table staged in LDS for <= 16 bins)
//                             RegionModality::CalculateCorrespondences
//   region_gradient_hessian_kernel RegionModality::CalculateGradientAndHessian
//   depth_correspondence_kernel / depth_gradient_hessian_kernel  DepthModality
//   rigid_optimize_kernel     Optimizer::CalculateOptimization (dof 6) + Link::UpdatePoses
//   tracking_step_kernel (+ _lds_); launched with 256-thread workgroups, two per CU, from two objects per CU on
//                             the whole ExecuteTrackingStep loop nest fused on device
//   tracking_step_split_kernel the same with 4 / 8 / 16 workgroups per object (batches that leave CUs idle)
//   (kinematic structures: m3t_links.hip)
//
// Arithmetic follows the reference expression by expression in IEEE f32
// (compile with -ffp-contract=off; hipcc's f32 divide / sqrt are correctly
// rounded by default), so every discrete decision (pixel truncation, validity
// tests, distribution index) matches the CPU restatement bit for bit, and the
// g/H sums over lines / points are formed in the reference's order (product
// rows in LDS, one sequential chain per gradient / Hessian entry: chain_sums),
// so do the poses.  Reference citations are relative to M3T/src/.

#include <hip/hip_runtime.h>
#include <limits.h>

#include "m3t_device.h"
#include "m3t_log.h"
#include "m3t_renderer_read.h"
#include "m3t_roi.h"

#ifdef M3T_PHASE_TIMING
// developer instrumentation: accumulated s_memtime cycles per phase, block 0 thread 0
__device__ unsigned long long g_phase_cycles[32];
__device__ unsigned long long g_exchange_times[3 * 16 * 16];  // [kind][round][part] of object 0, last frame
#define EXCHANGE_STAMP(kind, round, part, object)                                                      \
  do {                                                                                                 \
    if ((object) == 0 && threadIdx.x == 0 && (round) < 16 && (part) < 16)                              \
      g_exchange_times[((kind) * 16 + (round)) * 16 + (part)] = clock64();                             \
  } while (0)
#define PHASE_T0() unsigned long long _pt = clock64()
#define PHASE_MARK(i)                                                         \
  do {                                                                        \
    unsigned long long _n = clock64();                                        \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_cycles[i] += _n - _pt;   \
    _pt = _n;                                                                 \
  } while (0)
// (a phase that runs on another wave than the first: `thread` records it)
#define PHASE_MARK_BY(i, thread)                                                        \
  do {                                                                                  \
    unsigned long long _n = clock64();                                                  \
    if (blockIdx.x == 0 && (int)threadIdx.x == (thread)) g_phase_cycles[i] += _n - _pt; \
    _pt = _n;                                                                           \
  } while (0)
#else
#define EXCHANGE_STAMP(kind, round, part, object) do {} while (0)
#define PHASE_T0() do {} while (0)
#define PHASE_MARK(i) do {} while (0)
#define PHASE_MARK_BY(i, thread) do {} while (0)
#endif

namespace {

// The per-object parameter tables are read-only while a kernel runs: address them through
// the constant address space so that uniform field reads become scalar loads (s_load -> SGPR)
// instead of per-lane global loads in the middle of every dependency chain.
typedef const __attribute__((address_space(4))) RegionModDev CRegion;
typedef const __attribute__((address_space(4))) DepthModDev CDepth;
typedef const __attribute__((address_space(4))) CameraDev CCam;
typedef const __attribute__((address_space(4))) RigidOptDev COpt;

// Pointers stored inside the parameter tables are generic; loads through them would be FLAT
// instructions with 64-bit per-lane addresses.  Re-type them as global (address space 1) so the
// compiler emits global_load with a scalar base + 32-bit lane offset.
// plain clang vectors: HIP's float4/float2 classes cannot be copied out of a non-generic address space
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <typename T>
using G = const __attribute__((address_space(1))) T*;
template <typename T>
using GW = __attribute__((address_space(1))) T*;
template <typename T>
__device__ __forceinline__ G<T> as_global(const T* p) { return (G<T>)p; }
template <typename T>
__device__ __forceinline__ GW<T> as_global_w(T* p) { return (GW<T>)p; }

constexpr int kWave = 64;
// misc LDS scratch layout (floats): [0,128) view search / counters, [128,160) reduced sums,
// [160,640) per-wave partials (16 waves x 27), [640,656) pose, [704,746) region g/H, [768,810) depth g/H
constexpr int kMiscRed = 128, kMiscPartials = 160, kMiscPose = 640, kMiscGhRegion = 704, kMiscGhDepth = 768,
              kMiscLookup = 832,  // function_lookup_f[16], function_lookup_b[16]
              kMiscLogTable = 288;  // 128 doubles: the table of m3t_log.h, misc[288, 544) (free again in the histogram tail)

// ---------------------------------------------------------------------------
// pose math (column-major like Eigen; same expression trees as the reference)
// ---------------------------------------------------------------------------
struct Affine {
  float l[9];  // linear, (r,c) = l[c*3+r]
  float t[3];
};

template <typename P>
__device__ __forceinline__ Affine load_pose(P p /*16 floats, column-major 4x4*/) {
  Affine a;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) a.l[c * 3 + r] = p[c * 4 + r];
#pragma unroll
  for (int r = 0; r < 3; ++r) a.t[r] = p[12 + r];
  return a;
}

// Transform * Transform: L = La*Lb, t = La*tb + ta
__device__ __forceinline__ Affine mul_pose(const Affine& a, const Affine& b) {
  Affine r;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      r.l[c * 3 + k] = (a.l[k] * b.l[c * 3] + a.l[3 + k] * b.l[c * 3 + 1]) + a.l[6 + k] * b.l[c * 3 + 2];
#pragma unroll
  for (int k = 0; k < 3; ++k) r.t[k] = ((a.l[k] * b.t[0] + a.l[3 + k] * b.t[1]) + a.l[6 + k] * b.t[2]) + a.t[k];
  return r;
}

// Transform * Vector3f: t + L*v
__device__ __forceinline__ void apply_pose(const Affine& a, float x, float y, float z, float& ox, float& oy,
                                           float& oz) {
  ox = a.t[0] + ((a.l[0] * x + a.l[3] * y) + a.l[6] * z);
  oy = a.t[1] + ((a.l[1] * x + a.l[4] * y) + a.l[7] * z);
  oz = a.t[2] + ((a.l[2] * x + a.l[5] * y) + a.l[8] * z);
}

// Eigen compute_inverse_size3: cofactors / determinant
__device__ __forceinline__ float cofactor3(const float* m, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[j1 * 3 + i1] * m[j2 * 3 + i2] - m[j2 * 3 + i1] * m[j1 * 3 + i2];
}
__device__ __forceinline__ void inverse3(const float* m, float* r) {
  float c00 = cofactor3(m, 0, 0), c10 = cofactor3(m, 1, 0), c20 = cofactor3(m, 2, 0);
  float det = (c00 * m[0] + c10 * m[1]) + c20 * m[2];
  float invdet = 1.0f / det;
#pragma unroll
  for (int rr = 0; rr < 3; ++rr)
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) r[cc * 3 + rr] = cofactor3(m, cc, rr) * invdet;
}
__device__ __forceinline__ Affine inverse_pose(const Affine& a) {
  Affine r;
  inverse3(a.l, r.l);
#pragma unroll
  for (int k = 0; k < 3; ++k) r.t[k] = -((r.l[k] * a.t[0] + r.l[3 + k] * a.t[1]) + r.l[6 + k] * a.t[2]);
  return r;
}

__device__ __forceinline__ int f2i(float v) { return (int)v; }  // truncation like C int(float)

__device__ __forceinline__ float i2f_bits(int v) { return __int_as_float(v); }
__device__ __forceinline__ int f2i_bits(float v) { return __float_as_int(v); }

template <typename P>
__device__ __forceinline__ auto last_valid(P values, int n, int idx) {  // common.h:171-176
  return idx < n ? values[idx] : values[n - 1];
}

// ---------------------------------------------------------------------------
// wave64 primitives on DPP (no LDS traffic): row_shr 1/2/4/8 inside each 16-lane row,
// then row_bcast:15 / row_bcast:31 across rows; the full-wave result lands in lane 63.
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_zero(float v) {  // lanes without a source read 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_zero_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_self(float v) {  // lanes without a source read themselves
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_self_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_sum_i(int v) {  // broadcast result
  v += dpp_zero_i<0x111, 0xf>(v);
  v += dpp_zero_i<0x112, 0xf>(v);
  v += dpp_zero_i<0x114, 0xf>(v);
  v += dpp_zero_i<0x118, 0xf>(v);
  v += dpp_zero_i<0x142, 0xa>(v);
  v += dpp_zero_i<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_max(float v) {  // broadcast result
  v = fmaxf(v, dpp_self<0x111, 0xf>(v));
  v = fmaxf(v, dpp_self<0x112, 0xf>(v));
  v = fmaxf(v, dpp_self<0x114, 0xf>(v));
  v = fmaxf(v, dpp_self<0x118, 0xf>(v));
  v = fmaxf(v, dpp_self<0x142, 0xa>(v));
  v = fmaxf(v, dpp_self<0x143, 0xc>(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_i(int v) {  // broadcast result
  v = min(v, dpp_self_i<0x111, 0xf>(v));
  v = min(v, dpp_self_i<0x112, 0xf>(v));
  v = min(v, dpp_self_i<0x114, 0xf>(v));
  v = min(v, dpp_self_i<0x118, 0xf>(v));
  v = min(v, dpp_self_i<0x142, 0xa>(v));
  v = min(v, dpp_self_i<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}

// ---------------------------------------------------------------------------
// ROI guard (tracking_step_*_guard_kernel): a step that reads frame slots holding only
// the trackers' rectangles checks, at every pose its searches (and its histogram lines) run at, that what a reader can
// touch there lies inside the rectangle that was uploaded.  A step that does not is not committed -- pose, modality
// state and histograms stay those before the step -- and the object is flagged: the library fetches its cameras' whole
// frames and repeats the step for the flagged objects (m3t_hip_execute_tracking_step).  No reference counterpart
// (the step either side of the path: loader_camera.cpp:76-98 hands over whole frames).
// ---------------------------------------------------------------------------
// body2camera (column-major 4 x 4) = world2camera * body2world
__device__ __forceinline__ void roi_body2camera(const float* w2c, const float* b2w, float* out) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 3; ++r)
      out[c * 4 + r] = ((w2c[r] * b2w[c * 4] + w2c[4 + r] * b2w[c * 4 + 1]) + w2c[8 + r] * b2w[c * 4 + 2]) +
                       (c == 3 ? w2c[12 + r] : 0.0f);
  out[3] = out[7] = out[11] = 0.0f;
  out[15] = 1.0f;
}
__device__ __forceinline__ m3t_intrinsics roi_intrinsics(const CameraDev& cam) {
  m3t_intrinsics k;
  k.fu = cam.fu; k.fv = cam.fv; k.ppu = cam.ppu; k.ppv = cam.ppv;
  k.width = cam.width; k.height = cam.height;
  return k;
}
// All 64 lanes of one wave: does a reader of the object (hdr: its readers) need pixels outside the rectangle in its
// camera's slot at the pose b2w (16 floats)?  Eight lanes per reader, one corner of the model's box each
// (m3t_roi_corner), joined by lane exchanges, widened as m3t_roi_body widens it.  The same answer in every lane.
__device__ __noinline__ int roi_guard_outside(const RoiItemDev* items, const m3t_roi_rect* rects, int n_cams, int n_rect_slots,
                                              const RoiGuardDev* hdr, const CameraDev* cams, const float* b2w) {
  const int lane = threadIdx.x & (kWave - 1);
  const int g = lane >> 3, corner = lane & 7;
  const bool active = g < hdr->n_items && g < M3T_ROI_GUARD_ITEMS;
  const RoiItemDev& it = items[hdr->item[active ? g : 0]];
  const CameraDev& cam = cams[it.camera];
  const m3t_intrinsics k = roi_intrinsics(cam);
  float b2c[16];
  roi_body2camera(cam.world2camera, b2w, b2c);
  float u = 0.0f, v = 0.0f, z = 0.0f;
  int front = m3t_roi_corner(b2c, it.box_min, it.box_max, corner, &k, &u, &v, &z);
  float u_min = u, u_max = u, v_min = v, v_max = v, z_min = z;
#pragma unroll
  for (int m = 1; m < 8; m <<= 1) {
    front &= __shfl_xor(front, m);
    float o;
    o = __shfl_xor(u_min, m); u_min = o < u_min ? o : u_min;
    o = __shfl_xor(u_max, m); u_max = o > u_max ? o : u_max;
    o = __shfl_xor(v_min, m); v_min = o < v_min ? o : v_min;
    o = __shfl_xor(v_max, m); v_max = o > v_max ? o : v_max;
    o = __shfl_xor(z_min, m); z_min = o < z_min ? o : z_min;
  }
  m3t_roi_rect need = {0, 0, k.width - 1, k.height - 1};  // a box that reaches the camera plane: the whole frame
  if (front) {
    m3t_roi_tighten(b2c, it.box_min, it.box_max, it.rho, &k, &u_min, &u_max, &v_min, &v_max, &z_min);
    need = m3t_roi_widen(u_min, u_max, v_min, v_max, z_min, &k, it.reach_px, it.reach_m);
  }
  bool outside = false;
  if (active && cam.slot < n_rect_slots)  // (a ring slot added since the tables were built holds whole frames only)
    outside = !m3t_roi_contains(rects[(size_t)cam.slot * n_cams + it.camera], need);
  return __builtin_amdgcn_ballot_w64(outside) != 0ull ? 1 : 0;
}
// what a workgroup does with the guard's verdict (one thread per object): mode 1 flags a miss and reports the body,
// mode 2 (the repeat over whole frames) clears the flag and reports a body that still misses
__device__ __forceinline__ void roi_guard_report(const RoiGuardArgs& a, RoiGuardDev* hdr, int body, bool miss) {
  // (2: still outside after the repeat on whole frames -- the step's separate histogram launch leaves such a body alone;
  // the next step's first pass writes the flag afresh)
  hdr->flag = miss ? (a.mode == 1 ? 1 : 2) : 0;
  if (miss) {
    int* list = a.mode == 1 ? a.misses : a.unrecovered;
    const int at = atomicAdd(&list[0], 1);
    if (at < a.miss_capacity) list[1 + at] = body;
  }
}

// The viewing direction of RegionModel::GetClosestView (region_model.cpp:113-119): o = R^-1 t / |t|; false when t = 0
// (the reference then returns the first view)
__device__ __forceinline__ bool view_direction(const Affine& b2c, float& o0, float& o1, float& o2) {
  const float tn = sqrtf((b2c.t[0] * b2c.t[0] + b2c.t[1] * b2c.t[1]) + b2c.t[2] * b2c.t[2]);
  if (tn == 0.0f) return false;
  const float tx = b2c.t[0] / tn, ty = b2c.t[1] / tn, tz = b2c.t[2] / tn;
  float ri[9];
  inverse3(b2c.l, ri);
  o0 = (ri[0] * tx + ri[3] * ty) + ri[6] * tz;
  o1 = (ri[1] * tx + ri[4] * ty) + ri[7] * tz;
  o2 = (ri[2] * tx + ri[5] * ty) + ri[8] * tz;
  return true;
}

// GetClosestView from the view of the previous search (round 4).  Between two searches the pose moves by a Newton
// step, so the closest view is the previous one or one next to it.  `neighbors` holds for every view its
// M3T_VIEW_NEIGHBORS nearest views and the cosine of the angle within which a direction is provably closer to the
// view than to any view outside that row (m3t_hip_api.hip, CreateModel).  Inside that angle the argmax over all views
// is the argmax over the row: the same dot products (same expression, same operands), the largest, the lowest index
// among equals -- the result of the full scan, bit for bit.  Outside it: -1, the caller scans all views.  Every wave
// works the row out for itself (19 lanes): no LDS, no barrier.
// LEAN (tracking_step_tree_kernel, which has no register to spare): the lane goes through an empty asm, so that the row
// offset is worked out here and not once in front of the caller's loops -- one VGPR pair less alive across them.  The
// other kernels keep the plain form: with it the 64-object step measured 2.5 % slower (r05a, same process).
template <bool LEAN = false>
__device__ __forceinline__ int closest_view_local(G<v4f> neighbors, int prev, float o0, float o1, float o2) {
  int lane = threadIdx.x & (kWave - 1);
  if constexpr (LEAN) asm volatile("" : "+v"(lane));
  const v4f e = neighbors[(uint32_t)prev * M3T_VIEW_ROW + (lane < M3T_VIEW_ROW ? lane : 0)];
  float d = (o0 * e.x + o1 * e.y) + o2 * e.z;
  const float d_prev = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 0));
  const float threshold = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e.x), M3T_VIEW_ROW - 1));
  if (!(d_prev > threshold)) return -1;  // wave-uniform (and the same in every wave: same inputs)
  d = lane <= M3T_VIEW_NEIGHBORS ? d : -2.0f;
  const float best = wave_max(d);
  return wave_min_i(d == best ? __float_as_int(e.w) : INT_MAX);
}

// ---------------------------------------------------------------------------
// RegionModel::GetClosestView (region_model.cpp:105-130), whole block.
// First maximum wins == (max dot, lowest index); result broadcast to all threads.
// orientations4: one float4 per view (xyz + pad).  misc: >= 128 floats of LDS scratch.
// Contains two __syncthreads().
// ---------------------------------------------------------------------------
__device__ __forceinline__ int closest_view(G<v4f> orientations4, int n_views, const Affine& b2c, float* misc) {
  float tn = sqrtf((b2c.t[0] * b2c.t[0] + b2c.t[1] * b2c.t[1]) + b2c.t[2] * b2c.t[2]);
  if (tn == 0.0f) return 0;  // block-uniform
  const int nt = blockDim.x;
  // the view table does not depend on the pose: its loads (up to six views per thread: 3072 views at 512 threads)
  // are in flight while the viewing direction is worked out
  constexpr int kPre = 6;
  v4f pre[kPre];
#pragma unroll
  for (int j = 0; j < kPre; ++j) {
    const int v = threadIdx.x + j * nt;
    pre[j] = orientations4[v < n_views ? v : 0];
  }
  float tx = b2c.t[0] / tn, ty = b2c.t[1] / tn, tz = b2c.t[2] / tn;
  float ri[9];
  inverse3(b2c.l, ri);
  float o0 = (ri[0] * tx + ri[3] * ty) + ri[6] * tz;
  float o1 = (ri[1] * tx + ri[4] * ty) + ri[7] * tz;
  float o2 = (ri[2] * tx + ri[5] * ty) + ri[8] * tz;
  float best = -1.0f;
  int bi = INT_MAX;
#pragma unroll
  for (int j = 0; j < kPre; ++j) {
    const int v = threadIdx.x + j * nt;
    float d = (o0 * pre[j].x + o1 * pre[j].y) + o2 * pre[j].z;
    if (v < n_views && d > best) { best = d; bi = v; }
  }
  for (int v0 = threadIdx.x + kPre * nt; v0 < n_views; v0 += 4 * nt) {
    v4f p[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int v = v0 + j * nt;
      p[j] = orientations4[v < n_views ? v : v0];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int v = v0 + j * nt;
      float d = (o0 * p[j].x + o1 * p[j].y) + o2 * p[j].z;
      if (v < n_views && d > best) { best = d; bi = v; }
    }
  }
  float wbest = wave_max(best);
  int wbi = wave_min_i(best == wbest ? bi : INT_MAX);
  int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave, n_waves = blockDim.x / kWave;
  int* imisc = reinterpret_cast<int*>(misc);
  if (lane == 0) { misc[wave] = wbest; imisc[32 + wave] = wbi; }
  __syncthreads();
  // (largest value, lowest index) over the <= 16 waves: one LDS read per lane and a 16-lane DPP row tree instead of a
  // serial walk over the per-wave results
  const int w = lane & 15;
  best = w < n_waves ? misc[w] : -2.0f;
  bi = w < n_waves ? imisc[32 + w] : INT_MAX;
#define M3T_ROW_ARGMAX_STEP(CTRL)                                             \
  {                                                                           \
    float ob = dpp_self<CTRL, 0xf>(best);                                     \
    int oi = dpp_self_i<CTRL, 0xf>(bi);                                       \
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }         \
  }
  M3T_ROW_ARGMAX_STEP(0x111)
  M3T_ROW_ARGMAX_STEP(0x112)
  M3T_ROW_ARGMAX_STEP(0x114)
  M3T_ROW_ARGMAX_STEP(0x118)
#undef M3T_ROW_ARGMAX_STEP
  bi = __builtin_amdgcn_readlane(bi, 15);
  __syncthreads();
  return bi == INT_MAX ? 0 : bi;
}

// RegionModality::CalculateCorrespondences :417-430 (adaptive coverage)
template <typename ExtentPtr>
__device__ __forceinline__ int number_of_lines(int n_max, int adaptive, float reference, ExtentPtr extents, int view,
                                               float max_extent, int n_points) {
  int n = n_max;
  if (adaptive) {  // only then the view's contour length / surface area is fetched
    const float extent = extents[view];
    if (reference > 0.0f) n = (int)((float)n_max * fminf(1.0f, extent / reference));
    else n = (int)((float)n_max * extent / max_extent);
  }
  if (n > n_points) n = n_points;
  return n;
}

// IsLineUnoccludedMeasured :1343-1389 / IsPointUnoccludedMeasured depth_modality.cpp:736-776
// (window of <= (kMaxNOcclusionStrides+1)^2 u16 samples around (center_u, center_v))
__device__ bool occlusion_window_clear(CCam& dc, float center_u, float center_v, float diameter,
                                       float depth, float depth_offset, float threshold) {
  int stride = f2i(diameter / M3T_MAX_N_OCCLUSION_STRIDES + 1.0f);
  int n_strides = f2i(diameter / stride + 0.5f);
  int rounded_diameter = n_strides * stride;
  float rounded_radius = 0.5f * (float)rounded_diameter;
  int u_min = f2i(center_u - rounded_radius + 0.5f);
  int v_min = f2i(center_v - rounded_radius + 0.5f);
  int u_max = u_min + rounded_diameter;
  int v_max = v_min + rounded_diameter;
  u_min = max(u_min, 0);
  v_min = max(v_min, 0);
  u_max = min(u_max, dc.width - 1);
  v_max = min(v_max, dc.height - 1);
  unsigned short min_depth = (unsigned short)f2i((depth - depth_offset - threshold) / dc.depth_scale);
  // <= 6 x 6 samples; the result is an OR over all of them, so they are fetched in independent
  // batches of 12 (no early exit inside a batch) instead of one dependent load per sample
  G<uint8_t> image = as_global(dc.image);
  const int n_u = u_max >= u_min ? (u_max - u_min) / stride + 1 : 0;
  const int n_v = v_max >= v_min ? (v_max - v_min) / stride + 1 : 0;
  const int total = n_u * n_v;
  for (int base = 0; base < total; base += 12) {
    unsigned short d[12];
    int ui = base % n_u, vi = base / n_u;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      d[k] = 0;
      if (base + k < total)
        d[k] = *reinterpret_cast<G<unsigned short>>(image + (uint32_t)(v_min + vi * stride) * dc.pitch +
                                                    (uint32_t)(u_min + ui * stride) * 2u);
      if (++ui == n_u) { ui = 0; ++vi; }
    }
    bool hit = false;
#pragma unroll
    for (int k = 0; k < 12; ++k) hit |= (d[k] > 0 && d[k] < min_depth);
    if (hit) return false;
  }
  return true;
}

// The same window for a scan by the 16 lanes of a DPP row.  Its limits -- two float and two integer divisions, a
// dozen conversions -- are worked out ONCE, by the thread that owns the line (occlusion_window_pack), and handed
// over as three words; the row then only turns sample numbers into addresses (occlusion_window_lane_hit: lane gl
// takes samples gl, gl + 16, gl + 32; true if one of THIS lane's samples occludes; the caller ORs the row).
struct OcclusionWindow {
  uint32_t base;    // byte offset of sample (0, 0)
  uint32_t packed;  // min_depth | n_samples << 16 | n_u << 24; n_samples == 0: nothing to test (never occluded)
  uint32_t stride;
};
__device__ __forceinline__ OcclusionWindow occlusion_window_pack(CCam& dc, float center_u, float center_v,
                                                                 float diameter, float depth, float depth_offset,
                                                                 float threshold) {
  int stride = f2i(diameter / M3T_MAX_N_OCCLUSION_STRIDES + 1.0f);
  if (stride < 1) stride = 1;  // (a body behind the camera: no meaningful window, but no division by zero either)
  int n_strides = f2i(diameter / stride + 0.5f);
  int rounded_diameter = n_strides * stride;
  float rounded_radius = 0.5f * (float)rounded_diameter;
  int u_min = f2i(center_u - rounded_radius + 0.5f);
  int v_min = f2i(center_v - rounded_radius + 0.5f);
  int u_max = u_min + rounded_diameter;
  int v_max = v_min + rounded_diameter;
  u_min = max(u_min, 0);
  v_min = max(v_min, 0);
  u_max = min(u_max, dc.width - 1);
  v_max = min(v_max, dc.height - 1);
  const unsigned short min_depth = (unsigned short)f2i((depth - depth_offset - threshold) / dc.depth_scale);
  const int n_u = u_max >= u_min ? (u_max - u_min) / stride + 1 : 0;
  const int n_v = v_max >= v_min ? (v_max - v_min) / stride + 1 : 0;
  int total = n_u * n_v;  // <= 36 (at most 6 x 6 samples: n_strides <= M3T_MAX_N_OCCLUSION_STRIDES)
  OcclusionWindow w;
  w.base = (uint32_t)v_min * dc.pitch + (uint32_t)u_min * 2u;
  w.packed = (uint32_t)min_depth | ((uint32_t)total << 16) | ((uint32_t)n_u << 24);
  w.stride = (uint32_t)stride;
  return w;
}
// (two steps, so that a caller can have the samples of several lines in flight before it looks at the first)
__device__ __forceinline__ void occlusion_window_lane_load(CCam& dc, uint32_t base, uint32_t packed, uint32_t stride,
                                                           int gl, unsigned short& d0, unsigned short& d1,
                                                           unsigned short& d2) {
  const int total = (int)((packed >> 16) & 0xffu), n_u = (int)(packed >> 24);
  G<uint8_t> image = as_global(dc.image);
  // sample number -> (row, column): (pos + 0.5) / n_u is at least 0.5 / n_u away from an integer, the approximate
  // reciprocal's 1 ulp cannot carry it across (pos < 48, n_u <= 6)
  const float inv_n_u = __builtin_amdgcn_rcpf((float)n_u);
  unsigned short d[3] = {0, 0, 0};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int pos = gl + 16 * k;
    if (pos < total) {
      const int vi = (int)(((float)pos + 0.5f) * inv_n_u), ui = pos - vi * n_u;
      d[k] = *reinterpret_cast<G<unsigned short>>(image + base + (uint32_t)vi * stride * dc.pitch +
                                                  (uint32_t)ui * stride * 2u);
    }
  }
  d0 = d[0]; d1 = d[1]; d2 = d[2];
}
__device__ __forceinline__ bool occlusion_window_lane_test(uint32_t packed, unsigned short d0, unsigned short d1,
                                                           unsigned short d2) {
  const unsigned short min_depth = (unsigned short)(packed & 0xffffu);
  return (d0 > 0 && d0 < min_depth) || (d1 > 0 && d1 < min_depth) || (d2 > 0 && d2 < min_depth);
}
__device__ __forceinline__ bool occlusion_window_lane_hit(CCam& dc, uint32_t base, uint32_t packed, uint32_t stride,
                                                          int gl) {
  unsigned short d0, d1, d2;
  occlusion_window_lane_load(dc, base, packed, stride, gl, d0, d1, d2);
  return occlusion_window_lane_test(packed, d0, d1, d2);
}

// ---------------------------------------------------------------------------
// renderer-fed branches: what the modalities read from a focused rendering (m3t_render.hip writes it)
// ---------------------------------------------------------------------------
// reading side (modalities)
__device__ __forceinline__ bool renderer_body_visible(const RendererDev* r, int slot) {
  return r != nullptr && slot >= 0 && r->state[RS_VISIBLE0 + slot] != 0.0f;
}
__device__ __forceinline__ float renderer_depth(const RendererDev& r, unsigned short value) {
  return r.state[RS_TERM_A] / (r.state[RS_TERM_B] - (float)value);
}
// The sampling loops themselves live in m3t_renderer_read.h (host-checkable): all samples of a loop are requested
// at once, the reference's decisions follow in the reference's order.  These wrappers hand them the rendering through
// address-space-1 pointers (global loads, not flat ones).
__device__ __forceinline__ FocusedCrop renderer_crop(const RendererDev& r) {
  FocusedCrop c;
  c.corner_u = r.state[RS_CORNER_U];
  c.corner_v = r.state[RS_CORNER_V];
  c.scale = r.state[RS_SCALE];
  c.image_size = r.image_size;
  return c;
}
__device__ __forceinline__ unsigned short modeled_window_min(const RendererDev& r, float center_u, float center_v,
                                                             float diameter) {
  return modeled_window_min(as_global(r.depth_image), renderer_crop(r), center_u, center_v, diameter);
}
__device__ __forceinline__ bool dynamic_line_region_sufficient(const RendererDev& r, int region_id,
                                                               float min_continuous_distance, float fscale, float center_u,
                                                               float center_v, float normal_u, float normal_v) {
  return dynamic_line_region_sufficient(as_global(r.silhouette_image), renderer_crop(r), region_id, min_continuous_distance,
                                        fscale, center_u, center_v, normal_u, normal_v);
}
__device__ __forceinline__ void dynamic_region_distance(const RendererDev& r, int region_id, float max_considered_line_length,
                                                        float unconsidered_line_length, float center_u, float center_v,
                                                        float normal_u, float normal_v, float* foreground, float* background) {
  dynamic_region_distance(as_global(r.silhouette_image), renderer_crop(r), region_id, max_considered_line_length,
                          unconsidered_line_length, center_u, center_v, normal_u, normal_v, foreground, background);
}


// ---------------------------------------------------------------------------
// iteration-dependent scalars (PrecalculateIterationDependentVariables :1011-1023)
// ---------------------------------------------------------------------------
struct RegionIter {
  int scale;
  float fscale;
  int line_length, line_length_minus_1;
  float line_length_minus_1_half, line_length_half_minus_1, variance;
};
__device__ __forceinline__ RegionIter region_iter(CRegion& m, int corr_iteration) {
  RegionIter it;
  it.scale = last_valid(m.scales, m.n_scales, corr_iteration);
  it.fscale = (float)it.scale;
  it.line_length = m.n_seg * it.scale;
  it.line_length_minus_1 = it.line_length - 1;
  it.line_length_minus_1_half = (float)(it.line_length - 1) * 0.5f;
  it.line_length_half_minus_1 = (float)(it.line_length) * 0.5f - 1.0f;
  float sd = last_valid(m.standard_deviations, m.n_standard_deviations, corr_iteration);
  it.variance = sd * sd;
  return it;
}

// LDS views of one tracked object
struct Lds {
  float* state;  // [LS_FIELDS][nl]
  float* chain;  // [nl][ns]   minor-axis coordinate at every segment start; later raw distribution [nl][DL]
  float* seg_f;  // [nl][ns]
  float* seg_b;  // [nl][ns]
  float* misc;   // M3T_MISC_FLOATS
  const float2* hist;  // LDS-staged histogram_norm or nullptr
  int nl, ns;
};
__device__ __forceinline__ Lds carve(float* base, const TrackLdsLayout& L) {
  Lds s;
  s.state = base + L.off_state;
  s.chain = base + L.off_chain;
  s.seg_f = base + L.off_seg_f;
  s.seg_b = base + L.off_seg_b;
  s.misc = base + L.off_misc;
  s.hist = L.off_hist >= 0 ? reinterpret_cast<const float2*>(base + L.off_hist) : nullptr;
  s.nl = L.nl;
  s.ns = L.ns;
  return s;
}

// ---------------------------------------------------------------------------
// Phase B of the correspondence search: one thread per (line, segment).
// MultiplyPixelColorProbability (:1575-1598) over SCALE pixels in walk order, then the
// per-segment renormalisation (:1556-1571).  B segments per thread are processed together
// so that ~10 independent pixel loads, then ~10 independent histogram gathers are in
// flight per lane (the walk is latency bound); the products keep the reference's order.
// A pixel is one unaligned 4-byte load (B,G,R + one byte of the next pixel).
// ---------------------------------------------------------------------------
struct __attribute__((packed)) PackedU32 { uint32_t v; };

typedef const __attribute__((address_space(3))) float* LdsF;
__device__ __forceinline__ v2f load_pair(G<v2f> hist, uint32_t idx) { return hist[idx]; }
__device__ __forceinline__ v2f load_pair(LdsF hist, uint32_t idx) {
  v2f r;
  r.x = hist[2 * idx];
  r.y = hist[2 * idx + 1];
  return r;
}

// the reference's v_f += v_step chain (region_modality.cpp:1464-1473), sampled at segment starts
template <int SCALE>
__device__ __forceinline__ void chain_fill(float* chain, float x, float step, int n_seg) {
  for (int seg = 0; seg < n_seg; ++seg) {
    chain[seg] = x;
#pragma unroll
    for (int j = 0; j < SCALE; ++j) x += step;
  }
}

// BMAX caps the batch: a workgroup that walks only its share of an object's lines (tracking_step_split_kernel) has
// one or two items per thread, a batch of 8 would be mostly empty slots.
template <int SCALE, int BMAX, typename HistPtr>
__device__ __forceinline__ void region_segments(CRegion& m, G<uint8_t> image, uint32_t pitch, HistPtr hist,
                                                int n_lines, int valid_mask, const Lds& s, int line_lo) {
  constexpr int B0 = SCALE >= 8 ? 2 : (SCALE >= 6 ? 3 : (SCALE >= 4 ? 4 : (SCALE == 3 ? 6 : 8)));  // <= 20 pixels in flight
  constexpr int B = B0 < BMAX ? B0 : BMAX;
  const int tid = threadIdx.x, nt = blockDim.x, nl = s.nl, n_seg = m.n_seg;
  const int bin_bits = 8 - m.bitshift;  // n_bins == 1 << bin_bits
  const int bitshift = m.bitshift;
  const int n_items = (n_lines - line_lo) * n_seg;  // lines [line_lo, n_lines)
  // (line, segment) of item = tid, advanced by nt per step without divisions
  const int q = nt / n_seg, r = nt - q * n_seg;
  int line0 = tid / n_seg, sw0 = tid - line0 * n_seg;
  line0 += line_lo;
  for (int base = tid; base < n_items; base += nt * B) {
    PHASE_T0();
    uint32_t px[B][SCALE];
    int out_index[B];
    // all LDS reads of the batch first (independent), then the address arithmetic
    int flags_b[B], start_b[B], sw_b[B], line_b[B];
    float step_b[B], x_b[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      int item = base + b * nt;
      int line = line0, sw = sw0;  // segment index in walk order
      line0 += q;
      sw0 += r;
      if (sw0 >= n_seg) { sw0 -= n_seg; ++line0; }
      if (item >= n_items) { line = 0; sw = 0; }
      line_b[b] = line;
      sw_b[b] = sw;
      flags_b[b] = item < n_items ? f2i_bits(s.state[LS_VALID * nl + line]) : 0;
      start_b[b] = f2i_bits(s.state[LS_WALK_START * nl + line]);
      step_b[b] = s.state[LS_WALK_STEP * nl + line];
      x_b[b] = s.chain[line * s.ns + sw];
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int flags = flags_b[b], sw = sw_b[b];
      out_index[b] = -1;
#pragma unroll
      for (int j = 0; j < SCALE; ++j) px[b][j] = 0;
      if (flags & valid_mask) {
        float x = x_b[b];
        const float step = step_b[b];
        const int major = start_b[b] + sw * SCALE;
        const bool horiz = flags & 4;
        // byte offset = minor * stride_minor + (major + j) * stride_major; minor < 2^16 and the
        // strides < 2^24, so the 24-bit multiply (full rate) is exact in its low 32 bits
        const uint32_t stride_minor = horiz ? pitch : 3u, stride_major = horiz ? 3u : pitch;
        uint32_t off_major = (uint32_t)major * stride_major;
#pragma unroll
        for (int j = 0; j < SCALE; ++j) {
          uint32_t off = __umul24((uint32_t)f2i(x), stride_minor) + off_major;
          px[b][j] = reinterpret_cast<G<PackedU32>>(image + off)->v;
          off_major += stride_major;
          x += step;
        }
        int seg = (flags & 8) ? (n_seg - 1 - sw) : sw;
        out_index[b] = line_b[b] * s.ns + seg;
      }
    }
    PHASE_MARK(7);  // LDS reads + address arithmetic + pixel load issue
    v2f h[B][SCALE];
#ifdef M3T_PHASE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE_MARK(8);  // pixel load latency
#endif
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
      for (int j = 0; j < SCALE; ++j) {
          uint32_t v = px[b][j];
        // (B >> s) * n^2 + (G >> s) * n + (R >> s) with n = 2^bin_bits (color_histograms.cpp:97-99)
        uint32_t idx = ((((v & 0xffu) >> bitshift) << bin_bits | ((v >> 8) & 0xffu) >> bitshift) << bin_bits) |
                       (((v >> 16) & 0xffu) >> bitshift);
        h[b][j] = load_pair(hist, idx);
      }
#ifdef M3T_PHASE_TIMING
    PHASE_MARK(9);  // histogram gather issue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PHASE_MARK(10);  // histogram gather latency
#endif
#pragma unroll
    for (int b = 0; b < B; ++b) {
      float pf = 1.0f, pb = 1.0f;
#pragma unroll
      for (int j = 0; j < SCALE; ++j) {
        pf *= h[b][j].x;
        pb *= h[b][j].y;
      }
      if (SCALE > 1) {
        if (pf || pb) {
          float sum = pf;
          sum += pb;
          pf /= sum;
          pb /= sum;
        } else {
          pf = 0.5f;
          pb = 0.5f;
        }
      }
      if (out_index[b] >= 0) {
        s.seg_f[out_index[b]] = pf;
        s.seg_b[out_index[b]] = pb;
      }
    }
    PHASE_MARK(11);  // products, normalisation, LDS stores
  }
}

// any scale (not unrolled); same arithmetic
template <typename HistPtr>
__device__ void region_segments_generic(CRegion& m, G<uint8_t> image, uint32_t pitch, HistPtr hist, int n_lines,
                                        int valid_mask, int scale, const Lds& s, int line_lo) {
  const int tid = threadIdx.x, nt = blockDim.x, nl = s.nl, n_seg = m.n_seg;
  const int bitshift = m.bitshift, n_bins = m.n_bins, n_bins2 = m.n_bins * m.n_bins;
  const int n_items = (n_lines - line_lo) * n_seg;
  for (int item = tid; item < n_items; item += nt) {
    int line = item / n_seg;
    int sw = item - line * n_seg;
    line += line_lo;
    int flags = f2i_bits(s.state[LS_VALID * nl + line]);
    if (!(flags & valid_mask)) continue;
    int start = f2i_bits(s.state[LS_WALK_START * nl + line]);
    float step = s.state[LS_WALK_STEP * nl + line];
    float x = s.chain[line * s.ns + sw];
    int major = start + sw * scale;
    const bool horiz = flags & 4;
    float pf = 1.0f, pb = 1.0f;
    for (int j = 0; j < scale; ++j, ++major, x += step) {
      int minor = f2i(x);
      G<uint8_t> p = horiz ? image + (uint32_t)minor * pitch + major * 3 : image + (uint32_t)major * pitch + minor * 3;
      int idx = (p[0] >> bitshift) * n_bins2 + (p[1] >> bitshift) * n_bins + (p[2] >> bitshift);
      v2f h = load_pair(hist, idx);
      pf *= h.x;
      pb *= h.y;
    }
    if (scale > 1) {
      if (pf || pb) {
        float sum = pf;
        sum += pb;
        pf /= sum;
        pb /= sum;
      } else {
        pf = 0.5f;
        pb = 0.5f;
      }
    }
    int seg = (flags & 8) ? (n_seg - 1 - sw) : sw;
    s.seg_f[line * s.ns + seg] = pf;
    s.seg_b[line * s.ns + seg] = pb;
  }
}

template <int BMAX, typename HistPtr>
__device__ __forceinline__ void region_segments_dispatch(CRegion& m, G<uint8_t> image, uint32_t pitch, HistPtr hist,
                                                         int scale, int n_lines, int valid_mask, const Lds& s,
                                                         int line_lo) {
  switch (scale) {
    case 1: region_segments<1, BMAX>(m, image, pitch, hist, n_lines, valid_mask, s, line_lo); break;
    case 2: region_segments<2, BMAX>(m, image, pitch, hist, n_lines, valid_mask, s, line_lo); break;
    case 3: region_segments<3, BMAX>(m, image, pitch, hist, n_lines, valid_mask, s, line_lo); break;
    case 4: region_segments<4, BMAX>(m, image, pitch, hist, n_lines, valid_mask, s, line_lo); break;
    case 5: region_segments<5, BMAX>(m, image, pitch, hist, n_lines, valid_mask, s, line_lo); break;
    case 6: region_segments<6, BMAX>(m, image, pitch, hist, n_lines, valid_mask, s, line_lo); break;
    case 7: region_segments<7, BMAX>(m, image, pitch, hist, n_lines, valid_mask, s, line_lo); break;
    case 8: region_segments<8, BMAX>(m, image, pitch, hist, n_lines, valid_mask, s, line_lo); break;
    case 9: region_segments<9, BMAX>(m, image, pitch, hist, n_lines, valid_mask, s, line_lo); break;
    default: region_segments_generic(m, image, pitch, hist, n_lines, valid_mask, scale, s, line_lo); break;
  }
}

// ---------------------------------------------------------------------------
// RegionModality::CalculateCorrespondences (:390-465) for one object, whole block.
// Phase A  one thread per line: CalculateBasicLineData :1231, IsLineValid :1252,
//          the geometric part of CalculateSegmentProbabilities :1441-1455,:1494-1496
//          and the sequential minor-axis chain (v_f += v_step) sampled at segment starts.
// Phase B  one thread per (line, segment): MultiplyPixelColorProbability :1575 over
//          `scale` pixels in walk order + per-segment renormalisation :1556-1571.
// Phase C  one thread per (line, d): CalculateDistribution :1600 products; then one
//          thread per line: normalisation + CalculateDistributionMoments :1639.
// ---------------------------------------------------------------------------
// (tracking_step_split_kernel: what the workgroups of one object need to hand each other their lines' results; the
// exchange itself follows further down -- split_exchange_publish / split_exchange_collect)
struct SplitExchange {
  __attribute__((address_space(1))) unsigned long long* granules;  // this object's [2 slots][parts][32 fields][1 << lshift]
  __attribute__((address_space(1))) unsigned* object_abort;        // launch sequence number of an aborted step
  unsigned* host_abort;                                            // mapped host word, set to the sequence number
  uint32_t seq;    // launch sequence number (> 0)
  uint32_t abort_id;
  int part, n_parts, lshift;
  int per_part_lines, per_part_points;
  int n_region_fields, first_region_row;    // rows LS_DIST0 .. (from LS_VALID on while the occlusion vote is deferred)
  int n_depth_fields, first_depth_row;      // rows first_depth_row .. PS_VALID of the point state
};
constexpr int kExchangeFieldBits = 5;

// RENDER = false: a launch shape that never runs with renderer-fed branches (the split kernel: m3t_hip_api.hip takes
// the per-search launches of tracking_step_kernel as soon as a modality reads a rendering) leaves their code out
template <bool HIST_LDS, int BMAX = 8, bool RENDER = true, bool LEAN = false>
__device__ __forceinline__ int region_correspondences(CRegion& m, CCam& cam, CCam* dcam, const Affine& b2c,
                                                       const Affine& b2dc, int iteration, int corr_iteration,
                                                       const Lds& s, int line_lo = 0, int line_hi = 1 << 30,
                                                       bool* vote_deferred = nullptr, int prev_view = -1,
                                                       const SplitExchange* early = nullptr) {
  // early: the thread that normalises a distribution value (phase C2) also sends it to the object's other workgroups
  // (the granule split_exchange_publish would write after the phase: same slot, same tag, same value), unless the
  // occlusion vote is deferred -- then the rows travel with the flags after the phase
  int tid = threadIdx.x;
  if constexpr (LEAN) asm volatile("" : "+v"(tid));  // (an opaque copy: the per-thread addresses of the phases are formed in them)
  const int nt = blockDim.x;
  const RegionIter it = region_iter(m, corr_iteration);
  const int nl = s.nl;
  PHASE_T0();
  if (tid < M3T_MAX_FUNCTION_LENGTH) {  // (read in phase C: the barrier behind phase A publishes them)
    s.misc[kMiscLookup + tid] = m.function_lookup_f[tid];
    s.misc[kMiscLookup + M3T_MAX_FUNCTION_LENGTH + tid] = m.function_lookup_b[tid];
  }
  // the view of the previous search first (closest_view_local: no barrier), the scan over all views if that cannot
  // vouch for its answer.  (Fetching this thread's data point of the previous view ahead of the search -- most
  // searches stay on their view -- was measured: no gain, 9 VGPRs.)
  int view = -1;
  if (prev_view >= 0 && m.view_neighbors != nullptr) {
    float o0, o1, o2;
    if (view_direction(b2c, o0, o1, o2)) view = closest_view_local<LEAN>((G<v4f>)m.view_neighbors, prev_view, o0, o1, o2);
  }
  if (view < 0) view = closest_view((G<v4f>)m.orientations4, m.n_views, b2c, s.misc);  // block-uniform; two barriers
  PHASE_MARK(0);
  const int n_lines =
      number_of_lines(m.n_lines_max, m.use_adaptive_coverage, m.reference_contour_length, as_global(m.extents), view,
                      m.max_extent, m.n_points);
  const bool handle_occlusions = (iteration - m.first_iteration) >= m.n_unoccluded_iterations;
  const bool measured_pass = m.measure_occlusions && handle_occlusions;
  const bool modeled_pass = RENDER && m.model_occlusions && handle_occlusions &&
                            renderer_body_visible(m.depth_renderer, m.depth_renderer_slot);
  const bool region_checking =
      RENDER && m.use_region_checking && renderer_body_visible(m.silhouette_renderer, m.silhouette_renderer_slot);
  const bool occlusion_pass = measured_pass || modeled_pass;
  // A workgroup that shares its object with others (line_hi set) runs the occlusion tests -- up to 36 depth samples per
  // line -- for its own lines only and defers the two-pass vote :435-463 until the flags of all lines have been
  // exchanged (region_finish_flags): meanwhile it walks every valid line of its part.  The lines the vote then drops
  // were walked in vain; the result is the reference's.
  const bool defer_vote = occlusion_pass && line_hi < (1 << 30);
  if (vote_deferred) *vote_deferred = defer_vote;  // the caller exchanges the flags and calls region_finish_flags
  const int n_seg = m.n_seg;
  G<uint8_t> image = as_global(cam.image);
  const uint32_t pitch = cam.pitch;

  // ---- phase A ----
  int my_valid_occ = 0;
  for (int line = tid; line < nl; line += nt) {
    int flags = 0;
    if (measured_pass) reinterpret_cast<uint32_t*>(s.seg_f)[line * 3 + 1] = 0u;  // no occlusion window to scan (yet)
    if (line < n_lines) {
      G<v4f> p8 = (G<v4f>)m.points8 + ((uint32_t)view * m.n_points + line) * 2;
      const v4f pa = p8[0], pb4 = p8[1];
      float cx = pa.x, cy = pa.y, cz = pa.z;
      float nx = pa.w, ny = pb4.x, nz = pb4.y;
      float fg = pb4.z, bg = pb4.w;
      float X, Y, Z;
      apply_pose(b2c, cx, cy, cz, X, Y, Z);
      float nu = (b2c.l[0] * nx + b2c.l[3] * ny) + b2c.l[6] * nz;
      float nv = (b2c.l[1] * nx + b2c.l[4] * ny) + b2c.l[7] * nz;
      float nn = sqrtf(nu * nu + nv * nv);
      if (nn > 0.0f) { nu = nu / nn; nv = nv / nn; }
      float center_u = X * cam.fu / Z + cam.ppu;
      float center_v = Y * cam.fv / Z + cam.ppv;
      float cont = ((fg < bg) ? fg : bg) * cam.fu / (Z * it.fscale);
      bool valid = !(cont < m.min_continuous_distance) && !(Z <= 0.0f);
      if (valid) {
        int icu = f2i(center_u + 0.5f), icv = f2i(center_v + 0.5f);
        valid = !(icu < 0 || icu > cam.width - 1 || icv < 0 || icv > cam.height - 1);
      }
      // geometric part of CalculateSegmentProbabilities
      bool horiz = fabsf(nv) < fabsf(nu);
      float step, cmaj, cmin, ndom;
      int maj_lim, min_lim1, min_lim2;
      if (horiz) {
        step = nv / nu; cmaj = center_u; cmin = center_v; ndom = nu;
        maj_lim = cam.width - 1; min_lim1 = cam.height - 1; min_lim2 = cam.height - 2;
      } else {
        step = nu / nv; cmaj = center_v; cmin = center_u; ndom = nv;
        maj_lim = cam.height - 1; min_lim1 = cam.width - 1; min_lim2 = cam.width - 2;
      }
      int start = f2i(cmaj - it.line_length_half_minus_1);
      int end = start + it.line_length_minus_1;
      float x0 = cmin + step * ((float)start - cmaj) + 0.5f;
      float xend = x0 + step * (float)it.line_length_minus_1;
      if (valid)
        valid = !(start < 0 || end > maj_lim || f2i(x0) < 0 || f2i(x0) > min_lim1 || f2i(xend) < 1 ||
                  f2i(xend) > min_lim2);
      // IsLineValid :1252-1291: region checking applies in both passes, the occlusion tests in the first
      if (valid && region_checking)
        valid = dynamic_line_region_sufficient(*m.silhouette_renderer, m.region_id, m.min_continuous_distance,
                                               it.fscale, center_u, center_v, nu, nv);
      bool valid_occ = valid;
      const bool test_occlusion = !defer_vote || (line >= line_lo && line < line_hi);
      if (valid && modeled_pass && test_occlusion) {
        G<float> p = as_global(m.points) + ((size_t)view * m.n_points + line) * M3T_REGION_POINT_FLOATS;
        float diameter = 2.0f * m.modeled_occlusion_radius * ((cam.fu / Z) * m.depth_renderer->state[RS_SCALE]);
        unsigned short min_value = modeled_window_min(*m.depth_renderer, center_u, center_v, diameter);
        valid_occ = renderer_depth(*m.depth_renderer, min_value) >
                    Z - p[8 + m.modeled_depth_offset_id] - m.modeled_occlusion_threshold;
      }
      // IsLineUnoccludedMeasured :1343-1389: the window of up to 36 depth samples is scanned by 16 lanes per line
      // after this loop (measured_occlusion_pass); here only its parameters are set aside (seg_f is free until phase B)
      if (measured_pass) {
        if (valid_occ && test_occlusion) {
          float dx, dy, dz;
          apply_pose(b2dc, cx, cy, cz, dx, dy, dz);
          float du = dx * dcam->fu / dz + dcam->ppu;
          float dv = dy * dcam->fv / dz + dcam->ppv;
          float meter_to_pixel = dcam->fu / dz;
          float diameter = 2.0f * m.measured_occlusion_radius * meter_to_pixel;
          G<float> p = as_global(m.points) + ((size_t)view * m.n_points + line) * M3T_REGION_POINT_FLOATS;
          const OcclusionWindow ow = occlusion_window_pack(*dcam, du, dv, diameter, dz, p[8 + m.measured_depth_offset_id],
                                                           m.measured_occlusion_threshold);
          uint32_t* w = reinterpret_cast<uint32_t*>(s.seg_f) + line * 3;
          w[0] = ow.base;
          w[1] = ow.packed;
          w[2] = ow.stride;
        }
      }
      if (valid) {
        flags = (valid_occ ? 1 : 0) | 2 | (horiz ? 4 : 0) | ((ndom > 0.0f) ? 0 : 8);
        my_valid_occ += valid_occ ? 1 : 0;
        s.state[LS_CX * nl + line] = cx;
        s.state[LS_CY * nl + line] = cy;
        s.state[LS_CZ * nl + line] = cz;
        s.state[LS_CENTER_U * nl + line] = center_u;
        s.state[LS_CENTER_V * nl + line] = center_v;
        s.state[LS_NORMAL_U * nl + line] = nu;
        s.state[LS_NORMAL_V * nl + line] = nv;
        s.state[LS_CONT * nl + line] = cont;
        s.state[LS_NCTS * nl + line] = fabsf(ndom) / it.fscale;
        s.state[LS_DELTA_R * nl + line] =
            (roundf(cmaj - it.line_length_minus_1_half) + it.line_length_minus_1_half - cmaj) / ndom;
        s.state[LS_WALK_START * nl + line] = i2f_bits(start);
        s.state[LS_WALK_STEP * nl + line] = step;
        // sequential chain: x_{k+1} = x_k + step, recorded at every segment start (only the lines this workgroup
        // walks need it: up to 19 x scale dependent additions per line)
        float* chain = s.chain + line * s.ns;
        if (line >= line_lo && line < line_hi)
        switch (it.scale) {
          case 1: chain_fill<1>(chain, x0, step, n_seg); break;
          case 2: chain_fill<2>(chain, x0, step, n_seg); break;
          case 3: chain_fill<3>(chain, x0, step, n_seg); break;
          case 4: chain_fill<4>(chain, x0, step, n_seg); break;
          case 5: chain_fill<5>(chain, x0, step, n_seg); break;
          case 6: chain_fill<6>(chain, x0, step, n_seg); break;
          case 7: chain_fill<7>(chain, x0, step, n_seg); break;
          default: {
            float x = x0;
            for (int seg = 0; seg < n_seg; ++seg) {
              chain[seg] = x;
              for (int j = 0; j < it.scale; ++j) x += step;
            }
          }
        }
      }
    }
    s.state[LS_VALID * nl + line] = i2f_bits(flags);
  }
  if (measured_pass) {  // 16 lanes per line: the occlusion windows
    __syncthreads();
    const int gl = tid & 15;
    const int lo = defer_vote ? line_lo : 0, hi = defer_vote ? (line_hi < nl ? line_hi : nl) : nl;
    const int groups = nt >> 4;
    for (int line0 = lo; line0 < hi; line0 += groups) {  // uniform trip count: the DPP row OR needs all lanes of a row
      const int line = line0 + (tid >> 4);
      int occluded = 0;
      if (line < hi) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(s.seg_f) + line * 3;
        if (occlusion_window_lane_hit(*dcam, w[0], w[1], w[2], gl)) occluded = 1;
      }
      occluded |= dpp_self_i<0x111, 0xf>(occluded);
      occluded |= dpp_self_i<0x112, 0xf>(occluded);
      occluded |= dpp_self_i<0x114, 0xf>(occluded);
      occluded |= dpp_self_i<0x118, 0xf>(occluded);
      if (gl == 15 && line < hi && occluded) {
        const int flags = f2i_bits(s.state[LS_VALID * nl + line]);
        s.state[LS_VALID * nl + line] = i2f_bits(flags & ~1);
      }
    }
    __syncthreads();
    my_valid_occ = 0;  // counted from the flags now
    // (LEAN: the loops over a thread's lines stay rolled -- at most one trip at 200 lines and 512 threads; unrolled by
    // eight their trip-count bookkeeping sits in front of the caller's loops, a VGPR each)
    if constexpr (LEAN) {
#pragma nounroll
      for (int line = tid; line < nl; line += nt) my_valid_occ += f2i_bits(s.state[LS_VALID * nl + line]) & 1;
    } else {
      for (int line = tid; line < nl; line += nt) my_valid_occ += f2i_bits(s.state[LS_VALID * nl + line]) & 1;
    }
  }
  // two-pass fallback :435-463: use the occlusion-handled set only if it has enough lines
  bool use_occ = false;
  if (occlusion_pass && !defer_vote) {
    int cnt = wave_sum_i(my_valid_occ);
    int* imisc = reinterpret_cast<int*>(s.misc);
    if (tid % kWave == 0) imisc[64 + tid / kWave] = cnt;
    __syncthreads();
    int total = 0;
    for (int w = 0; w < nt / kWave; ++w) total += imisc[64 + w];
    use_occ = total >= m.min_n_unoccluded_lines;
  }
  __syncthreads();
  PHASE_MARK(1);
  const int valid_mask = use_occ ? 1 : 2;

  // ---- phase B ----
  // a workgroup that shares its object with others (tracking_step_split_kernel) walks only the lines
  // [line_lo, line_hi) from here on; phase A above ran for all lines (its two-pass vote needs them)
  const int n_lines_b = n_lines < line_hi ? n_lines : line_hi;
  const int nl_b = nl < line_hi ? nl : line_hi;
  if (HIST_LDS) {  // pair table staged in LDS (n_bins <= 16)
    region_segments_dispatch<BMAX>(m, image, pitch, (LdsF)s.hist, it.scale, n_lines_b, valid_mask, s, line_lo);
  } else {         // pair table gathered from L2 / HBM
    region_segments_dispatch<BMAX>(m, image, pitch, (G<v2f>)m.histogram_norm, it.scale, n_lines_b, valid_mask, s, line_lo);
  }
  __syncthreads();
  PHASE_MARK(2);

  // ---- phase C1: raw distribution products (aliases the chain buffer) ----
  const int dl = m.distribution_length, fl = m.function_length;
  float* raw = s.chain;  // [nl][M3T_MAX_DISTRIBUTION_LENGTH] needs ns >= dl (n_seg = fl + dl - 1 >= dl)
  {
    const float* lf = s.misc + kMiscLookup;
    const float* lb = s.misc + kMiscLookup + M3T_MAX_FUNCTION_LENGTH;
    // (line, d) of item = tid, advanced by nt per step without divisions
    const int q = nt / dl, r = nt - q * dl;
    int line = tid / dl, d = tid - line * dl;
    line += line_lo;
    const int n_items_c = (n_lines_b - line_lo) * dl;
    if (fl == 8) {  // default function_length: lookups as uniform scalars, independent LDS reads, ordered product
      float lfr[8], lbr[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { lfr[k] = m.function_lookup_f[k]; lbr[k] = m.function_lookup_b[k]; }
      for (int item = tid; item < n_items_c; item += nt) {
        int flags = f2i_bits(s.state[LS_VALID * nl + line]);
        if (flags & valid_mask) {
          const float* sf = s.seg_f + line * s.ns + d;
          const float* sb = s.seg_b + line * s.ns + d;
          float f[8], b[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) { f[k] = sf[k]; b[k] = sb[k]; }
          float value = 1.0f;
#pragma unroll
          for (int k = 0; k < 8; ++k) value *= f[k] * lfr[k] + b[k] * lbr[k];
          raw[line * s.ns + d] = value;
        }
        line += q;
        d += r;
        if (d >= dl) { d -= dl; ++line; }
      }
    } else {
      for (int item = tid; item < n_items_c; item += nt) {
        int flags = f2i_bits(s.state[LS_VALID * nl + line]);
        if (flags & valid_mask) {
          const float* sf = s.seg_f + line * s.ns + d;
          const float* sb = s.seg_b + line * s.ns + d;
          float value = 1.0f;
          for (int k = 0; k < fl; ++k) value *= sf[k] * lf[k] + sb[k] * lb[k];
          raw[line * s.ns + d] = value;
        }
        line += q;
        d += r;
        if (d >= dl) { d -= dl; ++line; }
      }
    }
  }
  __syncthreads();
  PHASE_MARK(3);
  // ---- phase C2: normalisation, one thread per (line, d): every thread adds up the line's raw values in the
  // reference's order (the same area in all of them) and divides its own (the moments follow in region_moments) ----
  {
    const bool send = early != nullptr && !defer_vote;
    __attribute__((address_space(1))) unsigned long long* mine = nullptr;
    unsigned long long tag_bits = 0;
    int lshift = 0;
    if (send) {
      lshift = early->lshift;
      mine = early->granules + ((size_t)(corr_iteration & 1) * early->n_parts << (kExchangeFieldBits + lshift)) +
             ((size_t)early->part << (kExchangeFieldBits + lshift));
      tag_bits = static_cast<unsigned long long>(early->seq * 64u + (uint32_t)corr_iteration + 1u) << 32;
    }
    const int q = nt / dl, r = nt - q * dl;
    int line = tid / dl, d = tid - line * dl;
    line += line_lo;
    const int n_items_c = (n_lines_b - line_lo) * dl;
    for (int item = tid; item < n_items_c; item += nt) {
      const int flags = f2i_bits(s.state[LS_VALID * nl + line]);
      float value = 0.0f;
      if (flags & valid_mask) {
        const float* rr = raw + line * s.ns;
        float area = 0.0f;
        if (dl == 12) {  // the default distribution length: independent LDS reads, then the ordered sum
          const float v0 = rr[0], v1 = rr[1], v2 = rr[2], v3 = rr[3], v4 = rr[4], v5 = rr[5], v6 = rr[6], v7 = rr[7],
                      v8 = rr[8], v9 = rr[9], v10 = rr[10], v11 = rr[11];
          area += v0; area += v1; area += v2; area += v3; area += v4; area += v5;
          area += v6; area += v7; area += v8; area += v9; area += v10; area += v11;
        } else {
          for (int k = 0; k < dl; ++k) area += rr[k];
        }
        value = rr[d] / area;
        s.state[(LS_DIST0 + d) * nl + line] = value;
      } else if (send) {
        value = s.state[(LS_DIST0 + d) * nl + line];  // (a line that is not walked: what the row holds, as the publish loop sends it)
      }
      if (send)
        __hip_atomic_store(mine + ((d << lshift) | (line - line_lo)), tag_bits | (unsigned)__float_as_int(value),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      line += q;
      d += r;
      if (d >= dl) { d -= dl; ++line; }
    }
    if (send) {  // the rows' entries of this part beyond the view's lines (the other workgroups wait for every granule)
      const int first = n_lines_b > line_lo ? n_lines_b : line_lo, last = line_hi < nl ? line_hi : nl;
      for (int item = tid; item < (last - first) * dl; item += nt) {
        const int l2 = first + item / dl, d2 = item - (item / dl) * dl;
        __hip_atomic_store(mine + ((d2 << lshift) | (l2 - line_lo)),
                           tag_bits | (unsigned)__float_as_int(s.state[(LS_DIST0 + d2) * nl + l2]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // the final flag (bit 0 = line is in data_lines_), for the lines of all parts: it follows from phase A, which every
  // workgroup ran in full.  (Threads above may still test `flags & valid_mask`: bit 0 only ever takes that test's value.)
  if (!defer_vote) {
    auto final_flag = [&](int line) {
      const int flags = f2i_bits(s.state[LS_VALID * nl + line]);
      s.state[LS_VALID * nl + line] = i2f_bits((flags & ~1) | ((flags & valid_mask) ? 1 : 0));
    };
    if constexpr (LEAN) {
#pragma nounroll
      for (int line = tid; line < nl; line += nt) final_flag(line);
    } else {
      for (int line = tid; line < nl; line += nt) final_flag(line);
    }
  }
  __syncthreads();
  PHASE_MARK(4);
  return view;  // the closest view (a depth modality of the same body with the same view table and camera pose reuses it)
}

// tracking_step_split_kernel, occlusion handling on: the two-pass vote :435-463 and the final flags once every
// workgroup's occlusion results (bit 0 of the flags) are in.  Barriers inside; ends without one.
__device__ __forceinline__ void region_finish_flags(CRegion& m, const Lds& s) {
  const int tid = threadIdx.x, nt = blockDim.x, nl = s.nl;
  int mine = 0;
  for (int line = tid; line < nl; line += nt) mine += f2i_bits(s.state[LS_VALID * nl + line]) & 1;
  const int cnt = wave_sum_i(mine);
  int* imisc = reinterpret_cast<int*>(s.misc);
  if (tid % kWave == 0) imisc[64 + tid / kWave] = cnt;
  __syncthreads();
  int total = 0;
  for (int w = 0; w < nt / kWave; ++w) total += imisc[64 + w];
  const int valid_mask = total >= m.min_n_unoccluded_lines ? 1 : 2;
  for (int line = tid; line < nl; line += nt) {
    const int flags = f2i_bits(s.state[LS_VALID * nl + line]);
    s.state[LS_VALID * nl + line] = i2f_bits((flags & ~1) | ((flags & valid_mask) ? 1 : 0));
  }
  __syncthreads();
}

// CalculateDistributionMoments :1639-1658 from the normalised distributions in LDS, for the lines [lo, hi) if
// `inside`, for all other lines if not (tracking_step_split_kernel: the own part's lines while the other parts'
// are still on their way, the received ones afterwards).  No barrier.
template <int DL>
__device__ __forceinline__ void region_moments_lines(CRegion& m, const Lds& s, int lo, int hi, bool inside, int dl) {
  const int nl = s.nl;
  for (int line = threadIdx.x; line < nl; line += blockDim.x) {
    if ((line >= lo && line < hi) != inside) continue;
    if (!(f2i_bits(s.state[LS_VALID * nl + line]) & 1)) continue;
    constexpr int N = DL > 0 ? DL : M3T_MAX_DISTRIBUTION_LENGTH;
    float dist[N];
#pragma unroll
    for (int d = 0; d < N; ++d) dist[d] = (DL > 0 || d < dl) ? s.state[(LS_DIST0 + d) * nl + line] : 0.0f;  // independent reads
    float mean_from_begin = 0.0f;
#pragma unroll
    for (int d = 0; d < N; ++d)
      if (DL > 0 || d < dl) mean_from_begin += (float)d * dist[d];
    float var = 0.0f;
#pragma unroll
    for (int d = 0; d < N; ++d)
      if (DL > 0 || d < dl) {
        float dd = (float)d - mean_from_begin;
        var += (dd * dd) * dist[d];
      }
    s.state[LS_MEAN * nl + line] = mean_from_begin - m.distribution_length_minus_1_half;
    s.state[LS_VAR * nl + line] = fmaxf(var, m.min_expected_variance);
  }
}
__device__ __forceinline__ void region_moments(CRegion& m, const Lds& s, int lo = 0, int hi = 1 << 30, bool inside = true) {
  const int dl = m.distribution_length;
  if (dl == 12) region_moments_lines<12>(m, s, lo, hi, inside, dl);  // the default length: straight-line code
  else region_moments_lines<0>(m, s, lo, hi, inside, dl);
}

// ---------------------------------------------------------------------------
// Gradient / Hessian sums in the reference's order.
// The reference adds the terms of one data line (or data point) after the other into gradient_ / hessian_
// (region_modality.cpp:550-554, depth_modality.cpp:361-377): 27 independent f32 chains (6 gradient entries,
// 21 entries of a Hessian triangle).  The products of every line are formed in parallel, one thread per line,
// and written as 27 rows P[row][line] to LDS; then the first 42 lanes of one wave (the layout of
// Modality::gradient() / hessian(): 6 + 36, mirrored entries read the same row) run down their rows with one
// dependent subtraction per line.  A line that does not contribute stores zeros (x -/+ 0 = x; a chain that
// started at +0 never holds -0).  Rows: r < 6: the gradient entry r; 6 + c*6 - c*(c-1)/2 + (r - c): H(r, c), r >= c.
// Every row stores what is to be SUBTRACTED from the running sum (the region gradient is negated: a - (-b) = a + b).
// ---------------------------------------------------------------------------
constexpr int kChainBlock = 24;  // lines per pipeline step of a single chain (6 x ds_read_b128, the next 6 in flight)

__host__ __device__ inline int chain_slots(int n) { return (n + kChainBlock - 1) / kChainBlock * kChainBlock; }
// row pitch in floats: a multiple of 4 with an odd quarter, so that 16 lanes reading 16 bytes of 16 rows hit 64 banks
__host__ __device__ inline int chain_pitch(int n) {
  const int slots = chain_slots(n);
  return ((slots / 4) & 1) ? slots : slots + 4;
}

__device__ __forceinline__ int gh_lane_row(int lane) {  // lane < 42 of the gradient()/hessian() layout -> product row
  if (lane < 6) return lane;
  const int idx = lane - 6, c = idx / 6, r = idx - c * 6;
  const int lo = r >= c ? r : c, hi = r >= c ? c : r;
  return 6 + hi * 6 - hi * (hi - 1) / 2 + (lo - hi);
}

// volatile: every quad is read exactly once and in program order, so the reads of the NEXT step really are in flight
// while the current step's subtractions run (a plain load is re-materialised at its use by the compiler, which puts
// the whole LDS latency in front of every step: measured 2.6 k instead of ~1.1 k cycles per 216-line chain)
typedef const volatile __attribute__((address_space(3))) v4f* LdsV4;
// one dependent subtraction per line; Q quads (4 lines each) per step, the next step's quads already in flight:
// two register sets take turns (no copies), a scheduling barrier keeps each set's reads ahead of the other set's sums
template <int Q>
__device__ __forceinline__ float chain_walk(LdsV4 p, int n_quads, float s) {
  // (round 4: a third register set -- the reads of the next TWO steps in flight, no full drain at the loop head -- was
  // measured: no change, 38.0 k vs 37.2 k cycles per frame; the chain is bound by its dependent subtractions)
  v4f a[Q], b[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) a[i] = p[i];
  int q = 0;
  while (true) {
    {
      const int nq = q + Q < n_quads ? q + Q : q;  // past the end: re-read the own quads (harmless, never summed)
#pragma unroll
      for (int i = 0; i < Q; ++i) b[i] = p[nq + i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < Q; ++i) { s -= a[i].x; s -= a[i].y; s -= a[i].z; s -= a[i].w; }
      q += Q;
      if (q >= n_quads) break;
    }
    {
      const int nq = q + Q < n_quads ? q + Q : q;
#pragma unroll
      for (int i = 0; i < Q; ++i) a[i] = p[nq + i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < Q; ++i) { s -= b[i].x; s -= b[i].y; s -= b[i].z; s -= b[i].w; }
      q += Q;
      if (q >= n_quads) break;
    }
  }
  return s;
}
// two chains at once (two independent dependency chains interleave for free), 3 quads each per step
__device__ __forceinline__ void chain_walk2(LdsV4 pa, LdsV4 pb, int n_quads, float& sa, float& sb) {
  constexpr int Q = 3;
  v4f a0[Q], b0[Q], a1[Q], b1[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) { a0[i] = pa[i]; b0[i] = pb[i]; }
  int q = 0;
  while (true) {
    {
      const int nq = q + Q < n_quads ? q + Q : q;
#pragma unroll
      for (int i = 0; i < Q; ++i) { a1[i] = pa[nq + i]; b1[i] = pb[nq + i]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        sa -= a0[i].x; sb -= b0[i].x;
        sa -= a0[i].y; sb -= b0[i].y;
        sa -= a0[i].z; sb -= b0[i].z;
        sa -= a0[i].w; sb -= b0[i].w;
      }
      q += Q;
      if (q >= n_quads) break;
    }
    {
      const int nq = q + Q < n_quads ? q + Q : q;
#pragma unroll
      for (int i = 0; i < Q; ++i) { a0[i] = pa[nq + i]; b0[i] = pb[nq + i]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        sa -= a1[i].x; sb -= b1[i].x;
        sa -= a1[i].y; sb -= b1[i].y;
        sa -= a1[i].z; sb -= b1[i].z;
        sa -= a1[i].w; sb -= b1[i].w;
      }
      q += Q;
      if (q >= n_quads) break;
    }
  }
}

// lanes < 42 of the calling wave; rows_a / rows_b: product tables of up to two modalities (either may be null)
__device__ __forceinline__ void chain_sums(const float* rows_a, int pitch_a, int slots_a, const float* rows_b,
                                           int pitch_b, int slots_b, int row, float& sum_a, float& sum_b) {
  float sa = 0.0f, sb = 0.0f;
  const int qa = rows_a ? slots_a / 4 : 0, qb = rows_b ? slots_b / 4 : 0;  // multiples of 6
  LdsV4 pa = (LdsV4)(rows_a ? rows_a + row * pitch_a : nullptr);
  LdsV4 pb = (LdsV4)(rows_b ? rows_b + row * pitch_b : nullptr);
  const int common = qa < qb ? qa : qb;
  if (common > 0) chain_walk2(pa, pb, common, sa, sb);
  if (qa > common) sa = chain_walk<6>(pa + common, qa - common, sa);
  if (qb > common) sb = chain_walk<6>(pb + common, qb - common, sb);
  sum_a = sa;
  sum_b = sb;
}

// ---------------------------------------------------------------------------
// RegionModality::CalculateGradientAndHessian (:485-558), the per-line part: one thread per line slot
// writes the line's 27 products to rows[row * pitch + line] (zeros for slots that do not contribute).
// ---------------------------------------------------------------------------
// (tid0 / nt0: the threads that take part -- all of the workgroup, or the half that works beside depth_products' half)
__device__ __forceinline__ void region_products(CRegion& m, CCam& cam, const Affine& b2c, int corr_iteration, int opt_iteration,
                                const Lds& s, float* rows, int pitch, int tid0 = -1, int nt0 = 0) {
  const int tid = tid0 >= 0 ? tid0 : (int)threadIdx.x, nt = tid0 >= 0 ? nt0 : (int)blockDim.x, nl = s.nl;
  const RegionIter it = region_iter(m, corr_iteration);
  const int slots = chain_slots(nl);
  for (int line = tid; line < slots; line += nt) {
    float J[6], wg = 0.0f, wh = 0.0f;
    bool ok = false;
    if (line < nl && (f2i_bits(s.state[LS_VALID * nl + line]) & 1)) {  // :497-513
      ok = true;
      float cx = s.state[LS_CX * nl + line], cy = s.state[LS_CY * nl + line], cz = s.state[LS_CZ * nl + line];
      float x, y, z;
      apply_pose(b2c, cx, cy, cz, x, y, z);
      float normal_u = s.state[LS_NORMAL_U * nl + line], normal_v = s.state[LS_NORMAL_V * nl + line];
      float center_u = s.state[LS_CENTER_U * nl + line], center_v = s.state[LS_CENTER_V * nl + line];
      float ncts = s.state[LS_NCTS * nl + line];
      float measured_variance = s.state[LS_VAR * nl + line];
      float fu_z = cam.fu / z;
      float fv_z = cam.fv / z;
      float xfu_z = x * fu_z;
      float yfv_z = y * fv_z;
      float delta_cs = (normal_u * (xfu_z + cam.ppu - center_u) + normal_v * (yfv_z + cam.ppv - center_v) -
                        s.state[LS_DELTA_R * nl + line]) *
                       ncts;
      float dll = 0.0f;
      if (opt_iteration < m.n_global_iterations) {
        dll = (s.state[LS_MEAN * nl + line] - delta_cs) / measured_variance;
      } else {
        int upper = f2i(delta_cs + m.distribution_length_plus_1_half);
        int lower = upper - 1;
        if (upper <= 0 || upper >= m.distribution_length) {
          ok = false;
        } else {
          // std::log(float): the f32 nearest to the logarithm, taken through f64 on both sides of the parity check.
          // m3t_log.h gets there with a fifth of a general double logarithm's instructions and says when it cannot
          // vouch for the rounding (one call in 10^5: the general logarithm decides)
          const float d_upper = s.state[(LS_DIST0 + upper) * nl + line], d_lower = s.state[(LS_DIST0 + lower) * nl + line];
          typedef const __attribute__((address_space(3))) double* LdsDoubles;
          LdsDoubles log_table = (LdsDoubles)(s.misc + kMiscLogTable);
          float log_upper = 0.0f, log_lower = 0.0f;
          const bool vouched = m3t_log_fast(d_upper, log_table, &log_upper) & m3t_log_fast(d_lower, log_table, &log_lower);
          if (!vouched) {
            log_upper = (float)log((double)d_upper);
            log_lower = (float)log((double)d_lower);
          }
          dll = (log_upper - log_lower) * m.learning_rate / measured_variance;
        }
      }
      float dc0 = ncts * normal_u * fu_z;
      float dc1 = ncts * normal_v * fv_z;
      float dc2 = ncts * (-normal_u * xfu_z - normal_v * yfv_z) / z;
      // RowVector3f * Matrix3f (body2camera_rotation_)
      float t0 = (dc0 * b2c.l[0] + dc1 * b2c.l[1]) + dc2 * b2c.l[2];
      float t1 = (dc0 * b2c.l[3] + dc1 * b2c.l[4]) + dc2 * b2c.l[5];
      float t2 = (dc0 * b2c.l[6] + dc1 * b2c.l[7]) + dc2 * b2c.l[8];
      J[0] = cy * t2 - cz * t1;
      J[1] = cz * t0 - cx * t2;
      J[2] = cx * t1 - cy * t0;
      J[3] = t0;
      J[4] = t1;
      J[5] = t2;
      float weight = m.min_expected_variance / (ncts * ncts * it.variance);
      wg = weight * dll;
      wh = weight / measured_variance;
    }
    float* out = rows + line;
    if (ok) {
#pragma unroll
      for (int r = 0; r < 6; ++r) out[r * pitch] = -(wg * J[r]);  // gradient_ += (weight * dll) * J^T
      int k = 6;
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int r = c; r < 6; ++r) out[(k++) * pitch] = (wh * J[r]) * J[c];  // hessian_ (lower) -= ((w / var) J^T) J
    } else {
#pragma unroll
      for (int k = 0; k < 27; ++k) out[k * pitch] = 0.0f;
    }
  }
}

// ---------------------------------------------------------------------------
// Line results of the workgroups that share one object (tracking_step_split_kernel).
// Every workgroup walks the pixels of its own part of the lines (and scans the depth windows of its own part of
// the points); what the Newton steps of ALL workgroups need from that -- mean, variance and distribution of a line,
// correspondence and validity of a point -- crosses CUs once per correspondence iteration as 8-byte {tag, value}
// granules: one write-through store each, re-read by the others until the tag matches.  The data is the flag, no
// fence (cdna_hip_programming.md guideline 16, form R2).  Slots alternate with the round: a workgroup can only be
// one round ahead of another, because it cannot leave a round before it has read everybody's granules of it.
// From there on every workgroup holds the same line / point state, forms the same sums in the reference's order
// and solves the same system: no exchange inside the Newton steps, bit-identical poses in every workgroup.
// A wait that runs out (a partner that is not resident: another process on the GPU) ends the whole object's step
// without writing anything and raises the context's abort flag, which the host reads at its next call.
// ---------------------------------------------------------------------------

struct SplitExchangeView {  // what publish and collect derive from the descriptor
  uint32_t tag;
  int lmask, nfr, nfd, nf, region_row0, depth_row0;
  __attribute__((address_space(1))) unsigned long long* slot;
  float* lds0;
};
__device__ __forceinline__ SplitExchangeView split_exchange_view(const SplitExchange& x, int round, const Lds& s,
                                                                 bool with_region, float* ps, int np, bool with_depth) {
  SplitExchangeView v;
  v.tag = x.seq * 64u + (uint32_t)round + 1u;
  v.lmask = (1 << x.lshift) - 1;
  v.nfr = with_region ? x.n_region_fields : 0;
  v.nfd = with_depth ? x.n_depth_fields : 0;
  v.nf = v.nfr + v.nfd;
  v.slot = x.granules + ((size_t)(round & 1) * x.n_parts << (kExchangeFieldBits + x.lshift));
  v.lds0 = s.misc;  // the lowest address of the workgroup's LDS carve-up
  // field f -> element offset of its row from lds0
  v.region_row0 = int(s.state - v.lds0) + x.first_region_row * s.nl;
  v.depth_row0 = int(ps - v.lds0) + x.first_depth_row * np;
  return v;
}
__device__ __forceinline__ void split_exchange_publish(const SplitExchange& x, int round, const Lds& s, bool with_region,
                                                       float* ps, int np, bool with_depth, int first_field = 0) {
  // first_field: the fields below it have been sent already (region_correspondences, early)
  const int tid = threadIdx.x, nt = blockDim.x;
  const SplitExchangeView v = split_exchange_view(x, round, s, with_region, ps, np, with_depth);
  const uint32_t tag = v.tag;
  const int lmask = v.lmask, nfr = v.nfr, nf = v.nf, region_row0 = v.region_row0, depth_row0 = v.depth_row0;
  float* const lds0 = v.lds0;
  auto* mine = v.slot + ((size_t)x.part << (kExchangeFieldBits + x.lshift));
  for (int idx = tid + (first_field << x.lshift); idx < (nf << x.lshift); idx += nt) {
    const int f = idx >> x.lshift, l = idx & lmask;
    const bool region = f < nfr;
    const int per_part = region ? x.per_part_lines : x.per_part_points, count = region ? s.nl : np;
    const int e = x.part * per_part + l;
    if (l < per_part && e < count) {
      const float val = lds0[(region ? region_row0 + f * s.nl : depth_row0 + (f - nfr) * np) + e];
      const unsigned long long g = (static_cast<unsigned long long>(tag) << 32) | (unsigned)__float_as_int(val);
      __hip_atomic_store(mine + idx, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// returns false when the exchange timed out (block-uniform); ends with a barrier
__device__ __forceinline__ bool split_exchange_collect(const SplitExchange& x, int round, const Lds& s, bool with_region,
                                                       float* ps, int np, bool with_depth) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const SplitExchangeView v = split_exchange_view(x, round, s, with_region, ps, np, with_depth);
  const uint32_t tag = v.tag;
  const int lmask = v.lmask, nfr = v.nfr, nf = v.nf, region_row0 = v.region_row0, depth_row0 = v.depth_row0;
  float* const lds0 = v.lds0;
  auto* slot = v.slot;
  // collect the other parts': thread -> (part q, element l) fixed; its wave takes the fields group, group + n_groups, ...
  // (the field index is wave-uniform: granule row and LDS row are scalar, per thread only one offset of each kind)
  bool timed_out = false;
  {
    const int lanes = x.n_parts << x.lshift;  // M3T_SPLIT_LANES, a multiple of the wave size
    const int ql = tid & (lanes - 1);
    const int q = ql >> x.lshift, l = ql & lmask;
    const int n_groups = nt / lanes, group = __builtin_amdgcn_readfirstlane(tid / lanes);
    const int e_r = q * x.per_part_lines + l, e_d = q * x.per_part_points + l;
    const bool ok_r = q != x.part && l < x.per_part_lines && e_r < s.nl;
    const bool ok_d = q != x.part && l < x.per_part_points && e_d < np;
    auto* theirs = slot + ((size_t)q << (kExchangeFieldBits + x.lshift)) + l;
    // batches of 8 granules per thread: all loads of a batch are issued before the first tag is looked at
    // (straight-line code on named registers: an indexed private array would live in scratch memory).
    // Round 5, tools/exchange_waits.py (the stamps without the phase marks): every part waits 4.4-5.3 k cycles between
    // its publish and its last granule, all parts alike -- no laggard.  Two other ways of polling were measured and are
    // slower at 64 objects (0.148 ms): re-reading all stale granules of a batch per poll (0.156: eight times the
    // polling traffic on the L2 the partners' stores have to get through) and waiting for the batch's first granule
    // alone before asking for the other seven (0.155: one more round trip in front of every batch).
    for (int f0 = group; f0 < nf; f0 += 8 * n_groups) {
      unsigned long long g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0, g6 = 0, g7 = 0;
#define M3T_EXCHANGE_LOAD(J, G)                                                                                      \
      {                                                                                                              \
        const int f = f0 + J * n_groups;                                                                             \
        if (f < nf && (f < nfr ? ok_r : ok_d))                                                                       \
          G = __hip_atomic_load(theirs + ((size_t)f << x.lshift), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        \
      }
#define M3T_EXCHANGE_WAIT(J, G)                                                                                      \
      {                                                                                                              \
        const int f = f0 + J * n_groups;                                                                             \
        if (f < nf && (f < nfr ? ok_r : ok_d)) {                                                                     \
          unsigned spins = 0;                                                                                        \
          while (static_cast<uint32_t>(G >> 32) != tag) {                                                            \
            if (++spins > (1u << 12) || /* ~5 ms: resident partners answer within microseconds */                    \
                ((spins & 255u) == 0 &&                                                                              \
                 __hip_atomic_load(x.object_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == x.seq)) {          \
              timed_out = true;                                                                                      \
              break;                                                                                                 \
            }                                                                                                        \
            __builtin_amdgcn_s_sleep(1);                                                                             \
            G = __hip_atomic_load(theirs + ((size_t)f << x.lshift), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      \
          }                                                                                                          \
          lds0[f < nfr ? region_row0 + f * s.nl + e_r : depth_row0 + (f - nfr) * np + e_d] =                         \
              __int_as_float(static_cast<int>(static_cast<uint32_t>(G)));                                            \
        }                                                                                                            \
      }
      M3T_EXCHANGE_LOAD(0, g0) M3T_EXCHANGE_LOAD(1, g1) M3T_EXCHANGE_LOAD(2, g2) M3T_EXCHANGE_LOAD(3, g3)
      M3T_EXCHANGE_LOAD(4, g4) M3T_EXCHANGE_LOAD(5, g5) M3T_EXCHANGE_LOAD(6, g6) M3T_EXCHANGE_LOAD(7, g7)
      M3T_EXCHANGE_WAIT(0, g0) M3T_EXCHANGE_WAIT(1, g1) M3T_EXCHANGE_WAIT(2, g2) M3T_EXCHANGE_WAIT(3, g3)
      M3T_EXCHANGE_WAIT(4, g4) M3T_EXCHANGE_WAIT(5, g5) M3T_EXCHANGE_WAIT(6, g6) M3T_EXCHANGE_WAIT(7, g7)
#undef M3T_EXCHANGE_LOAD
#undef M3T_EXCHANGE_WAIT
    }
  }
  if (__syncthreads_or(timed_out ? 1 : 0)) {
    if (tid == 0) {
      __hip_atomic_store(x.object_abort, x.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(x.host_abort, x.abort_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return false;
  }
  return true;
}

// ---------------------------------------------------------------------------
// serial 3x3 helpers (column-major, Eigen coefficient order): used by the kinematic-structure kernels (m3t_links.hip)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void mul3(const float* a, const float* b, float* r) {
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int k = 0; k < 3; ++k) r[c * 3 + k] = (a[k] * b[c * 3] + a[3 + k] * b[c * 3 + 1]) + a[6 + k] * b[c * 3 + 2];
}
__device__ __forceinline__ void add_scaled3(const float* a, float sa, const float* b, float sb, float* r) {
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = sa * a[i] + sb * b[i];
}
// X = A^-1 B, Gaussian elimination with partial pivoting (first largest |a(i,k)| wins)
__device__ void solve3(float* a, float* b, float* x) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int p = k;
    float best = fabsf(a[k * 3 + k]);
#pragma unroll
    for (int i = k + 1; i < 3; ++i) {
      float v = fabsf(a[k * 3 + i]);
      if (v > best) { best = v; p = i; }
    }
#pragma unroll
    for (int j = k + 1; j < 3; ++j) {
      if (p == j) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float t = a[c * 3 + k]; a[c * 3 + k] = a[c * 3 + j]; a[c * 3 + j] = t;
          t = b[c * 3 + k]; b[c * 3 + k] = b[c * 3 + j]; b[c * 3 + j] = t;
        }
      }
    }
#pragma unroll
    for (int i = k + 1; i < 3; ++i) {
      float f = a[k * 3 + i] / a[k * 3 + k];
      a[k * 3 + i] = f;
#pragma unroll
      for (int c = k + 1; c < 3; ++c) a[c * 3 + i] -= f * a[c * 3 + k];
#pragma unroll
      for (int c = 0; c < 3; ++c) b[c * 3 + i] -= f * b[c * 3 + k];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 2; i >= 0; --i) {
      float sacc = b[c * 3 + i];
#pragma unroll
      for (int j = i + 1; j < 3; ++j) sacc -= a[j * 3 + i] * x[c * 3 + j];
      x[c * 3 + i] = sacc / a[i * 3 + i];
    }
}
__device__ void expm3(const float* a_in, float* r) {
  float l1 = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float sacc = 0.0f;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) sacc += fabsf(a_in[c * 3 + rr]);
    l1 = fmaxf(l1, sacc);
  }
  const float I[9] = {1.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.0f, 0.0f, 1.0f};
  float U[9], V[9], a[9], a2[9], tmp[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = a_in[i];
  int squarings = 0;
  if (l1 < 4.258730016922831e-001f) {
    mul3(a, a, a2);
    add_scaled3(a2, 1.0f, I, 60.0f, tmp);
    mul3(a, tmp, U);
    add_scaled3(a2, 12.0f, I, 120.0f, V);
  } else if (l1 < 1.880152677804762e+000f) {
    float a4[9], t2[9];
    mul3(a, a, a2);
    mul3(a2, a2, a4);
    add_scaled3(a4, 1.0f, a2, 420.0f, t2);
    add_scaled3(t2, 1.0f, I, 15120.0f, tmp);
    mul3(a, tmp, U);
    add_scaled3(a4, 30.0f, a2, 3360.0f, t2);
    add_scaled3(t2, 1.0f, I, 30240.0f, V);
  } else {
    int e;
    (void)frexpf(l1 / 3.925724783138660f, &e);
    squarings = e > 0 ? e : 0;
    float sc = ldexpf(1.0f, -squarings);
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] *= sc;
    float a4[9], a6[9], t2[9], t3[9];
    mul3(a, a, a2);
    mul3(a2, a2, a4);
    mul3(a4, a2, a6);
    add_scaled3(a6, 1.0f, a4, 1512.0f, t2);
    add_scaled3(t2, 1.0f, a2, 277200.0f, t3);
    add_scaled3(t3, 1.0f, I, 8648640.0f, tmp);
    mul3(a, tmp, U);
    add_scaled3(a6, 56.0f, a4, 25200.0f, t2);
    add_scaled3(t2, 1.0f, a2, 1995840.0f, t3);
    add_scaled3(t3, 1.0f, I, 17297280.0f, V);
  }
  float num[9], den[9];
  add_scaled3(U, 1.0f, V, 1.0f, num);
  add_scaled3(U, -1.0f, V, 1.0f, den);
  solve3(den, num, r);
  for (int i = 0; i < squarings; ++i) {
    float t[9];
    mul3(r, r, t);
#pragma unroll
    for (int j = 0; j < 9; ++j) r[j] = t[j];
  }
}

// ---------------------------------------------------------------------------
// Optimizer::CalculateOptimization (optimizer.cpp:144-167) for dof = 6, J = I, plus
// Link::UpdatePoses (link.cpp:205-241) with body2joint = I, executed by ONE WAVE.
//
// Eigen's LDLT<Lower> (the oracle's LdltSolve) is the bordered, "lazy" variant: step k only writes
// column k, the trailing matrix is never updated, so the diagonal pivoting looks at the ORIGINAL
// diagonal entries and the whole transposition sequence is known before the factorisation starts.
// With distinct, non-NaN |a_ii| the pivots are simply the diagonal in descending order: lane j finds
// its rank with six compares, the permuted matrix P A P^T is gathered from LDS, and the factorisation
// runs without pivoting on six lanes (lane i = row i): element for element the operations of the
// pivoted in-place algorithm, only moved to where the swaps would have carried them.  Equal or NaN
// diagonal entries (H = 0: no valid line) take the serial restatement below.  exp(skew) and the pose
// product run column-per-lane.  Everything is IEEE f32, -ffp-contract=off: bit-identical to the oracle.
// ---------------------------------------------------------------------------
constexpr int kMiscSolve = 160;  // 128 floats of LDS scratch inside `misc`

__device__ __forceinline__ float rlf(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// Serial fallback (lane 0): the oracle's LdltSolve for n = 6 on LDS arrays.  a: column-major, lower.
__device__ void ldlt6_fallback(float* a, float* x, int* trans, float* temp) {
  constexpr int n = 6;
#pragma nounroll
  for (int k = 0; k < n; ++k) {
    int piv = k;
    float best = fabsf(a[k * n + k]);
#pragma nounroll
    for (int i = k + 1; i < n; ++i) {
      float v = fabsf(a[i * n + i]);
      if (v > best) { best = v; piv = i; }
    }
    trans[k] = piv;
    if (piv != k) {  // symmetric swap of k and piv inside the lower triangle (Eigen ldlt_inplace)
#pragma nounroll
      for (int c = 0; c < k; ++c) { float t = a[c * n + k]; a[c * n + k] = a[c * n + piv]; a[c * n + piv] = t; }
#pragma nounroll
      for (int i = piv + 1; i < n; ++i) { float t = a[k * n + i]; a[k * n + i] = a[piv * n + i]; a[piv * n + i] = t; }
      { float t = a[k * n + k]; a[k * n + k] = a[piv * n + piv]; a[piv * n + piv] = t; }
#pragma nounroll
      for (int i = k + 1; i < piv; ++i) { float t = a[k * n + i]; a[k * n + i] = a[i * n + piv]; a[i * n + piv] = t; }
    }
    if (k > 0) {
#pragma nounroll
      for (int c = 0; c < k; ++c) temp[c] = a[c * n + c] * a[c * n + k];
      float acc = 0.0f;
#pragma nounroll
      for (int c = 0; c < k; ++c) acc += a[c * n + k] * temp[c];
      a[k * n + k] -= acc;
#pragma nounroll
      for (int i = k + 1; i < n; ++i) {
        float sacc = 0.0f;
#pragma nounroll
        for (int c = 0; c < k; ++c) sacc += a[c * n + i] * temp[c];
        a[k * n + i] -= sacc;
      }
    }
    float akk = a[k * n + k];
    bool pivot_valid = fabsf(akk) > 0.0f;
    if (k == 0 && !pivot_valid) {
#pragma nounroll
      for (int j = 0; j < n; ++j) trans[j] = j;
      break;
    }
    if (pivot_valid) {
#pragma nounroll
      for (int i = k + 1; i < n; ++i) a[k * n + i] /= akk;
    }
  }
#pragma nounroll
  for (int k = 0; k < n; ++k) { float t = x[k]; x[k] = x[trans[k]]; x[trans[k]] = t; }
#pragma nounroll
  for (int i = 0; i < n; ++i) {
    float sacc = x[i];
#pragma nounroll
    for (int c = 0; c < i; ++c) sacc -= a[c * n + i] * x[c];
    x[i] = sacc;
  }
#pragma nounroll
  for (int i = 0; i < n; ++i) {
    if (fabsf(a[i * n + i]) > 1.17549435e-38f) x[i] /= a[i * n + i];
    else x[i] = 0.0f;
  }
#pragma nounroll
  for (int i = n - 1; i >= 0; --i) {
    float sacc = x[i];
#pragma nounroll
    for (int r = i + 1; r < n; ++r) sacc -= a[i * n + r] * x[r];
    x[i] = sacc;
  }
#pragma nounroll
  for (int k = n - 1; k >= 0; --k) { float t = x[k]; x[k] = x[trans[k]]; x[trans[k]] = t; }
}

// ---- column-per-lane 3x3 helpers (column c of a matrix lives in lane c; `u` = the same matrix in every lane,
// column-major like Eigen).  The expression trees are those of the oracle's Mul3 / AddScaled / Solve3.
__device__ __forceinline__ void colmul(const float (&u)[9], const float (&col)[3], float (&r)[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) r[k] = (u[k] * col[0] + u[3 + k] * col[1]) + u[6 + k] * col[2];
}
// QUAD: the three columns live in lanes 0..2 of every group of four lanes (a matrix per group), not in lanes 0..2 of
// the wave (one matrix per wave): the gather is a DPP quad broadcast instead of a v_readlane
template <int C>
__device__ __forceinline__ float quad_lane(float v) {  // lane C of the caller's group of four
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), C * 0x55, 0xf, 0xf, false));
}
template <bool QUAD = false>
__device__ __forceinline__ void colgather(const float (&col)[3], float (&u)[9]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if constexpr (QUAD) {
      u[r] = quad_lane<0>(col[r]);
      u[3 + r] = quad_lane<1>(col[r]);
      u[6 + r] = quad_lane<2>(col[r]);
    } else {
      u[r] = rlf(col[r], 0);
      u[3 + r] = rlf(col[r], 1);
      u[6 + r] = rlf(col[r], 2);
    }
  }
}
__device__ __forceinline__ void coladd(const float (&a)[3], float sa, const float (&b)[3], float sb, float (&r)[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) r[i] = sa * a[i] + sb * b[i];
}
// X = A^-1 B: Gaussian elimination with partial pivoting (first largest |a(i,k)| wins) on the uniform A,
// every lane carries its own column of B through the same row operations
__device__ __forceinline__ void colsolve3(float (&a)[9], float (&b)[3], float (&x)[3]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int p = k;
    float best = fabsf(a[k * 3 + k]);
#pragma unroll
    for (int i = k + 1; i < 3; ++i) {
      float v = fabsf(a[k * 3 + i]);
      if (v > best) { best = v; p = i; }
    }
#pragma unroll
    for (int j = k + 1; j < 3; ++j) {
      if (p == j) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { float t = a[c * 3 + k]; a[c * 3 + k] = a[c * 3 + j]; a[c * 3 + j] = t; }
        float t = b[k]; b[k] = b[j]; b[j] = t;
      }
    }
#pragma unroll
    for (int i = k + 1; i < 3; ++i) {
      float f = a[k * 3 + i] / a[k * 3 + k];
      a[k * 3 + i] = f;
#pragma unroll
      for (int c = k + 1; c < 3; ++c) a[c * 3 + i] -= f * a[c * 3 + k];
      b[i] -= f * b[k];
    }
  }
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    float sacc = b[i];
#pragma unroll
    for (int j = i + 1; j < 3; ++j) sacc -= a[j * 3 + i] * x[j];
    x[i] = sacc / a[i * 3 + i];
  }
}

// exp(skew(theta_r)): Pade approximant with scaling and squaring like Eigen's MatrixFunctions (link.cpp:224),
// operation for operation the oracle's Expm3; the result's column c is returned in lane c (c < 3).
template <bool QUAD = false>
__device__ __forceinline__ void colexpm3(const float (&K)[9], int c, float (&R)[3]) {
  float l1 = 0.0f;
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) {
    float sacc = 0.0f;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) sacc += fabsf(K[cc * 3 + rr]);
    l1 = fmaxf(l1, sacc);
  }
  const float Ic[3] = {c == 0 ? 1.0f : 0.0f, c == 1 ? 1.0f : 0.0f, c == 2 ? 1.0f : 0.0f};
  float a[9], ac[3], U[3], V[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) a[i] = K[i];
  int squarings = 0;
  if (l1 >= 1.880152677804762e+000f) {
    int e;
    (void)frexpf(l1 / 3.925724783138660f, &e);
    squarings = e > 0 ? e : 0;
    float sc = ldexpf(1.0f, -squarings);
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] *= sc;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) ac[r] = c == 0 ? a[r] : (c == 1 ? a[3 + r] : a[6 + r]);
  float a2c[3], tmp[3];
  colmul(a, ac, a2c);  // a2 = a * a
  if (l1 < 4.258730016922831e-001f) {
    coladd(a2c, 1.0f, Ic, 60.0f, tmp);
    colmul(a, tmp, U);
    coladd(a2c, 12.0f, Ic, 120.0f, V);
  } else {
    float a2[9], a4c[3], t2[3];
    colgather<QUAD>(a2c, a2);
    colmul(a2, a2c, a4c);  // a4 = a2 * a2
    if (l1 < 1.880152677804762e+000f) {
      coladd(a4c, 1.0f, a2c, 420.0f, t2);
      coladd(t2, 1.0f, Ic, 15120.0f, tmp);
      colmul(a, tmp, U);
      coladd(a4c, 30.0f, a2c, 3360.0f, t2);
      coladd(t2, 1.0f, Ic, 30240.0f, V);
    } else {
      float a4[9], a6c[3], t3[3];
      colgather<QUAD>(a4c, a4);
      colmul(a4, a2c, a6c);  // a6 = a4 * a2
      coladd(a6c, 1.0f, a4c, 1512.0f, t2);
      coladd(t2, 1.0f, a2c, 277200.0f, t3);
      coladd(t3, 1.0f, Ic, 8648640.0f, tmp);
      colmul(a, tmp, U);
      coladd(a6c, 56.0f, a4c, 25200.0f, t2);
      coladd(t2, 1.0f, a2c, 1995840.0f, t3);
      coladd(t3, 1.0f, Ic, 17297280.0f, V);
    }
  }
  float num[3], denc[3], den[9];
  coladd(U, 1.0f, V, 1.0f, num);
  coladd(U, -1.0f, V, 1.0f, denc);
  colgather<QUAD>(denc, den);
  colsolve3(den, num, R);
#pragma nounroll
  for (int i = 0; i < squarings; ++i) {
    float r9[9], t[3];
    colgather<QUAD>(R, r9);
    colmul(r9, R, t);
#pragma unroll
    for (int j = 0; j < 3; ++j) R[j] = t[j];
  }
}

// gh: lane l < 6 holds the link's gradient sum g[l], lane 6 + c * 6 + r the Hessian sum H(r, c) (full, symmetric:
// the layout of Modality::gradient() / hessian()).  pose: 16 floats in LDS (column-major), updated in place.
// scratch: 128 floats of LDS.  All 64 lanes of the wave must call this together.
typedef __attribute__((address_space(3))) float* LdsW;  // LDS pointers stay LDS pointers across the (non-inlined) call:
                                                         // ds_read / ds_write instead of flat accesses in the solve
__device__ __noinline__ void rigid_solve_wave(float gh, float lambda_rot, float lambda_trans, LdsW pose, LdsW scratch) {
  const int lane = threadIdx.x & (kWave - 1);
  PHASE_T0();
  if (lane < 42) scratch[lane] = gh;
  __builtin_amdgcn_wave_barrier();
  // A = -H + diag(lambda) (only the lower triangle is ever read), b = g
  float v[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) v[j] = fabsf((0.0f - scratch[6 + j * 7]) + (j < 3 ? lambda_rot : lambda_trans));
  const int lj = lane < 6 ? lane : 5;
  const float my = fabsf((0.0f - scratch[6 + lj * 7]) + (lj < 3 ? lambda_rot : lambda_trans));
  int rank = 0, equal = 0;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    rank += v[j] > my ? 1 : 0;
    equal += v[j] == my ? 1 : 0;
  }
  const bool slow = __builtin_amdgcn_ballot_w64(lane < 6 && equal != 1) != 0;  // ties or NaN on the diagonal
  float th[6];  // theta, uniform
  if (!slow) {
    // position i of the pivoted order holds the original row pos[i]; lane i < 6 works on row i of P A P^T
    // (lanes >= 6 push to themselves: the six ranks are a permutation of 0..5, nothing collides)
    const int pi = __builtin_amdgcn_ds_permute((lane < 6 ? rank : lane) << 2, lane);
    int pos[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) pos[c] = __builtin_amdgcn_readlane(lane < 6 ? pi : 0, c);
    float b[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      b[c] = 0.0f - scratch[6 + pos[c] * 6 + (lane < 6 ? pi : 0)];
      const float lam = pos[c] < 3 ? lambda_rot : lambda_trans;
      b[c] = c == lane ? b[c] + lam : b[c];
    }
    float x = 0.0f + scratch[lane < 6 ? pi : 0];
    PHASE_MARK(12);  // build P A P^T, b
    float D[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k > 0) {
        float T[5];
#pragma unroll
        for (int c = 0; c < k; ++c) T[c] = rlf(D[c] * b[c], k);  // temp = D * A10^T, from row k
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < k; ++c) acc += b[c] * T[c];
        b[k] -= acc;  // row k: the pivot D_k; rows below: A21 -= A20 * temp
      }
      D[k] = rlf(b[k], k);
      if (k < 5) {
        const bool pivot_valid = fabsf(D[k]) > 0.0f;
        const float q = b[k] / D[k];
        b[k] = (pivot_valid && lane > k) ? q : b[k];
      }
    }
    PHASE_MARK(13);  // LDLT factorisation
    // L y = P b (row i: c ascending)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float xc = rlf(x, c);
      x = lane > c ? x - b[c] * xc : x;
    }
    // D z = y with the pseudo-inverse of D
    float dself = D[5];
#pragma unroll
    for (int k = 4; k >= 0; --k) dself = lane == k ? D[k] : dself;
    x = fabsf(dself) > 1.17549435e-38f ? x / dself : 0.0f;
    // L^T w = z: row i needs x[i+1] first, so this part is a chain; every lane runs it on broadcast values
    float X[6], Lt[15];
#pragma unroll
    for (int i = 0; i < 6; ++i) X[i] = rlf(x, i);
    {
      int n = 0;
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int r = i + 1; r < 6; ++r) Lt[n++] = rlf(b[i], r);  // L(r, i)
    }
    {
      int base = 15;
#pragma unroll
      for (int i = 4; i >= 0; --i) {
        base -= 5 - i;
        float sacc = X[i];
#pragma unroll
        for (int r = i + 1; r < 6; ++r) sacc -= Lt[base + (r - i - 1)] * X[r];
        X[i] = sacc;
      }
    }
    // theta = P^T w: original row j sits at position rank[j]
    float mine = X[5];
#pragma unroll
    for (int k = 4; k >= 0; --k) mine = rank == k ? X[k] : mine;
#pragma unroll
    for (int j = 0; j < 6; ++j) th[j] = rlf(mine, j);
  } else {
    float* a = (float*)(scratch + 48);
    float* xs = (float*)(scratch + 84);
    int* trans = reinterpret_cast<int*>((float*)(scratch + 90));
    float* temp = (float*)(scratch + 96);
    if (lane < 36) {
      const int c = lane / 6, r = lane - c * 6;
      float e = r >= c ? 0.0f - scratch[6 + lane] : 0.0f;
      if (r == c) e += c < 3 ? lambda_rot : lambda_trans;
      a[lane] = e;
    }
    if (lane < 6) xs[lane] = 0.0f + scratch[lane];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) ldlt6_fallback(a, xs, trans, temp);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 6; ++j) th[j] = xs[j];
  }
  PHASE_MARK(14);  // triangular solves
  // NaN guard (optimizer.cpp:165): skip the update, still success
  bool has_nan = false;
#pragma unroll
  for (int i = 0; i < 6; ++i) has_nan |= (th[i] != th[i]);
  if (has_nan) return;
  // dT = [exp(skew(theta_r)) | theta_t], pose <- pose * dT (link.cpp:218-232 with body2joint = I)
  float K[9];  // column-major skew(theta_r), common.h:62-68
  K[0] = 0.0f;   K[3] = -th[2]; K[6] = th[1];
  K[1] = th[2];  K[4] = 0.0f;   K[7] = -th[0];
  K[2] = -th[1]; K[5] = th[0];  K[8] = 0.0f;
  const int c = lane & 3;
  float col[3];
  colexpm3(K, c, col);
  PHASE_MARK(15);  // expm
  if (c == 3) { col[0] = th[3]; col[1] = th[4]; col[2] = th[5]; }
  const Affine T = load_pose(pose);
  float N[3];
  colmul(T.l, col, N);
  if (c == 3) {
#pragma unroll
    for (int k = 0; k < 3; ++k) N[k] = N[k] + T.t[k];
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < 4) {
#pragma unroll
    for (int k = 0; k < 3; ++k) pose[lane * 4 + k] = N[k];
    pose[lane * 4 + 3] = lane == 3 ? 1.0f : 0.0f;
  }
}

// ---------------------------------------------------------------------------
// DepthModality::CalculateCorrespondences (depth_modality.cpp:252-315), whole block.
// One thread per point: CalculateBasicPointData :656, IsPointValid :697,
// IsPointUnoccludedMeasured :736, FindCorrespondence :826 (strided window scan,
// first strictly smaller distance wins in (v outer, u inner) order).
// Point state lives in `ps` (LDS, [PS_FIELDS][np]).
// ---------------------------------------------------------------------------
// Part 1 (depth_correspondences_scan): everything up to the per-point flags {bit 0: unoccluded, bit 1: valid} for the
// points [pt_lo, pt_hi); a workgroup that shares its object with others (tracking_step_split_kernel) also loads the
// model data of the other parts' points, whose correspondences arrive through split_exchange_state().
// Part 2 (depth_correspondences_vote): the two-pass fallback :282-313 over all points.
template <bool RENDER = true>
__device__ __forceinline__ void depth_correspondences_scan(CDepth& m, CCam& cam, const Affine& b2c, int iteration,
                                           int corr_iteration, float* ps, int np, float* misc, int pt_lo = 0,
                                           int pt_hi = 1 << 30, int known_view = -1) {
  // 16 lanes (one DPP row) per model point: the strided search window of FindCorrespondence
  // (<= 15x15 depth samples) and the occlusion window (<= 6x6) are scanned by the row in
  // parallel; the winner is the smallest distance, ties to the lowest scan index == the
  // reference's "first strictly smaller in (v outer, u inner) order".
  constexpr int kGroup = 16;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int gl = tid % kGroup;
  PHASE_T0();
  const int view = known_view >= 0 ? known_view : closest_view((G<v4f>)m.orientations4, m.n_views, b2c, misc);
  PHASE_MARK(17);
  int n_points = number_of_lines(m.n_points_max, m.use_adaptive_coverage, m.reference_surface_area, as_global(m.extents), view,
                                 m.max_extent, m.n_points);
  const float considered_distance0 = last_valid(m.considered_distances, m.n_considered_distances, corr_iteration);
  const int max_n_strides = f2i(considered_distance0 / m.stride_length + 0.5f);
  const bool handle_occlusions = (iteration - m.first_iteration) >= m.n_unoccluded_iterations;
  const bool measured_pass = m.measure_occlusions && handle_occlusions;
  const bool modeled_pass = RENDER && m.model_occlusions && handle_occlusions &&
                            renderer_body_visible(m.depth_renderer, m.depth_renderer_slot);
  const bool silhouette_checking =
      RENDER && m.use_silhouette_checking && renderer_body_visible(m.silhouette_renderer, m.silhouette_renderer_slot);
  const bool occlusion_pass = measured_pass || modeled_pass;
  G<uint8_t> image = as_global(cam.image);
  const int own_hi = pt_hi < np ? pt_hi : np;
  const int own_count = own_hi > pt_lo ? own_hi - pt_lo : 0;
  const int np_round = (own_count + nt / kGroup - 1) / (nt / kGroup) * (nt / kGroup);
  for (int i = pt_lo + tid / kGroup; i < pt_lo + np_round; i += nt / kGroup) {
    int flags = 0;
    const bool in_model = i < n_points && i < own_hi;
    const int ip = in_model ? i : 0;
    G<float> p = as_global(m.points) + ((size_t)view * m.n_points + ip) * M3T_DEPTH_POINT_FLOATS;
    G<v4f> p8 = (G<v4f>)m.points8 + ((uint32_t)view * m.n_points + ip) * 2;
    const v4f pa = p8[0], pb4 = p8[1];
    float cx = pa.x, cy = pa.y, cz = pa.z;
    float X, Y, Z;
    apply_pose(b2c, cx, cy, cz, X, Y, Z);
    float center_u = X * cam.fu / Z + cam.ppu;
    float center_v = Y * cam.fv / Z + cam.ppv;
    float depth = Z;
    bool valid = in_model && !(depth <= 0.0f);
    if (valid) {
      int icu = f2i(center_u + 0.5f), icv = f2i(center_v + 0.5f);
      valid = !(icu < 0 || icu > cam.width - 1 || icv < 0 || icv > cam.height - 1);
      if (valid && silhouette_checking) {  // IsPointOnValidSilhouette :728-734, SilhouetteValue silhouette_renderer.cpp:394-399
        const RendererDev& sr = *m.silhouette_renderer;
        int fu_ = f2i(((float)icu - sr.state[RS_CORNER_U]) * sr.state[RS_SCALE] + 0.5f);
        int fv_ = f2i(((float)icv - sr.state[RS_CORNER_V]) * sr.state[RS_SCALE] + 0.5f);
        valid = fu_ >= 0 && fu_ < sr.image_size && fv_ >= 0 && fv_ < sr.image_size &&
                sr.silhouette_image[(size_t)fv_ * sr.image_size + fu_] == m.body_id;
      }
    }
    // FindCorrespondence: window limits (uniform inside the row)
    float cd = considered_distance0;
    if (m.use_depth_scaling) cd *= depth;
    int stride = 1, u_min = 0, v_min = 0, n_u = 0, n_v = 0;
    if (valid) {
      float meter_to_pixel = cam.fu / depth;
      float diameter = 2.0f * cd * meter_to_pixel;
      stride = f2i(diameter / max_n_strides + 1.0f);
      int n_strides = f2i(diameter / stride + 0.5f);
      int rounded_diameter = n_strides * stride;
      float rounded_radius = 0.5f * (float)rounded_diameter;
      u_min = f2i(center_u - rounded_radius + 0.5f);
      v_min = f2i(center_v - rounded_radius + 0.5f);
      int u_max = u_min + rounded_diameter;
      int v_max = v_min + rounded_diameter;
      u_min = max(u_min, 0);
      v_min = max(v_min, 0);
      u_max = min(u_max, cam.width - 1);
      v_max = min(v_max, cam.height - 1);
      if (stride < 1) stride = 1;
      n_u = u_max >= u_min ? (u_max - u_min) / stride + 1 : 0;
      n_v = v_max >= v_min ? (v_max - v_min) / stride + 1 : 0;
    }
    PHASE_MARK(18);
    const float min_depth_value = fminf(0.0f, (depth - cd) / cam.depth_scale);
    const float max_depth_value = (depth + cd) / cam.depth_scale;
    const float min_considered = cd * cd;
    float best = min_considered;
    int best_pos = INT_MAX;
    const int total = n_u * n_v;
    {
      // this lane's samples: pos = gl, gl + 16, ... in scan order; 8 loads in flight
      int ui = 0, vi = 0;
      if (n_u > 0) { vi = gl / n_u; ui = gl - vi * n_u; }
      const int du = kGroup % (n_u > 0 ? n_u : 1), dvq = kGroup / (n_u > 0 ? n_u : 1);
      for (int pos0 = gl; pos0 < total; pos0 += 8 * kGroup) {
        unsigned short raw[8];
        int us[8], vs[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          us[k] = u_min + ui * stride;
          vs[k] = v_min + vi * stride;
          raw[k] = 0;
          if (pos0 + k * kGroup < total)
            raw[k] = *reinterpret_cast<G<unsigned short>>(image + (uint32_t)vs[k] * cam.pitch + (uint32_t)us[k] * 2u);
          ui += du;
          vi += dvq;
          if (ui >= n_u) { ui -= n_u; ++vi; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (pos0 + k * kGroup >= total) continue;
          float d = (float)raw[k];
          if (d > min_depth_value && d < max_depth_value) {
            d *= cam.depth_scale;
            float t0 = ((float)us[k] - cam.ppu) * d / cam.fu;
            float t1 = ((float)vs[k] - cam.ppv) * d / cam.fv;
            float e0 = t0 - X, e1 = t1 - Y, e2 = d - Z;
            float dist2 = (e0 * e0 + e1 * e1) + e2 * e2;
            if (dist2 < best) {  // positions of one lane are visited in increasing scan order
              best = dist2;
              best_pos = pos0 + k * kGroup;
            }
          }
        }
      }
    }
    PHASE_MARK(19);
    // row reduction (min distance, lowest scan index); the result lands in lane 15 of the row
#define M3T_ROW_MIN_STEP(CTRL)                                                  \
    {                                                                         \
      float ob = dpp_self<CTRL, 0xf>(best);                                   \
      int op = dpp_self_i<CTRL, 0xf>(best_pos);                               \
      if (ob < best || (ob == best && op < best_pos)) { best = ob; best_pos = op; } \
    }
    M3T_ROW_MIN_STEP(0x111)
    M3T_ROW_MIN_STEP(0x112)
    M3T_ROW_MIN_STEP(0x114)
    M3T_ROW_MIN_STEP(0x118)
#undef M3T_ROW_MIN_STEP
    // measured occlusion (IsPointUnoccludedMeasured :736-776), the row ORs its samples
    int occluded = 0;
    if (measured_pass && valid) {
      float radius = m.measured_depth_offset_radius;
      if (m.use_depth_scaling) radius *= depth;
      int id = f2i(radius / m.stride_depth_offset + 0.5f);
      if (id >= M3T_N_DEPTH_OFFSETS) id = M3T_N_DEPTH_OFFSETS - 1;
      float diameter = 2.0f * m.measured_occlusion_radius * cam.fu;
      if (!m.use_depth_scaling) diameter /= depth;
      float threshold = m.measured_occlusion_threshold;
      if (m.use_depth_scaling) threshold *= depth;
      int ostride = f2i(diameter / M3T_MAX_N_OCCLUSION_STRIDES + 1.0f);
      int n_strides = f2i(diameter / ostride + 0.5f);
      int rounded_diameter = n_strides * ostride;
      float rounded_radius = 0.5f * (float)rounded_diameter;
      int ou_min = f2i(center_u - rounded_radius + 0.5f);
      int ov_min = f2i(center_v - rounded_radius + 0.5f);
      int ou_max = ou_min + rounded_diameter;
      int ov_max = ov_min + rounded_diameter;
      ou_min = max(ou_min, 0);
      ov_min = max(ov_min, 0);
      ou_max = min(ou_max, cam.width - 1);
      ov_max = min(ov_max, cam.height - 1);
      unsigned short min_depth = (unsigned short)f2i((depth - p[6 + id] - threshold) / cam.depth_scale);
      int on_u = ou_max >= ou_min ? (ou_max - ou_min) / ostride + 1 : 0;
      int on_v = ov_max >= ov_min ? (ov_max - ov_min) / ostride + 1 : 0;
      for (int pos = gl; pos < on_u * on_v; pos += kGroup) {
        int vi = pos / on_u, ui = pos - vi * on_u;
        unsigned short d = *reinterpret_cast<G<unsigned short>>(
            image + (uint32_t)(ov_min + vi * ostride) * cam.pitch + (uint32_t)(ou_min + ui * ostride) * 2u);
        if (d > 0 && d < min_depth) occluded = 1;
      }
    }
    occluded |= dpp_self_i<0x111, 0xf>(occluded);
    occluded |= dpp_self_i<0x112, 0xf>(occluded);
    occluded |= dpp_self_i<0x114, 0xf>(occluded);
    occluded |= dpp_self_i<0x118, 0xf>(occluded);
    PHASE_MARK(20);
    if (gl == kGroup - 1 && i < own_hi) {  // lane 15 of the row owns the reduced result
      valid = valid && best != min_considered;
      if (valid) {
        // recompute the winning sample (same operations as in the scan)
        int vi = best_pos / n_u, ui = best_pos - vi * n_u;
        int u = u_min + ui * stride, v = v_min + vi * stride;
        float d = (float)(*reinterpret_cast<G<unsigned short>>(image + (uint32_t)v * cam.pitch + (uint32_t)u * 2u));
        d *= cam.depth_scale;
        float t0 = ((float)u - cam.ppu) * d / cam.fu;
        float t1 = ((float)v - cam.ppv) * d / cam.fv;
        bool valid_occ = !occluded;
        if (valid_occ && modeled_pass) {  // IsPointUnoccludedModeled :778-824
          const RendererDev& dr = *m.depth_renderer;
          float radius = m.modeled_depth_offset_radius;
          if (m.use_depth_scaling) radius *= depth;
          int id = f2i(radius / m.stride_depth_offset + 0.5f);
          if (id >= M3T_N_DEPTH_OFFSETS) id = M3T_N_DEPTH_OFFSETS - 1;
          float meter_to_pixel = cam.fu * dr.state[RS_SCALE];
          if (!m.use_depth_scaling) meter_to_pixel /= depth;
          float diameter = 2.0f * m.modeled_occlusion_radius * meter_to_pixel;
          unsigned short min_value = modeled_window_min(dr, center_u, center_v, diameter);
          float threshold = m.modeled_occlusion_threshold;
          if (m.use_depth_scaling) threshold *= depth;
          valid_occ = renderer_depth(dr, min_value) > depth - p[6 + id] - threshold;
        }
        flags = (valid_occ ? 1 : 0) | 2;
        ps[PS_CX * np + i] = cx; ps[PS_CY * np + i] = cy; ps[PS_CZ * np + i] = cz;
        ps[PS_NX * np + i] = pa.w; ps[PS_NY * np + i] = pb4.x; ps[PS_NZ * np + i] = pb4.y;
        ps[PS_CENTER_U * np + i] = center_u;
        ps[PS_CENTER_V * np + i] = center_v;
        ps[PS_DEPTH * np + i] = depth;
        ps[PS_CORR_X * np + i] = t0; ps[PS_CORR_Y * np + i] = t1; ps[PS_CORR_Z * np + i] = d;
      }
      ps[PS_VALID * np + i] = i2f_bits(flags);
    }
    PHASE_MARK(21);
  }
  // model data of the points other workgroups scan (their correspondences and flags arrive by exchange)
  if (pt_lo > 0 || own_hi < np) {
    for (int i = tid; i < n_points && i < np; i += nt) {
      if (i >= pt_lo && i < own_hi) continue;
      G<v4f> p8 = (G<v4f>)m.points8 + ((uint32_t)view * m.n_points + i) * 2;
      const v4f pa = p8[0], pb4 = p8[1];
      ps[PS_CX * np + i] = pa.x; ps[PS_CY * np + i] = pa.y; ps[PS_CZ * np + i] = pa.z;
      ps[PS_NX * np + i] = pa.w; ps[PS_NY * np + i] = pb4.x; ps[PS_NZ * np + i] = pb4.y;
    }
  }
  __syncthreads();
}

template <bool RENDER = true, bool LEAN = false>
__device__ __forceinline__ void depth_correspondences_vote(CDepth& m, int iteration, float* ps, int np, float* misc) {
  int tid = threadIdx.x;
  if constexpr (LEAN) asm volatile("" : "+v"(tid));  // (the flag addresses are formed here, not once in front of the caller's loops)
  const int nt = blockDim.x;
  const bool handle_occlusions = (iteration - m.first_iteration) >= m.n_unoccluded_iterations;
  const bool measured_pass = m.measure_occlusions && handle_occlusions;
  const bool modeled_pass = RENDER && m.model_occlusions && handle_occlusions &&
                            renderer_body_visible(m.depth_renderer, m.depth_renderer_slot);
  bool use_occ = false;
  if (measured_pass || modeled_pass) {
    int mine = 0;
    if constexpr (LEAN) {
#pragma nounroll
      for (int i = tid; i < np; i += nt) mine += f2i_bits(ps[PS_VALID * np + i]) & 1;
    } else {
      for (int i = tid; i < np; i += nt) mine += f2i_bits(ps[PS_VALID * np + i]) & 1;
    }
    int cnt = wave_sum_i(mine);
    int* imisc = reinterpret_cast<int*>(misc);
    if (tid % kWave == 0) imisc[64 + tid / kWave] = cnt;
    __syncthreads();
    int total = 0;
    for (int w = 0; w < nt / kWave; ++w) total += imisc[64 + w];
    use_occ = total >= m.min_n_unoccluded_points;
  }
  const int valid_mask = use_occ ? 1 : 2;
  auto final_flag = [&](int i) {
    int flags = f2i_bits(ps[PS_VALID * np + i]);
    ps[PS_VALID * np + i] = i2f_bits((flags & ~1) | ((flags & valid_mask) ? 1 : 0));
  };
  if constexpr (LEAN) {
#pragma nounroll
    for (int i = tid; i < np; i += nt) final_flag(i);
  } else {
    for (int i = tid; i < np; i += nt) final_flag(i);
  }
  __syncthreads();
}
__device__ __forceinline__ void depth_correspondences(CDepth& m, CCam& cam, const Affine& b2c, int iteration, int corr_iteration,
                                      float* ps, int np, float* misc) {
  depth_correspondences_scan(m, cam, b2c, iteration, corr_iteration, ps, np, misc);
  depth_correspondences_vote(m, iteration, ps, np, misc);
}

// DepthModality::CalculateGradientAndHessian (depth_modality.cpp:333-381), the per-point part: one thread per
// point slot writes the point's 27 products to rows[row * pitch + point] (see chain_sums above).
__device__ __forceinline__ void depth_products(CDepth& m, const Affine& b2c, int corr_iteration, const float* ps, int np, float* rows,
                               int pitch, int tid0 = -1, int nt0 = 0) {
  const int tid = tid0 >= 0 ? tid0 : (int)threadIdx.x, nt = tid0 >= 0 ? nt0 : (int)blockDim.x;
  const Affine c2b = inverse_pose(b2c);
  const float standard_deviation = last_valid(m.standard_deviations, m.n_standard_deviations, corr_iteration);
  const int slots = chain_slots(np);
  for (int i = tid; i < slots; i += nt) {
    float* out = rows + i;
    if (i < np && (f2i_bits(ps[PS_VALID * np + i]) & 1)) {
      float qx, qy, qz;
      apply_pose(c2b, ps[PS_CORR_X * np + i], ps[PS_CORR_Y * np + i], ps[PS_CORR_Z * np + i], qx, qy, qz);
      float nx = ps[PS_NX * np + i], ny = ps[PS_NY * np + i], nz = ps[PS_NZ * np + i];
      float d0 = ps[PS_CX * np + i] - qx, d1 = ps[PS_CY * np + i] - qy, d2 = ps[PS_CZ * np + i] - qz;
      float epsilon = (nx * d0 + ny * d1) + nz * d2;
      float c0 = qy * nz - qz * ny, c1 = qz * nx - qx * nz, c2 = qx * ny - qy * nx;
      float weight = 1.0f / (standard_deviation * ps[PS_CORR_Z * np + i]);
      float se = (weight * weight) * epsilon;
      const float v[6] = {c0, c1, c2, nx, ny, nz};
      float w[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) w[r] = weight * v[r];
#pragma unroll
      for (int r = 0; r < 6; ++r) out[r * pitch] = se * v[r];  // gradient_ -= (w^2 epsilon) * [p x n; n]
      int k = 6;
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int r = c; r < 6; ++r) out[(k++) * pitch] = w[c] * w[r];  // hessian_ (upper, (c, r)) -= w_c w_r
    } else {
#pragma unroll
      for (int k = 0; k < 27; ++k) out[k * pitch] = 0.0f;
    }
  }
}

// ---------------------------------------------------------------------------
// AddLinePixelColorsToTempHistograms (:1025-1155) + ColorHistograms::CalculateHistogram
// (color_histograms.cpp:174-214) for one object.  counts: packed u32 per bin
// (foreground count in the low 16 bits, background in the high 16 bits; each is
// <= n_lines * max_considered_line_length < 65536), in LDS when it fits.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void count_add(__attribute__((address_space(3))) uint32_t* p, uint32_t v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // ds_add_u32
}
__device__ __forceinline__ void count_add(__attribute__((address_space(1))) uint32_t* p, uint32_t v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void count_add(__attribute__((address_space(1))) unsigned long long* p,
                                          unsigned long long v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// SHARED: the modality adds into a ColorHistograms object it shares with others (64-bit words, foreground
// count in the low half, background in the high half); shared_histogram_finish() turns them into histograms
// LIST: for a workgroup whose LDS cannot hold a count word per bin (tracking_step_compact_kernel: four objects per
// CU).  The walk writes the bin of every sample into an LDS list (16 bits per sample = the bin, 0xffff = no sample;
// one row of `list_row` entries per walker, the rows of the foreground walkers first, then those of the background
// walkers: which half an entry sits in says what it counts for -- a flag inside the entry would make a background
// sample of bin 0x7fff, a saturated white pixel at 32 bins, look like the sentinel); then the bins are taken in passes
// of `pass_bins`: the pass's share of the list is counted into the (small) table `counts`, its bins are blended and
// written.  Same sums, same blend arithmetic, any number of passes.  Needs n_bins <= 32 (15-bit bin numbers).
template <bool SHARED = false, bool LIST = false, bool RENDER = true, typename CountPtr>
__device__ __forceinline__ void region_histogram_update(CRegion& m, CCam& cam, CCam* dcam, const Affine& b2c,
                                                        const Affine& b2dc, bool handle_occlusions, bool initialize,
                                                        CountPtr counts, float* misc, int bin_lo = 0,
                                                        int bin_hi = -1, uint16_t* list = nullptr, int list_row = 0,
                                                        int pass_bins = 0, int prev_view = -1) {
  // [bin_lo, bin_hi): the bins this workgroup counts and blends (all of them unless it shares its object with
  // others: then every workgroup walks all lines, keeps the samples that fall into its bins, and no table
  // has to be merged); counts[0] is bin_lo's word
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n_bins3 = m.n_bins * m.n_bins * m.n_bins;
  if (bin_hi < 0) bin_hi = n_bins3;
  const uint32_t n_own_bins = (uint32_t)(bin_hi - bin_lo);
  if (!SHARED && !LIST)
    for (int i = tid; i < (int)n_own_bins; i += nt) counts[i] = 0;
  unsigned sf = 0, sb = 0;  // this thread's foreground / background samples (all bins)
  PHASE_T0();
  int view = -1;
  if (prev_view >= 0 && m.view_neighbors != nullptr) {
    float o0, o1, o2;
    if (view_direction(b2c, o0, o1, o2)) view = closest_view_local((G<v4f>)m.view_neighbors, prev_view, o0, o1, o2);
    if (view >= 0) __syncthreads();  // (the counts are zeroed behind this)
  }
  if (view < 0) view = closest_view((G<v4f>)m.orientations4, m.n_views, b2c, misc);  // syncs: counts are zeroed after this
  const int n_lines = number_of_lines(m.n_lines_max, m.use_adaptive_coverage, m.reference_contour_length,
                                      as_global(m.extents), view, m.max_extent, m.n_points);
  const int bitshift = m.bitshift, n_bins = m.n_bins, n_bins2 = n_bins * n_bins;
  const int w1 = cam.width - 1, h1 = cam.height - 1;
  const bool visible_depth = RENDER && m.model_occlusions && renderer_body_visible(m.depth_renderer, m.depth_renderer_slot);
  const bool visible_silhouette =
      RENDER && m.use_region_checking && renderer_body_visible(m.silhouette_renderer, m.silhouette_renderer_slot);
  // IsLineUnoccludedMeasured at the final pose :1084-1087: the windows of all lines first -- their limits by one thread
  // per line, the samples by 16 lanes per line (three words per line in the misc block, which holds 256 lines; the
  // verdict replaces the first word; longer models test inside the walk)
  int* line_occluded = reinterpret_cast<int*>(misc) + 256;  // [3 * line]
  const bool measured = handle_occlusions && m.measure_occlusions;
  const bool windows_first = measured && 3 * n_lines <= M3T_MISC_FLOATS - 256;
  PHASE_MARK(27);  // tail: view search
  if (windows_first) {
    uint32_t* windows = reinterpret_cast<uint32_t*>(line_occluded);
    for (int line = tid; line < n_lines; line += nt) {
      G<v4f> p8 = (G<v4f>)m.points8 + ((uint32_t)view * m.n_points + line) * 2;
      const v4f pa = p8[0];
      float dx, dy, dz;
      apply_pose(b2dc, pa.x, pa.y, pa.z, dx, dy, dz);
      const float du = dx * dcam->fu / dz + dcam->ppu;
      const float dv = dy * dcam->fv / dz + dcam->ppv;
      const float diameter = 2.0f * m.measured_occlusion_radius * (dcam->fu / dz);
      const float offset = as_global(m.points)[((size_t)view * m.n_points + line) * M3T_REGION_POINT_FLOATS + 8 +
                                               m.measured_depth_offset_id];
      const OcclusionWindow ow =
          occlusion_window_pack(*dcam, du, dv, diameter, dz, offset, m.measured_occlusion_threshold);
      windows[3 * line] = ow.base;
      windows[3 * line + 1] = ow.packed;
      windows[3 * line + 2] = ow.stride;
    }
    __syncthreads();
    const int gl = tid & 15, groups = nt >> 4;
    for (int line0 = 0; line0 < n_lines; line0 += 8 * groups) {  // eight lines per row and trip: 24 samples in flight
      uint32_t packed[8];
      unsigned short d0[8], d1[8], d2[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int line = line0 + r * groups + (tid >> 4);
        packed[r] = 0u;
        d0[r] = d1[r] = d2[r] = 0;
        if (line < n_lines) {
          packed[r] = windows[3 * line + 1];
          occlusion_window_lane_load(*dcam, windows[3 * line], packed[r], windows[3 * line + 2], gl, d0[r], d1[r], d2[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int line = line0 + r * groups + (tid >> 4);
        int occluded = occlusion_window_lane_test(packed[r], d0[r], d1[r], d2[r]) ? 1 : 0;
        occluded |= dpp_self_i<0x111, 0xf>(occluded);
        occluded |= dpp_self_i<0x112, 0xf>(occluded);
        occluded |= dpp_self_i<0x114, 0xf>(occluded);
        occluded |= dpp_self_i<0x118, 0xf>(occluded);
        if (gl == 15 && line < n_lines) line_occluded[3 * line] = occluded;
      }
    }
    __syncthreads();
  }
  PHASE_MARK(28);  // tail: occlusion windows
  if constexpr (LIST) {  // 0xffff = no sample (two entries per word; rows of the walkers that end early stay empty)
    uint32_t* words = reinterpret_cast<uint32_t*>(list);
    for (int i = tid; i < n_lines * list_row; i += nt) words[i] = 0xffffffffu;  // 2 walkers x list_row / 2 words per line
    __syncthreads();
  }
  // two lanes per line: even lane = foreground walk (inwards), odd lane = background walk
  for (int item = tid; item < 2 * n_lines; item += nt) {
    const int line = item >> 1;
    const bool background = item & 1;
    const float* p = m.points + ((size_t)view * m.n_points + line) * M3T_REGION_POINT_FLOATS;
    G<v4f> p8 = (G<v4f>)m.points8 + ((uint32_t)view * m.n_points + line) * 2;
    const v4f pa = p8[0], pb4 = p8[1];
    float cx = pa.x, cy = pa.y, cz = pa.z;
    float X, Y, Z;
    apply_pose(b2c, cx, cy, cz, X, Y, Z);
    if (Z <= 0.0f) continue;
    float center_u = X * cam.fu / Z + cam.ppu;
    float center_v = Y * cam.fv / Z + cam.ppv;
    int icu = f2i(center_u + 0.5f), icv = f2i(center_v + 0.5f);
    if ((float)icu < 0.0f || icu > w1 || icv < 0 || icv > h1) continue;
    if (handle_occlusions && m.model_occlusions && visible_depth) {  // :1079-1083
      float diameter = 2.0f * m.modeled_occlusion_radius * ((cam.fu / Z) * m.depth_renderer->state[RS_SCALE]);
      unsigned short min_value = modeled_window_min(*m.depth_renderer, center_u, center_v, diameter);
      if (!(renderer_depth(*m.depth_renderer, min_value) >
            Z - p[8 + m.modeled_depth_offset_id] - m.modeled_occlusion_threshold))
        continue;
    }
    if (windows_first) {
      if (line_occluded[3 * line]) continue;
    } else if (measured) {
      float dx, dy, dz;
      apply_pose(b2dc, cx, cy, cz, dx, dy, dz);
      float du = dx * dcam->fu / dz + dcam->ppu;
      float dv = dy * dcam->fv / dz + dcam->ppv;
      float diameter = 2.0f * m.measured_occlusion_radius * (dcam->fu / dz);
      if (!occlusion_window_clear(*dcam, du, dv, diameter, dz, p[8 + m.measured_depth_offset_id],
                                  m.measured_occlusion_threshold))
        continue;
    }
    float length_f = m.max_considered_line_length, length_b = m.max_considered_line_length;
    if (m.use_region_checking && visible_silhouette) {  // :1094-1099
      float ru = (b2c.l[0] * pa.w + b2c.l[3] * pb4.x) + b2c.l[6] * pb4.y;
      float rv = (b2c.l[1] * pa.w + b2c.l[4] * pb4.x) + b2c.l[7] * pb4.y;
      float rn = sqrtf(ru * ru + rv * rv);
      if (rn > 0.0f) { ru = ru / rn; rv = rv / rn; }
      dynamic_region_distance(*m.silhouette_renderer, m.region_id, m.max_considered_line_length,
                              m.unconsidered_line_length, center_u, center_v, ru, rv, &length_f, &length_b);
    }
    float l_f = pb4.z * cam.fu / Z;
    float l_b = pb4.w * cam.fu / Z;
    length_f = fminf(length_f, l_f - 2.0f * m.unconsidered_line_length);
    length_b = fminf(length_b, l_b - 2.0f * m.unconsidered_line_length);
    float nu = (b2c.l[0] * pa.w + b2c.l[3] * pb4.x) + b2c.l[6] * pb4.y;
    float nv = (b2c.l[1] * pa.w + b2c.l[4] * pb4.x) + b2c.l[7] * pb4.y;
    float nn = sqrtf(nu * nu + nv * nv);
    if (nn > 0.0f) { nu = nu / nn; nv = nv / nn; }
    float u_step, v_step;
    int projected_length_f, projected_length_b;
    float abs_nu = fabsf(nu), abs_nv = fabsf(nv);
    if (abs_nu > abs_nv) {
      u_step = nu < 0.0f ? -1.0f : (nu > 0.0f ? 1.0f : 0.0f);
      v_step = nv / abs_nu;
      projected_length_f = f2i(length_f * abs_nu + 0.5f);
      projected_length_b = f2i(length_b * abs_nu + 0.5f);
    } else {
      u_step = nu / abs_nv;
      v_step = nv < 0.0f ? -1.0f : (nv > 0.0f ? 1.0f : 0.0f);
      projected_length_f = f2i(length_f * abs_nv + 0.5f);
      projected_length_b = f2i(length_b * abs_nv + 0.5f);
    }
    // one walk per lane; first count the steps that stay on the image (the reference breaks at the
    // first step off the image), then replay the same float chain with independent loads + LDS atomics
    const float sgn = background ? 1.0f : -1.0f;
    const int projected_length = background ? projected_length_b : projected_length_f;
    const float u0 = background ? center_u + nu * m.unconsidered_line_length + 0.5f
                                : center_u - nu * m.unconsidered_line_length + 0.5f;
    const float v0 = background ? center_v + nv * m.unconsidered_line_length + 0.5f
                                : center_v - nv * m.unconsidered_line_length + 0.5f;
    const float du = sgn * u_step, dv = sgn * v_step;  // u -= u_step == u += (-u_step), exactly
    const auto inc = [&]() {
      if constexpr (SHARED) return background ? (1ull << 32) : 1ull;
      else return background ? 65536u : 1u;
    }();
    // One pass, eight steps at a time: the float chain u += du only feeds the addresses, so the loads of a batch are
    // independent; `alive` reproduces the reference's break at the first step off the image (nothing after it
    // counts), the count-table atomics follow once a batch's loads are issued.
    int n_valid = 0;
    float u = u0, v = v0;
    bool alive = true;
    G<uint8_t> image = as_global(cam.image);
    for (int k0 = 0; k0 < projected_length && alive; k0 += 8) {
      uint32_t px[8];
      uint32_t taken = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int iu = f2i(u), iv = f2i(v);
        alive = alive && k0 + j < projected_length && !(iu < 0 || iu > w1 || iv < 0 || iv > h1);
        px[j] = 0;
        if (alive) {
          px[j] = reinterpret_cast<G<PackedU32>>(image + (__umul24((uint32_t)iv, cam.pitch) + (uint32_t)iu * 3u))->v;
          taken |= 1u << j;
        }
        u += du;
        v += dv;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t bin = ((px[j] & 0xffu) >> bitshift) * n_bins2 + (((px[j] >> 8) & 0xffu) >> bitshift) * n_bins +
                             (((px[j] >> 16) & 0xffu) >> bitshift) - (uint32_t)bin_lo;
        if constexpr (LIST) {
          if (((taken >> j) & 1u) && k0 + j < list_row)
            list[(background ? n_lines * list_row : 0) + line * list_row + k0 + j] = (uint16_t)bin;
        } else {
          if (((taken >> j) & 1u) && (SHARED || bin < n_own_bins)) count_add(&counts[bin], inc);
        }
      }
      n_valid += __builtin_popcount(taken);
    }
    if (background) sb += (unsigned)n_valid;
    else sf += (unsigned)n_valid;
  }
  if constexpr (!SHARED) {
  __syncthreads();
  PHASE_MARK(29);  // tail: pixel walk
  // sums of the count tables = numbers of samples (exact: integers)
  sf = (unsigned)wave_sum_i((int)sf);
  sb = (unsigned)wave_sum_i((int)sb);
  unsigned* umisc = reinterpret_cast<unsigned*>(misc);
  if (tid % kWave == 0) { umisc[2 * (tid / kWave)] = sf; umisc[2 * (tid / kWave) + 1] = sb; }
  __syncthreads();
  sf = 0; sb = 0;
  for (int w = 0; w < nt / kWave; ++w) { sf += umisc[2 * w]; sb += umisc[2 * w + 1]; }
  const float sum_f = (float)sf, sum_b = (float)sb;
  const float lr_f = initialize ? 1.0f : m.learning_rate_f, lr_b = initialize ? 1.0f : m.learning_rate_b;
  const float comp_f = 1.0f - lr_f, comp_b = 1.0f - lr_b;
  const float scale_f = lr_f / sum_f, scale_b = lr_b / sum_b;
  const float uniform_value = 1.0f / (float)n_bins3;
  // blend (color_histograms.cpp:174-214) + per-bin normalisation, 4 bins per lane: 16-byte global
  // loads / stores through global-address-space pointers (n_bins^3 is a multiple of 8)
  GW<v4f> hist_f4 = (GW<v4f>)m.histogram_f;
  GW<v4f> hist_b4 = (GW<v4f>)m.histogram_b;
  GW<v4f> norm4 = (GW<v4f>)m.histogram_norm;
  // (the old histograms of four trips are requested before the first is blended: one memory round trip for a
  // workgroup's 8192 bins at 512 threads instead of four)
  const int n_passes = LIST ? (n_bins3 + pass_bins - 1) / pass_bins : 1;
  for (int pass = 0; pass < n_passes; ++pass) {
  if constexpr (LIST) {
    bin_lo = pass * pass_bins;
    bin_hi = min(bin_lo + pass_bins, n_bins3);
    if (pass > 0) __syncthreads();  // the previous pass's blend has read its counts
    for (int i = tid; i < pass_bins; i += nt) counts[i] = 0;
    __syncthreads();
    const uint32_t* words = reinterpret_cast<const uint32_t*>(list);
    const uint32_t first_background = (uint32_t)(n_lines * list_row);  // entry index
    for (int i = tid; i < n_lines * list_row; i += nt) {
      const uint32_t w = words[i];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const uint32_t v = half ? w >> 16 : w & 0xffffu;
        const uint32_t bin = v - (uint32_t)bin_lo;
        if (v != 0xffffu && bin < (uint32_t)(bin_hi - bin_lo))
          count_add(&counts[bin], 2u * (uint32_t)i + (uint32_t)half >= first_background ? 65536u : 1u);
      }
    }
    __syncthreads();
  }
  const int i4_end = bin_hi / 4;
  // One byte per group of four bins says whether the group holds a non-zero value.  A group that is empty and got no
  // sample stays as it is -- 0 * (1 - lr) + 0 * scale = 0, pair (0.5, 0.5) -- and is neither read nor written: most of
  // a 32^3 table, frame after frame.  (Not while initialising, and not in the no-sample cases, which may set the
  // uniform value.)  The occupancy bytes and the counts of four trips are looked at first, then the old histograms
  // of the groups that need them are requested together.
  GW<uint8_t> occupancy = as_global_w(m.occupancy);
  const bool every_group = initialize || sf == 0 || sb == 0;
  for (int i40 = bin_lo / 4 + tid; i40 < i4_end; i40 += 4 * nt) {
  uint32_t occupied[4];
#pragma unroll
  for (int trip = 0; trip < 4; ++trip) {
    const int i4 = i40 + trip * nt;
    occupied[trip] = 0u;
    if (i4 < i4_end) occupied[trip] = every_group ? 1u : (uint32_t)occupancy[i4];
  }
  uint32_t c4s[4][4];
  bool need[4];
#pragma unroll
  for (int trip = 0; trip < 4; ++trip) {
    const int i4 = i40 + trip * nt;
    need[trip] = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) c4s[trip][k] = 0u;
    if (i4 < i4_end) {
      const int c0 = 4 * i4 - bin_lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) c4s[trip][k] = counts[c0 + k];
      need[trip] = occupied[trip] != 0u || (c4s[trip][0] | c4s[trip][1] | c4s[trip][2] | c4s[trip][3]) != 0u;
    }
  }
  v4f old_f[4], old_b[4];
#pragma unroll
  for (int trip = 0; trip < 4; ++trip) {
    const int i4 = i40 + trip * nt;
    if (need[trip]) { old_f[trip] = hist_f4[i4]; old_b[trip] = hist_b4[i4]; }
  }
#pragma unroll
  for (int trip = 0; trip < 4; ++trip) {
    const int i4 = i40 + trip * nt;
    if (need[trip]) {  // (no `continue`: it would send the unrolled loop's arrays to scratch memory)
    v4f hf4 = old_f[trip], hb4 = old_b[trip];
    v4f n0, n1;
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float hf = hf4[k], hb = hb4[k];
      uint32_t c = c4s[trip][k];
      if (sf == 0) {
        if (lr_f == 1.0f) hf = uniform_value;
      } else if (comp_f == 0.0f) {
        hf = (float)(c & 0xffffu) * scale_f;
      } else {
        hf *= comp_f;
        hf += (float)(c & 0xffffu) * scale_f;
      }
      if (sb == 0) {
        if (lr_b == 1.0f) hb = uniform_value;
      } else if (comp_b == 0.0f) {
        hb = (float)(c >> 16) * scale_b;
      } else {
        hb *= comp_b;
        hb += (float)(c >> 16) * scale_b;
      }
      hf4[k] = hf;
      hb4[k] = hb;
      any = any || hf != 0.0f || hb != 0.0f;
      // MultiplyPixelColorProbability :1585-1593 hoisted from per pixel to per bin
      float nx = 0.5f, ny = 0.5f;
      if (hf || hb) {
        float sum = hf;
        sum += hb;
        nx = hf / sum;
        ny = hb / sum;
      }
      if (k < 2) { n0[2 * k] = nx; n0[2 * k + 1] = ny; }
      else { n1[2 * (k - 2)] = nx; n1[2 * (k - 2) + 1] = ny; }
    }
    hist_f4[i4] = hf4;
    hist_b4[i4] = hb4;
    norm4[2 * i4] = n0;
    norm4[2 * i4 + 1] = n1;
    occupancy[i4] = any ? 1 : 0;
    }
  }
  }
  }  // passes
  }  // !SHARED
}

// ColorHistograms::InitializeHistograms / UpdateHistograms (color_histograms.cpp:70-92,174-214) of a shared
// object, one workgroup: sums, blend, per-bin normalisation, then ClearMemory
__device__ void shared_histogram_finish(const SharedHistogramsDev& h, bool initialize, float* misc) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n_bins3 = h.n_bins * h.n_bins * h.n_bins;
  unsigned sf = 0, sb = 0;
  for (int i = tid; i < n_bins3; i += nt) {
    unsigned long long c = h.counts[i];
    sf += (unsigned)(c & 0xffffffffull);
    sb += (unsigned)(c >> 32);
  }
  sf = (unsigned)wave_sum_i((int)sf);
  sb = (unsigned)wave_sum_i((int)sb);
  unsigned* umisc = reinterpret_cast<unsigned*>(misc);
  if (tid % kWave == 0) { umisc[2 * (tid / kWave)] = sf; umisc[2 * (tid / kWave) + 1] = sb; }
  __syncthreads();
  sf = 0; sb = 0;
  for (int w = 0; w < nt / kWave; ++w) { sf += umisc[2 * w]; sb += umisc[2 * w + 1]; }
  const float sum_f = (float)sf, sum_b = (float)sb;
  const float lr_f = initialize ? 1.0f : h.learning_rate_f, lr_b = initialize ? 1.0f : h.learning_rate_b;
  const float comp_f = 1.0f - lr_f, comp_b = 1.0f - lr_b;
  const float scale_f = lr_f / sum_f, scale_b = lr_b / sum_b;
  const float uniform_value = 1.0f / (float)n_bins3;
  for (int i = tid; i < n_bins3; i += nt) {
    const unsigned long long c = h.counts[i];
    const float cf = (float)(unsigned)(c & 0xffffffffull), cb = (float)(unsigned)(c >> 32);
    float hf = h.histogram_f[i], hb = h.histogram_b[i];
    if (sf == 0) {
      if (lr_f == 1.0f) hf = uniform_value;
    } else if (comp_f == 0.0f) {
      hf = cf * scale_f;
    } else {
      hf *= comp_f;
      hf += cf * scale_f;
    }
    if (sb == 0) {
      if (lr_b == 1.0f) hb = uniform_value;
    } else if (comp_b == 0.0f) {
      hb = cb * scale_b;
    } else {
      hb *= comp_b;
      hb += cb * scale_b;
    }
    h.histogram_f[i] = hf;
    h.histogram_b[i] = hb;
    float nx = 0.5f, ny = 0.5f;
    if (hf || hb) {
      float sum = hf;
      sum += hb;
      nx = hf / sum;
      ny = hb / sum;
    }
    h.histogram_norm[i] = make_float2(nx, ny);
    h.counts[i] = 0ull;
  }
}

// the table of m3t_log.h into the misc block (the kernels that form region products call this before a barrier)
__device__ const uint64_t g_log_table_bits[M3T_LOG_TABLE_DOUBLES] = M3T_LOG_TABLE_INIT;
__device__ __forceinline__ void stage_log_table(float* misc) {
  if (threadIdx.x < M3T_LOG_TABLE_DOUBLES)
    reinterpret_cast<uint64_t*>(misc + kMiscLogTable)[threadIdx.x] = g_log_table_bits[threadIdx.x];
}
__device__ __forceinline__ void stage_histogram(CRegion& m, float* lds_hist) {
  const int n2 = m.n_bins * m.n_bins * m.n_bins * 2;
  const float* src = reinterpret_cast<const float*>(m.histogram_norm);
  for (int i = threadIdx.x; i < n2; i += blockDim.x) lds_hist[i] = src[i];
}

}  // namespace

// ===========================================================================
// kernels
// ===========================================================================
extern "C" {

// One block per region modality.  initialize != 0: StartModality (:375-388), else CalculateResults (:572-583).
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
region_histogram_kernel(const RegionModDev* mods, const CameraDev* cams, const float* body_poses, int iteration,
                        int initialize, int counts_in_lds, const RigidOptDev* guarded_opts, const int* opt_of_region,
                        int guard_poses) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // behind a guarded step (ROI ingest): a body whose step was not committed keeps its histograms as well
  if (guarded_opts && opt_of_region) {
    const int oi = opt_of_region[blockIdx.x];
    if (oi >= 0 && guarded_opts[oi].search_poses &&
        reinterpret_cast<const RoiGuardDev*>(guarded_opts[oi].search_poses + 16 * guard_poses)->flag != 0)
      return;
  }
  CRegion& m = *(CRegion*)(mods + blockIdx.x);
  CCam& cam = *(CCam*)(cams + m.camera);
  CCam* dcam = m.measure_occlusions ? (CCam*)(cams + m.depth_camera) : nullptr;
  const Affine b2w = load_pose(body_poses + 16 * m.body);
  const Affine b2c = mul_pose(load_pose(cam.world2camera), b2w);
  Affine b2dc = b2c;
  if (dcam) b2dc = mul_pose(load_pose(dcam->world2camera), b2w);
  // first_iteration is maintained by the host table (StartModality :378)
  bool handle_occlusions = initialize ? (m.n_unoccluded_iterations == 0)
                                      : ((iteration - m.first_iteration) >= m.n_unoccluded_iterations);
  float* misc = lds;
  if (m.shared_counts) {  // UseSharedColorHistograms: only add this modality's samples
    region_histogram_update<true>(m, cam, dcam, b2c, b2dc, handle_occlusions, initialize != 0,
                                  (__attribute__((address_space(1))) unsigned long long*)m.shared_counts, misc);
    return;
  }
  if (counts_in_lds) {  // ds_add_u32 on the LDS count table
    region_histogram_update(m, cam, dcam, b2c, b2dc, handle_occlusions, initialize != 0,
                            (__attribute__((address_space(3))) uint32_t*)(lds + M3T_MISC_FLOATS), misc);
  } else {              // 64 bins: 1 MB table in HBM, global atomics
    region_histogram_update(m, cam, dcam, b2c, b2dc, handle_occlusions, initialize != 0,
                            (__attribute__((address_space(1))) uint32_t*)m.count_scratch, misc);
  }
}

// one workgroup per shared ColorHistograms object, after every modality has added its samples
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
shared_histogram_finish_kernel(const SharedHistogramsDev* shared, int initialize) {
  __shared__ float misc[64];
  shared_histogram_finish(shared[blockIdx.x], initialize != 0, misc);
}

extern "C++" {
template <bool HIST_LDS>
__device__ __forceinline__ void region_correspondence_body(const RegionModDev* mods, const CameraDev* cams, const float* body_poses,
                             TrackLdsLayout layout, int iteration, int corr_iteration) {
  extern __shared__ __attribute__((aligned(16))) float lds_t[];
  CRegion& m = *(CRegion*)(mods + blockIdx.x);
  CCam& cam = *(CCam*)(cams + m.camera);
  CCam* dcam = m.measure_occlusions ? (CCam*)(cams + m.depth_camera) : nullptr;
  Lds s = carve(lds_t, layout);
  if (HIST_LDS) {
    stage_histogram(m, lds_t + layout.off_hist);
    __syncthreads();
  }
  const Affine b2w = load_pose(body_poses + 16 * m.body);
  const Affine b2c = mul_pose(load_pose(cam.world2camera), b2w);
  Affine b2dc = b2c;
  if (dcam) b2dc = mul_pose(load_pose(dcam->world2camera), b2w);
  region_correspondences<HIST_LDS>(m, cam, dcam, b2c, b2dc, iteration, corr_iteration, s);
  region_moments(m, s);
  __syncthreads();
  // LDS -> global line state (compact stride n_lines_max)
  for (int i = threadIdx.x; i < LS_FIELDS * m.n_lines_max; i += blockDim.x) {
    int f = i / m.n_lines_max, l = i - f * m.n_lines_max;
    m.line_state[i] = s.state[f * s.nl + l];
  }
}

}  // extern "C++"
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
region_correspondence_kernel(const RegionModDev* mods, const CameraDev* cams, const float* body_poses,
                             TrackLdsLayout layout, int iteration, int corr_iteration) {
  region_correspondence_body<false>(mods, cams, body_poses, layout, iteration, corr_iteration);
}
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
region_correspondence_lds_kernel(const RegionModDev* mods, const CameraDev* cams, const float* body_poses,
                                 TrackLdsLayout layout, int iteration, int corr_iteration) {
  region_correspondence_body<true>(mods, cams, body_poses, layout, iteration, corr_iteration);
}

__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
region_gradient_hessian_kernel(const RegionModDev* mods, const CameraDev* cams, const float* body_poses,
                               TrackLdsLayout layout, int corr_iteration, int opt_iteration) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  CRegion& m = *(CRegion*)(mods + blockIdx.x);
  CCam& cam = *(CCam*)(cams + m.camera);
  Lds s = carve(lds, layout);
  for (int i = threadIdx.x; i < LS_FIELDS * m.n_lines_max; i += blockDim.x) {
    int f = i / m.n_lines_max, l = i - f * m.n_lines_max;
    s.state[f * s.nl + l] = m.line_state[i];
  }
  for (int l = m.n_lines_max + threadIdx.x; l < s.nl; l += blockDim.x) s.state[LS_VALID * s.nl + l] = i2f_bits(0);
  stage_log_table(s.misc);
  __syncthreads();
  const Affine b2c = mul_pose(load_pose(cam.world2camera), load_pose(body_poses + 16 * m.body));
  float* rows = lds + layout.off_rows_r;
  region_products(m, cam, b2c, corr_iteration, opt_iteration, s, rows, layout.pitch_r);
  __syncthreads();
  if (threadIdx.x < 42) {  // the first wave: the sums in the layout of gradient() / hessian()
    float sum, unused;
    chain_sums(rows, layout.pitch_r, chain_slots(s.nl), nullptr, 0, 0, gh_lane_row(threadIdx.x), sum, unused);
    m.gradient_hessian[threadIdx.x] = sum;
  }
}

__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
depth_correspondence_kernel(const DepthModDev* mods, const CameraDev* cams, const float* body_poses, int np,
                            int iteration, int corr_iteration) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  CDepth& m = *(CDepth*)(mods + blockIdx.x);
  CCam& cam = *(CCam*)(cams + m.camera);
  float* misc = lds;
  float* ps = lds + M3T_MISC_FLOATS;
  const Affine b2c = mul_pose(load_pose(cam.world2camera), load_pose(body_poses + 16 * m.body));
  depth_correspondences(m, cam, b2c, iteration, corr_iteration, ps, np, misc);
  for (int i = threadIdx.x; i < PS_FIELDS * m.n_points_max; i += blockDim.x) {
    int f = i / m.n_points_max, l = i - f * m.n_points_max;
    m.point_state[i] = ps[f * np + l];
  }
}

// LDS: misc | point state [PS_FIELDS][np] (rounded up to 16 bytes) | product rows [27][chain_pitch(np)]
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
depth_gradient_hessian_kernel(const DepthModDev* mods, const CameraDev* cams, const float* body_poses, int np,
                              int corr_iteration) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  CDepth& m = *(CDepth*)(mods + blockIdx.x);
  CCam& cam = *(CCam*)(cams + m.camera);
  float* ps = lds + M3T_MISC_FLOATS;
  float* rows = ps + (PS_FIELDS * np + 3) / 4 * 4;
  for (int i = threadIdx.x; i < PS_FIELDS * m.n_points_max; i += blockDim.x) {
    int f = i / m.n_points_max, l = i - f * m.n_points_max;
    ps[f * np + l] = m.point_state[i];
  }
  for (int l = m.n_points_max + threadIdx.x; l < np; l += blockDim.x) ps[PS_VALID * np + l] = i2f_bits(0);
  __syncthreads();
  const Affine b2c = mul_pose(load_pose(cam.world2camera), load_pose(body_poses + 16 * m.body));
  const int pitch = chain_pitch(np);
  depth_products(m, b2c, corr_iteration, ps, np, rows, pitch);
  __syncthreads();
  if (threadIdx.x < 42) {
    float sum, unused;
    chain_sums(rows, pitch, chain_slots(np), nullptr, 0, 0, gh_lane_row(threadIdx.x), sum, unused);
    m.gradient_hessian[threadIdx.x] = sum;
  }
}

// One wave per rigid optimizer (Link::CalculateGradientAndHessian link.cpp:184-193 + solve + update).
__global__ void __launch_bounds__(64)
rigid_optimize_kernel(const RigidOptDev* opts, int n_opts, const RegionModDev* rmods, const DepthModDev* dmods,
                      float* body_poses) {
  __shared__ float pose[16];
  __shared__ float scratch[128];
  COpt& o = *(COpt*)(opts + blockIdx.x);
  const int lane = threadIdx.x;
  float gh = 0.0f;
  if (lane < 42) {
    if (o.region_modality >= 0) gh += rmods[o.region_modality].gradient_hessian[lane];
    if (o.depth_modality >= 0) gh += dmods[o.depth_modality].gradient_hessian[lane];
  }
  if (lane < 16) pose[lane] = body_poses[16 * o.body + lane];
  __builtin_amdgcn_wave_barrier();
  rigid_solve_wave(gh, o.tikhonov_rotation, o.tikhonov_translation, (LdsW)pose, (LdsW)scratch);
  __builtin_amdgcn_wave_barrier();
  if (lane < 16) body_poses[16 * o.body + lane] = pose[lane];
}

// ---------------------------------------------------------------------------
// Fused Tracker::ExecuteTrackingStep (tracker.cpp:344-364) minus CalculateResults:
// one block per rigid optimizer; n_corr x (correspondences + n_update x (g/H + solve)).
// Everything between the image/model gathers and the final pose stays in LDS.
// ---------------------------------------------------------------------------
struct SplitParams {               // tracking_step_split_kernel: n_parts workgroups (on as many CUs) per object
  unsigned long long* granules;    // [objects][2 slots][n_parts][32 fields][1 << lshift]
  unsigned* object_abort;          // [objects]
  unsigned* host_abort;            // mapped host word
  unsigned seq;                    // launch sequence number (restarts when the granule tags are cleared)
  unsigned abort_id;               // what an aborting workgroup writes to *host_abort: unique per launch, never reset
  int n_objects;                   // objects of the launch (the grid is padded to a multiple of 8 x n_parts blocks)
  int n_parts, lshift;             // n_parts << lshift == 256
  int per_part_lines, per_part_points;
};

extern "C++" {
// PAIR (round 6; the _pair_ kernels, for batches whose bodies carry a RegionModality AND a DepthModality): per Newton step
// the lines' products by the first half of the workgroup BESIDE the points' products by the second (one round instead of
// two: 200 lines and 200 points never fill 512 threads one after the other), and the two modalities' sums on two waves --
// each in the reference's order, added like Link::CalculateGradientAndHessian adds them.  A template parameter, not a
// run-time choice: the Region-only step keeps its machine code (with the choice inside, tracking_step_split_kernel went
// from 234 to 241 VGPRs and the 64-object headline from 0.149 to 0.151 ms).
template <bool HIST_LDS, bool SPLIT = false, bool RENDER = !SPLIT, bool GUARD = false, bool PAIR = false>
__device__ __forceinline__ void tracking_step_body(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, const SplitParams* split = nullptr, int first_corr_iteration = 0,
                     const RoiGuardArgs* guard = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds_t[];
  // SPLIT: n_parts workgroups share one object.  Each runs the whole step, but walks the pixels (and scans the depth
  // windows) of its own part of the lines (points) only; the line results are exchanged once per correspondence
  // iteration (split_exchange_state), after which every workgroup holds the same state, forms the same sums and
  // solves redundantly.  With a multiple of 8 objects the workgroups of one object sit on one XCD (block b runs on
  // XCD b % 8): a speed bonus, not a correctness condition.
  int object = blockIdx.x, part = 0, n_parts = 1;
  if constexpr (SPLIT) {
    // All workgroups of an object on one XCD (block b runs on XCD b % 8: observed, a speed matter only), so that the
    // object's image regions, model rows and histograms are fetched into ONE L2 and not into eight: XCD x hosts the
    // objects x, x + 8, ...; its blocks b = x, x + 8, ... take them part after part.  The grid is padded to eight
    // times the blocks of the fullest XCD (split->n_objects is the real count); surplus blocks leave at once.
    n_parts = split->n_parts;
    const int b = blockIdx.x, j = b >> 3;
    object = (j / n_parts) * 8 + (b & 7);
    part = j % n_parts;
    // (the test for surplus blocks only where the grid is padded: with it unconditional the compiler's code for the
    // 64-object launch came out 3.5 % slower, measured)
    if ((int)gridDim.x / n_parts != split->n_objects && object >= split->n_objects) return;
  }
  COpt& o = *(COpt*)(opts + object);
  // GUARD (ROI ingest, roi_guard_outside above): the last wave checks every pose the step reads its frames at
  RoiGuardDev* guard_hdr = nullptr;
  int guard_miss = 0;
  if constexpr (GUARD) {
    guard_hdr = reinterpret_cast<RoiGuardDev*>(o.search_poses + 16 * guard->n_poses);
    if (guard->mode == 2 && guard_hdr->flag == 0) return;  // the repeat: flagged objects only (all parts read the same flag)
  }
  const bool guard_wave = GUARD && threadIdx.x >= blockDim.x - kWave;
  CRegion* rm = o.region_modality >= 0 ? (CRegion*)(rmods + o.region_modality) : nullptr;
  CDepth* dm = o.depth_modality >= 0 ? (CDepth*)(dmods + o.depth_modality) : nullptr;
  Lds s = carve(lds_t, layout);
  float* ps = lds_t + off_points;
  float* rows_r = lds_t + layout.off_rows_r;
  float* rows_d = lds_t + layout.off_rows_d;
  float* pose = s.misc + kMiscPose;           // 16 floats
  float* gh_region = s.misc + kMiscGhRegion;  // 42
  float* gh_depth = s.misc + kMiscGhDepth;    // 42
  if (threadIdx.x < 16) pose[threadIdx.x] = body_poses[16 * o.body + threadIdx.x];
  if (rm) stage_log_table(s.misc);
  if (HIST_LDS && rm) stage_histogram(*rm, lds_t + layout.off_hist);
  __syncthreads();
  CCam* cam = rm ? (CCam*)(cams + rm->camera) : nullptr;
  CCam* rdcam = (rm && rm->measure_occlusions) ? (CCam*)(cams + rm->depth_camera) : nullptr;
  CCam* dcam = dm ? (CCam*)(cams + dm->camera) : nullptr;
  int line_lo = 0, line_hi = 1 << 30, pt_lo = 0, pt_hi = 1 << 30;
  SplitExchange exchange{};
  if constexpr (SPLIT) {
    line_lo = part * split->per_part_lines;
    line_hi = line_lo + split->per_part_lines;
    pt_lo = part * split->per_part_points;
    pt_hi = pt_lo + split->per_part_points;
    exchange.granules = (__attribute__((address_space(1))) unsigned long long*)split->granules +
                        ((size_t)object * 2 * n_parts << (kExchangeFieldBits + split->lshift));
    exchange.object_abort = (__attribute__((address_space(1))) unsigned*)split->object_abort + object;
    exchange.host_abort = split->host_abort;
    exchange.seq = split->seq;
    exchange.abort_id = split->abort_id;
    exchange.part = part;
    exchange.n_parts = n_parts;
    exchange.lshift = split->lshift;
    exchange.per_part_lines = split->per_part_lines;
    exchange.per_part_points = split->per_part_points;
    exchange.n_region_fields = rm ? rm->distribution_length : 0;
    exchange.first_region_row = LS_DIST0;
    exchange.n_depth_fields = write_state ? PS_VALID + 1 - PS_CENTER_U : PS_VALID + 1 - PS_CORR_X;
    exchange.first_depth_row = write_state ? PS_CENTER_U : PS_CORR_X;
  }
  // (first_corr_iteration > 0: a host that refreshes renderer-fed inputs between the correspondence searches
  // launches the loop one search at a time)
  // ROI ingest: the poses this step reads its frames at (m3t_ingest.hip checks them against what was uploaded)
  GW<float> search_poses = as_global_w(o.search_poses);
  const bool record_poses = o.search_poses != nullptr && part == 0 && threadIdx.x < 16;
  if (record_poses && first_corr_iteration == 0) search_poses[threadIdx.x] = pose[threadIdx.x];
  // the view of the modality's previous search (of the previous frame for the first one): closest_view_local
  int region_view = rm ? *as_global(rm->last_view) : -1;
  for (int c = first_corr_iteration; c < first_corr_iteration + n_corr_iterations; ++c) {
    if (record_poses) search_poses[(c + 1) * 16 + threadIdx.x] = pose[threadIdx.x];
    if constexpr (GUARD) {
      if (guard_wave)  // (the pose in LDS stands until the solve of this search's first Newton step, barriers away)
        guard_miss |= roi_guard_outside(guard->items, guard->rects, guard->n_cams, guard->n_rect_slots, guard_hdr, cams, pose);
    }
    {
      const Affine b2w = load_pose(pose);
      bool vote_deferred = false;  // decided by region_correspondences (the one predicate for both sides)
      if (rm) {
        const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
        Affine b2dc = b2c;
        if (rdcam) b2dc = mul_pose(load_pose(rdcam->world2camera), b2w);
        region_view = region_correspondences<HIST_LDS, SPLIT ? 2 : 8, RENDER>(*rm, *cam, rdcam, b2c, b2dc, iteration, c, s,
                                                                              line_lo, line_hi, &vote_deferred, region_view,
                                                                              SPLIT ? &exchange : nullptr);
      }
      if (dm) {
        PHASE_T0();
        const Affine b2c = mul_pose(load_pose(dcam->world2camera), b2w);
        // (same view table and same camera pose as the region modality of this body: its search is this one's)
        depth_correspondences_scan<RENDER>(*dm, *dcam, b2c, iteration, c, ps, np, s.misc, pt_lo, pt_hi,
                                           (rm && dm->view_search_shared) ? region_view : -1);
        PHASE_MARK(16);
      }
      if constexpr (SPLIT) {
        PHASE_T0();
        // with occlusion handling on, the flags of the own lines (their occlusion results) travel too and the vote
        // over all lines follows the exchange (region_correspondences, defer_vote)
        if (rm) {
          exchange.n_region_fields = rm->distribution_length + (vote_deferred ? 1 : 0);
          exchange.first_region_row = vote_deferred ? LS_VALID : LS_DIST0;
        }
        // publish the own part's results, take the moments of the own lines while the other parts' results are on
        // their way, collect them, then the moments of the received lines
        EXCHANGE_STAMP(0, c, part, object);
        // (the distribution rows left with phase C2 unless the vote is deferred)
        split_exchange_publish(exchange, c, s, rm != nullptr, ps, np, dm != nullptr,
                               (rm && !vote_deferred) ? rm->distribution_length : 0);
        EXCHANGE_STAMP(1, c, part, object);
        if (rm && !vote_deferred) region_moments(*rm, s, line_lo, line_hi, true);
        if (!split_exchange_collect(exchange, c, s, rm != nullptr, ps, np, dm != nullptr)) return;
        EXCHANGE_STAMP(2, c, part, object);
        PHASE_MARK(22);
        if (vote_deferred) {
          region_finish_flags(*rm, s);
          region_moments(*rm, s);
        } else if (rm) {
          region_moments(*rm, s, line_lo, line_hi, false);
        }
      } else {
        if (rm) region_moments(*rm, s);
      }
      {
        PHASE_T0();
        if (dm) depth_correspondences_vote<RENDER>(*dm, iteration, ps, np, s.misc);
        else __syncthreads();
        PHASE_MARK(25);
      }
    }
    for (int u = 0; u < n_update_iterations; ++u) {
      PHASE_T0();
      const Affine b2w = load_pose(pose);
      bool side_by_side = false;
      float sum_early = 0.0f;
      if constexpr (PAIR) {
        const int half = (int)blockDim.x >> 1;
        side_by_side = rm && dm && (half & (kWave - 1)) == 0 && chain_slots(s.nl) <= half && chain_slots(np) <= half;
        if (side_by_side) {
          if ((int)threadIdx.x < half) {
            const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
            region_products(*rm, *cam, b2c, c, u, s, rows_r, layout.pitch_r, (int)threadIdx.x, half);
          } else {
            const Affine b2c = mul_pose(load_pose(dcam->world2camera), b2w);
            depth_products(*dm, b2c, c, ps, np, rows_d, layout.pitch_d, (int)threadIdx.x - half, half);
          }
        }
      }
      if (!side_by_side) {
      if (rm) {
        const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
        region_products(*rm, *cam, b2c, c, u, s, rows_r, layout.pitch_r);
      }
      if (dm) {
        const Affine b2c = mul_pose(load_pose(dcam->world2camera), b2w);
        depth_products(*dm, b2c, c, ps, np, rows_d, layout.pitch_d);
      }
      }
      __syncthreads();
#ifdef M3T_PHASE_TIMING
      if (u == 0) { PHASE_MARK(5); } else { PHASE_MARK(24); }
#endif
      if constexpr (PAIR) {
        if (side_by_side) {  // the lines' sums by the first wave, the points' by the second (handed over in LDS)
          if (threadIdx.x < 2 * kWave) {
            const bool second = threadIdx.x >= kWave;
            const int lane = (int)threadIdx.x & (kWave - 1);
            float unused = 0.0f;
            chain_sums(second ? rows_d : rows_r, second ? layout.pitch_d : layout.pitch_r, chain_slots(second ? np : s.nl),
                       nullptr, 0, 0, gh_lane_row(lane < 42 ? lane : 0), sum_early, unused);
            if (second && lane < 42) gh_depth[lane] = sum_early;
          }
          __syncthreads();
        }
      }
#ifdef M3T_TREE_SUMS
      // EXPERIMENT (tools/variants, never the product): the g/H sums as north_star literally
      // prescribes them -- LDS + wavefront tree reductions -- instead of the reference-order chain: 16 lanes per
      // product row add every sixteenth slot, a DPP row tree adds the sixteen partial sums.  Not the reference's
      // order: the poses leave the oracle's bits (tools/tree_sums_deviation.py measures by how much;
      // profiles/r05_headline_floor.txt has the result).
      float* tree_r = s.misc + kMiscPartials;       // 27 sums per modality (the per-wave partials' scratch)
      float* tree_d = s.misc + kMiscPartials + 32;
      {
        const int row = threadIdx.x >> 4, gl = threadIdx.x & 15;
        if (row < 27) {
          float acc_r = 0.0f, acc_d = 0.0f;
          if (rm) for (int i = gl; i < chain_slots(s.nl); i += 16) acc_r -= rows_r[row * layout.pitch_r + i];
          if (dm) for (int i = gl; i < chain_slots(np); i += 16) acc_d -= rows_d[row * layout.pitch_d + i];
          acc_r += dpp_zero<0x111, 0xf>(acc_r); acc_r += dpp_zero<0x112, 0xf>(acc_r);
          acc_r += dpp_zero<0x114, 0xf>(acc_r); acc_r += dpp_zero<0x118, 0xf>(acc_r);
          acc_d += dpp_zero<0x111, 0xf>(acc_d); acc_d += dpp_zero<0x112, 0xf>(acc_d);
          acc_d += dpp_zero<0x114, 0xf>(acc_d); acc_d += dpp_zero<0x118, 0xf>(acc_d);
          if (gl == 15) { tree_r[row] = acc_r; tree_d[row] = acc_d; }
        }
      }
      __syncthreads();
#endif
      if (threadIdx.x < kWave) {  // one wave: the sums in the reference's order, Link sum, solve, pose update
        float sum_r = 0.0f, sum_d = 0.0f;
#ifdef M3T_TREE_SUMS
        sum_r = rm ? tree_r[gh_lane_row(threadIdx.x < 42 ? threadIdx.x : 0)] : 0.0f;
        sum_d = dm ? tree_d[gh_lane_row(threadIdx.x < 42 ? threadIdx.x : 0)] : 0.0f;
#else
        if (PAIR && side_by_side) {
          sum_r = sum_early;
          sum_d = gh_depth[threadIdx.x < 42 ? threadIdx.x : 0];
        } else {
          chain_sums(rm ? rows_r : nullptr, layout.pitch_r, chain_slots(s.nl), dm ? rows_d : nullptr, layout.pitch_d,
                     chain_slots(np), gh_lane_row(threadIdx.x < 42 ? threadIdx.x : 0), sum_r, sum_d);
        }
#endif
        float gh = 0.0f;  // Link::CalculateGradientAndHessian link.cpp:184-193
        if (rm) gh += sum_r;
        if (dm) gh += sum_d;
        if (write_state && threadIdx.x < 42) {
          gh_region[threadIdx.x] = sum_r;
          gh_depth[threadIdx.x] = sum_d;
        }
        PHASE_MARK(23);
        rigid_solve_wave(gh, o.tikhonov_rotation, o.tikhonov_translation, (LdsW)pose, (LdsW)(s.misc + kMiscSolve));
      }
      __syncthreads();
      PHASE_MARK(6);
    }
  }
  if constexpr (GUARD) {
    // the final pose (the histogram lines run there), then the verdict for the whole workgroup: a step that left its
    // rectangles is dropped before anything of it reaches memory
    int* verdict = reinterpret_cast<int*>(s.misc) + 96;
    if (guard_wave) {
      guard_miss |= roi_guard_outside(guard->items, guard->rects, guard->n_cams, guard->n_rect_slots, guard_hdr, cams, pose);
      if ((threadIdx.x & (kWave - 1)) == 0) *verdict = guard_miss;
    }
    __syncthreads();
    const bool miss = *verdict != 0;
    if (part == 0 && threadIdx.x == 0) roi_guard_report(*guard, guard_hdr, o.body, miss);
    if (miss) return;
  }
  // (every workgroup of a split object holds the same pose: the first one writes it; no other workgroup can
  // still be waiting to read the old one, it had to publish its first results before this one got here)
  if (threadIdx.x < 16 && part == 0) body_poses[16 * o.body + threadIdx.x] = pose[threadIdx.x];
  if (rm && threadIdx.x == 0 && part == 0) *as_global_w(rm->last_view) = region_view;
  if (record_poses) search_poses[(first_corr_iteration + n_corr_iterations + 1) * 16 + threadIdx.x] = pose[threadIdx.x];
  if (write_state && part == 0) {
    if (rm) {
      for (int i = threadIdx.x; i < LS_FIELDS * rm->n_lines_max; i += blockDim.x) {
        int f = i / rm->n_lines_max, l = i - f * rm->n_lines_max;
        rm->line_state[i] = s.state[f * s.nl + l];
      }
      if (threadIdx.x < 42) rm->gradient_hessian[threadIdx.x] = gh_region[threadIdx.x];
    }
    if (dm) {
      for (int i = threadIdx.x; i < PS_FIELDS * dm->n_points_max; i += blockDim.x) {
        int f = i / dm->n_points_max, l = i - f * dm->n_points_max;
        dm->point_state[i] = ps[f * np + l];
      }
      if (threadIdx.x < 42) dm->gradient_hessian[threadIdx.x] = gh_depth[threadIdx.x];
    }
  }
  PHASE_T0();
  if (fuse_histogram && rm) {
    // RegionModality::CalculateResults :572-583 in the same launch: the packed count table takes over the LDS
    // of the line buffers (misc block first, as in region_histogram_kernel)
    const Affine b2w = load_pose(pose);
    __syncthreads();
    const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
    Affine b2dc = b2c;
    if (rdcam) b2dc = mul_pose(load_pose(rdcam->world2camera), b2w);
    const bool handle_occlusions = (iteration - rm->first_iteration) >= rm->n_unoccluded_iterations;
    // (the workgroups of a split object take an equal share of the bins each: n_bins^3 is a multiple of 64)
    const int n_bins3 = rm->n_bins * rm->n_bins * rm->n_bins;
    const int bin_lo = SPLIT ? part * (n_bins3 / n_parts) : 0;
    const int bin_hi = SPLIT ? bin_lo + n_bins3 / n_parts : n_bins3;
    region_histogram_update<false, false, RENDER>(*rm, *cam, rdcam, b2c, b2dc, handle_occlusions, false,
                                                  (__attribute__((address_space(3))) uint32_t*)(lds_t + M3T_MISC_FLOATS),
                                                  lds_t, bin_lo, bin_hi, nullptr, 0, 0, region_view);
  }
  PHASE_MARK(26);
}

}  // extern "C++"
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, int first_corr_iteration) {
  tracking_step_body<false>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration, n_corr_iterations,
                            n_update_iterations, write_state, fuse_histogram, nullptr, first_corr_iteration);
}
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_lds_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, int first_corr_iteration) {
  tracking_step_body<true>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration, n_corr_iterations,
                            n_update_iterations, write_state, fuse_histogram, nullptr, first_corr_iteration);
}
// several workgroups per object: for batches that leave most CUs idle
// (round 4: a 128-VGPR build -- two of these workgroups per CU, so 8 instead of 4 per object at 64 objects -- spills
// 496 bytes per lane: 0.194 ms with 4 and 0.232 ms with 8 workgroups per object against 0.150 ms)
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_split_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, SplitParams split) {
  tracking_step_body<false, true>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration,
                                  n_corr_iterations, n_update_iterations, write_state, fuse_histogram, &split);
}

// Bodies with a RegionModality and a DepthModality: the same three kernels with PAIR (tracking_step_body): ycb21 0.157 ->
// 0.143 ms per step, 64 / 128 / 256 Region + Depth objects 0.207 / 0.295 / 0.429 -> 0.193 / 0.281 / 0.415
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_pair_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, int first_corr_iteration) {
  tracking_step_body<false, false, false, false, true>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration,
                                                       n_corr_iterations, n_update_iterations, write_state, fuse_histogram,
                                                       nullptr, first_corr_iteration);
}
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_lds_pair_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, int first_corr_iteration) {
  tracking_step_body<true, false, false, false, true>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration,
                                                      n_corr_iterations, n_update_iterations, write_state, fuse_histogram,
                                                      nullptr, first_corr_iteration);
}
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_split_pair_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, SplitParams split) {
  tracking_step_body<false, true, false, false, true>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration,
                                                      n_corr_iterations, n_update_iterations, write_state, fuse_histogram, &split);
}

// ROI ingest: the same three kernels with the guard compiled in (frame slots that hold the trackers' rectangles only;
// the kernels above stay what they are for whole frames)
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_guard_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, RoiGuardArgs guard) {
  tracking_step_body<false, false, false, true>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration,
                                                n_corr_iterations, n_update_iterations, write_state, fuse_histogram,
                                                nullptr, 0, &guard);
}
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_lds_guard_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, RoiGuardArgs guard) {
  tracking_step_body<true, false, false, true>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration,
                                               n_corr_iterations, n_update_iterations, write_state, fuse_histogram,
                                               nullptr, 0, &guard);
}
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_split_guard_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_corr_iterations, int n_update_iterations, int write_state,
                     int fuse_histogram, SplitParams split, RoiGuardArgs guard) {
  tracking_step_body<false, true, false, true>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration,
                                               n_corr_iterations, n_update_iterations, write_state, fuse_histogram, &split,
                                               0, &guard);
}

// the same with the renderer-fed branches compiled in (and PAIR: 1 object 0.570 -> 0.545 ms per step, 64 objects 1.209 -> 1.181), one correspondence search per launch (the host redraws the
// focused renderings between the searches): round 4 -- the renderer-fed step of a small batch was ONE workgroup per
// object (87 us per search for the reference's test scene)
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_split_render_kernel(const RigidOptDev* opts, const RegionModDev* rmods, const DepthModDev* dmods,
                     const CameraDev* cams, float* body_poses, TrackLdsLayout layout, int off_points, int np,
                     int iteration, int n_update_iterations, int write_state, int first_corr_iteration,
                     SplitParams split) {
  tracking_step_body<false, true, true, false, true>(opts, rmods, dmods, cams, body_poses, layout, off_points, np, iteration, 1,
                                                     n_update_iterations, write_state, 0, &split, first_corr_iteration);
}

// Test hook (m3t_hip_debug_log_checksum): the logarithm exactly as region_products takes it -- m3t_log_fast on the
// table in LDS, the general double logarithm where the fast path does not vouch -- for every float whose bit pattern
// lies in [first_bits, last_bits].  out[0] += sum of result_bits * (input_bits | 1) mod 2^64, out[1] += calls that
// took the general logarithm, out[2] += the same sum over those calls alone (ocml's logarithm by itself).
// tests/cpp/log_check.cpp forms the same sums from float(std::log(double(x))) on the host.
__global__ void __launch_bounds__(256)
log_checksum_kernel(unsigned first_bits, unsigned last_bits, unsigned long long* out) {
  __shared__ double table[M3T_LOG_TABLE_DOUBLES];
  if (threadIdx.x < M3T_LOG_TABLE_DOUBLES)
    reinterpret_cast<uint64_t*>(table)[threadIdx.x] = g_log_table_bits[threadIdx.x];
  __syncthreads();
  typedef const __attribute__((address_space(3))) double* LdsDoubles;
  LdsDoubles log_table = (LdsDoubles)table;
  unsigned long long sum = 0ull, fallbacks = 0ull, fallback_sum = 0ull;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long b = (unsigned long long)first_bits + blockIdx.x * blockDim.x + threadIdx.x; b <= last_bits;
       b += stride) {
    const unsigned ix = (unsigned)b;
    const float x = __uint_as_float(ix);
    float y = 0.0f;
    const bool vouched = m3t_log_fast(x, log_table, &y);
    if (!vouched) y = (float)log((double)x);
    const unsigned long long term = (unsigned long long)__float_as_uint(y) * (unsigned long long)(ix | 1u);
    sum += term;
    if (!vouched) {
      ++fallbacks;
      fallback_sum += term;
    }
  }
  atomicAdd(&out[0], sum);
  atomicAdd(&out[1], fallbacks);
  atomicAdd(&out[2], fallback_sum);
}

}  // extern "C"
