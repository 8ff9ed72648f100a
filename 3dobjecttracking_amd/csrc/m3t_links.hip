// m3t_links.hip — kinematic structures on the device: Link::CalculateJacobian /
// CalculateGradientAndHessian / UpdatePoses (src/link.cpp:159-241), Constraint
// (src/constraint.cpp:81-102,176-274) and Optimizer::CalculateOptimization for any dof with
// constraint rows (src/optimizer.cpp:144-167, 281-346).  Included by m3t_hip_api.hip after
// m3t_kernels.hip (same translation unit, shares its pose helpers).
//
// These systems are tiny (dof <= a few dozen): one wave per kinematic structure executes them with
// its work arrays in LDS (lanes over matrix elements and links); the data-parallel work
// (correspondences, g/H) stays in the modality kernels.  The optimisation is split in two kernels at the only point where a
// structure spread over several GPUs exchanges data:
//   links_project_kernel : J per link, link g/H, partial A = sum J^T H J (lower), b = sum J^T g
//   [ one all-reduce(sum) over the stacked [dof*dof | dof] buffers of all structures ]
//   links_solve_kernel   : constraint rows, Tikhonov, pivoted LDL^T, NaN guard, pose update
// Arithmetic mirrors the CPU restatement operation for operation.

#include "m3t_exact_math.h"  // atan2f / tanf / tan of the constraint code, the implementation the oracle includes too

#define M3T_MAX_LINK_MODALITIES 4

struct LinkDev {
  int body;    // body id or -1
  int parent;  // index inside this structure's link array (DFS order: parents first) or -1
  float body2joint[16], joint2parent[16], link2world[16];
  int free_directions[6];
  int fixed_body2joint_pose;
  int first_jacobian_index;
  int n_gh;
  const float* gh[M3T_MAX_LINK_MODALITIES];  // modality gradient_hessian buffers (6 + 36)
};
struct ConstraintDev {
  int link1, link2;  // indices inside the structure
  float body12joint1[16], body22joint2[16];
  int directions[6];
  int n;  // number of constrained directions
};
struct SoftConstraintDev {  // soft_constraint.h
  ConstraintDev joint;
  float max_distance_rotation, max_distance_translation;
  float standard_deviation_rotation, standard_deviation_translation;
};
struct TreeOptDev {
  int n_links;
  LinkDev* links;
  int dof;
  int n_constraints;
  ConstraintDev* constraints;
  int n_rows;  // sum of constraint rows
  int n_soft;  // soft constraints
  SoftConstraintDev* soft;
  float tikhonov_rotation, tikhonov_translation;
  float* work;     // scratch, layout in tree_work_floats()
  float* partial;  // [dof*dof | dof]
  // tracking_step_tree_kernel: the links that carry modalities ("tracked" links, one workgroup each) and the buffer
  // through which their workgroups hand each other the link sums: [2 slots][n_tracked][64] {tag, value} granules,
  // then one abort word
  int n_tracked;
  const int* tracked_links;     // [n_tracked] link index inside the structure
  unsigned long long* exchange;
  // tracking_step_tree_segment_kernel (a structure spread over processes): the second copy of the link table -- a launch
  // reads one and writes the other, so that no workgroup can read joints another workgroup of the same launch has
  // already moved -- and where this structure's links start in the stacked link sums
  LinkDev* links_alt;
  int first_link;
};
#define M3T_TREE_GRANULES 64  /* granules per tracked link and slot (42 used) */
#define M3T_TREE_LANE_FIELDS 9 /* per-lane constants of a structure (tree_tables) */
// what the one-launch step's structure code (tracking_step_tree_kernel) is laid out for: a link per wave in the
// system assembly, two lanes per link for the adjoints, four per link for exp(), a Jacobian column per lane
#define M3T_TREE_FUSED_MAX_LINKS 16
#define M3T_TREE_FUSED_MAX_SIZE 64

// one workgroup of tracking_step_tree_kernel
struct TreeStepDev {
  int opt;                 // structure (index into the TreeOptDev table)
  int link;                // this workgroup's link inside the structure
  int tracked;             // ... and its number among the structure's tracked links
  int region_modality;     // index into the RegionModDev table or -1
  int depth_modality;      // index into the DepthModDev table or -1
  int region_first;        // order of the two in Link::modalities (Link::CalculateGradientAndHessian adds in that order)
};
struct TreeStepParams {
  unsigned seq;            // launch sequence number inside the granule tags
  unsigned abort_id;       // what a workgroup that waited in vain writes to *host_abort
  unsigned* host_abort;    // mapped host word
};
// one launch of tracking_step_tree_segment_kernel: [solve from the summed link sums of the previous Newton step] ->
// [correspondence search | line state from the search's launch] -> [g/H products, the link's sums]
enum {
  TSEG_SOLVE = 1,        // apply the (all-reduced) link sums in `sums_in` first: project, solve, update the joints
  TSEG_SEARCH = 2,       // run the correspondence search (the first Newton step of a correspondence iteration)
  TSEG_STORE_STATE = 4,  // ... and leave its line / point state in global memory for the following launches
  TSEG_LOAD_STATE = 8,   // a later Newton step of the same search: line / point state from global memory
  TSEG_SUMS = 16,        // form this link's g/H sums and write them to `sums_out`
  TSEG_FINAL = 32,       // the frame's last launch: bodies written back, histogram update
  TSEG_FIRST = 64,       // no launch of this frame has moved the bodies yet: links with a body stand where the body stands
  TSEG_LINKS_FROM_ALT = 128,  // read the link table from links_alt (else links) ...
  TSEG_LINKS_TO_ALT = 256,    // ... and write the solved one to links_alt (else links)
};
struct TreeSegmentParams {
  int flags;
  int corr_iteration, opt_iteration;
  const float* sums_in;  // [all links of all structures][42], summed over the ranks
  float* sums_out;       // this rank's link sums of this Newton step (links without local modalities: zero)
  const float* first_poses;  // TSEG_FIRST: the bodies' poses the frame starts from -- body_poses itself, or (a frame of ONE
                             // Newton step: FIRST and FINAL in the same launch) a snapshot that launch does not write
};

namespace {

__host__ __device__ inline size_t tree_work_floats(int n_links, int dof, int n_rows) {
  size_t size = size_t(dof) + n_rows;
  return size_t(n_links) * (12 * dof + 42 + 72) + size * size + 4 * size + size_t(n_rows) * (dof + 1) + 2 * 6 * 6 + 64 +
         size_t(n_links) * (size_t(dof) * dof + dof) + dof +  // tree_system_block: per-link terms of A and b, Tikhonov vector
         size_t(dof) * (dof + 1) / 2 + M3T_TREE_LANE_FIELDS * 64 +  // tree_tables: (row, column) of the lower triangle's elements, per-lane constants
         size_t(n_links) * 12;  // ... and the inverse of a root's body2joint
}

__device__ inline void affine_to_array(const Affine& a, float* p) {
  for (int c = 0; c < 3; ++c) {
    for (int r = 0; r < 3; ++r) p[c * 4 + r] = a.l[c * 3 + r];
    p[c * 4 + 3] = 0.0f;
  }
  p[12] = a.t[0]; p[13] = a.t[1]; p[14] = a.t[2]; p[15] = 1.0f;
}

// Link::Adjoint link.cpp:341-348: [[R, 0], [skew(t) R, R]], 6x6 column-major
__device__ void adjoint6(const Affine& pose, float* out) {
  float sk[9];
  sk[0] = 0.0f;        sk[3] = -pose.t[2]; sk[6] = pose.t[1];
  sk[1] = pose.t[2];   sk[4] = 0.0f;       sk[7] = -pose.t[0];
  sk[2] = -pose.t[1];  sk[5] = pose.t[0];  sk[8] = 0.0f;
  float tr[9];
  mul3(sk, pose.l, tr);
  for (int i = 0; i < 36; ++i) out[i] = 0.0f;
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) {
      out[c * 6 + r] = pose.l[c * 3 + r];
      out[c * 6 + 3 + r] = tr[c * 3 + r];
      out[(3 + c) * 6 + 3 + r] = pose.l[c * 3 + r];
    }
}

__device__ Affine link_pose(const LinkDev& l, const float* body_poses) {  // Link::link2world_pose link.cpp:296-301
  return l.body >= 0 ? load_pose(body_poses + 16 * l.body) : load_pose(l.link2world);
}

// Eigen::AngleAxisf(Matrix3f) via quaternion
__device__ void angle_axis(const float* m /*3x3 col-major*/, float* angle, float* axis) {
  float q[4];
  float t = m[0] + m[4] + m[8];
  if (t > 0.0f) {
    t = sqrtf(t + 1.0f);
    q[3] = 0.5f * t;
    t = 0.5f / t;
    q[0] = (m[1 * 3 + 2] - m[2 * 3 + 1]) * t;  // (2,1) - (1,2)
    q[1] = (m[2 * 3 + 0] - m[0 * 3 + 2]) * t;  // (0,2) - (2,0)
    q[2] = (m[0 * 3 + 1] - m[1 * 3 + 0]) * t;  // (1,0) - (0,1)
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrtf(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0f);
    q[i] = 0.5f * t;
    t = 0.5f / t;
    q[3] = (m[j * 3 + k] - m[k * 3 + j]) * t;  // (k,j) - (j,k)
    q[j] = (m[i * 3 + j] + m[j * 3 + i]) * t;  // (j,i) + (i,j)
    q[k] = (m[i * 3 + k] + m[k * 3 + i]) * t;  // (k,i) + (i,k)
  }
  float n = sqrtf((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
  if (n != 0.0f) {
    *angle = 2.0f * m3t_atan2f_pos(n, fabsf(q[3]));  // (the f32 nearest to the f64 value: m3t_exact_math.h, shared with the oracle)
    if (q[3] < 0.0f) n = -n;
    for (int c = 0; c < 3; ++c) axis[c] = q[c] / n;
  } else {
    *angle = 0.0f;
    axis[0] = 1.0f; axis[1] = 0.0f; axis[2] = 0.0f;
  }
}

__device__ float xcotx_dev(float x) { return m3t_xcotx(x); }  // common.h:73-77 (m3t_exact_math.h)

// Constraint::UnprojectedConstraintJacobian constraint.cpp:211-274 -> jac [n x 6] column-major
__device__ void constraint_unprojected_jacobian(const ConstraintDev& c, const Affine& joint22joint1,
                                                const Affine& body2joint1, float* jac) {
  const int n = c.n;
  Affine body2joint2 = mul_pose(inverse_pose(joint22joint1), body2joint1);
  Affine inv = inverse_pose(body2joint2);
  float angle, axis[3];
  angle_axis(joint22joint1.l, &angle, axis);
  float angle_half = 0.5f * angle;
  float xc = xcotx_dev(angle_half);
  float sk[9];
  sk[0] = 0.0f;      sk[3] = -axis[2]; sk[6] = axis[1];
  sk[1] = axis[2];   sk[4] = 0.0f;     sk[7] = -axis[0];
  sk[2] = -axis[1];  sk[5] = axis[0];  sk[8] = 0.0f;
  float vm[9];
  for (int cc = 0; cc < 3; ++cc)
    for (int r = 0; r < 3; ++r)
      vm[cc * 3 + r] = (xc * (r == cc ? 1.0f : 0.0f) - angle_half * sk[cc * 3 + r]) + ((1.0f - xc) * axis[r]) * axis[cc];
  for (int i = 0; i < n * 6; ++i) jac[i] = 0.0f;
  int idx = 0;
  for (int d = 0; d < 6; ++d) {
    if (!c.directions[d]) continue;
    if (d < 3) {
      for (int col = 0; col < 3; ++col)
        jac[col * n + idx] = (vm[0 * 3 + d] * body2joint1.l[col * 3 + 0] + vm[1 * 3 + d] * body2joint1.l[col * 3 + 1]) +
                             vm[2 * 3 + d] * body2joint1.l[col * 3 + 2];
    } else {
      float row[3] = {body2joint1.l[0 * 3 + d - 3], body2joint1.l[1 * 3 + d - 3], body2joint1.l[2 * 3 + d - 3]};
      float cr[3] = {inv.t[1] * row[2] - inv.t[2] * row[1], inv.t[2] * row[0] - inv.t[0] * row[2],
                     inv.t[0] * row[1] - inv.t[1] * row[0]};
      for (int col = 0; col < 3; ++col) {
        jac[col * n + idx] = cr[col];
        jac[(3 + col) * n + idx] = row[col];
      }
    }
    idx++;
  }
}

}  // namespace

// SoftConstraint::AddGradientsAndHessiansToLink soft_constraint.cpp:220-272, one residual group
// (rotation: directions 0-2, translation: 3-5); gh = link gradient[6] | hessian[36]
__device__ void soft_constraint_add_group(const SoftConstraintDev& sc, bool rotation, const Affine& joint22joint1,
                                          const Affine& body2joint1, float sign, float* g, float* h) {
  ConstraintDev group = sc.joint;
  group.n = 0;
  for (int d = 0; d < 6; ++d) {
    group.directions[d] = sc.joint.directions[d] && ((d < 3) == rotation);
    group.n += group.directions[d] ? 1 : 0;
  }
  const int n = group.n;
  if (n == 0) return;
  float full[3];
  if (rotation) {
    float angle, axis[3];
    angle_axis(joint22joint1.l, &angle, axis);
    for (int k = 0; k < 3; ++k) full[k] = angle * axis[k];
  } else {
    for (int k = 0; k < 3; ++k) full[k] = joint22joint1.t[k];
  }
  float v[3] = {0.0f, 0.0f, 0.0f};
  for (int d = 0, idx = 0; d < 3; ++d)
    if (group.directions[d + (rotation ? 0 : 3)]) v[idx++] = full[d];
  const float squared = n == 1 ? v[0] * v[0] : (n == 2 ? v[0] * v[0] + v[1] * v[1] : v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]));
  const float distance = sqrtf(squared);
  const float max_distance = rotation ? sc.max_distance_rotation : sc.max_distance_translation;
  const float sd = rotation ? sc.standard_deviation_rotation : sc.standard_deviation_translation;
  if (!(distance > max_distance)) return;
  float jac[18];
  constraint_unprojected_jacobian(group, joint22joint1, body2joint1, jac);
  float vn[3], r[3];
  for (int k = 0; k < n; ++k) {
    vn[k] = v[k] / distance;
    r[k] = v[k] - vn[k] * max_distance;
  }
  const float cg = sign / (sd * sd), ch = 1.0f / (sd * sd), ratio = max_distance / distance;
  float mm[9];
  for (int c = 0; c < n; ++c)
    for (int k = 0; k < n; ++k) {
      float id = k == c ? 1.0f : 0.0f;
      mm[c * 3 + k] = id - ratio * (id - vn[k] * vn[c]);
    }
  for (int i = 0; i < 6; ++i) {
    float s = 0.0f;
    for (int k = 0; k < n; ++k) s += (cg * jac[i * n + k]) * r[k];
    g[i] -= s;
  }
  float jm[18];
  for (int c = 0; c < n; ++c)
    for (int i = 0; i < 6; ++i) {
      float s = 0.0f;
      for (int k = 0; k < n; ++k) s += (ch * jac[i * n + k]) * mm[c * 3 + k];
      jm[c * 6 + i] = s;
    }
  for (int c = 0; c < 6; ++c)
    for (int i = 0; i < 6; ++i) {
      float s = 0.0f;
      for (int k = 0; k < n; ++k) s += jm[k * 6 + i] * jac[c * n + k];
      h[c * 6 + i] -= s;
    }
}

// ---------------------------------------------------------------------------
// One wave (a 64-thread workgroup) per kinematic structure, all structures of a frame in one launch.  The work arrays
// live in LDS when they fit (tree_work_floats <= ~38 K floats: up to ~40 bodies), else in the structure's global
// scratch; lanes share the loops over matrix elements / links, the accumulation order of every element stays the
// serial one (the oracle's), so the poses equal the CPU restatement bit for bit.
// ---------------------------------------------------------------------------
struct TreeWork {
  float *J, *GH, *AD, *HJ, *A, *b, *temp, *cres, *j1, *j2;
  float *terms, *tikhonov;  // [n_links][dof * dof + dof], [dof] (tree_system_block)
  int* trans;
  int *lower, *lanes;       // [dof (dof + 1) / 2], [M3T_TREE_LANE_FIELDS][64] (tree_tables)
  float* root_inverse;      // [n_links][12]: body2joint^-1 of a root, columns 0..2 | translation (tree_tables)
};
__device__ __forceinline__ TreeWork tree_carve(float* w, int n_links, int dof, int n_rows) {
  const int size = dof + n_rows;
  TreeWork t;
  t.J = w;                                  // [n_links][6 * dof]
  t.GH = t.J + (size_t)n_links * 6 * dof;   // [n_links][42]
  t.AD = t.GH + (size_t)n_links * 42;       // [n_links][72]: adjoint(parent2body) | adjoint(joint2body); later [n_links][12] variations
  t.HJ = t.AD + (size_t)n_links * 72;       // [n_links][6 * dof]
  t.A = t.HJ + (size_t)n_links * 6 * dof;   // [size][size]
  t.b = t.A + (size_t)size * size;
  t.temp = t.b + size;
  t.trans = reinterpret_cast<int*>(t.temp + size);
  t.cres = t.temp + 2 * size + size;        // [n_rows]
  t.j1 = t.cres + n_rows + (size_t)n_rows * dof;
  t.j2 = t.j1 + 36;
  t.terms = t.j2 + 36 + 64;
  t.tikhonov = t.terms + (size_t)n_links * ((size_t)dof * dof + dof);
  t.lower = reinterpret_cast<int*>(t.tikhonov + dof);
  t.lanes = t.lower + (size_t)dof * (dof + 1) / 2;
  t.root_inverse = reinterpret_cast<float*>(t.lanes + M3T_TREE_LANE_FIELDS * 64);
  return t;
}

// The structure code below is executed by ONE wave.  As the 64-thread workgroup of links_project_kernel /
// links_solve_kernel its phases are separated by __syncthreads() (the work arrays may live in global memory);
// inside tracking_step_tree_kernel (WAVE: the first wave of a 512-thread workgroup, work arrays in LDS) a wave-level
// barrier is enough: the LDS executes one wave's instructions in order.
// MODE of the structure code: 0 = the 64-thread workgroup of links_project_kernel / links_solve_kernel, 1 = ONE wave of
// a larger workgroup (work arrays in LDS), 2 = the WHOLE workgroup of tracking_step_tree_kernel (every loop over
// matrix elements / links spread over all its threads; the wave-level factorisation stays on the first wave).
template <int MODE>
__device__ __forceinline__ int tree_lane() { return MODE == 2 ? (int)threadIdx.x : (int)(threadIdx.x & (kWave - 1)); }
template <int MODE>
__device__ __forceinline__ int tree_width() { return MODE == 2 ? (int)blockDim.x : kWave; }
template <int MODE>
__device__ __forceinline__ void tree_sync_mode() {
  if constexpr (MODE == 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}
template <bool WAVE>
__device__ __forceinline__ void tree_sync() { tree_sync_mode<WAVE ? 1 : 0>(); }
// MODE != 0: every link's link2world in the (LDS) link table is current, body or not
template <int MODE>
__device__ __forceinline__ float tree_link_pose_element(const LinkDev& l, const float* body_poses, int e) {
  if constexpr (MODE != 0) return l.link2world[e];
  else return l.body >= 0 ? body_poses[16 * l.body + e] : l.link2world[e];
}
template <int MODE>
__device__ __forceinline__ Affine tree_link_pose(const LinkDev& l, const float* body_poses) {
  if constexpr (MODE != 0) return load_pose(l.link2world);
  else return link_pose(l, body_poses);
}

// Eigen::LDLT<MatrixXf, Lower> + solve (optimizer.cpp:162-163) by one wave: the oracle's LdltSolve, its loops over
// rows / columns spread over the lanes
template <bool WAVE>
__device__ __forceinline__ void ldlt_solve_wave(float* a, float* x, int n, float* temp, int* trans, int lane) {
#define A_(r, c) a[(size_t)(c) * n + (r)]
  bool degenerate = false;
  for (int k = 0; k < n && !degenerate; ++k) {
    // first largest |A(i,i)|, i >= k (a NaN never beats `best`; a NaN at k keeps piv = k)
    float v = -1.0f;
    int vi = INT_MAX;
    for (int i = k + lane; i < n; i += kWave) {
      const float d = fabsf(A_(i, i));
      const float key = i == k ? (d != d ? __int_as_float(0x7f800000) : d) : (d != d ? -1.0f : d);
      if (key > v) { v = key; vi = i; }
    }
    const float m = wave_max(v);
    const int piv = wave_min_i(v == m ? vi : INT_MAX);
    if (lane == 0) trans[k] = piv;
    if (piv != k) {
      for (int c = lane; c < k; c += kWave) { float t = A_(k, c); A_(k, c) = A_(piv, c); A_(piv, c) = t; }
      for (int i = piv + 1 + lane; i < n; i += kWave) { float t = A_(i, k); A_(i, k) = A_(i, piv); A_(i, piv) = t; }
      if (lane == 0) { float t = A_(k, k); A_(k, k) = A_(piv, piv); A_(piv, piv) = t; }
      for (int i = k + 1 + lane; i < piv; i += kWave) { float t = A_(i, k); A_(i, k) = A_(piv, i); A_(piv, i) = t; }
    }
    tree_sync<WAVE>();
    if (k > 0) {
      for (int c = lane; c < k; c += kWave) temp[c] = A_(c, c) * A_(k, c);
      tree_sync<WAVE>();
      for (int i = k + lane; i < n; i += kWave) {  // i == k: the pivot; below: A21 -= A20 * temp
        float sacc = 0.0f;
        for (int c = 0; c < k; ++c) sacc += A_(i, c) * temp[c];
        A_(i, k) -= sacc;
      }
      tree_sync<WAVE>();
    }
    const float akk = A_(k, k);
    const bool pivot_valid = fabsf(akk) > 0.0f;
    if (k == 0 && !pivot_valid) {
      for (int j = lane; j < n; j += kWave) trans[j] = j;
      degenerate = true;
    } else if (pivot_valid) {
      for (int i = k + 1 + lane; i < n; i += kWave) A_(i, k) /= akk;
    }
    tree_sync<WAVE>();
  }
  if (lane == 0)
    for (int k = 0; k < n; ++k) { float t = x[k]; x[k] = x[trans[k]]; x[trans[k]] = t; }
  tree_sync<WAVE>();
  for (int c = 0; c + 1 < n; ++c) {  // L y = P b, column sweep: row i sees its c in ascending order
    const float xc = x[c];
    for (int i = c + 1 + lane; i < n; i += kWave) x[i] -= A_(i, c) * xc;
    tree_sync<WAVE>();
  }
  for (int i = lane; i < n; i += kWave) {
    if (fabsf(A_(i, i)) > 1.17549435e-38f) x[i] /= A_(i, i);
    else x[i] = 0.0f;
  }
  tree_sync<WAVE>();
  for (int i = n - 1; i >= 0; --i) {  // L^T w = z: products by the lanes, the ordered subtraction by one
    for (int r = i + 1 + lane; r < n; r += kWave) temp[r] = A_(r, i) * x[r];
    tree_sync<WAVE>();
    if (lane == 0) {
      float sacc = x[i];
      for (int r = i + 1; r < n; ++r) sacc -= temp[r];
      x[i] = sacc;
    }
    tree_sync<WAVE>();
  }
  if (lane == 0)
    for (int k = n - 1; k >= 0; --k) { float t = x[k]; x[k] = x[trans[k]]; x[trans[k]] = t; }
  tree_sync<WAVE>();
#undef A_
}

// Lane L (a constant after unrolling, L < 16) of the caller's ROW of sixteen lanes, by DPP row_newbcast: for the rows
// of a system of at most sixteen unknowns (lanes 0..15) what v_readlane gives, without the trip through an SGPR (round 5:
// v_readlane -> VALU costs ~23 cycles on the dependent path, DPP ~14, and the move folds into the consuming multiply).
template <int L>
__device__ __forceinline__ int row_lane_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + L, 0xf, 0xf, true); }
__device__ __forceinline__ int row_lane(int v, int l) {
  switch (l) {
    case 0: return row_lane_i<0>(v);   case 1: return row_lane_i<1>(v);   case 2: return row_lane_i<2>(v);
    case 3: return row_lane_i<3>(v);   case 4: return row_lane_i<4>(v);   case 5: return row_lane_i<5>(v);
    case 6: return row_lane_i<6>(v);   case 7: return row_lane_i<7>(v);   case 8: return row_lane_i<8>(v);
    case 9: return row_lane_i<9>(v);   case 10: return row_lane_i<10>(v); case 11: return row_lane_i<11>(v);
    case 12: return row_lane_i<12>(v); case 13: return row_lane_i<13>(v); case 14: return row_lane_i<14>(v);
    default: return row_lane_i<15>(v);
  }
}
__device__ __forceinline__ float row_lane(float v, int l) { return __int_as_float(row_lane(__float_as_int(v), l)); }

// The same solve for systems of at most N unknowns with the matrix in REGISTERS: lane = row, one register per column
// (the scheme of rigid_solve_wave for any size).  Eigen's LDLT<Lower> is the bordered variant -- step k writes column
// k only -- so its diagonal pivoting looks at input diagonal entries only and the whole transposition sequence can be
// found first, on the diagonal alone (a wave-wide "first largest" per step); P A P^T is then gathered from LDS and
// factorised without pivoting: element for element the operations of the pivoted in-place algorithm, moved to where
// the swaps would have carried them.  A zero or NaN diagonal (first pivot invalid, or a NaN that the reference's
// comparisons treat specially) takes the general routine.  ~8 k cycles instead of ~50 k for the 13 x 13 system of an
// 8-body chain, where every step of the general routine pays several LDS round trips and barriers.
// FULL: the caller stored BOTH triangles of the (symmetric) matrix -- the gather of P A P^T then needs no (larger, smaller)
// index pair per element.
template <int N, bool WAVE, bool FULL = false>
__device__ __forceinline__ void ldlt_solve_rows(float* a, float* x, int n, float* temp, int* trans, int lane) {
  const bool row = lane < n;
  float d = row ? fabsf(a[(size_t)lane * n + lane]) : -1.0f;  // (a padded lane: below every |A(i,i)|)
  const bool bad = (d != d) || (__builtin_amdgcn_ballot_w64(row && d > 0.0f) == 0);
  if (__builtin_amdgcn_ballot_w64(row && bad) != 0) {  // uniform
    ldlt_solve_wave<WAVE>(a, x, n, temp, trans, lane);
    return;
  }
  // 1. the transposition sequence: position p holds input row src.  Eigen's selection (the first largest of the
  // rest, swapped to the front) leaves DISTINCT values in descending order whatever the swaps were: position p holds
  // the row of rank p, and the ranks take N broadcasts instead of n dependent wave-wide maxima.  Equal values are
  // reordered by the swaps themselves: those take the selection step by step.  (Two rows with the same value have the
  // same rank, so some position finds no row: that is the test for equal values.  Padded lanes rank behind all rows.)
  int rank = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) rank += row_lane(d, j) > d ? 1 : 0;
  int src = -1;
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (row_lane(rank, j) == lane) src = j;
  const bool distinct = __builtin_amdgcn_ballot_w64(row && src < 0) == 0;
  if (!distinct) src = lane;
  if (!row) src = 0;
#pragma nounroll
  for (int k = 0; k < (distinct ? 0 : n); ++k) {
    const float key = (lane >= k && row) ? d : -1.0f;
    const float m = wave_max(key);
    const unsigned long long hits = __builtin_amdgcn_ballot_w64(key == m);
    const int piv = __builtin_ctzll(hits);  // the first position that holds the largest |A(i,i)|, i >= k
    const float dk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), k));
    const float dp = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), piv));
    const int sk = __builtin_amdgcn_readlane(src, k), sp = __builtin_amdgcn_readlane(src, piv);
    if (lane == k) { d = dp; src = sp; }
    else if (lane == piv) { d = dk; src = sk; }
  }
  // 2. row `lane` of P A P^T (lower triangle; the input holds its lower triangle) and the permuted right-hand side
  float b[N];
  if constexpr (FULL) {
    // (the elements above the diagonal come along: they stay in their lane's registers and only ever feed each other)
    const int srcn = src * n;
#pragma unroll
    for (int c = 0; c < N; ++c) {
      const float v = a[row_lane(srcn, c) + src];
      b[c] = (row && c < n) ? v : 0.0f;
    }
  } else {
#pragma unroll
    for (int c = 0; c < N; ++c) {
      const int sc = row_lane(src, c);
      const int hi = src > sc ? src : sc, lo = src > sc ? sc : src;
      b[c] = (row && c <= lane && c < n) ? a[(size_t)lo * n + hi] : 0.0f;
    }
  }
  float xp = row ? x[src] : 0.0f;
  // 3. factorisation, right-looking: as soon as column c is final, its term enters the running sums acc[k] of all
  // later columns -- the same terms in the same order (c = 0, 1, ...) as the bordered algorithm's dot product
  // sum_c A(i, c) * (D_c * A(k, c)) taken when column k's turn comes, but only the LAST term of a column's sum sits
  // on the dependent path (broadcast, division, broadcast) instead of the whole dot product.  Padded rows / columns
  // (lane, c >= n) hold zeros and a unit pivot: they add +-0 and are never read; one basic block, no branches.
  float D[N], acc[N];
#pragma unroll
  for (int k = 0; k < N; ++k) acc[k] = 0.0f;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    b[c] -= acc[c];  // row c: the pivot D_c; rows below: A21 -= A20 * temp  (c = 0: minus +0)
    const float dc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b[c]), c));  // (kept: an SGPR, used by every lane below)
    D[c] = c < n ? dc : 1.0f;
    const bool pivot_valid = fabsf(D[c]) > 0.0f;
    const float q = b[c] / D[c];
    b[c] = (pivot_valid && lane > c) ? q : b[c];
    const float tv = D[c] * b[c];  // lane k: temp_k[c] = D_c * L(k, c)
#pragma unroll
    for (int k = c + 1; k < N; ++k) {
      const float t = row_lane(tv, k);
      acc[k] += b[c] * t;
    }
  }
  // 4. L y = P b (column sweep), D z = y (pseudo-inverse), L^T w = z (ordered subtraction on broadcast values)
#pragma unroll
  for (int c = 0; c < N - 1; ++c) {
    const float xc = row_lane(xp, c);
    xp = lane > c ? xp - b[c] * xc : xp;
  }
  float dself = 1.0f;
#pragma unroll
  for (int k = 0; k < N; ++k) dself = lane == k ? D[k] : dself;
  xp = fabsf(dself) > 1.17549435e-38f ? xp / dself : 0.0f;
  float X[N];
#pragma unroll
  for (int i = 0; i < N; ++i) X[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xp), i));
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    float sacc = X[i];
#pragma unroll
    for (int r = i + 1; r < N; ++r)
      sacc -= row_lane(b[i], r) * X[r];  // L(r, i) * w_r
    X[i] = sacc;
  }
  // 5. un-permute: position p is input row src
  float mine = 0.0f;
#pragma unroll
  for (int i = 0; i < N; ++i) mine = lane == i ? X[i] : mine;
  if (row) x[src] = mine;
  tree_sync<WAVE>();
}
template <bool WAVE, bool FULL = false>
__device__ __forceinline__ void ldlt_solve_any(float* a, float* x, int n, float* temp, int* trans, int lane) {
  // lane: the caller's lane inside the wave that runs the solve (a parameter so that a caller can keep the compiler
  // from working its addresses out far ahead: tree_step_body)
  if (n <= 8) ldlt_solve_rows<8, WAVE, FULL>(a, x, n, temp, trans, lane);
  else if (n <= 13) ldlt_solve_rows<13, WAVE, FULL>(a, x, n, temp, trans, lane);  // (13: a free root and seven one-dof joints)
  else if (n <= 16) ldlt_solve_rows<16, WAVE, FULL>(a, x, n, temp, trans, lane);
  else ldlt_solve_wave<WAVE>(a, x, n, temp, trans, lane);
}

// Optimizer::CalculateDataLinks (:281-296) + AddProjectedGradientsAndHessians (:309-321).
// links: the structure's link table (global, or its LDS copy); gh_links: nullptr = every link sums its modalities'
// buffers (Link::CalculateGradientAndHessian), else [n_links][42] link sums already formed; A / b: where the
// [dof x dof] (lower) and [dof] sums go.
// The part of the projection that depends on the joint poses only: the adjoints and Jacobians of all links
// (Link::CalculateJacobian link.cpp:159-182).  tracking_step_tree_kernel runs it on an otherwise idle wave (MODE 1)
// while the first wave forms the link's sums and the other links' sums are on their way.
template <int MODE>
__device__ __forceinline__ void tree_kinematics(const TreeOptDev& o, const LinkDev* links, const TreeWork& w) {
  const int lane = tree_lane<MODE>(), width = tree_width<MODE>(), dof = o.dof, n_links = o.n_links;
  PHASE_T0();
  // adjoints of every link, one lane per link (they depend on the link's own joint poses only)
  for (int li = lane; li < n_links; li += width) {
    const LinkDev& l = links[li];
    if (l.parent >= 0)
      adjoint6(inverse_pose(mul_pose(load_pose(l.joint2parent), load_pose(l.body2joint))), w.AD + (size_t)li * 72);
    adjoint6(inverse_pose(load_pose(l.body2joint)), w.AD + (size_t)li * 72 + 36);
  }
  tree_sync_mode<MODE>();
  PHASE_MARK(17);
  // Link::CalculateJacobian link.cpp:159-182, parents before children
  for (int li = 0; li < n_links; ++li) {
    const LinkDev& l = links[li];
    float* J = w.J + (size_t)li * 6 * dof;
    const float* ad = w.AD + (size_t)li * 72;
    const float* Jp = l.parent >= 0 ? w.J + (size_t)l.parent * 6 * dof : nullptr;
    for (int e = lane; e < 6 * dof; e += width) {
      const int c = e / 6, r = e - c * 6;
      float v = 0.0f;
      if (Jp) {
        float sacc = 0.0f;
        for (int k = 0; k < 6; ++k) sacc += ad[k * 6 + r] * Jp[(size_t)c * 6 + k];
        v = sacc;
      }
      J[e] = v;
    }
    tree_sync_mode<MODE>();
    if (lane < 36) {
      const int d = lane / 6, r = lane - d * 6;
      if (l.free_directions[d]) {
        int jidx = l.first_jacobian_index;
        for (int dd = 0; dd < d; ++dd) jidx += l.free_directions[dd] ? 1 : 0;
        J[(size_t)jidx * 6 + r] = ad[36 + d * 6 + r];
      }
    }
    tree_sync_mode<MODE>();
  }
  for (int li = lane; li < n_links; li += width) {  // Tikhonov vector optimizer.cpp:252-271 (free-direction order)
    const LinkDev& l = links[li];
    int j = l.first_jacobian_index;
    for (int d = 0; d < 6; ++d)
      if (l.free_directions[d]) w.tikhonov[j++] = d < 3 ? o.tikhonov_rotation : o.tikhonov_translation;
  }
  PHASE_MARK(18);
}

template <int MODE>
__device__ __forceinline__ void tree_project(const TreeOptDev& o, const LinkDev* links, const TreeWork& w, const float* gh_links, float* A,
                             float* b, const float* body_poses, bool kinematics_done = false) {
  const int lane = tree_lane<MODE>(), width = tree_width<MODE>(), dof = o.dof, n_links = o.n_links;
  if (!kinematics_done) tree_kinematics<MODE>(o, links, w);
  PHASE_T0();
  // Link::CalculateGradientAndHessian link.cpp:184-193
  for (int e = lane; e < n_links * 42; e += width) {
    const int li = e / 42, i = e - li * 42;
    const LinkDev& l = links[li];
    float sacc = 0.0f;
    if (gh_links) sacc = gh_links[e];
    else
      for (int m = 0; m < l.n_gh; ++m) sacc += l.gh[m][i];
    w.GH[e] = sacc;
  }
  tree_sync_mode<MODE>();
  // SoftConstraint::AddGradientsAndHessiansToLinks soft_constraint.cpp:113-131 (optimizer.cpp:283-284)
  if (lane == 0) {
    for (int si = 0; si < o.n_soft; ++si) {
      const SoftConstraintDev& sc = o.soft[si];
      Affine b12j1 = load_pose(sc.joint.body12joint1);
      Affine body22joint1 = mul_pose(mul_pose(b12j1, inverse_pose(tree_link_pose<MODE>(links[sc.joint.link1], body_poses))),
                                     tree_link_pose<MODE>(links[sc.joint.link2], body_poses));
      Affine joint22joint1 = mul_pose(body22joint1, inverse_pose(load_pose(sc.joint.body22joint2)));
      for (int which = 0; which < 2; ++which) {
        float g[6], h[36];
        for (int i = 0; i < 6; ++i) g[i] = 0.0f;
        for (int i = 0; i < 36; ++i) h[i] = 0.0f;
        const Affine& body2joint1 = which == 0 ? b12j1 : body22joint1;
        const float sign = which == 0 ? -1.0f : 1.0f;
        soft_constraint_add_group(sc, true, joint22joint1, body2joint1, sign, g, h);
        soft_constraint_add_group(sc, false, joint22joint1, body2joint1, sign, g, h);
        float* gh = w.GH + (size_t)(which == 0 ? sc.joint.link1 : sc.joint.link2) * 42;
        for (int i = 0; i < 6; ++i) gh[i] += g[i];
        for (int i = 0; i < 36; ++i) gh[6 + i] += h[i];
      }
    }
  }
  tree_sync_mode<MODE>();
  // H J of every link, then b = sum J^T g and A = -sum J^T (H J) (lower), link after link per element
  for (int e = lane; e < n_links * 6 * dof; e += width) {
    const int li = e / (6 * dof), rem = e - li * 6 * dof, c = rem / 6, r = rem - c * 6;
    const float* H = w.GH + (size_t)li * 42 + 6;
    const float* J = w.J + (size_t)li * 6 * dof;
    float sacc = 0.0f;
    for (int k = 0; k < 6; ++k) sacc += H[k * 6 + r] * J[(size_t)c * 6 + k];
    w.HJ[e] = sacc;
  }
  tree_sync_mode<MODE>();
  PHASE_MARK(19);
  for (int i = lane; i < dof; i += width) {
    float acc = 0.0f;
    for (int li = 0; li < n_links; ++li) {
      const float* J = w.J + (size_t)li * 6 * dof;
      const float* g = w.GH + (size_t)li * 42;
      float sacc = 0.0f;
      for (int k = 0; k < 6; ++k) sacc += J[(size_t)i * 6 + k] * g[k];
      acc += sacc;
    }
    b[i] = acc;
  }
  for (int e = lane; e < dof * dof; e += width) {
    const int c = e / dof, r = e - c * dof;
    float acc = 0.0f;
    if (r >= c) {
      for (int li = 0; li < n_links; ++li) {
        const float* J = w.J + (size_t)li * 6 * dof;
        const float* hj = w.HJ + (size_t)li * 6 * dof;
        float sacc = 0.0f;
        for (int k = 0; k < 6; ++k) sacc += J[(size_t)r * 6 + k] * hj[(size_t)c * 6 + k];
        acc -= sacc;
      }
    }
    A[e] = acc;
  }
  PHASE_MARK(20);
}

// tracking_step_tree_kernel's form of the sums of tree_project (the kinematics are done: tree_kinematics), by the WHOLE
// workgroup, straight into the system of tree_solve (w.A: [size x size], w.b): the dof x dof block -sum J^T H J (lower
// triangle) with the Tikhonov diagonal, b = sum J^T g, zeros elsewhere (constraint rows are added by tree_solve).
// Every number is formed by the operations of tree_project in their order -- a 6-term dot per (element, link), the
// links added one after the other -- only that the dots of all links are taken side by side first (w.terms) instead
// of inside the element's loop over the links: 3 short rounds over 512 threads instead of a 48-term serial loop.
__device__ __forceinline__ void tree_system_block(const TreeOptDev& o, const LinkDev* links, const TreeWork& w,
                                                  const float* gh_links) {
  const int tid = threadIdx.x, nt = blockDim.x, dof = o.dof, n_links = o.n_links, size = o.dof + o.n_rows;
  const int per_link = dof * dof + dof;
  PHASE_T0();
  // Link::CalculateGradientAndHessian link.cpp:184-193: the link sums as they were collected (gh_links), unless soft
  // constraints add to them: then a copy (w.GH)
  const float* GH = gh_links;
  if (o.n_soft > 0) {  // SoftConstraint::AddGradientsAndHessiansToLinks soft_constraint.cpp:113-131 (optimizer.cpp:283-284)
    GH = w.GH;
    for (int e = tid; e < n_links * 42; e += nt) w.GH[e] = gh_links[e];
    __syncthreads();
    if (tid == 0) {
      for (int si = 0; si < o.n_soft; ++si) {
        const SoftConstraintDev& sc = o.soft[si];
        Affine b12j1 = load_pose(sc.joint.body12joint1);
        Affine body22joint1 = mul_pose(mul_pose(b12j1, inverse_pose(load_pose(links[sc.joint.link1].link2world))),
                                       load_pose(links[sc.joint.link2].link2world));
        Affine joint22joint1 = mul_pose(body22joint1, inverse_pose(load_pose(sc.joint.body22joint2)));
        for (int which = 0; which < 2; ++which) {
          float g[6], h[36];
          for (int i = 0; i < 6; ++i) g[i] = 0.0f;
          for (int i = 0; i < 36; ++i) h[i] = 0.0f;
          const Affine& body2joint1 = which == 0 ? b12j1 : body22joint1;
          const float sign = which == 0 ? -1.0f : 1.0f;
          soft_constraint_add_group(sc, true, joint22joint1, body2joint1, sign, g, h);
          soft_constraint_add_group(sc, false, joint22joint1, body2joint1, sign, g, h);
          float* gh = w.GH + (size_t)(which == 0 ? sc.joint.link1 : sc.joint.link2) * 42;
          for (int i = 0; i < 6; ++i) gh[i] += g[i];
          for (int i = 0; i < 36; ++i) gh[6 + i] += h[i];
        }
      }
    }
    __syncthreads();
  }
  // H J of every link
  for (int e = tid; e < n_links * 6 * dof; e += nt) {
    const int li = e / (6 * dof), rem = e - li * 6 * dof, c = rem / 6, r = rem - c * 6;
    const float* H = GH + (size_t)li * 42 + 6;
    const float* J = w.J + (size_t)li * 6 * dof;
    float sacc = 0.0f;
    for (int k = 0; k < 6; ++k) sacc += H[k * 6 + r] * J[(size_t)c * 6 + k];
    w.HJ[e] = sacc;
  }
  __syncthreads();
  PHASE_MARK(19);
  // the terms of every link: J^T g (dof) and, for the lower triangle, J^T (H J) (dof x dof)
  for (int x = tid; x < n_links * per_link; x += nt) {
    const int li = x / per_link, e = x - li * per_link;
    const float* J = w.J + (size_t)li * 6 * dof;
    float sacc = 0.0f;
    if (e < dof) {
      const float* g = GH + (size_t)li * 42;
      for (int k = 0; k < 6; ++k) sacc += J[(size_t)e * 6 + k] * g[k];
    } else {
      const int a = e - dof, c = a / dof, r = a - c * dof;
      if (r >= c) {
        const float* hj = w.HJ + (size_t)li * 6 * dof;
        for (int k = 0; k < 6; ++k) sacc += J[(size_t)r * 6 + k] * hj[(size_t)c * 6 + k];
      }
    }
    w.terms[x] = sacc;
  }
  __syncthreads();
  // the links one after the other, then the Tikhonov term of the diagonal (tree_solve adds it after the sums too)
  for (int i = tid; i < size; i += nt) {
    float acc = 0.0f;
    if (i < dof)
      for (int li = 0; li < n_links; ++li) acc += w.terms[(size_t)li * per_link + i];
    w.b[i] = acc;
  }
  for (int e = tid; e < size * size; e += nt) {
    const int c = e / size, r = e - c * size;
    float acc = 0.0f;
    if (c < dof && r < dof && r >= c) {
      for (int li = 0; li < n_links; ++li) acc -= w.terms[(size_t)li * per_link + dof + c * dof + r];
      if (r == c) acc += w.tikhonov[c];
    }
    w.A[e] = acc;
  }
  __syncthreads();
  PHASE_MARK(20);
}

// the rest of Optimizer::CalculateOptimization + Optimizer::UpdatePoses (:335-346).  partial: [dof x dof | dof] sums
// (after the all-reduce when the structure spans GPUs); w.J must hold the links' Jacobians.  Returns false when the
// NaN guard skipped the update.
template <int MODE>
__device__ __forceinline__ bool tree_solve(const TreeOptDev& o, LinkDev* links, const TreeWork& w, const float* partial, float* body_poses,
                           int zero_theta, bool assembled = false) {
  // assembled: w.A / w.b already hold the sums and the Tikhonov diagonal (tree_system_block); only the constraint
  // rows are still to be added
  const int lane = tree_lane<MODE>(), width = tree_width<MODE>(), dof = o.dof, n_links = o.n_links, size = o.dof + o.n_rows;
  PHASE_T0();
  float* A = w.A;
  float* b = w.b;
  if (!assembled) {
    for (int e = lane; e < size * size; e += width) {
      const int c = e / size, r = e - c * size;
      A[e] = (!zero_theta && c < dof && r < dof) ? partial[(size_t)c * dof + r] : 0.0f;
    }
    for (int i = lane; i < size; i += width) b[i] = (!zero_theta && i < dof) ? partial[(size_t)dof * dof + i] : 0.0f;
    tree_sync_mode<MODE>();
  }
  if (!zero_theta) {  // zero_theta: Optimizer::CalculateConsistentPoses optimizer.cpp:135 (theta = 0)
    // constraints: Constraint::CalculateResidualAndConstraintJacobian constraint.cpp:81-102
    int idx = dof;
    for (int ci = 0; ci < o.n_constraints; ++ci) {
      const ConstraintDev& c = o.constraints[ci];
      const int n_c = c.n;
      if (lane == 0) {
        const LinkDev& l1 = links[c.link1];
        const LinkDev& l2 = links[c.link2];
        Affine b12j1 = load_pose(c.body12joint1);
        Affine body22joint1 = mul_pose(mul_pose(b12j1, inverse_pose(tree_link_pose<MODE>(l1, body_poses))), tree_link_pose<MODE>(l2, body_poses));
        Affine joint22joint1 = mul_pose(body22joint1, inverse_pose(load_pose(c.body22joint2)));
        float angle, axis[3];
        angle_axis(joint22joint1.l, &angle, axis);
        float rv[3] = {angle * axis[0], angle * axis[1], angle * axis[2]};
        int ri = 0;
        for (int d = 0; d < 6; ++d)
          if (c.directions[d]) w.cres[ri++] = d < 3 ? rv[d] : joint22joint1.t[d - 3];
        constraint_unprojected_jacobian(c, joint22joint1, body22joint1, w.j2);
        constraint_unprojected_jacobian(c, joint22joint1, b12j1, w.j1);
      }
      tree_sync_mode<MODE>();
      const float* J1 = w.J + (size_t)c.link1 * 6 * dof;
      const float* J2 = w.J + (size_t)c.link2 * 6 * dof;
      for (int e = lane; e < dof * n_c; e += width) {
        const int col = e / n_c, r = e - col * n_c;
        float s2 = 0.0f, s1 = 0.0f;
        for (int k = 0; k < 6; ++k) {
          s2 += w.j2[k * n_c + r] * J2[(size_t)col * 6 + k];
          s1 += w.j1[k * n_c + r] * J1[(size_t)col * 6 + k];
        }
        // AddResidualsAndConstraintJacobians optimizer.cpp:323-333
        A[(size_t)col * size + idx + r] = -(s2 - s1);
      }
      if (lane < n_c) b[idx + lane] = w.cres[lane];
      tree_sync_mode<MODE>();
      idx += n_c;
    }
    // Tikhonov vector optimizer.cpp:252-271 (free-direction order, rotation first)
    if (!assembled) {
      for (int li = lane; li < n_links; li += width) {
        const LinkDev& l = links[li];
        int j = l.first_jacobian_index;
        for (int d = 0; d < 6; ++d)
          if (l.free_directions[d]) {
            A[(size_t)j * size + j] += d < 3 ? o.tikhonov_rotation : o.tikhonov_translation;
            j++;
          }
      }
      tree_sync_mode<MODE>();
    }
    PHASE_MARK(12);
    // the factorisation is wave-level code: in the whole-workgroup mode the first wave runs it, the others wait
    if (MODE != 2 || threadIdx.x < kWave) ldlt_solve_any<MODE != 0>(A, b, size, w.temp, w.trans, threadIdx.x & (kWave - 1));
    if constexpr (MODE == 2) __syncthreads();
    PHASE_MARK(13);
    int has_nan = 0;
    for (int i = lane; i < size; i += width) has_nan |= (b[i] != b[i]) ? 1 : 0;
    // NaN guard optimizer.cpp:165
    if constexpr (MODE == 1) {
      if (__builtin_amdgcn_ballot_w64(has_nan != 0) != 0) return false;
    } else {
      if (__syncthreads_or(has_nan)) return false;
    }
  }
  // Link::UpdatePoses link.cpp:205-241: the variations of all links at once (one lane per link) ...
  float* var_all = w.AD;  // [n_links][12]
  if constexpr (MODE == 2) {
    for (int li = lane; li < n_links; li += width) {
      const LinkDev& l = links[li];
      float th[6];
      int j = l.first_jacobian_index;
      for (int d = 0; d < 6; ++d) th[d] = l.free_directions[d] ? b[j++] : 0.0f;
      float K[9], R[9];
      K[0] = 0.0f;   K[3] = -th[2]; K[6] = th[1];
      K[1] = th[2];  K[4] = 0.0f;   K[7] = -th[0];
      K[2] = -th[1]; K[5] = th[0];  K[8] = 0.0f;
      expm3(K, R);
      float* v = var_all + (size_t)li * 12;
      for (int i = 0; i < 9; ++i) v[i] = R[i];
      v[9] = th[3]; v[10] = th[4]; v[11] = th[5];
    }
  } else {
    // four lanes per link: lane c < 3 of the group works out column c of exp(skew(theta_r)) (colexpm3: the oracle's
    // Expm3 operation for operation, a third of its serial length), lane 3 carries the translation
    for (int base = 0; base < n_links; base += kWave / 4) {
      const int li = base + (lane >> 2), c = lane & 3;
      if (li < n_links) {
        const LinkDev& l = links[li];
        float th[6];
        int j = l.first_jacobian_index;
#pragma unroll
        for (int d = 0; d < 6; ++d) th[d] = l.free_directions[d] ? b[j++] : 0.0f;
        float K[9], col[3];
        K[0] = 0.0f;   K[3] = -th[2]; K[6] = th[1];
        K[1] = th[2];  K[4] = 0.0f;   K[7] = -th[0];
        K[2] = -th[1]; K[5] = th[0];  K[8] = 0.0f;
        colexpm3<true>(K, c, col);
        float* v = var_all + (size_t)li * 12;
        if (c == 3) { col[0] = th[3]; col[1] = th[4]; col[2] = th[5]; }
        v[c * 3] = col[0]; v[c * 3 + 1] = col[1]; v[c * 3 + 2] = col[2];
      }
    }
  }
  tree_sync_mode<MODE>();
  PHASE_MARK(14);
  // ... then the joints (all links at once, one lane per matrix element: mul_pose's expression per element) ...
  // (results staged in w.HJ, dead here: every element of the old matrix is read before any is overwritten)
  float* joint2body_chain = w.HJ;
  for (int e = lane; e < n_links * 12; e += width) {
    const int li = e / 12, q = e - li * 12;
    LinkDev& l = links[li];
    if (l.parent < 0) continue;
    const float* v = var_all + (size_t)li * 12;  // variation: l[9] | t[3]
    // joint2parent <- joint2parent * var (fixed body2joint), or body2joint <- var * body2joint
    const float* A4 = l.fixed_body2joint_pose ? l.joint2parent : nullptr;  // 4 x 4 column-major
    const float* B4 = l.fixed_body2joint_pose ? nullptr : l.body2joint;
    float r;
    if (q < 9) {
      const int c = q / 3, k = q - c * 3;
      if (A4) r = (A4[k] * v[c * 3] + A4[4 + k] * v[c * 3 + 1]) + A4[8 + k] * v[c * 3 + 2];
      else r = (v[k] * B4[c * 4] + v[3 + k] * B4[c * 4 + 1]) + v[6 + k] * B4[c * 4 + 2];
    } else {
      const int k = q - 9;
      if (A4) r = ((A4[k] * v[9] + A4[4 + k] * v[10]) + A4[8 + k] * v[11]) + A4[12 + k];
      else r = ((v[k] * B4[12] + v[3 + k] * B4[13]) + v[6 + k] * B4[14]) + v[9 + k];
    }
    joint2body_chain[e] = r;
  }
  tree_sync_mode<MODE>();
  for (int e = lane; e < n_links * 12; e += width) {
    const int li = e / 12, q = e - li * 12;
    LinkDev& l = links[li];
    if (l.parent < 0) continue;
    float* M = l.fixed_body2joint_pose ? l.joint2parent : l.body2joint;
    if (q < 9) M[(q / 3) * 4 + (q % 3)] = joint2body_chain[e];
    else M[12 + (q - 9)] = joint2body_chain[e];
    if (q < 4) M[q * 4 + 3] = q == 3 ? 1.0f : 0.0f;
  }
  tree_sync_mode<MODE>();
  if constexpr (MODE == 2) {
    // ... then link2world.  The root links first (each by its own thread) ...
    for (int li = lane; li < n_links; li += width) {
      LinkDev& l = links[li];
      if (l.parent >= 0) continue;
      const float* v = var_all + (size_t)li * 12;
      Affine var;
      for (int i = 0; i < 9; ++i) var.l[i] = v[i];
      var.t[0] = v[9]; var.t[1] = v[10]; var.t[2] = v[11];
      Affine b2j = load_pose(l.body2joint);
      Affine l2w = mul_pose(mul_pose(mul_pose(tree_link_pose<MODE>(l, body_poses), inverse_pose(b2j)), var), b2j);
      affine_to_array(l2w, l.link2world);
    }
    tree_sync_mode<MODE>();
    // ... then every other link by ITS OWN thread, which walks down from the link's root: link2world of a link is
    // (parent's link2world * joint2parent) * body2joint, left to right like the reference (the products are not
    // associative in floating point).  A thread forms the products of all links on its path itself -- the same
    // operations on the same values as the threads of those links, so the same bits -- and needs no barrier on the way.
    for (int li = lane; li < n_links; li += width) {
      if (links[li].parent < 0) continue;
      int depth = 0, root = li;
      while (links[root].parent >= 0) { root = links[root].parent; ++depth; }
      float P[12];  // 3 x 4: element (k, c) at P[c * 3 + k], the translation in P[9 .. 11]
      {
        const float* R4 = links[root].link2world;
        for (int c = 0; c < 4; ++c)
          for (int k = 0; k < 3; ++k) P[c * 3 + k] = R4[c * 4 + k];
      }
      for (int step = depth - 1; step >= 0; --step) {
        int a = li;  // the ancestor `step` levels above li (step == 0: li itself)
        for (int up = 0; up < step; ++up) a = links[a].parent;
        const LinkDev& la = links[a];
        for (int pass = 0; pass < 2; ++pass) {
          const float* B4 = pass == 0 ? la.joint2parent : la.body2joint;
          float Q[12];
          for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 3; ++k) Q[c * 3 + k] = (P[k] * B4[c * 4] + P[3 + k] * B4[c * 4 + 1]) + P[6 + k] * B4[c * 4 + 2];
          for (int k = 0; k < 3; ++k) Q[9 + k] = ((P[k] * B4[12] + P[3 + k] * B4[13]) + P[6 + k] * B4[14]) + P[9 + k];
          for (int i = 0; i < 12; ++i) P[i] = Q[i];
        }
      }
      float* out = links[li].link2world;
      for (int c = 0; c < 4; ++c) {
        for (int k = 0; k < 3; ++k) out[c * 4 + k] = P[c * 3 + k];
        out[c * 4 + 3] = c == 3 ? 1.0f : 0.0f;
      }
    }
    tree_sync_mode<MODE>();
  } else {
    // ... then link2world, parents before children: (parent's link2world * joint2parent) * body2joint, left to right
    // like the reference (the products are not associative in floating point).  Twelve lanes hold a pose, element
    // (k, c) of its 3 x 4 block in lane 4 k + c: a row sits in one group of four lanes, so the four values of row k
    // that mul_pose's expression for element (k, c) needs come by DPP quad broadcasts -- no LDS round trip inside a
    // product -- and a child whose parent was the previous link takes the parent's pose from the registers it was just
    // formed in.  Same operations on the same values as mul_pose, element for element.
    {
      const int k = (lane >> 2) < 3 ? (lane >> 2) : 0, c = lane & 3;
      const bool holds = lane < 12;
      // out(k, c) = (P(k,0) b0 + P(k,1) b1) + P(k,2) b2 [+ P(k,3) for the translation column]; b = column c of B
      auto quadmul = [&](float P, float b0, float b1, float b2) {
        const float p0 = quad_lane<0>(P), p1 = quad_lane<1>(P), p2 = quad_lane<2>(P), p3 = quad_lane<3>(P);
        const float v = (p0 * b0 + p1 * b1) + p2 * b2;
        return c == 3 ? v + p3 : v;
      };
      float cur = 0.0f;
      int cur_link = -2;
      for (int li = 0; li < n_links; ++li) {
        LinkDev& l = links[li];
        const float* B1 = l.joint2parent;
        const float* B2 = l.body2joint;
        const float e0 = B2[c * 4], e1 = B2[c * 4 + 1], e2 = B2[c * 4 + 2];  // column c of body2joint
        float result;
        if (l.parent >= 0) {
          const float d0 = B1[c * 4], d1 = B1[c * 4 + 1], d2 = B1[c * 4 + 2];  // column c of joint2parent
          const float P = l.parent == cur_link ? cur : links[l.parent].link2world[c * 4 + k];
          result = quadmul(quadmul(P, d0, d1, d2), e0, e1, e2);
        } else {
          // a root: link2world <- ((link2world * body2joint^-1) * variation) * body2joint (link.cpp:226-230)
          const float* v = var_all + (size_t)li * 12;  // variation: l[9] | t[3], column c at v[3 c ..]
          const Affine inv = inverse_pose(load_pose(B2));
          const float i0 = c < 3 ? inv.l[c * 3] : inv.t[0], i1 = c < 3 ? inv.l[c * 3 + 1] : inv.t[1],
                      i2 = c < 3 ? inv.l[c * 3 + 2] : inv.t[2];
          const float T = tree_link_pose_element<MODE>(l, body_poses, c * 4 + k);
          result = quadmul(quadmul(quadmul(T, i0, i1, i2), v[c * 3], v[c * 3 + 1], v[c * 3 + 2]), e0, e1, e2);
        }
        if (holds) l.link2world[c * 4 + k] = result;
        if (lane < 4) l.link2world[lane * 4 + 3] = lane == 3 ? 1.0f : 0.0f;
        cur = result;
        cur_link = li;
        if constexpr (MODE == 0) {  // (the fused kernel's caller writes the bodies)
          tree_sync_mode<MODE>();
          if (l.body >= 0 && lane < 16) body_poses[16 * l.body + lane] = l.link2world[lane];
        }
      }
      tree_sync_mode<MODE>();
    }
  }
  PHASE_MARK(15);
  return true;
}

// ---------------------------------------------------------------------------
// Round 5: the structure code of tracking_step_tree_kernel, laid out for the wavefront.  The same numbers as
// tree_kinematics / tree_project / tree_solve above, operation for operation (the 8-body chain's poses stay equal to
// the oracle's bit for bit), but
//   * the adjoints by two lanes per link, kept in registers; the Jacobians by ONE LANE PER COLUMN, which carries its
//     column down the tree in registers (the adjoint of the link in turn broadcast by v_readlane): no barrier and no LDS
//     round trip between two links, 8 x ~80 instructions instead of 8 x two barriers -- and on the workgroup's LAST wave,
//     which has no correspondence line to form products for: it runs beside the products, the chain and the exchange;
//   * the system -sum J^T H J | sum J^T g by a wave per link (H J and the link's terms need no workgroup barrier in
//     between), the (row, column) of a lower-triangle element from a table instead of integer divisions;
//   * the joint update by twelve lanes per link, four links at a time, and the link2world chain with the next link's
//     joint poses already loaded: DPP quad broadcasts only on the dependent path.
// CONSTRAINED = false leaves Constraint / SoftConstraint code out of the kernel altogether (a chain never runs it; in
// round 4 its private arrays were 40 % of the kernel's scratch instructions).
// ---------------------------------------------------------------------------
enum { TL_COL_OWNER = 0, TL_COL_DIR = 1, TL_PARENT = 2, TL_QUAD_FIRST = 3, TL_QUAD_MASK = 4, TL_JOINT = 5 /* .. 8 */ };
static_assert(M3T_TREE_LANE_FIELDS == TL_JOINT + M3T_TREE_FUSED_MAX_LINKS / 4, "lane table: a joint row per four links");

__device__ __forceinline__ void tree_wave_sync() { tree_sync_mode<1>(); }

// Constants of the structure, once per launch (the link table keeps its shape while a kernel runs): per lane L the link
// and direction that own Jacobian column L (optimizer.cpp:237-250: free directions in link order), the parent of link
// L, first Jacobian index and free-direction bits of link L / 4; the (row | column << 8) of the elements of the lower
// triangle, column after column.  Ends without a barrier.
__device__ __forceinline__ void tree_tables(const TreeOptDev& o, const LinkDev* links, const TreeWork& w) {
  const int tid = threadIdx.x, nt = blockDim.x, dof = o.dof, n_links = o.n_links;
  if (tid < kWave) {
    int owner = -1, dir = 0;
    for (int li = 0; li < n_links; ++li) {
      int j = links[li].first_jacobian_index;
      for (int d = 0; d < 6; ++d)
        if (links[li].free_directions[d]) {
          if (j == tid) { owner = li; dir = d; }
          ++j;
        }
    }
    w.lanes[TL_COL_OWNER * kWave + tid] = owner;
    w.lanes[TL_COL_DIR * kWave + tid] = dir;
    w.lanes[TL_PARENT * kWave + tid] = tid < n_links ? links[tid].parent : -1;
    const int ql = tid >> 2;
    int first = 0, mask = 0;
    if (ql < n_links) {
      first = links[ql].first_jacobian_index;
      for (int d = 0; d < 6; ++d) mask |= links[ql].free_directions[d] ? (1 << d) : 0;
    }
    w.lanes[TL_QUAD_FIRST * kWave + tid] = first;
    w.lanes[TL_QUAD_MASK * kWave + tid] = mask;
    // a root's body2joint does not move while a kernel runs (Link::UpdatePoses link.cpp:226-230 moves link2world): its
    // inverse once per launch, not once per Newton step
    if (tid < n_links && links[tid].parent < 0) {
      const Affine inv = inverse_pose(load_pose(links[tid].body2joint));
      float* ri = w.root_inverse + (size_t)tid * 12;
      for (int i = 0; i < 9; ++i) ri[i] = inv.l[i];
      for (int i = 0; i < 3; ++i) ri[9 + i] = inv.t[i];
    }
    // the joint update (tree_solve_fast): link 4 pass + tid / 16 moves joint2parent (a fixed body2joint) or body2joint
    for (int pass = 0; pass < M3T_TREE_FUSED_MAX_LINKS / 4; ++pass) {
      const int li = pass * 4 + (tid >> 4);
      int code = -1;
      if (li < n_links && links[li].parent >= 0) {
        const bool fixed = links[li].fixed_body2joint_pose != 0;
        const float* M = fixed ? links[li].joint2parent : links[li].body2joint;
        code = (int)(M - reinterpret_cast<const float*>(links)) * 2 + (fixed ? 1 : 0);
      }
      w.lanes[(TL_JOINT + pass) * kWave + tid] = code;
    }
  }
  for (int e = tid; e < dof * dof; e += nt) {
    const int c = e / dof, r = e - c * dof;
    if (r >= c) w.lower[c * dof - c * (c - 1) / 2 + (r - c)] = r | (c << 8);
  }
}

// Link::Adjoint (link.cpp:341-348) of one pose as its two blocks: [[R, 0], [skew(t) R, R]]
struct TreeKin {
  float R[9], TR[9];  // lane 2 li: adjoint(parent2body) of link li, lane 2 li + 1: adjoint(joint2body)
  float own[6];       // lane c: Jacobian column c where its link sets it (link.cpp:176-180)
};
// Part 1 (one wave, before the Jacobians; depends on the joint poses only): the adjoints, the columns the links set
// themselves, the Tikhonov vector (optimizer.cpp:252-271)
__device__ __forceinline__ void tree_kin_adjoints(const TreeOptDev& o, const LinkDev* links, const TreeWork& w, TreeKin& kin,
                                                  int tid) {
  const int lane = tid & (kWave - 1), n_links = o.n_links;
  const int li = lane >> 1, which = lane & 1;
#pragma unroll
  for (int i = 0; i < 9; ++i) { kin.R[i] = 0.0f; kin.TR[i] = 0.0f; }
  if (li < n_links) {
    const LinkDev& l = links[li];
    const Affine b2j = load_pose(l.body2joint);
    Affine m = b2j;
    if (which == 0) m = mul_pose(load_pose(l.joint2parent), b2j);  // (a root's is never read)
    const Affine p = inverse_pose(m);
    float sk[9];
    sk[0] = 0.0f;     sk[3] = -p.t[2]; sk[6] = p.t[1];
    sk[1] = p.t[2];   sk[4] = 0.0f;    sk[7] = -p.t[0];
    sk[2] = -p.t[1];  sk[5] = p.t[0];  sk[8] = 0.0f;
    mul3(sk, p.l, kin.TR);
#pragma unroll
    for (int i = 0; i < 9; ++i) kin.R[i] = p.l[i];
    if (which == 1) {
      float* ad = w.AD + (size_t)li * 72 + 36;  // R | skew(t) R of adjoint(joint2body)
#pragma unroll
      for (int i = 0; i < 9; ++i) { ad[i] = kin.R[i]; ad[9 + i] = kin.TR[i]; }
    }
  }
  tree_wave_sync();
  const int owner = w.lanes[TL_COL_OWNER * kWave + lane], dir = w.lanes[TL_COL_DIR * kWave + lane];
#pragma unroll
  for (int r = 0; r < 6; ++r) kin.own[r] = 0.0f;
  if (owner >= 0) {
    // column `dir` of [[R, 0], [skew(t) R, R]]: rows 0..2 | rows 3..5
    const float* ad = w.AD + (size_t)owner * 72 + 36;
    const int cc = dir < 3 ? dir : dir - 3;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float top = ad[cc * 3 + r], low = ad[9 + cc * 3 + r];
      kin.own[r] = dir < 3 ? top : 0.0f;
      kin.own[3 + r] = dir < 3 ? low : top;
    }
    w.tikhonov[lane] = dir < 3 ? o.tikhonov_rotation : o.tikhonov_translation;
  }
}
// Part 2: Link::CalculateJacobian (link.cpp:159-182), parents before children.  Lane c carries column c:
// J_link[:, c] = adjoint(parent2body) J_parent[:, c] -- sum over k = 0..5 of ad(r, k) J(k, c) in that order; the upper
// right block of the adjoint is zero, and adding a product with zero to a sum that started at +0 changes nothing -- or
// what the link sets itself.  Ends without a barrier (the caller's workgroup barrier publishes w.J).
__device__ __forceinline__ void tree_kin_jacobians(const TreeOptDev& o, const TreeWork& w, const TreeKin& kin, int tid) {
  const int lane = tid & (kWave - 1), n_links = o.n_links, dof = o.dof;
  const int parents = w.lanes[TL_PARENT * kWave + lane], owner = w.lanes[TL_COL_OWNER * kWave + lane];
  float v[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) v[r] = 0.0f;
  int held = -2;  // the link whose column is in v
#pragma nounroll
  for (int li = 0; li < n_links; ++li) {
    const int p = __builtin_amdgcn_readlane(parents, li);
    float nv[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) nv[r] = 0.0f;
    if (p >= 0) {  // uniform
      float in[6];
      if (p == held) {
#pragma unroll
        for (int r = 0; r < 6; ++r) in[r] = v[r];
      } else {  // (a branching tree: this lane's own store of the parent's column)
        const float* Jp = w.J + (size_t)p * 6 * dof + (size_t)(lane < dof ? lane : 0) * 6;
#pragma unroll
        for (int r = 0; r < 6; ++r) in[r] = Jp[r];
      }
      float R[9], T[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        R[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kin.R[i]), 2 * li));
        T[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kin.TR[i]), 2 * li));
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float top = 0.0f, low = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) top += R[k * 3 + r] * in[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) low += T[k * 3 + r] * in[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) low += R[k * 3 + r] * in[3 + k];
        nv[r] = top;
        nv[3 + r] = low;
      }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) v[r] = owner == li ? kin.own[r] : nv[r];
    held = li;
    if (lane < dof) {
      float* J = w.J + (size_t)li * 6 * dof + (size_t)lane * 6;
#pragma unroll
      for (int r = 0; r < 6; ++r) J[r] = v[r];
    }
  }
}

// SoftConstraint::AddGradientsAndHessiansToLinks soft_constraint.cpp:113-131 (optimizer.cpp:283-284) onto a copy of
// the link sums; thread 0, between two workgroup barriers
__device__ __noinline__ void tree_soft_constraints(int n_soft, const SoftConstraintDev* soft, const LinkDev* links,
                                                   float* GH) {
  for (int si = 0; si < n_soft; ++si) {
    const SoftConstraintDev& sc = soft[si];
    Affine b12j1 = load_pose(sc.joint.body12joint1);
    Affine body22joint1 = mul_pose(mul_pose(b12j1, inverse_pose(load_pose(links[sc.joint.link1].link2world))),
                                   load_pose(links[sc.joint.link2].link2world));
    Affine joint22joint1 = mul_pose(body22joint1, inverse_pose(load_pose(sc.joint.body22joint2)));
    for (int which = 0; which < 2; ++which) {
      float g[6], h[36];
      for (int i = 0; i < 6; ++i) g[i] = 0.0f;
      for (int i = 0; i < 36; ++i) h[i] = 0.0f;
      const Affine& body2joint1 = which == 0 ? b12j1 : body22joint1;
      const float sign = which == 0 ? -1.0f : 1.0f;
      soft_constraint_add_group(sc, true, joint22joint1, body2joint1, sign, g, h);
      soft_constraint_add_group(sc, false, joint22joint1, body2joint1, sign, g, h);
      float* gh = GH + (size_t)(which == 0 ? sc.joint.link1 : sc.joint.link2) * 42;
      for (int i = 0; i < 6; ++i) gh[i] += g[i];
      for (int i = 0; i < 36; ++i) gh[6 + i] += h[i];
    }
  }
}

// Optimizer::AddProjectedGradientsAndHessians (optimizer.cpp:309-321) + the Tikhonov diagonal, by the whole workgroup,
// straight into the system of the solve (w.A lower triangle, w.b).  A wave per link: H J, then the link's terms
// J^T g and J^T (H J) -- 6-term dots, the expressions of tree_project -- ; then one thread per element adds the links
// one after the other.  Needs w.J (tree_kin_jacobians) and ends with a workgroup barrier.
template <bool CONSTRAINED>
__device__ __forceinline__ void tree_system_fast(const TreeOptDev& o, const LinkDev* links, const TreeWork& w,
                                                 const float* gh_links, int tid) {
  PHASE_T0();
  const int nt = blockDim.x, wave = tid >> 6, lane = tid & (kWave - 1), n_waves = nt >> 6;
  const int dof = o.dof, n_links = o.n_links, size = o.dof + o.n_rows;
  const int n_lower = dof * (dof + 1) / 2, stride = dof + n_lower;
  const float* GH = gh_links;
  if constexpr (CONSTRAINED) {
    // (the pivoted in-place factorisation may run: everything below the diagonal is read, the constraint block too)
    for (int e = tid; e < size * size; e += nt) w.A[e] = 0.0f;
    for (int i = dof + tid; i < size; i += nt) w.b[i] = 0.0f;
    if (o.n_soft > 0) {
      GH = w.GH;
      for (int e = tid; e < n_links * 42; e += nt) w.GH[e] = gh_links[e];
      __syncthreads();
      if (tid == 0) tree_soft_constraints(o.n_soft, o.soft, links, w.GH);
      __syncthreads();
    }
  }
  // (row | column << 8 of the lane's first two terms: asked for in front of the H J products, not between them and the dots)
  int rc_first[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int x = lane + q * kWave;
    rc_first[q] = (x >= dof && x < stride) ? w.lower[x - dof] : x;
  }
  for (int li = wave; li < n_links; li += n_waves) {
    const float* g = GH + (size_t)li * 42;
    const float* H = g + 6;
    const float* J = w.J + (size_t)li * 6 * dof;
    float* HJ = w.HJ + (size_t)li * 6 * dof;
    for (int e = lane; e < 6 * dof; e += kWave) {
      const int c = e / 6, r = e - c * 6;
      float sacc = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k) sacc += H[k * 6 + r] * J[c * 6 + k];
      HJ[e] = sacc;
    }
    tree_wave_sync();
    // J^T g (x < dof: row x of J^T with g) and J^T (H J) (row r with column c of H J): ONE six-term dot for both kinds
    for (int x = lane, q = 0; x < stride; x += kWave, ++q) {
      const int rc = q < 2 ? (q == 0 ? rc_first[0] : rc_first[1]) : (x >= dof ? w.lower[x - dof] : x);
      const int r = rc & 255, c = rc >> 8;
      const float* pa = J + r * 6;
      const float* pb = x < dof ? g : HJ + c * 6;
      float sacc = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k) sacc += pa[k] * pb[k];
      w.terms[(size_t)li * stride + x] = sacc;
    }
  }
  PHASE_MARK(19);
  // one thread per element adds the links' terms one after the other: b = sum, A = 0 - term - term ... (x - t is
  // x + (-t) bit for bit, so both kinds of element run ONE loop), four terms asked for at a time; where the element
  // goes and its Tikhonov value do not depend on the terms: fetched in front of the barrier
  const int x0 = tid < stride ? tid : 0;
  const bool is_b = x0 < dof;
  int r0 = 0, c0 = 0;
  float tikhonov = 0.0f;
  if (!is_b) {
    const int rc = w.lower[x0 - dof];
    r0 = rc & 255; c0 = rc >> 8;
    if (r0 == c0) tikhonov = w.tikhonov[c0];
  }
  __syncthreads();
  PHASE_MARK(20);
  for (int x = tid; x < stride; x += nt) {
    int r = r0, c = c0;
    float tik = tikhonov;
    if (x != x0) {  // (more elements than threads: not the fused kernels' shapes)
      r = c = 0; tik = 0.0f;
      if (x >= dof) {
        const int rc = w.lower[x - dof];
        r = rc & 255; c = rc >> 8;
        if (r == c) tik = w.tikhonov[c];
      }
    }
    const bool to_b = x < dof;
    float acc = 0.0f;
    const int last = n_links - 1;
#pragma nounroll
    for (int l0 = 0; l0 < n_links; l0 += 4) {
      float t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = w.terms[(size_t)(l0 + j < last ? l0 + j : last) * stride + x];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (l0 + j < n_links) acc += to_b ? t[j] : -t[j];  // uniform
    }
    if (to_b) {
      w.b[x] = acc;
    } else {
      if (r == c) acc += tik;
      w.A[(size_t)c * size + r] = acc;
      if constexpr (!CONSTRAINED) w.A[(size_t)r * size + c] = acc;  // (ldlt_solve_rows<FULL> gathers from either triangle)
    }
  }
  __syncthreads();
}

// The rest of Optimizer::CalculateOptimization + Link::UpdatePoses (link.cpp:205-241) on the workgroup's FIRST wave:
// constraint rows (CONSTRAINED), LDL^T, NaN guard, exp() with four lanes per link, the joints, link2world down the
// tree.  w.A / w.b hold the assembled system.  False when the NaN guard skipped the update.
template <bool CONSTRAINED>
__device__ __forceinline__ bool tree_solve_fast(const TreeOptDev& o, LinkDev* links, const TreeWork& w, int tid) {
  const int lane = tid & (kWave - 1), dof = o.dof, n_links = o.n_links, size = o.dof + o.n_rows;
  PHASE_T0();
  float* A = w.A;
  float* b = w.b;
  if constexpr (CONSTRAINED) {  // Constraint::CalculateResidualAndConstraintJacobian constraint.cpp:81-102
    int idx = dof;
    for (int ci = 0; ci < o.n_constraints; ++ci) {
      const ConstraintDev& c = o.constraints[ci];
      const int n_c = c.n;
      if (lane == 0) {
        Affine b12j1 = load_pose(c.body12joint1);
        Affine body22joint1 = mul_pose(mul_pose(b12j1, inverse_pose(load_pose(links[c.link1].link2world))),
                                       load_pose(links[c.link2].link2world));
        Affine joint22joint1 = mul_pose(body22joint1, inverse_pose(load_pose(c.body22joint2)));
        float angle, axis[3];
        angle_axis(joint22joint1.l, &angle, axis);
        float rv[3] = {angle * axis[0], angle * axis[1], angle * axis[2]};
        int ri = 0;
        for (int d = 0; d < 6; ++d)
          if (c.directions[d]) w.cres[ri++] = d < 3 ? rv[d] : joint22joint1.t[d - 3];
        constraint_unprojected_jacobian(c, joint22joint1, body22joint1, w.j2);
        constraint_unprojected_jacobian(c, joint22joint1, b12j1, w.j1);
      }
      tree_wave_sync();
      const float* J1 = w.J + (size_t)c.link1 * 6 * dof;
      const float* J2 = w.J + (size_t)c.link2 * 6 * dof;
      for (int e = lane; e < dof * n_c; e += kWave) {
        const int col = e / n_c, r = e - col * n_c;
        float s2 = 0.0f, s1 = 0.0f;
        for (int k = 0; k < 6; ++k) {
          s2 += w.j2[k * n_c + r] * J2[(size_t)col * 6 + k];
          s1 += w.j1[k * n_c + r] * J1[(size_t)col * 6 + k];
        }
        A[(size_t)col * size + idx + r] = -(s2 - s1);  // AddResidualsAndConstraintJacobians optimizer.cpp:323-333
      }
      if (lane < n_c) b[idx + lane] = w.cres[lane];
      tree_wave_sync();
      idx += n_c;
    }
  }
  PHASE_MARK(12);
  ldlt_solve_any<true, !CONSTRAINED>(A, b, size, w.temp, w.trans, lane);  // (tree_system_fast: both triangles)
  PHASE_MARK(13);
  {  // NaN guard optimizer.cpp:165 (size <= M3T_TREE_FUSED_MAX_SIZE = the wave: one element per lane)
    const float bl = b[lane < size ? lane : 0];
    if (__builtin_amdgcn_ballot_w64(bl != bl) != 0) return false;
  }
  // the variations of all links: four lanes per link, lane c < 3 of the group works out column c of
  // exp(skew(theta_r)) (colexpm3: the oracle's Expm3 operation for operation), lane 3 carries the translation
  float* var_all = w.AD;  // [n_links][12]: l[9] | t[3]
  {
    const int li = lane >> 2, c = lane & 3;
    if (li < n_links) {
      const int first = w.lanes[TL_QUAD_FIRST * kWave + lane], mask = w.lanes[TL_QUAD_MASK * kWave + lane];
      float th[6];
#pragma unroll
      for (int d = 0; d < 6; ++d) {
        const int j = first + __builtin_popcount(mask & ((1 << d) - 1));
        const float t = b[(mask >> d) & 1 ? j : 0];
        th[d] = (mask >> d) & 1 ? t : 0.0f;
      }
      float K[9], col[3];
      K[0] = 0.0f;   K[3] = -th[2]; K[6] = th[1];
      K[1] = th[2];  K[4] = 0.0f;   K[7] = -th[0];
      K[2] = -th[1]; K[5] = th[0];  K[8] = 0.0f;
      colexpm3<true>(K, c, col);
      float* v = var_all + (size_t)li * 12;
      if (c == 3) { col[0] = th[3]; col[1] = th[4]; col[2] = th[5]; }
      v[c * 3] = col[0]; v[c * 3 + 1] = col[1]; v[c * 3 + 2] = col[2];
    }
  }
  tree_wave_sync();
  PHASE_MARK(14);
  // out(k, c) = (P(k,0) b0 + P(k,1) b1) + P(k,2) b2 [+ P(k,3) for the translation column], b = column c of the right
  // factor: mul_pose's expression for element (k, c), the row's four values by DPP quad broadcasts
  const int qc = lane & 3;
  auto quadmul = [&](float P, float b0, float b1, float b2) {
    const float p0 = quad_lane<0>(P), p1 = quad_lane<1>(P), p2 = quad_lane<2>(P), p3 = quad_lane<3>(P);
    const float v = (p0 * b0 + p1 * b1) + p2 * b2;
    return qc == 3 ? v + p3 : v;
  };
  // the joints: joint2parent <- joint2parent * variation (fixed body2joint), or body2joint <- variation * body2joint;
  // twelve lanes per link (element (k, c) in lane 4 k + c of a row of sixteen), four links at a time
  // (which matrix a row of lanes updates comes from the lane table: no trip to the link's parent / fixed fields first)
  for (int pass = 0; pass * 4 < n_links; ++pass) {
    const int code = w.lanes[(TL_JOINT + pass) * kWave + lane];  // (offset of the matrix) * 2 + fixed, or -1
    if (code >= 0) {  // (the same in all lanes of a row)
      const int li = pass * 4 + (lane >> 4), k = (lane >> 2) & 3, kk = k < 3 ? k : 0;
      const float* v = var_all + (size_t)li * 12;
      const bool fixed = (code & 1) != 0;
      float* M = reinterpret_cast<float*>(links) + (code >> 1);
      const float m_own = M[qc * 4 + kk], m0 = M[qc * 4], m1 = M[qc * 4 + 1], m2 = M[qc * 4 + 2];
      const float v_own = v[qc * 3 + kk], v0 = v[qc * 3], v1 = v[qc * 3 + 1], v2 = v[qc * 3 + 2];
      const float r = fixed ? quadmul(m_own, v0, v1, v2) : quadmul(v_own, m0, m1, m2);
      M[qc * 4 + k] = k < 3 ? r : (qc == 3 ? 1.0f : 0.0f);
    }
  }
  tree_wave_sync();
  PHASE_MARK(21);
  // link2world, parents before children: (parent's link2world * joint2parent) * body2joint, left to right like the
  // reference.  Sixteen lanes hold a pose (element (k, c) in lane 4 k + c; the fourth group of four stores the constant
  // bottom row); a child whose parent was the previous link takes the parent's pose from the registers it was just
  // formed in.  The roots first (they depend on nothing but themselves), then the other links four at a time without a
  // divergent branch: parents from the lane table, the four links' joint poses loaded before the first product, the four
  // stores behind the last.
  {
    const int k3 = lane >> 2, k = k3 < 3 ? k3 : 0, c = qc;
    const int parents = w.lanes[TL_PARENT * kWave + lane];  // lane li: the parent of link li
    float cur = 0.0f;
    int cur_link = -2;
    const float bottom = c == 3 ? 1.0f : 0.0f;
#pragma nounroll
    for (int li = 0; li < n_links; ++li) {
      if (__builtin_amdgcn_readlane(parents, li) >= 0) continue;  // uniform
      // a root: link2world <- ((link2world * body2joint^-1) * variation) * body2joint (link.cpp:226-230)
      LinkDev& l = links[li];
      const float* v = var_all + (size_t)li * 12;
      const float* ri = w.root_inverse + (size_t)li * 12 + c * 3;  // column c of body2joint^-1 (tree_tables)
      const float i0 = ri[0], i1 = ri[1], i2 = ri[2];
      const float T = l.link2world[c * 4 + k];
      const float g0 = l.body2joint[c * 4], g1 = l.body2joint[c * 4 + 1], g2 = l.body2joint[c * 4 + 2];
      const float result = quadmul(quadmul(quadmul(T, i0, i1, i2), v[c * 3], v[c * 3 + 1], v[c * 3 + 2]), g0, g1, g2);
      if (lane < 16) l.link2world[c * 4 + k3] = k3 < 3 ? result : bottom;
      cur = result;
      cur_link = li;
    }
    if (lane < 16) {
      // Four links at a time: their joint poses are all asked for first, then four products in straight-line code (the
      // LDS answers in order: the compiler's wait in front of a product is a count that leaves the later links' loads
      // in flight; behind the uniform branches of a rolled, software-pipelined loop it drained the queue per link).
      // (a column of a pose is sixteen bytes at an eight-byte boundary of the link table: one ds_read2_b64)
      typedef float Column __attribute__((ext_vector_type(4), aligned(8)));
      typedef const volatile __attribute__((address_space(3))) Column* LdsVC;
      static_assert(offsetof(LinkDev, joint2parent) % 8 == 0 && offsetof(LinkDev, body2joint) % 8 == 0 && sizeof(LinkDev) % 8 == 0,
                    "pose columns of the link table at eight-byte boundaries");
      const int last = n_links - 1;
#pragma nounroll
      for (int base = 0; base < n_links; base += 4) {
        float J[4][3], B[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int li = base + j < last ? base + j : last;
          const Column cj = *(LdsVC)(links[li].joint2parent + c * 4);
          const Column cb = *(LdsVC)(links[li].body2joint + c * 4);
          J[j][0] = cj.x; J[j][1] = cj.y; J[j][2] = cj.z;
          B[j][0] = cb.x; B[j][1] = cb.y; B[j][2] = cb.z;
        }
        __builtin_amdgcn_sched_barrier(0);
        // (the four poses stay in registers until all four are formed -- a store in front of the next product would be
        // waited for with it; a parent formed in this group that is not the previous link comes from those registers)
        float res[4];
        int formed = 0;  // bit j: res[j] holds link base + j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int li = base + j;
          const int p = li < n_links ? __builtin_amdgcn_readlane(parents, li) : -1;
          res[j] = 0.0f;
          if (p >= 0) {  // uniform
            float P = cur;
            if (p != cur_link) {
              if (p >= base && ((formed >> (p - base)) & 1)) {
                P = res[0];
#pragma unroll
                for (int i = 1; i < j; ++i) P = p == base + i ? res[i] : P;
              } else {
                P = links[p].link2world[c * 4 + k];
              }
            }
            const float result = quadmul(quadmul(P, J[j][0], J[j][1], J[j][2]), B[j][0], B[j][1], B[j][2]);
            res[j] = result;
            formed |= 1 << j;
            cur = result;
            cur_link = li;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((formed >> j) & 1) links[base + j].link2world[c * 4 + k3] = k3 < 3 ? res[j] : bottom;
      }
    }
    tree_wave_sync();
  }
  PHASE_MARK(15);
  return true;
}

// ---------------------------------------------------------------------------
// Tracker::ExecuteTrackingStep (tracker.cpp:344-364) for kinematic structures in ONE launch: one workgroup per link
// that carries modalities runs that body's correspondence searches, products and reference-order sums (the device
// functions of tracking_step_kernel); per Newton step the workgroups of a structure hand each other their link's 42
// sums as {tag, value} granules (the exchange of tracking_step_split_kernel), after which EVERY workgroup holds all
// link sums and its first wave runs Optimizer::CalculateOptimization on its own LDS copy of the structure -- same
// inputs, same operations, same poses in all of them, so nothing has to be sent back.  Replaces 50 dependent
// launches per frame (8-body chain, 7 x 2 iterations).  All workgroups of the grid must be resident (checked by the
// host); a wait that runs out abandons the step and raises the context's abort word like the split kernel does.
// ---------------------------------------------------------------------------
// SPLIT (round 5; tracking_step_tree_split_kernel): the bodies of a structure get the split kernel's treatment -- n_parts
// workgroups per tracked link, each walks the pixels and builds the distributions of its part of the link's lines
// (scans the depth windows of its part of the points), the parts hand each other their results once per
// correspondence iteration (split_exchange_publish / _collect, m3t_kernels.hip), after which every part holds the whole
// link state, forms the same sums in the same order and -- with the other links' sums, which the link's first part
// publishes for everybody -- solves the same system.  Workgroup b works for tracked link b mod n_steps, part
// b / n_steps: with eight tracked links a link's parts share an XCD.
template <bool CONSTRAINED, bool SPLIT = false>
__device__ __forceinline__ void tree_step_body(const TreeStepDev* steps, const TreeOptDev* opts, const RegionModDev* rmods,
                                               const DepthModDev* dmods, const CameraDev* cams, float* body_poses,
                                               TrackLdsLayout layout, int off_points, int np, int off_tree, int iteration,
                                               int n_corr_iterations, int n_update_iterations, int fuse_histogram,
                                               TreeStepParams xp, const SplitParams* split = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds_tree[];
  // the workgroup's and the structure's parameters through the constant address space: scalar loads, the values in
  // SGPRs for the whole launch (round 4 read them with per-lane loads: every field a VGPR alive across the search loop
  // -- 119 spill stores in front of it)
  typedef const __attribute__((address_space(4))) TreeStepDev CTreeStep;
  typedef const __attribute__((address_space(4))) TreeOptDev CTreeOpt;
  int step_index = blockIdx.x, part = 0, n_parts = 1;
  if constexpr (SPLIT) {
    n_parts = split->n_parts;
    const int n_steps = (int)gridDim.x / n_parts;
    part = (int)blockIdx.x / n_steps;
    step_index = (int)blockIdx.x - part * n_steps;
  }
  CTreeStep& stc = *(CTreeStep*)(steps + step_index);
  TreeStepDev st;
  st.opt = stc.opt; st.link = stc.link; st.tracked = stc.tracked;
  st.region_modality = stc.region_modality; st.depth_modality = stc.depth_modality; st.region_first = stc.region_first;
  CTreeOpt& oc = *(CTreeOpt*)(opts + st.opt);
  TreeOptDev o;
  o.n_links = oc.n_links; o.links = oc.links; o.dof = oc.dof;
  o.n_constraints = oc.n_constraints; o.constraints = oc.constraints; o.n_rows = oc.n_rows;
  o.n_soft = oc.n_soft; o.soft = oc.soft;
  o.tikhonov_rotation = oc.tikhonov_rotation; o.tikhonov_translation = oc.tikhonov_translation;
  o.work = oc.work; o.partial = oc.partial;
  o.n_tracked = oc.n_tracked; o.tracked_links = oc.tracked_links; o.exchange = oc.exchange;
  if constexpr (!CONSTRAINED) { o.n_constraints = 0; o.n_rows = 0; o.n_soft = 0; }
  CRegion* rm = st.region_modality >= 0 ? (CRegion*)(rmods + st.region_modality) : nullptr;
  CDepth* dm = st.depth_modality >= 0 ? (CDepth*)(dmods + st.depth_modality) : nullptr;
  const int tid = threadIdx.x, nt = blockDim.x, n_links = o.n_links, dof = o.dof;
  // the last wave has no correspondence line to form products for (216 slots at most): it keeps the structure's
  // kinematics; the first runs the chain and the solve
  const int kin_first = nt - kWave;
  const bool kin_wave = tid >= kin_first;
  Lds s = carve(lds_tree, layout);
  float* ps = lds_tree + off_points;
  float* rows_r = lds_tree + layout.off_rows_r;
  float* rows_d = lds_tree + layout.off_rows_d;
  // the structure: link table | link sums [n_links][42] | [dof x dof | dof] | work arrays
  LinkDev* links = reinterpret_cast<LinkDev*>(lds_tree + off_tree);
  float* gh_links = reinterpret_cast<float*>(links + n_links);
  float* partial = gh_links + n_links * 42;
  TreeWork w = tree_carve(partial + dof * dof + dof, n_links, dof, o.n_rows);
  // The LDS carve-up is formed anew -- from offsets that went through an empty asm -- at the top of every
  // correspondence iteration and every Newton step: left alone, the compiler works the per-thread LDS addresses of all
  // phases out once, in front of the loops, keeps them alive across the loops and spills them (the same reason as for
  // `ltid` below)
  auto recarve = [&]() {
    TrackLdsLayout L = layout;
    int op = off_points, ot = off_tree;
    asm volatile("" : "+s"(L.off_state), "+s"(L.off_chain), "+s"(L.off_seg_f), "+s"(L.off_seg_b), "+s"(L.off_misc),
                      "+s"(L.off_rows_r), "+s"(L.off_rows_d), "+s"(op), "+s"(ot));
    s = carve(lds_tree, L);
    ps = lds_tree + op;
    rows_r = lds_tree + L.off_rows_r;
    rows_d = lds_tree + L.off_rows_d;
    links = reinterpret_cast<LinkDev*>(lds_tree + ot);
    gh_links = reinterpret_cast<float*>(links + n_links);
    partial = gh_links + n_links * 42;
    w = tree_carve(partial + dof * dof + dof, n_links, dof, o.n_rows);
  };
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(o.links);
    uint32_t* dst = reinterpret_cast<uint32_t*>(links);
    for (int i = tid; i < n_links * (int)(sizeof(LinkDev) / 4); i += nt) dst[i] = src[i];
    for (int i = tid; i < n_links * 42; i += nt) gh_links[i] = 0.0f;  // links without modalities add nothing
  }
  if (rm) stage_log_table(s.misc);
  __syncthreads();
  for (int i = tid; i < n_links * 16; i += nt) {  // a link with a body stands where its body stands (link.cpp:296-301)
    const int li = i >> 4;
    if (links[li].body >= 0) links[li].link2world[i & 15] = body_poses[16 * links[li].body + (i & 15)];
  }
  tree_tables(o, links, w);
  __syncthreads();
  float* pose = links[st.link].link2world;  // this workgroup's body2world, kept current by its own solve (re-formed by recarve)
  CCam* cam = rm ? (CCam*)(cams + rm->camera) : nullptr;
  CCam* rdcam = (rm && rm->measure_occlusions) ? (CCam*)(cams + rm->depth_camera) : nullptr;
  CCam* dcam = dm ? (CCam*)(cams + dm->camera) : nullptr;
  auto* granules = (__attribute__((address_space(1))) unsigned long long*)o.exchange;
  auto* abort_word = (__attribute__((address_space(1))) unsigned*)(o.exchange + 2 * (size_t)o.n_tracked * M3T_TREE_GRANULES);
  int line_lo = 0, line_hi = 1 << 30, pt_lo = 0, pt_hi = 1 << 30;
  SplitExchange exchange{};
  if constexpr (SPLIT) {  // (as in tracking_step_body: the link's step index is the "object" of the granule buffer)
    line_lo = part * split->per_part_lines;
    line_hi = line_lo + split->per_part_lines;
    pt_lo = part * split->per_part_points;
    pt_hi = pt_lo + split->per_part_points;
    exchange.granules = (__attribute__((address_space(1))) unsigned long long*)split->granules +
                        ((size_t)step_index * 2 * n_parts << (kExchangeFieldBits + split->lshift));
    exchange.object_abort = (__attribute__((address_space(1))) unsigned*)split->object_abort + step_index;
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
    exchange.n_depth_fields = PS_VALID + 1 - PS_CORR_X;
    exchange.first_depth_row = PS_CORR_X;
  }
  int round = 0;
  int region_view = rm ? *as_global(rm->last_view) : -1;  // the view of the modality's previous search (closest_view_local)
  for (int c = 0; c < n_corr_iterations; ++c) {
    {
      recarve();
      pose = links[st.link].link2world;
      const Affine b2w = load_pose(pose);
      bool vote_deferred = false;  // decided by region_correspondences (the one predicate for all parts)
      if (rm) {
        const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
        Affine b2dc = b2c;
        if (rdcam) b2dc = mul_pose(load_pose(rdcam->world2camera), b2w);
        region_view = region_correspondences<false, SPLIT ? 2 : 8, true, true>(
            *rm, *cam, rdcam, b2c, b2dc, iteration, c, s, line_lo, line_hi, SPLIT ? &vote_deferred : nullptr, region_view,
            SPLIT ? &exchange : nullptr);
        if constexpr (!SPLIT) region_moments(*rm, s);
      }
      if (dm) {
        const Affine b2c = mul_pose(load_pose(dcam->world2camera), b2w);
        depth_correspondences_scan(*dm, *dcam, b2c, iteration, c, ps, np, s.misc, pt_lo, pt_hi,
                                   (rm && dm->view_search_shared) ? region_view : -1);
      }
      if constexpr (SPLIT) {
        // publish the own part's results, take the moments of the own lines while the other parts' results are on
        // their way, collect them, then the moments of the received lines (tracking_step_body, SPLIT)
        if (rm) {
          exchange.n_region_fields = rm->distribution_length + (vote_deferred ? 1 : 0);
          exchange.first_region_row = vote_deferred ? LS_VALID : LS_DIST0;
        }
        split_exchange_publish(exchange, c, s, rm != nullptr, ps, np, dm != nullptr,
                               (rm && !vote_deferred) ? rm->distribution_length : 0);
        if (rm && !vote_deferred) region_moments(*rm, s, line_lo, line_hi, true);
        if (!split_exchange_collect(exchange, c, s, rm != nullptr, ps, np, dm != nullptr)) return;
        if (vote_deferred) {
          region_finish_flags(*rm, s);
          region_moments(*rm, s);
        } else if (rm) {
          region_moments(*rm, s, line_lo, line_hi, false);
        }
      }
      if (dm) {
        depth_correspondences_vote<true, true>(*dm, iteration, ps, np, s.misc);
      } else {
        __syncthreads();
      }
    }
    for (int u = 0; u < n_update_iterations; ++u, ++round) {
      PHASE_T0();
      recarve();
      pose = links[st.link].link2world;
      const Affine b2w = load_pose(pose);
      // The structure code addresses LDS by thread: left alone, the compiler works all those addresses out once, in
      // front of the search loop, keeps them alive across it and spills them (round 4: 119 stores there, the reloads
      // on the first wave's serial path).  An opaque copy of the thread index per Newton step keeps them where they
      // are used.
      int ltid = tid;
      asm volatile("" : "+v"(ltid));
      TreeKin kin;
      if (kin_wave) {
        // adjoints of the structure (they depend on the joint poses only) beside the other waves' products
        tree_kin_adjoints(o, links, w, kin, ltid);
        PHASE_MARK_BY(17, kin_first);
      }
      if (rm) {
        const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
        region_products(*rm, *cam, b2c, c, u, s, rows_r, layout.pitch_r);
      }
      if (dm) {
        const Affine b2c = mul_pose(load_pose(dcam->world2camera), b2w);
        depth_products(*dm, b2c, c, ps, np, rows_d, layout.pitch_d);
      }
      __syncthreads();
      PHASE_MARK(5);
      const uint32_t tag = xp.seq * 64u + (uint32_t)round + 1u;
      auto* slot = granules + (size_t)(round & 1) * o.n_tracked * M3T_TREE_GRANULES;
      bool timed_out = false;
      if (kin_wave) {
        // ... and the Jacobians, a column per lane, while the first wave runs down this link's sums and the other
        // links' sums are on their way
        tree_kin_jacobians(o, w, kin, ltid);
        PHASE_MARK_BY(18, kin_first);
      } else {
        if (tid < kWave) {  // one wave: the sums in the reference's order, the link's sum, publish
          float sum_r = 0.0f, sum_d = 0.0f;
          chain_sums(rm ? rows_r : nullptr, layout.pitch_r, chain_slots(s.nl), dm ? rows_d : nullptr, layout.pitch_d,
                     chain_slots(np), gh_lane_row(ltid < 42 ? ltid : 0), sum_r, sum_d);
          float gh = 0.0f;  // Link::CalculateGradientAndHessian link.cpp:184-193, in the order of Link::modalities
          if (rm && dm && !st.region_first) { gh += sum_d; gh += sum_r; }
          else { if (rm) gh += sum_r; if (dm) gh += sum_d; }
          if (tid < 42) {
            gh_links[st.link * 42 + ltid] = gh;
            if (part == 0)  // (every part of the link holds these sums: the first one hands them to the other links)
              __hip_atomic_store(slot + (size_t)st.tracked * M3T_TREE_GRANULES + ltid,
                                 (static_cast<unsigned long long>(tag) << 32) | (unsigned)__float_as_int(gh),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          PHASE_MARK(23);
        }
        // collect the other tracked links' sums (one granule per thread and trip, re-read until its tag matches)
        for (int idx = ltid; idx < o.n_tracked * 42; idx += kin_first) {
          const int t = idx / 42, i = idx - t * 42;
          if (t == st.tracked) continue;
          auto* g = slot + (size_t)t * M3T_TREE_GRANULES + i;
          const int dst_link = o.tracked_links[t];  // (a global load: on its way while the granule is awaited, not after)
          unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned spins = 0;
          while (static_cast<uint32_t>(v >> 32) != tag) {
            if (++spins > (1u << 12) ||
                ((spins & 255u) == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == xp.seq)) {
              timed_out = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          gh_links[dst_link * 42 + i] = __int_as_float(static_cast<int>(static_cast<uint32_t>(v)));
        }
      }
      if (__syncthreads_or(timed_out ? 1 : 0)) {
        if (tid == 0) {
          __hip_atomic_store(abort_word, xp.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(xp.host_abort, xp.abort_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
      }
      PHASE_MARK(22);
      // Optimizer::CalculateOptimization + UpdatePoses on this workgroup's copy of the structure: the sums by all
      // waves, the solve and the pose updates -- short dependent stages -- by the first
      tree_system_fast<CONSTRAINED>(o, links, w, gh_links, ltid);
      PHASE_MARK(30);
      if (tid < kWave) (void)tree_solve_fast<CONSTRAINED>(o, links, w, ltid);
      __syncthreads();
      PHASE_MARK(31);
    }
  }
  recarve();
  pose = links[st.link].link2world;
  if (rm && tid == 0 && part == 0) *as_global_w(rm->last_view) = region_view;
  // every workgroup of a structure holds the same link table: the first one writes it (and the bodies) back
  if (st.tracked == 0 && part == 0) {
    for (int i = tid; i < n_links * 48; i += nt) {
      const int li = i / 48, k = i - li * 48;
      float* dst = k < 16 ? o.links[li].body2joint : (k < 32 ? o.links[li].joint2parent : o.links[li].link2world);
      const float* src = k < 16 ? links[li].body2joint : (k < 32 ? links[li].joint2parent : links[li].link2world);
      dst[k & 15] = src[k & 15];
    }
    for (int i = tid; i < n_links * 16; i += nt) {
      const int li = i >> 4;
      if (links[li].body >= 0) body_poses[16 * links[li].body + (i & 15)] = links[li].link2world[i & 15];
    }
  }
  if (fuse_histogram && rm) {  // RegionModality::CalculateResults :572-583 in the same launch
    PHASE_T0();
    const Affine b2w = load_pose(pose);
    __syncthreads();
    const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
    Affine b2dc = b2c;
    if (rdcam) b2dc = mul_pose(load_pose(rdcam->world2camera), b2w);
    const bool handle_occlusions = (iteration - rm->first_iteration) >= rm->n_unoccluded_iterations;
    // (the parts of a split link take an equal share of the bins each: n_bins^3 is a multiple of 64)
    const int n_bins3 = rm->n_bins * rm->n_bins * rm->n_bins;
    const int bin_lo = SPLIT ? part * (n_bins3 / n_parts) : 0;
    const int bin_hi = SPLIT ? bin_lo + n_bins3 / n_parts : -1;
    region_histogram_update(*rm, *cam, rdcam, b2c, b2dc, handle_occlusions, false,
                            (__attribute__((address_space(3))) uint32_t*)(lds_tree + M3T_MISC_FLOATS), lds_tree, bin_lo, bin_hi,
                            nullptr, 0, 0, region_view);
    PHASE_MARK(26);
  }
}

// ---------------------------------------------------------------------------
// A structure spread over processes, round 5: ONE launch and ONE all-reduce per Newton step instead of
// the per-sub-step launches (correspondences, g/H per modality, link sums, all-reduce, project + solve: 50 launches per
// frame of the 8-body chain, 0.70 ms per step before any transport).  Launch k = [solve: every workgroup applies the
// summed link sums of Newton step k - 1 to its own LDS copy of the structure, exactly what tracking_step_tree_kernel
// does after its in-kernel exchange] -> [the search of a new correspondence iteration, or its line state from global
// memory] -> [products, the link's sums in the reference's order -> sums_out]; the host all-reduces sums_out on the
// stream and launches k + 1.  Links whose modalities live on another rank contribute zeros (written by the structure's
// first workgroup), so the sum is exact and N ranks compute the poses of one process bit for bit, as before.
// No workgroup waits for another one inside a launch: the grid needs no co-residency.  Two copies of the link table
// and of the sums take turns, so that a workgroup that starts late never reads what a faster one of the same launch
// has already written.
// ---------------------------------------------------------------------------
template <bool CONSTRAINED>
__device__ __forceinline__ void tree_segment_body(const TreeStepDev* steps, const TreeOptDev* opts, const RegionModDev* rmods,
                                                  const DepthModDev* dmods, const CameraDev* cams, float* body_poses,
                                                  TrackLdsLayout layout, int off_points, int np, int off_tree,
                                                  int iteration, int fuse_histogram, TreeSegmentParams sp) {
  extern __shared__ __attribute__((aligned(16))) float lds_tree[];
  typedef const __attribute__((address_space(4))) TreeStepDev CTreeStep;
  typedef const __attribute__((address_space(4))) TreeOptDev CTreeOpt;
  CTreeStep& stc = *(CTreeStep*)(steps + blockIdx.x);
  TreeStepDev st;
  st.opt = stc.opt; st.link = stc.link; st.tracked = stc.tracked;
  st.region_modality = stc.region_modality; st.depth_modality = stc.depth_modality; st.region_first = stc.region_first;
  CTreeOpt& oc = *(CTreeOpt*)(opts + st.opt);
  TreeOptDev o;
  o.n_links = oc.n_links; o.links = oc.links; o.dof = oc.dof;
  o.n_constraints = oc.n_constraints; o.constraints = oc.constraints; o.n_rows = oc.n_rows;
  o.n_soft = oc.n_soft; o.soft = oc.soft;
  o.tikhonov_rotation = oc.tikhonov_rotation; o.tikhonov_translation = oc.tikhonov_translation;
  o.work = oc.work; o.partial = oc.partial;
  o.n_tracked = oc.n_tracked; o.tracked_links = oc.tracked_links; o.exchange = oc.exchange;
  o.links_alt = oc.links_alt; o.first_link = oc.first_link;
  if constexpr (!CONSTRAINED) { o.n_constraints = 0; o.n_rows = 0; o.n_soft = 0; }
  CRegion* rm = st.region_modality >= 0 ? (CRegion*)(rmods + st.region_modality) : nullptr;
  CDepth* dm = st.depth_modality >= 0 ? (CDepth*)(dmods + st.depth_modality) : nullptr;
  const int tid = threadIdx.x, nt = blockDim.x, n_links = o.n_links, dof = o.dof, flags = sp.flags;
  const int c = sp.corr_iteration, u = sp.opt_iteration;
  const bool kin_wave = tid >= nt - kWave;
  Lds s = carve(lds_tree, layout);
  float* ps = lds_tree + off_points;
  float* rows_r = lds_tree + layout.off_rows_r;
  float* rows_d = lds_tree + layout.off_rows_d;
  LinkDev* links = reinterpret_cast<LinkDev*>(lds_tree + off_tree);
  float* gh_links = reinterpret_cast<float*>(links + n_links);
  float* partial = gh_links + n_links * 42;
  const TreeWork w = tree_carve(partial + dof * dof + dof, n_links, dof, o.n_rows);
  LinkDev* table_in = (flags & TSEG_LINKS_FROM_ALT) ? o.links_alt : o.links;
  LinkDev* table_out = (flags & TSEG_LINKS_TO_ALT) ? o.links_alt : o.links;
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(table_in);
    uint32_t* dst = reinterpret_cast<uint32_t*>(links);
    for (int i = tid; i < n_links * (int)(sizeof(LinkDev) / 4); i += nt) dst[i] = src[i];
    if (flags & TSEG_SOLVE) {
      const float* sums = sp.sums_in + (size_t)o.first_link * 42;
      for (int i = tid; i < n_links * 42; i += nt) gh_links[i] = sums[i];
    }
  }
  if (rm) stage_log_table(s.misc);
  __syncthreads();
  if (flags & TSEG_FIRST) {
    for (int i = tid; i < n_links * 16; i += nt) {  // a link with a body stands where its body stands (link.cpp:296-301)
      const int li = i >> 4;
      if (links[li].body >= 0) links[li].link2world[i & 15] = sp.first_poses[16 * links[li].body + (i & 15)];
    }
  }
  tree_tables(o, links, w);
  __syncthreads();
  if (flags & TSEG_SOLVE) {
    // Optimizer::CalculateOptimization + UpdatePoses from the summed link sums, on this workgroup's copy
    TreeKin kin;
    if (kin_wave) {
      tree_kin_adjoints(o, links, w, kin, tid);
      tree_kin_jacobians(o, w, kin, tid);
    }
    __syncthreads();
    tree_system_fast<CONSTRAINED>(o, links, w, gh_links, tid);
    if (tid < kWave) (void)tree_solve_fast<CONSTRAINED>(o, links, w, tid);
    __syncthreads();
    if (st.tracked == 0) {  // every workgroup of the structure holds the same table: the first one writes it
      for (int i = tid; i < n_links * 48; i += nt) {
        const int li = i / 48, k = i - li * 48;
        float* dst = k < 16 ? table_out[li].body2joint : (k < 32 ? table_out[li].joint2parent : table_out[li].link2world);
        const float* src = k < 16 ? links[li].body2joint : (k < 32 ? links[li].joint2parent : links[li].link2world);
        dst[k & 15] = src[k & 15];
      }
      if (flags & TSEG_FINAL)
        for (int i = tid; i < n_links * 16; i += nt) {
          const int li = i >> 4;
          if (links[li].body >= 0) body_poses[16 * links[li].body + (i & 15)] = links[li].link2world[i & 15];
        }
    }
  }
  float* pose = links[st.link].link2world;
  CCam* cam = rm ? (CCam*)(cams + rm->camera) : nullptr;
  CCam* rdcam = (rm && rm->measure_occlusions) ? (CCam*)(cams + rm->depth_camera) : nullptr;
  CCam* dcam = dm ? (CCam*)(cams + dm->camera) : nullptr;
  int region_view = rm ? *as_global(rm->last_view) : -1;
  if (flags & TSEG_SEARCH) {
    const Affine b2w = load_pose(pose);
    if (rm) {
      const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
      Affine b2dc = b2c;
      if (rdcam) b2dc = mul_pose(load_pose(rdcam->world2camera), b2w);
      region_view = region_correspondences<false, 8, true, true>(*rm, *cam, rdcam, b2c, b2dc, iteration, c, s, 0, 1 << 30,
                                                                 nullptr, region_view);
      region_moments(*rm, s);
    }
    if (dm) {
      const Affine b2c = mul_pose(load_pose(dcam->world2camera), b2w);
      depth_correspondences_scan(*dm, *dcam, b2c, iteration, c, ps, np, s.misc, 0, 1 << 30,
                                 (rm && dm->view_search_shared) ? region_view : -1);
      depth_correspondences_vote<true, true>(*dm, iteration, ps, np, s.misc);
    } else {
      __syncthreads();
    }
    if (rm && tid == 0) *as_global_w(rm->last_view) = region_view;
    if (flags & TSEG_STORE_STATE) {  // (the layout of the unfused kernels' line / point state: compact stride)
      if (rm)
        for (int i = tid; i < LS_FIELDS * rm->n_lines_max; i += nt) {
          const int f = i / rm->n_lines_max, l = i - f * rm->n_lines_max;
          rm->line_state[i] = s.state[f * s.nl + l];
        }
      if (dm)
        for (int i = tid; i < PS_FIELDS * dm->n_points_max; i += nt) {
          const int f = i / dm->n_points_max, l = i - f * dm->n_points_max;
          dm->point_state[i] = ps[f * np + l];
        }
    }
  } else if (flags & TSEG_LOAD_STATE) {
    if (rm) {
      for (int i = tid; i < LS_FIELDS * rm->n_lines_max; i += nt) {
        const int f = i / rm->n_lines_max, l = i - f * rm->n_lines_max;
        s.state[f * s.nl + l] = rm->line_state[i];
      }
      for (int l = rm->n_lines_max + tid; l < s.nl; l += nt) s.state[LS_VALID * s.nl + l] = i2f_bits(0);
    }
    if (dm) {
      for (int i = tid; i < PS_FIELDS * dm->n_points_max; i += nt) {
        const int f = i / dm->n_points_max, l = i - f * dm->n_points_max;
        ps[f * np + l] = dm->point_state[i];
      }
      for (int l = dm->n_points_max + tid; l < np; l += nt) ps[PS_VALID * np + l] = i2f_bits(0);
    }
    __syncthreads();
  }
  if (flags & TSEG_SUMS) {
    const Affine b2w = load_pose(pose);
    if (rm) {
      const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
      region_products(*rm, *cam, b2c, c, u, s, rows_r, layout.pitch_r);
    }
    if (dm) {
      const Affine b2c = mul_pose(load_pose(dcam->world2camera), b2w);
      depth_products(*dm, b2c, c, ps, np, rows_d, layout.pitch_d);
    }
    __syncthreads();
    float* out = sp.sums_out + (size_t)o.first_link * 42;
    if (tid < kWave) {  // one wave: the sums in the reference's order, the link's sum
      float sum_r = 0.0f, sum_d = 0.0f;
      chain_sums(rm ? rows_r : nullptr, layout.pitch_r, chain_slots(s.nl), dm ? rows_d : nullptr, layout.pitch_d,
                 chain_slots(np), gh_lane_row(tid < 42 ? tid : 0), sum_r, sum_d);
      float gh = 0.0f;  // Link::CalculateGradientAndHessian link.cpp:184-193, in the order of Link::modalities
      if (rm && dm && !st.region_first) { gh += sum_d; gh += sum_r; }
      else { if (rm) gh += sum_r; if (dm) gh += sum_d; }
      if (tid < 42) out[st.link * 42 + tid] = gh;
    } else if (st.tracked == 0) {
      // links without modalities on this rank add nothing (x + 0 = x on every rank: the all-reduce is exact)
      for (int i = tid - kWave; i < n_links * 42; i += nt - kWave)
        if (links[i / 42].n_gh == 0) out[i] = 0.0f;
    }
  }
  if ((flags & TSEG_FINAL) && fuse_histogram && rm) {  // RegionModality::CalculateResults :572-583 in the same launch
    const Affine b2w = load_pose(pose);
    __syncthreads();
    const Affine b2c = mul_pose(load_pose(cam->world2camera), b2w);
    Affine b2dc = b2c;
    if (rdcam) b2dc = mul_pose(load_pose(rdcam->world2camera), b2w);
    const bool handle_occlusions = (iteration - rm->first_iteration) >= rm->n_unoccluded_iterations;
    region_histogram_update(*rm, *cam, rdcam, b2c, b2dc, handle_occlusions, false,
                            (__attribute__((address_space(3))) uint32_t*)(lds_tree + M3T_MISC_FLOATS), lds_tree, 0, -1,
                            nullptr, 0, 0, region_view);
  }
}

extern "C" {

// kinematic structures without Constraint / SoftConstraint objects (a chain, a tree): no constraint code in the kernel
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_tree_kernel(const TreeStepDev* steps, const TreeOptDev* opts, const RegionModDev* rmods,
                          const DepthModDev* dmods, const CameraDev* cams, float* body_poses, TrackLdsLayout layout,
                          int off_points, int np, int off_tree, int iteration, int n_corr_iterations,
                          int n_update_iterations, int fuse_histogram, TreeStepParams xp) {
  tree_step_body<false>(steps, opts, rmods, dmods, cams, body_poses, layout, off_points, np, off_tree, iteration,
                        n_corr_iterations, n_update_iterations, fuse_histogram, xp);
}
// ... with several workgroups per tracked link (open structures that leave CUs idle: SPLIT above)
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_tree_split_kernel(const TreeStepDev* steps, const TreeOptDev* opts, const RegionModDev* rmods,
                                const DepthModDev* dmods, const CameraDev* cams, float* body_poses, TrackLdsLayout layout,
                                int off_points, int np, int off_tree, int iteration, int n_corr_iterations,
                                int n_update_iterations, int fuse_histogram, TreeStepParams xp, SplitParams split) {
  tree_step_body<false, true>(steps, opts, rmods, dmods, cams, body_poses, layout, off_points, np, off_tree, iteration,
                              n_corr_iterations, n_update_iterations, fuse_histogram, xp, &split);
}
// ... and with them (closed chains: constraint rows, an indefinite system; soft constraints onto the link sums)
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_tree_constrained_kernel(const TreeStepDev* steps, const TreeOptDev* opts, const RegionModDev* rmods,
                                      const DepthModDev* dmods, const CameraDev* cams, float* body_poses,
                                      TrackLdsLayout layout, int off_points, int np, int off_tree, int iteration,
                                      int n_corr_iterations, int n_update_iterations, int fuse_histogram,
                                      TreeStepParams xp) {
  tree_step_body<true>(steps, opts, rmods, dmods, cams, body_poses, layout, off_points, np, off_tree, iteration,
                       n_corr_iterations, n_update_iterations, fuse_histogram, xp);
}

// (a structure spread over processes: one launch per Newton step, the all-reduce of the link sums between two launches)
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_tree_segment_kernel(const TreeStepDev* steps, const TreeOptDev* opts, const RegionModDev* rmods,
                                  const DepthModDev* dmods, const CameraDev* cams, float* body_poses, TrackLdsLayout layout,
                                  int off_points, int np, int off_tree, int iteration, int fuse_histogram,
                                  TreeSegmentParams sp) {
  tree_segment_body<false>(steps, opts, rmods, dmods, cams, body_poses, layout, off_points, np, off_tree, iteration,
                           fuse_histogram, sp);
}
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
tracking_step_tree_segment_constrained_kernel(const TreeStepDev* steps, const TreeOptDev* opts, const RegionModDev* rmods,
                                              const DepthModDev* dmods, const CameraDev* cams, float* body_poses,
                                              TrackLdsLayout layout, int off_points, int np, int off_tree, int iteration,
                                              int fuse_histogram, TreeSegmentParams sp) {
  tree_segment_body<true>(steps, opts, rmods, dmods, cams, body_poses, layout, off_points, np, off_tree, iteration,
                          fuse_histogram, sp);
}

__global__ void __launch_bounds__(64)
links_project_kernel(const TreeOptDev* opts, int n_opts, const float* body_poses, int work_in_lds) {
  extern __shared__ __attribute__((aligned(16))) float tree_lds[];
  const TreeOptDev& o = opts[blockIdx.x];
  const TreeWork w = tree_carve(work_in_lds ? tree_lds : o.work, o.n_links, o.dof, o.n_rows);
  tree_project<0>(o, o.links, w, nullptr, o.partial, o.partial + (size_t)o.dof * o.dof, body_poses);
  if (work_in_lds)  // the Jacobians are needed again by the solve kernel (constraint rows)
    for (int e = threadIdx.x; e < o.n_links * 6 * o.dof; e += kWave) o.work[e] = w.J[e];
}

__global__ void __launch_bounds__(64)
links_solve_kernel(const TreeOptDev* opts, int n_opts, float* body_poses, int zero_theta, int work_in_lds) {
  extern __shared__ __attribute__((aligned(16))) float tree_lds[];
  const TreeOptDev& o = opts[blockIdx.x];
  const TreeWork w = tree_carve(work_in_lds ? tree_lds : o.work, o.n_links, o.dof, o.n_rows);
  if (work_in_lds) {
    for (int e = threadIdx.x; e < o.n_links * 6 * o.dof; e += kWave) w.J[e] = o.work[e];
    __syncthreads();
  }
  (void)tree_solve<0>(o, o.links, w, o.partial, body_poses, zero_theta);
}

// A structure spread over processes.  What crosses between the ranks is the LINK SUMS themselves --
// Link::CalculateGradientAndHessian's 6 + 36 floats per link (link.cpp:184-193), stacked over all links of all
// structures: link_sums[first_link[structure] + link][42] -- and not the projected system: a link's modalities live on
// one rank, every other rank contributes +0.0 to its 42 numbers, so the all-reduce returns on every rank exactly the
// floats one process has (x + 0 + ... + 0 = x in any order; the sums are never -0: they start from +0).  Soft
// constraints, projection, constraint rows, Tikhonov and the solve then run on every rank as they run in one process:
// the poses are the single-process poses bit for bit for any number of ranks.  (Summing the projected [dof x dof | dof]
// blocks instead is a reassociation of the sum over the links, and the tracker's discrete decisions amplify it: the
// 8-body chain on two ranks ends 6e-3 away from one process after ONE frame.)
__global__ void __launch_bounds__(64)
links_gather_kernel(const TreeOptDev* opts, int n_opts, float* link_sums, const int* first_link) {
  const TreeOptDev& o = opts[blockIdx.x];
  float* out = link_sums + (size_t)first_link[blockIdx.x] * 42;
  for (int e = threadIdx.x; e < o.n_links * 42; e += kWave) {
    const int li = e / 42, i = e - li * 42;
    const LinkDev& l = o.links[li];
    float sacc = 0.0f;  // (the expression of tree_project)
    for (int m = 0; m < l.n_gh; ++m) sacc += l.gh[m][i];
    out[e] = sacc;
  }
}

// ... and the rest of Optimizer::CalculateOptimization from the (all-reduced) link sums: links_project_kernel and
// links_solve_kernel in one launch
__global__ void __launch_bounds__(64)
links_solve_sums_kernel(const TreeOptDev* opts, int n_opts, float* body_poses, int zero_theta, int work_in_lds,
                        const float* link_sums, const int* first_link) {
  extern __shared__ __attribute__((aligned(16))) float tree_lds[];
  const TreeOptDev& o = opts[blockIdx.x];
  const TreeWork w = tree_carve(work_in_lds ? tree_lds : o.work, o.n_links, o.dof, o.n_rows);
  tree_project<0>(o, o.links, w, link_sums + (size_t)first_link[blockIdx.x] * 42, o.partial,
                  o.partial + (size_t)o.dof * o.dof, body_poses);
  __threadfence_block();
  __syncthreads();  // (o.partial is read back by other lanes)
  (void)tree_solve<0>(o, o.links, w, o.partial, body_poses, zero_theta);
}

}  // extern "C"
