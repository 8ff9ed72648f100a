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
  int n_soft;  // soft constraints (0 while they are switched off for this process)
  SoftConstraintDev* soft;
  float tikhonov_rotation, tikhonov_translation;
  float* work;     // scratch, layout in tree_work_floats()
  float* partial;  // [dof*dof | dof]
};

namespace {

__host__ __device__ inline size_t tree_work_floats(int n_links, int dof, int n_rows) {
  size_t size = size_t(dof) + n_rows;
  return size_t(n_links) * (12 * dof + 42 + 72) + size * size + 4 * size + size_t(n_rows) * (dof + 1) + 2 * 6 * 6 + 64;
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
    *angle = 2.0f * atan2f(n, fabsf(q[3]));
    if (q[3] < 0.0f) n = -n;
    for (int c = 0; c < 3; ++c) axis[c] = q[c] / n;
  } else {
    *angle = 0.0f;
    axis[0] = 1.0f; axis[1] = 0.0f; axis[2] = 0.0f;
  }
}

__device__ float xcotx_dev(float x) {  // common.h:73-77
  if (tanf(x) <= 1.17549435e-38f) return 1.0f;
  if (tanf(x) >= 3.40282347e+38f) return 0.0f;
  return (float)((double)x / tan((double)x));
}

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
  int* trans;
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
  return t;
}

// Eigen::LDLT<MatrixXf, Lower> + solve (optimizer.cpp:162-163) by one wave: the oracle's LdltSolve, its loops over
// rows / columns spread over the lanes
__device__ void ldlt_solve_wave(float* a, float* x, int n, float* temp, int* trans) {
#define A_(r, c) a[(size_t)(c) * n + (r)]
  const int lane = threadIdx.x;
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
    __syncthreads();
    if (k > 0) {
      for (int c = lane; c < k; c += kWave) temp[c] = A_(c, c) * A_(k, c);
      __syncthreads();
      for (int i = k + lane; i < n; i += kWave) {  // i == k: the pivot; below: A21 -= A20 * temp
        float sacc = 0.0f;
        for (int c = 0; c < k; ++c) sacc += A_(i, c) * temp[c];
        A_(i, k) -= sacc;
      }
      __syncthreads();
    }
    const float akk = A_(k, k);
    const bool pivot_valid = fabsf(akk) > 0.0f;
    if (k == 0 && !pivot_valid) {
      for (int j = lane; j < n; j += kWave) trans[j] = j;
      degenerate = true;
    } else if (pivot_valid) {
      for (int i = k + 1 + lane; i < n; i += kWave) A_(i, k) /= akk;
    }
    __syncthreads();
  }
  if (lane == 0)
    for (int k = 0; k < n; ++k) { float t = x[k]; x[k] = x[trans[k]]; x[trans[k]] = t; }
  __syncthreads();
  for (int c = 0; c + 1 < n; ++c) {  // L y = P b, column sweep: row i sees its c in ascending order
    const float xc = x[c];
    for (int i = c + 1 + lane; i < n; i += kWave) x[i] -= A_(i, c) * xc;
    __syncthreads();
  }
  for (int i = lane; i < n; i += kWave) {
    if (fabsf(A_(i, i)) > 1.17549435e-38f) x[i] /= A_(i, i);
    else x[i] = 0.0f;
  }
  __syncthreads();
  for (int i = n - 1; i >= 0; --i) {  // L^T w = z: products by the lanes, the ordered subtraction by one
    for (int r = i + 1 + lane; r < n; r += kWave) temp[r] = A_(r, i) * x[r];
    __syncthreads();
    if (lane == 0) {
      float sacc = x[i];
      for (int r = i + 1; r < n; ++r) sacc -= temp[r];
      x[i] = sacc;
    }
    __syncthreads();
  }
  if (lane == 0)
    for (int k = n - 1; k >= 0; --k) { float t = x[k]; x[k] = x[trans[k]]; x[trans[k]] = t; }
  __syncthreads();
#undef A_
}

extern "C" {

// Optimizer::CalculateDataLinks (:281-296) + AddProjectedGradientsAndHessians (:309-321)
__global__ void __launch_bounds__(64)
links_project_kernel(const TreeOptDev* opts, int n_opts, const float* body_poses, int work_in_lds) {
  extern __shared__ __attribute__((aligned(16))) float tree_lds[];
  const TreeOptDev& o = opts[blockIdx.x];
  const int lane = threadIdx.x, dof = o.dof, n_links = o.n_links;
  const TreeWork w = tree_carve(work_in_lds ? tree_lds : o.work, n_links, dof, o.n_rows);
  // adjoints of every link, one lane per link (they depend on the link's own joint poses only)
  for (int li = lane; li < n_links; li += kWave) {
    const LinkDev& l = o.links[li];
    if (l.parent >= 0)
      adjoint6(inverse_pose(mul_pose(load_pose(l.joint2parent), load_pose(l.body2joint))), w.AD + (size_t)li * 72);
    adjoint6(inverse_pose(load_pose(l.body2joint)), w.AD + (size_t)li * 72 + 36);
  }
  __syncthreads();
  // Link::CalculateJacobian link.cpp:159-182, parents before children
  for (int li = 0; li < n_links; ++li) {
    const LinkDev& l = o.links[li];
    float* J = w.J + (size_t)li * 6 * dof;
    const float* ad = w.AD + (size_t)li * 72;
    const float* Jp = l.parent >= 0 ? w.J + (size_t)l.parent * 6 * dof : nullptr;
    for (int e = lane; e < 6 * dof; e += kWave) {
      const int c = e / 6, r = e - c * 6;
      float v = 0.0f;
      if (Jp) {
        float sacc = 0.0f;
        for (int k = 0; k < 6; ++k) sacc += ad[k * 6 + r] * Jp[(size_t)c * 6 + k];
        v = sacc;
      }
      J[e] = v;
    }
    __syncthreads();
    if (lane < 36) {
      const int d = lane / 6, r = lane - d * 6;
      if (l.free_directions[d]) {
        int jidx = l.first_jacobian_index;
        for (int dd = 0; dd < d; ++dd) jidx += l.free_directions[dd] ? 1 : 0;
        J[(size_t)jidx * 6 + r] = ad[36 + d * 6 + r];
      }
    }
    __syncthreads();
  }
  // Link::CalculateGradientAndHessian link.cpp:184-193
  for (int e = lane; e < n_links * 42; e += kWave) {
    const int li = e / 42, i = e - li * 42;
    const LinkDev& l = o.links[li];
    float sacc = 0.0f;
    for (int m = 0; m < l.n_gh; ++m) sacc += l.gh[m][i];
    w.GH[e] = sacc;
  }
  __syncthreads();
  // SoftConstraint::AddGradientsAndHessiansToLinks soft_constraint.cpp:113-131 (optimizer.cpp:283-284)
  if (lane == 0) {
    for (int si = 0; si < o.n_soft; ++si) {
      const SoftConstraintDev& sc = o.soft[si];
      Affine b12j1 = load_pose(sc.joint.body12joint1);
      Affine body22joint1 = mul_pose(mul_pose(b12j1, inverse_pose(link_pose(o.links[sc.joint.link1], body_poses))),
                                     link_pose(o.links[sc.joint.link2], body_poses));
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
  // H J of every link, then b = sum J^T g and A = -sum J^T (H J) (lower), link after link per element
  for (int e = lane; e < n_links * 6 * dof; e += kWave) {
    const int li = e / (6 * dof), rem = e - li * 6 * dof, c = rem / 6, r = rem - c * 6;
    const float* H = w.GH + (size_t)li * 42 + 6;
    const float* J = w.J + (size_t)li * 6 * dof;
    float sacc = 0.0f;
    for (int k = 0; k < 6; ++k) sacc += H[k * 6 + r] * J[(size_t)c * 6 + k];
    w.HJ[e] = sacc;
  }
  __syncthreads();
  float* A = o.partial;
  float* b = o.partial + (size_t)dof * dof;
  for (int i = lane; i < dof; i += kWave) {
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
  for (int e = lane; e < dof * dof; e += kWave) {
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
  if (work_in_lds)  // the Jacobians are needed again by the solve kernel (constraint rows)
    for (int e = lane; e < n_links * 6 * dof; e += kWave) o.work[e] = w.J[e];
}

// the rest of Optimizer::CalculateOptimization + Optimizer::UpdatePoses (:335-346)
__global__ void __launch_bounds__(64)
links_solve_kernel(const TreeOptDev* opts, int n_opts, float* body_poses, int zero_theta, int work_in_lds) {
  extern __shared__ __attribute__((aligned(16))) float tree_lds[];
  const TreeOptDev& o = opts[blockIdx.x];
  const int lane = threadIdx.x, dof = o.dof, n_links = o.n_links, size = o.dof + o.n_rows;
  const TreeWork w = tree_carve(work_in_lds ? tree_lds : o.work, n_links, dof, o.n_rows);
  float* A = w.A;
  float* b = w.b;
  if (work_in_lds)
    for (int e = lane; e < n_links * 6 * dof; e += kWave) w.J[e] = o.work[e];
  for (int e = lane; e < size * size; e += kWave) {
    const int c = e / size, r = e - c * size;
    A[e] = (!zero_theta && c < dof && r < dof) ? o.partial[(size_t)c * dof + r] : 0.0f;
  }
  for (int i = lane; i < size; i += kWave) b[i] = (!zero_theta && i < dof) ? o.partial[(size_t)dof * dof + i] : 0.0f;
  __syncthreads();
  if (!zero_theta) {  // zero_theta: Optimizer::CalculateConsistentPoses optimizer.cpp:135 (theta = 0)
    // constraints: Constraint::CalculateResidualAndConstraintJacobian constraint.cpp:81-102
    int idx = dof;
    for (int ci = 0; ci < o.n_constraints; ++ci) {
      const ConstraintDev& c = o.constraints[ci];
      const int n_c = c.n;
      if (lane == 0) {
        const LinkDev& l1 = o.links[c.link1];
        const LinkDev& l2 = o.links[c.link2];
        Affine b12j1 = load_pose(c.body12joint1);
        Affine body22joint1 = mul_pose(mul_pose(b12j1, inverse_pose(link_pose(l1, body_poses))), link_pose(l2, body_poses));
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
      __syncthreads();
      const float* J1 = w.J + (size_t)c.link1 * 6 * dof;
      const float* J2 = w.J + (size_t)c.link2 * 6 * dof;
      for (int e = lane; e < dof * n_c; e += kWave) {
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
      __syncthreads();
      idx += n_c;
    }
    // Tikhonov vector optimizer.cpp:252-271 (free-direction order, rotation first)
    for (int li = lane; li < n_links; li += kWave) {
      const LinkDev& l = o.links[li];
      int j = l.first_jacobian_index;
      for (int d = 0; d < 6; ++d)
        if (l.free_directions[d]) {
          A[(size_t)j * size + j] += d < 3 ? o.tikhonov_rotation : o.tikhonov_translation;
          j++;
        }
    }
    __syncthreads();
    ldlt_solve_wave(A, b, size, w.temp, w.trans);
    int has_nan = 0;
    for (int i = lane; i < size; i += kWave) has_nan |= (b[i] != b[i]) ? 1 : 0;
    if (__syncthreads_or(has_nan)) return;  // NaN guard optimizer.cpp:165
  }
  // Link::UpdatePoses link.cpp:205-241: the variations of all links at once (one lane per link) ...
  float* var_all = w.AD;  // [n_links][12]
  for (int li = lane; li < n_links; li += kWave) {
    const LinkDev& l = o.links[li];
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
  __syncthreads();
  // ... then the poses, parents before children
  if (lane == 0) {
    for (int li = 0; li < n_links; ++li) {
      LinkDev& l = o.links[li];
      const float* v = var_all + (size_t)li * 12;
      Affine var;
      for (int i = 0; i < 9; ++i) var.l[i] = v[i];
      var.t[0] = v[9]; var.t[1] = v[10]; var.t[2] = v[11];
      Affine l2w;
      if (l.parent >= 0) {
        if (l.fixed_body2joint_pose) {
          Affine j2p = mul_pose(load_pose(l.joint2parent), var);
          affine_to_array(j2p, l.joint2parent);
        } else {
          Affine b2j = mul_pose(var, load_pose(l.body2joint));
          affine_to_array(b2j, l.body2joint);
        }
        l2w = mul_pose(mul_pose(link_pose(o.links[l.parent], body_poses), load_pose(l.joint2parent)),
                       load_pose(l.body2joint));
      } else {
        Affine b2j = load_pose(l.body2joint);
        l2w = mul_pose(mul_pose(mul_pose(link_pose(l, body_poses), inverse_pose(b2j)), var), b2j);
      }
      affine_to_array(l2w, l.link2world);
      if (l.body >= 0) affine_to_array(l2w, body_poses + 16 * l.body);
    }
  }
}

}  // extern "C"
