// m3t_links.hip — kinematic structures on the device: Link::CalculateJacobian /
// CalculateGradientAndHessian / UpdatePoses (src/link.cpp:159-241), Constraint
// (src/constraint.cpp:81-102,176-274) and Optimizer::CalculateOptimization for any dof with
// constraint rows (src/optimizer.cpp:144-167, 281-346).  Included by m3t_hip_api.hip after
// m3t_kernels.hip (same translation unit, shares its pose helpers).
//
// These systems are tiny (dof <= a few dozen) and strictly sequential, so one lane per
// kinematic structure executes them; the data-parallel work (correspondences, g/H) stays in
// the modality kernels.  The optimisation is split in two kernels at the only point where a
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
  return size_t(n_links) * (6 * dof + 42) + size * size + 4 * size + size_t(n_rows) * (dof + 1) + 2 * 6 * 6 + 64;
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

// Eigen::LDLT<MatrixXf, Lower> + solve (optimizer.cpp:162-163), any size, in global scratch
__device__ void ldlt_solve_dynamic(float* a, float* x, int n, float* temp, int* trans) {
#define A_(r, c) a[(size_t)(c) * n + (r)]
  bool degenerate = false;
  for (int k = 0; k < n && !degenerate; ++k) {
    int piv = k;
    float best = fabsf(A_(k, k));
    for (int i = k + 1; i < n; ++i)
      if (fabsf(A_(i, i)) > best) { best = fabsf(A_(i, i)); piv = i; }
    trans[k] = piv;
    if (piv != k) {
      int s = n - piv - 1;
      for (int c = 0; c < k; ++c) { float t = A_(k, c); A_(k, c) = A_(piv, c); A_(piv, c) = t; }
      for (int i = 0; i < s; ++i) { float t = A_(piv + 1 + i, k); A_(piv + 1 + i, k) = A_(piv + 1 + i, piv); A_(piv + 1 + i, piv) = t; }
      { float t = A_(k, k); A_(k, k) = A_(piv, piv); A_(piv, piv) = t; }
      for (int i = k + 1; i < piv; ++i) { float t = A_(i, k); A_(i, k) = A_(piv, i); A_(piv, i) = t; }
    }
    int rs = n - k - 1;
    if (k > 0) {
      for (int c = 0; c < k; ++c) temp[c] = A_(c, c) * A_(k, c);
      float acc = 0.0f;
      for (int c = 0; c < k; ++c) acc += A_(k, c) * temp[c];
      A_(k, k) -= acc;
      for (int i = 0; i < rs; ++i) {
        float s = 0.0f;
        for (int c = 0; c < k; ++c) s += A_(k + 1 + i, c) * temp[c];
        A_(k + 1 + i, k) -= s;
      }
    }
    float akk = A_(k, k);
    bool pivot_valid = fabsf(akk) > 0.0f;
    if (k == 0 && !pivot_valid) {
      for (int j = 0; j < n; ++j) trans[j] = j;
      degenerate = true;
    } else if (rs > 0 && pivot_valid) {
      for (int i = 0; i < rs; ++i) A_(k + 1 + i, k) /= akk;
    }
  }
  for (int k = 0; k < n; ++k) { float t = x[k]; x[k] = x[trans[k]]; x[trans[k]] = t; }
  for (int i = 0; i < n; ++i) {
    float s = x[i];
    for (int c = 0; c < i; ++c) s -= A_(i, c) * x[c];
    x[i] = s;
  }
  for (int i = 0; i < n; ++i) {
    if (fabsf(A_(i, i)) > 1.17549435e-38f) x[i] /= A_(i, i);
    else x[i] = 0.0f;
  }
  for (int i = n - 1; i >= 0; --i) {
    float s = x[i];
    for (int r = i + 1; r < n; ++r) s -= A_(r, i) * x[r];
    x[i] = s;
  }
  for (int k = n - 1; k >= 0; --k) { float t = x[k]; x[k] = x[trans[k]]; x[trans[k]] = t; }
#undef A_
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

extern "C" {

// Optimizer::CalculateDataLinks (:281-296) + AddProjectedGradientsAndHessians (:309-321)
__global__ void links_project_kernel(const TreeOptDev* opts, int n_opts, const float* body_poses) {
  int oi = blockIdx.x * blockDim.x + threadIdx.x;
  if (oi >= n_opts) return;
  const TreeOptDev& o = opts[oi];
  const int dof = o.dof;
  float* jac_all = o.work;                                // [n_links][6 * dof]
  float* gh_all = jac_all + (size_t)o.n_links * 6 * dof;  // [n_links][42]
  for (int li = 0; li < o.n_links; ++li) {
    const LinkDev& l = o.links[li];
    float* J = jac_all + (size_t)li * 6 * dof;
    // Link::CalculateJacobian link.cpp:159-182
    for (int i = 0; i < 6 * dof; ++i) J[i] = 0.0f;
    float ad[36];
    if (l.parent >= 0) {
      const float* Jp = jac_all + (size_t)l.parent * 6 * dof;
      Affine parent2body = inverse_pose(mul_pose(load_pose(l.joint2parent), load_pose(l.body2joint)));
      adjoint6(parent2body, ad);
      for (int c = 0; c < dof; ++c)
        for (int r = 0; r < 6; ++r) {
          float s = 0.0f;
          for (int k = 0; k < 6; ++k) s += ad[k * 6 + r] * Jp[(size_t)c * 6 + k];
          J[(size_t)c * 6 + r] = s;
        }
    }
    adjoint6(inverse_pose(load_pose(l.body2joint)), ad);
    int jidx = l.first_jacobian_index;
    for (int d = 0; d < 6; ++d)
      if (l.free_directions[d]) {
        for (int r = 0; r < 6; ++r) J[(size_t)jidx * 6 + r] = ad[d * 6 + r];
        jidx++;
      }
    // Link::CalculateGradientAndHessian link.cpp:184-193
    float* gh = gh_all + (size_t)li * 42;
    for (int i = 0; i < 42; ++i) gh[i] = 0.0f;
    for (int m = 0; m < l.n_gh; ++m)
      for (int i = 0; i < 42; ++i) gh[i] += l.gh[m][i];
  }
  // SoftConstraint::AddGradientsAndHessiansToLinks soft_constraint.cpp:113-131 (optimizer.cpp:283-284)
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
      float* gh = gh_all + (size_t)(which == 0 ? sc.joint.link1 : sc.joint.link2) * 42;
      for (int i = 0; i < 6; ++i) gh[i] += g[i];
      for (int i = 0; i < 36; ++i) gh[6 + i] += h[i];
    }
  }
  float* A = o.partial;
  float* b = o.partial + (size_t)dof * dof;
  for (int i = 0; i < dof * dof + dof; ++i) o.partial[i] = 0.0f;
  float* hj = gh_all + (size_t)o.n_links * 42;  // reuse the head of the solve scratch: 6 * dof floats
  for (int li = 0; li < o.n_links; ++li) {
    const float* J = jac_all + (size_t)li * 6 * dof;
    const float* g = gh_all + (size_t)li * 42;
    const float* H = g + 6;
    for (int i = 0; i < dof; ++i) {
      float s = 0.0f;
      for (int k = 0; k < 6; ++k) s += J[(size_t)i * 6 + k] * g[k];
      b[i] += s;
    }
    for (int c = 0; c < dof; ++c)
      for (int r = 0; r < 6; ++r) {
        float s = 0.0f;
        for (int k = 0; k < 6; ++k) s += H[k * 6 + r] * J[(size_t)c * 6 + k];
        hj[(size_t)c * 6 + r] = s;
      }
    for (int c = 0; c < dof; ++c)
      for (int r = c; r < dof; ++r) {
        float s = 0.0f;
        for (int k = 0; k < 6; ++k) s += J[(size_t)r * 6 + k] * hj[(size_t)c * 6 + k];
        A[(size_t)c * dof + r] -= s;
      }
  }
}

// the rest of Optimizer::CalculateOptimization + Optimizer::UpdatePoses (:335-346)
__global__ void links_solve_kernel(const TreeOptDev* opts, int n_opts, float* body_poses, int zero_theta) {
  int oi = blockIdx.x * blockDim.x + threadIdx.x;
  if (oi >= n_opts) return;
  const TreeOptDev& o = opts[oi];
  const int dof = o.dof, size = o.dof + o.n_rows;
  float* jac_all = o.work;
  float* gh_all = jac_all + (size_t)o.n_links * 6 * dof;
  float* A = gh_all + (size_t)o.n_links * 42;
  float* b = A + (size_t)size * size;
  float* temp = b + size;
  int* trans = reinterpret_cast<int*>(temp + size);
  float* cres = temp + 2 * size + size;  // [n_rows]
  float* cjac = cres + o.n_rows;         // [n_rows x dof] (per constraint block, column-major n_c x dof)
  float* j1 = cjac + (size_t)o.n_rows * dof;
  float* j2 = j1 + 36;
  for (size_t i = 0; i < (size_t)size * size; ++i) A[i] = 0.0f;
  for (int i = 0; i < size; ++i) b[i] = 0.0f;
  if (!zero_theta) {  // zero_theta: Optimizer::CalculateConsistentPoses optimizer.cpp:135 (theta = 0)
  for (int c = 0; c < dof; ++c)
    for (int r = 0; r < dof; ++r) A[(size_t)c * size + r] = o.partial[(size_t)c * dof + r];
  for (int i = 0; i < dof; ++i) b[i] = o.partial[(size_t)dof * dof + i];
  // constraints: Constraint::CalculateResidualAndConstraintJacobian constraint.cpp:81-102
  int idx = dof;
  for (int ci = 0; ci < o.n_constraints; ++ci) {
    const ConstraintDev& c = o.constraints[ci];
    const LinkDev& l1 = o.links[c.link1];
    const LinkDev& l2 = o.links[c.link2];
    Affine b12j1 = load_pose(c.body12joint1);
    Affine body22joint1 = mul_pose(mul_pose(b12j1, inverse_pose(link_pose(l1, body_poses))), link_pose(l2, body_poses));
    Affine joint22joint1 = mul_pose(body22joint1, inverse_pose(load_pose(c.body22joint2)));
    float angle, axis[3];
    angle_axis(joint22joint1.l, &angle, axis);
    float rv[3] = {angle * axis[0], angle * axis[1], angle * axis[2]};
    int n_c = c.n, ri = 0;
    for (int d = 0; d < 6; ++d)
      if (c.directions[d]) cres[ri++] = d < 3 ? rv[d] : joint22joint1.t[d - 3];
    constraint_unprojected_jacobian(c, joint22joint1, body22joint1, j2);
    constraint_unprojected_jacobian(c, joint22joint1, b12j1, j1);
    const float* J1 = jac_all + (size_t)c.link1 * 6 * dof;
    const float* J2 = jac_all + (size_t)c.link2 * 6 * dof;
    for (int col = 0; col < dof; ++col)
      for (int r = 0; r < n_c; ++r) {
        float s2 = 0.0f, s1 = 0.0f;
        for (int k = 0; k < 6; ++k) {
          s2 += j2[k * n_c + r] * J2[(size_t)col * 6 + k];
          s1 += j1[k * n_c + r] * J1[(size_t)col * 6 + k];
        }
        cjac[(size_t)col * n_c + r] = s2 - s1;
      }
    // AddResidualsAndConstraintJacobians optimizer.cpp:323-333
    for (int r = 0; r < n_c; ++r) {
      b[idx + r] = cres[r];
      for (int col = 0; col < dof; ++col) A[(size_t)col * size + idx + r] = -cjac[(size_t)col * n_c + r];
    }
    idx += n_c;
  }
  // Tikhonov vector optimizer.cpp:252-271 (free-direction order, rotation first)
  for (int li = 0; li < o.n_links; ++li) {
    const LinkDev& l = o.links[li];
    int j = l.first_jacobian_index;
    for (int d = 0; d < 6; ++d)
      if (l.free_directions[d]) {
        A[(size_t)j * size + j] += d < 3 ? o.tikhonov_rotation : o.tikhonov_translation;
        j++;
      }
  }
  ldlt_solve_dynamic(A, b, size, temp, trans);
  for (int i = 0; i < size; ++i)
    if (b[i] != b[i]) return;  // NaN guard optimizer.cpp:165
  }
  // Link::UpdatePoses link.cpp:205-241, parents before children
  for (int li = 0; li < o.n_links; ++li) {
    LinkDev& l = o.links[li];
    float th[6];
    int j = l.first_jacobian_index;
    for (int d = 0; d < 6; ++d) th[d] = l.free_directions[d] ? b[j++] : 0.0f;
    float K[9], R[9];
    K[0] = 0.0f;   K[3] = -th[2]; K[6] = th[1];
    K[1] = th[2];  K[4] = 0.0f;   K[7] = -th[0];
    K[2] = -th[1]; K[5] = th[0];  K[8] = 0.0f;
    expm3(K, R);
    Affine var;
    for (int i = 0; i < 9; ++i) var.l[i] = R[i];
    var.t[0] = th[3]; var.t[1] = th[4]; var.t[2] = th[5];
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

}  // extern "C"
