// m3t_render.hip — FocusedBasicDepthRenderer / FocusedSilhouetteRenderer for the renderer-fed
// branches: the square crop around the referenced bodies
// (FocusedRenderer::CalculateProjectionMatrix renderer.cpp:348-405) rasterised with OpenGL's
// rules: pixel centres at integer image coordinates (renderer.cpp:396-404), window coordinates
// snapped to 1/256 pixel, top-left fill rule, 16-bit depth, GL_LESS in draw order
// (basic_depth_renderer.cpp:45-84, silhouette_renderer.cpp:54-100).  One workgroup per renderer;
// the z-buffer holds packed (depth16 << 16 | draw order << 8 | id) words that triangles reach with
// atomicMin, in LDS when image_size^2 words fit (the default 200 x 200 does).  Included by
// m3t_hip_api.hip after m3t_kernels.hip.
#ifndef M3T_RENDER_HIP_
#define M3T_RENDER_HIP_

namespace {

struct M44 {
  float m[16];  // column-major
  __device__ float operator()(int r, int c) const { return m[c * 4 + r]; }
  __device__ float& operator()(int r, int c) { return m[c * 4 + r]; }
};
__device__ M44 mul44(const M44& a, const M44& b) {
  M44 o;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r)
      o(r, c) = ((a(r, 0) * b(0, c) + a(r, 1) * b(1, c)) + a(r, 2) * b(2, c)) + a(r, 3) * b(3, c);
  return o;
}
__device__ M44 load44(const float* p) {
  M44 o;
  for (int i = 0; i < 16; ++i) o.m[i] = p[i];
  return o;
}
__device__ __forceinline__ long long floor_div256(long long a) { return a >> 8; }  // arithmetic shift = floor

}  // namespace

extern "C" {

__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
focused_render_kernel(const RendererDev* renderers, const int* which, const CameraDev* cams, const float* body_poses,
                      int z_buffer_in_lds) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_z[];
  const RendererDev& r = renderers[which[blockIdx.x]];
  const CameraDev& cam = cams[r.camera];
  const int S = r.image_size, n_px = S * S;
  const int tid = threadIdx.x, nt = blockDim.x;
  uint32_t* z_buffer = z_buffer_in_lds ? lds_z : r.packed;

  // FocusedRenderer::CalculateProjectionMatrix renderer.cpp:348-405 (every thread, identical arithmetic)
  const Affine w2c = load_pose(cam.world2camera);
  float u_min = 3.402823466e+38f, u_max = 1.175494351e-38f, v_min = 3.402823466e+38f, v_max = 1.175494351e-38f;
  int n_visible = 0;
  unsigned visible_mask = 0;
  for (int k = 0; k < r.n_referenced; ++k) {
    const float* b2w = body_poses + 16 * r.referenced[k];
    float rr = 0.5f * r.referenced_diameter[k];
    float x, y, z;
    apply_pose(w2c, b2w[12], b2w[13], b2w[14], x, y, z);
    if (z < rr * 1.5f || z - rr < r.z_min || z + rr > r.z_max) continue;
    float abs_x = fabsf(x), abs_y = fabsf(y);
    float x2 = x * x, y2 = y * y, z2 = z * z, r2 = rr * rr;
    float rz = rr * z;
    float z2_r2 = z2 - r2;
    float z3_zr2 = z2_r2 * z;
    float r_u = cam.fu * (abs_x * r2 + rz * sqrtf(z2_r2 + x2)) / z3_zr2;
    float r_v = cam.fv * (abs_y * r2 + rz * sqrtf(z2_r2 + y2)) / z3_zr2;
    float center_u = x * cam.fu / z + cam.ppu;
    float center_v = y * cam.fv / z + cam.ppv;
    float u_min_body = center_u - r_u, u_max_body = center_u + r_u;
    float v_min_body = center_v - r_v, v_max_body = center_v + r_v;
    if (u_min_body > (float)cam.width || u_max_body < 0.0f || v_min_body > (float)cam.height || v_max_body < 0.0f)
      continue;
    u_min = fminf(u_min, u_min_body);
    u_max = fmaxf(u_max, u_max_body);
    v_min = fminf(v_min, v_min_body);
    v_max = fmaxf(v_max, v_max_body);
    visible_mask |= 1u << k;
    ++n_visible;
  }
  for (int i = tid; i < n_px; i += nt) z_buffer[i] = 0xffffffffu;
  __syncthreads();

  float corner_u = 0.0f, corner_v = 0.0f, scale = 1.0f;
  if (n_visible > 0) {
    const float d = fmaxf(u_max - u_min, v_max - v_min) * 1.05f;  // kImageSizeSafetyMargin
    corner_u = 0.5f * (u_min + u_max - d);
    corner_v = 0.5f * (v_min + v_max - d);
    scale = (float)S / d;
    const float ppu_scaled = (cam.ppu - corner_u) * scale;
    const float ppv_scaled = (cam.ppv - corner_v) * scale;
    M44 P;
    for (int i = 0; i < 16; ++i) P.m[i] = 0.0f;
    P(0, 0) = 2.0f * cam.fu / d;
    P(0, 2) = 2.0f * (ppu_scaled + 0.5f) / (float)S - 1.0f;
    P(1, 1) = 2.0f * cam.fv / d;
    P(1, 2) = 2.0f * (ppv_scaled + 0.5f) / (float)S - 1.0f;
    P(2, 2) = (r.z_max + r.z_min) / (r.z_max - r.z_min);
    P(2, 3) = -2.0f * r.z_max * r.z_min / (r.z_max - r.z_min);
    P(3, 2) = 1.0f;
    const float half_s = 0.5f * (float)S;
    for (int order = 0; order < r.n_bodies; ++order) {
      const M44 trans = mul44(P, mul44(load44(cam.world2camera),
                                       mul44(load44(body_poses + 16 * r.body[order]), load44(r.geometry2body[order]))));
      const uint32_t id = r.silhouette ? (uint32_t)r.id[order] : 0u;
      const float* vertices = r.vertices[order];
      const int* triangles = r.triangles[order];
      const bool culling = r.culling[order] != 0;
      for (int t = tid; t < r.n_triangles[order]; t += nt) {
        long long sx[3], sy[3];
        float wz[3];
        bool behind = false;
        for (int k = 0; k < 3; ++k) {
          const float* p = vertices + (size_t)triangles[t * 3 + k] * 3;
          float cx = ((trans(0, 0) * p[0] + trans(0, 1) * p[1]) + trans(0, 2) * p[2]) + trans(0, 3);
          float cy = ((trans(1, 0) * p[0] + trans(1, 1) * p[1]) + trans(1, 2) * p[2]) + trans(1, 3);
          float cz = ((trans(2, 0) * p[0] + trans(2, 1) * p[1]) + trans(2, 2) * p[2]) + trans(2, 3);
          float cw = ((trans(3, 0) * p[0] + trans(3, 1) * p[1]) + trans(3, 2) * p[2]) + trans(3, 3);
          if (!(cw > 0.0f)) behind = true;  // no near-plane clipping: such triangles are dropped
          float wx = (cx / cw + 1.0f) * half_s;
          float wy = (cy / cw + 1.0f) * half_s;
          wz[k] = (cz / cw + 1.0f) * 0.5f;
          sx[k] = (long long)floor((double)wx * 256.0 + 0.5);
          sy[k] = (long long)floor((double)wy * 256.0 + 0.5);
        }
        if (behind) continue;
        long long area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sy[1] - sy[0]) * (sx[2] - sx[0]);
        if (area == 0) continue;
        // counter-clockwise meshes seen from outside have negative area in the y-down image
        if (area > 0 && culling) continue;
        int i1 = 1, i2 = 2;
        if (area < 0) { i1 = 2; i2 = 1; area = -area; }
        const long long ax[3] = {sx[0], sx[i1], sx[i2]}, ay[3] = {sy[0], sy[i1], sy[i2]};
        const double z0 = (double)wz[0], z1 = (double)wz[i1], z2 = (double)wz[i2];
        const long long min_x = min(ax[0], min(ax[1], ax[2])), max_x = max(ax[0], max(ax[1], ax[2]));
        const long long min_y = min(ay[0], min(ay[1], ay[2])), max_y = max(ay[0], max(ay[1], ay[2]));
        const int x0 = (int)max(floor_div256(min_x) - 1, 0LL), x1 = (int)min(floor_div256(max_x) + 1, (long long)(S - 1));
        const int y0 = (int)max(floor_div256(min_y) - 1, 0LL), y1 = (int)min(floor_div256(max_y) + 1, (long long)(S - 1));
        const double a2 = (double)area;
        for (int py = y0; py <= y1; ++py)
          for (int px = x0; px <= x1; ++px) {
            const long long cx = (long long)px * 256 + 128, cy = (long long)py * 256 + 128;
            long long e[3];
            bool inside = true;
            for (int k = 0; k < 3; ++k) {
              const int k1 = (k + 1) % 3;
              const long long dx = ax[k1] - ax[k], dy = ay[k1] - ay[k];
              e[k] = dx * (cy - ay[k]) - dy * (cx - ax[k]);
              const bool owns = dy < 0 || (dy == 0 && dx > 0);  // top-left rule, y down
              inside = inside && (e[k] > 0 || (e[k] == 0 && owns));
            }
            if (!inside) continue;
            const double z = ((double)e[1] / a2) * z0 + ((double)e[2] / a2) * z1 + ((double)e[0] / a2) * z2;
            if (!(z >= 0.0 && z <= 1.0)) continue;
            const uint32_t d16 = (uint32_t)floor(z * 65535.0 + 0.46);  // see the oracle / gl_model.py
            atomicMin(&z_buffer[py * S + px], (d16 << 16) | ((uint32_t)order << 8) | id);
          }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < n_px; i += nt) {
    const uint32_t v = z_buffer[i];
    r.depth_image[i] = v == 0xffffffffu ? (uint16_t)65535 : (uint16_t)(v >> 16);
    r.silhouette_image[i] = v == 0xffffffffu ? (uint8_t)0 : (uint8_t)(v & 0xffu);
  }
  if (tid == 0) {
    r.state[RS_CORNER_U] = corner_u;
    r.state[RS_CORNER_V] = corner_v;
    r.state[RS_SCALE] = scale;
    r.state[RS_TERM_A] = r.z_max * r.z_min * 65535.0f / (r.z_max - r.z_min);  // renderer.cpp:567-570
    r.state[RS_TERM_B] = r.z_max * 65535.0f / (r.z_max - r.z_min);
    r.state[RS_N_VISIBLE] = (float)n_visible;
    for (int k = 0; k < M3T_MAX_RENDERER_BODIES; ++k) r.state[RS_VISIBLE0 + k] = (visible_mask >> k & 1u) ? 1.0f : 0.0f;
  }
}

}  // extern "C"
#endif  // M3T_RENDER_HIP_
