// m3t_render.hip — FocusedBasicDepthRenderer / FocusedSilhouetteRenderer for the renderer-fed
// branches: the square crop around the referenced bodies
// (FocusedRenderer::CalculateProjectionMatrix renderer.cpp:348-405) rasterised with OpenGL's
// rules: pixel centres at integer image coordinates (renderer.cpp:396-404), window coordinates
// snapped to 1/256 pixel, top-left fill rule, 16-bit depth, GL_LESS in draw order
// (basic_depth_renderer.cpp:45-84, silhouette_renderer.cpp:54-100).  Three launches per set of
// renderers: clear + crop, rasterise (the triangle lists split over 32 workgroups per renderer; the
// z-buffer holds packed (depth16 << 16 | draw order << 8 | id) words that triangles reach with
// atomicMin, so the result does not depend on the order), unpack.  Included by m3t_hip_api.hip after
// m3t_kernels.hip.
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

// FocusedRenderer::CalculateProjectionMatrix renderer.cpp:348-405 (identical arithmetic in every thread that
// needs it); returns the number of visible referenced bodies
struct FocusedProjection {
  float corner_u, corner_v, scale;
  unsigned visible_mask;
  int n_visible;
  M44 P;
};
__device__ FocusedProjection focused_projection(const RendererDev& r, const CameraDev& cam, const float* body_poses) {
  FocusedProjection f;
  const Affine w2c = load_pose(cam.world2camera);
  float u_min = 3.402823466e+38f, u_max = 1.175494351e-38f, v_min = 3.402823466e+38f, v_max = 1.175494351e-38f;
  f.n_visible = 0;
  f.visible_mask = 0;
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
    f.visible_mask |= 1u << k;
    ++f.n_visible;
  }
  f.corner_u = 0.0f;
  f.corner_v = 0.0f;
  f.scale = 1.0f;
  for (int i = 0; i < 16; ++i) f.P.m[i] = 0.0f;
  if (f.n_visible > 0) {
    const int S = r.image_size;
    const float d = fmaxf(u_max - u_min, v_max - v_min) * 1.05f;  // kImageSizeSafetyMargin
    f.corner_u = 0.5f * (u_min + u_max - d);
    f.corner_v = 0.5f * (v_min + v_max - d);
    f.scale = (float)S / d;
    const float ppu_scaled = (cam.ppu - f.corner_u) * f.scale;
    const float ppv_scaled = (cam.ppv - f.corner_v) * f.scale;
    f.P(0, 0) = 2.0f * cam.fu / d;
    f.P(0, 2) = 2.0f * (ppu_scaled + 0.5f) / (float)S - 1.0f;
    f.P(1, 1) = 2.0f * cam.fv / d;
    f.P(1, 2) = 2.0f * (ppv_scaled + 0.5f) / (float)S - 1.0f;
    f.P(2, 2) = (r.z_max + r.z_min) / (r.z_max - r.z_min);
    f.P(2, 3) = -2.0f * r.z_max * r.z_min / (r.z_max - r.z_min);
    f.P(3, 2) = 1.0f;
  }
  return f;
}

extern "C" {

// 1/3: clear the packed z-buffer, publish the crop (grid: 16 x renderers)
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
focused_clear_kernel(const RendererDev* renderers, const int* which, const CameraDev* cams, const float* body_poses) {
  const RendererDev& r = renderers[which[blockIdx.y]];
  const int n_px = r.image_size * r.image_size;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += gridDim.x * blockDim.x) r.packed[i] = 0xffffffffu;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const FocusedProjection f = focused_projection(r, cams[r.camera], body_poses);
    r.state[RS_CORNER_U] = f.corner_u;
    r.state[RS_CORNER_V] = f.corner_v;
    r.state[RS_SCALE] = f.scale;
    r.state[RS_TERM_A] = r.z_max * r.z_min * 65535.0f / (r.z_max - r.z_min);  // renderer.cpp:567-570
    r.state[RS_TERM_B] = r.z_max * 65535.0f / (r.z_max - r.z_min);
    r.state[RS_N_VISIBLE] = (float)f.n_visible;
    for (int k = 0; k < M3T_MAX_RENDERER_BODIES; ++k)
      r.state[RS_VISIBLE0 + k] = (f.visible_mask >> k & 1u) ? 1.0f : 0.0f;
  }
}

// One triangle after projection, snapping, culling: vertices re-ordered to positive area
struct RasterTriangle {
  double ax[3], ay[3], z[3], area;
  int x0, x1, y0, y1;
};
__device__ bool raster_setup(const M44& trans, const float* vertices, const int* triangles, int t, bool culling,
                             int S, RasterTriangle& o) {
  const float half_s = 0.5f * (float)S;
  double sx[3], sy[3];
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
    sx[k] = floor((double)wx * 256.0 + 0.5);
    sy[k] = floor((double)wy * 256.0 + 0.5);
  }
  if (behind) return false;
  // anything this far off the image cannot touch it and would leave the exact-integer range
  if (!(fabs(sx[0]) < 3.0e7 && fabs(sx[1]) < 3.0e7 && fabs(sx[2]) < 3.0e7 && fabs(sy[0]) < 3.0e7 &&
        fabs(sy[1]) < 3.0e7 && fabs(sy[2]) < 3.0e7))
    return false;
  double area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sy[1] - sy[0]) * (sx[2] - sx[0]);
  if (area == 0.0) return false;
  // counter-clockwise meshes seen from outside have negative area in the y-down image
  if (area > 0.0 && culling) return false;
  int i1 = 1, i2 = 2;
  if (area < 0.0) { i1 = 2; i2 = 1; area = -area; }
  o.ax[0] = sx[0]; o.ax[1] = sx[i1]; o.ax[2] = sx[i2];
  o.ay[0] = sy[0]; o.ay[1] = sy[i1]; o.ay[2] = sy[i2];
  o.z[0] = (double)wz[0]; o.z[1] = (double)wz[i1]; o.z[2] = (double)wz[i2];
  o.area = area;
  const double min_x = fmin(o.ax[0], fmin(o.ax[1], o.ax[2])), max_x = fmax(o.ax[0], fmax(o.ax[1], o.ax[2]));
  const double min_y = fmin(o.ay[0], fmin(o.ay[1], o.ay[2])), max_y = fmax(o.ay[0], fmax(o.ay[1], o.ay[2]));
  // pixels whose centre (256 p + 128) lies inside the bounding box: nothing else can pass the edge tests
  o.x0 = (int)fmax(ceil((min_x - 128.0) / 256.0), 0.0);
  o.x1 = (int)fmin(floor((max_x - 128.0) / 256.0), (double)(S - 1));
  o.y0 = (int)fmax(ceil((min_y - 128.0) / 256.0), 0.0);
  o.y1 = (int)fmin(floor((max_y - 128.0) / 256.0), (double)(S - 1));
  return o.x1 >= o.x0 && o.y1 >= o.y0;
}
// edge functions in f64: the snapped coordinates are integers below 2^26, products and their differences are
// exact, so the coverage is the integer result of the oracle
__device__ __forceinline__ void raster_pixel(const RasterTriangle& t, int px, int py, uint32_t low_bits, int S,
                                             uint32_t* z_buffer) {
  const double cx = (double)px * 256.0 + 128.0, cy = (double)py * 256.0 + 128.0;
  double e[3];
  bool inside = true;
  for (int k = 0; k < 3; ++k) {
    const int k1 = (k + 1) % 3;
    const double ex = t.ax[k1] - t.ax[k], ey = t.ay[k1] - t.ay[k];
    e[k] = ex * (cy - t.ay[k]) - ey * (cx - t.ax[k]);
    const bool owns = ey < 0.0 || (ey == 0.0 && ex > 0.0);  // top-left rule, y down
    inside = inside && (e[k] > 0.0 || (e[k] == 0.0 && owns));
  }
  if (!inside) return;
#if M3T_RASTER_PROBE == 5
  if (t.area > 0.0) { atomicMin(&z_buffer[py * S + px], 0xfffffffeu); return; }
#endif
  const double z = (e[1] / t.area) * t.z[0] + (e[2] / t.area) * t.z[1] + (e[0] / t.area) * t.z[2];
  if (!(z >= 0.0 && z <= 1.0)) return;
  const uint32_t d16 = (uint32_t)floor(z * 65535.0 + 0.46);  // see the oracle / gl_model.py
  atomicMin(&z_buffer[py * S + px], (d16 << 16) | low_bits);
}

// 2/3: rasterise (grid: slices x renderers; a slice is a contiguous part of every body's triangle list).
// A triangle with a small bounding box is finished by the thread that owns it; larger ones are queued in
// LDS and rasterised by the whole workgroup.
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
focused_raster_kernel(const RendererDev* renderers, const int* which, const CameraDev* cams, const float* body_poses) {
#ifndef M3T_RASTER_QUEUE
#define M3T_RASTER_QUEUE 64
#endif
#ifndef M3T_RASTER_SERIAL_MAX
#define M3T_RASTER_SERIAL_MAX 192
#endif
  constexpr int kQueue = M3T_RASTER_QUEUE;
  __shared__ RasterTriangle queue[kQueue];
  __shared__ int n_queued;
  const RendererDev& r = renderers[which[blockIdx.y]];
  const CameraDev& cam = cams[r.camera];
  const FocusedProjection f = focused_projection(r, cam, body_poses);
  if (f.n_visible == 0) return;  // block-uniform
#if M3T_RASTER_PROBE == 1
  if (f.n_visible > 0) return;
#endif
  const int S = r.image_size;
  uint32_t* z_buffer = r.packed;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int order = 0; order < r.n_bodies; ++order) {
    const M44 trans = mul44(f.P, mul44(load44(cam.world2camera),
                                       mul44(load44(body_poses + 16 * r.body[order]), load44(r.geometry2body[order]))));
    const uint32_t low_bits = ((uint32_t)order << 8) | (r.silhouette ? (uint32_t)r.id[order] : 0u);
    const float* vertices = r.vertices[order];
    const int* triangles = r.triangles[order];
    const bool culling = r.culling[order] != 0;
#if M3T_RASTER_PROBE == 3
    if (r.n_triangles[order] > 64) continue;   // few-triangle bodies only
#elif M3T_RASTER_PROBE == 4
    if (r.n_triangles[order] <= 64) continue;  // many-triangle bodies only
#endif
    const int per_slice = (r.n_triangles[order] + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per_slice;
    const int t_end = min(t_begin + per_slice, r.n_triangles[order]);
    for (int base = t_begin; base < t_end; base += nt) {  // block-uniform trip count
      if (tid == 0) n_queued = 0;
      __syncthreads();
      const int t = base + tid;
      RasterTriangle tri;
      if (t < t_end && raster_setup(trans, vertices, triangles, t, culling, S, tri)) {
        const int pixels = (tri.x1 - tri.x0 + 1) * (tri.y1 - tri.y0 + 1);
        int slot = kQueue;
#if M3T_RASTER_PROBE == 2
        atomicMin(&z_buffer[tri.y0 * S + tri.x0], 0xfffffffeu);
        continue;
#endif
        if (pixels > M3T_RASTER_SERIAL_MAX) slot = atomicAdd(&n_queued, 1);
        if (slot < kQueue) {
          queue[slot] = tri;
        } else {
          for (int py = tri.y0; py <= tri.y1; ++py)
            for (int px = tri.x0; px <= tri.x1; ++px) raster_pixel(tri, px, py, low_bits, S, z_buffer);
        }
      }
      __syncthreads();
      const int nq = min(n_queued, kQueue);
      for (int q = 0; q < nq; ++q) {
        const RasterTriangle big = queue[q];
        const int w = big.x1 - big.x0 + 1, total = w * (big.y1 - big.y0 + 1);
        for (int i = tid; i < total; i += nt) raster_pixel(big, big.x0 + i % w, big.y0 + i / w, low_bits, S, z_buffer);
      }
      __syncthreads();
    }
  }
}

// 3/3: unpack into the u16 depth image and the u8 id image (grid: 16 x renderers)
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
focused_unpack_kernel(const RendererDev* renderers, const int* which) {
  const RendererDev& r = renderers[which[blockIdx.y]];
  const int n_px = r.image_size * r.image_size;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += gridDim.x * blockDim.x) {
    const uint32_t v = r.packed[i];
    r.depth_image[i] = v == 0xffffffffu ? (uint16_t)65535 : (uint16_t)(v >> 16);
    r.silhouette_image[i] = v == 0xffffffffu ? (uint8_t)0 : (uint8_t)(v & 0xffu);
  }
}

}  // extern "C"
#endif  // M3T_RENDER_HIP_
