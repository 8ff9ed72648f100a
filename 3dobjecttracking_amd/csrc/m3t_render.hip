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

#include "m3t_raster.h"

namespace {

using M44 = RasterM44;
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

// 2/3: rasterise (grid: slices x renderers; a slice is a contiguous part of every body's triangle list).
// A triangle with a small bounding box is finished by the thread that owns it, row after row (raster_row: three
// additions per pixel, and a row is left behind its covered span -- the slivers of a finely tessellated body have
// boxes that are mostly empty); larger ones are queued in LDS and rasterised by the whole workgroup, a thread taking
// 32-pixel pieces of rows.
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
focused_raster_kernel(const RendererDev* renderers, const int* which, const CameraDev* cams, const float* body_poses) {
  constexpr int kQueue = 64, kPiece = 32;
  __shared__ RasterTriangle queue[kQueue];
  __shared__ int n_queued;
  const RendererDev& r = renderers[which[blockIdx.y]];
  const CameraDev& cam = cams[r.camera];
  const FocusedProjection f = focused_projection(r, cam, body_poses);
  if (f.n_visible == 0) return;  // block-uniform
  const int S = r.image_size;
  uint32_t* z_buffer = r.packed;
  auto sink = [z_buffer, S](int px, int py, uint32_t word) { atomicMin(&z_buffer[py * S + px], word); };
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int order = 0; order < r.n_bodies; ++order) {
    const M44 trans = mul44(f.P, mul44(load44(cam.world2camera),
                                       mul44(load44(body_poses + 16 * r.body[order]), load44(r.geometry2body[order]))));
    const uint32_t low_bits = ((uint32_t)order << 8) | (r.silhouette ? (uint32_t)r.id[order] : 0u);
    const float* vertices = r.vertices[order];
    const int* triangles = r.triangles[order];
    const bool culling = r.culling[order] != 0;
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
        if (pixels > 192) slot = atomicAdd(&n_queued, 1);
        if (slot < kQueue) {
          queue[slot] = tri;
        } else {
          for (int py = tri.y0; py <= tri.y1; ++py) raster_row(tri, py, tri.x0, tri.x1, low_bits, sink);
        }
      }
      __syncthreads();
      const int nq = min(n_queued, kQueue);
      for (int q = 0; q < nq; ++q) {
        const RasterTriangle big = queue[q];
        const int pieces = (big.x1 - big.x0 + kPiece) / kPiece, total = pieces * (big.y1 - big.y0 + 1);
        for (int i = tid; i < total; i += nt) {
          const int row = i / pieces, xa = big.x0 + (i - row * pieces) * kPiece;
          raster_row(big, big.y0 + row, xa, min(xa + kPiece - 1, big.x1), low_bits, sink);
        }
      }
      __syncthreads();
    }
  }
}

// ---- the two-launch form, used when the z-buffer of a rendering fits the LDS of a CU (image_size <= 200) ----
// Of a finely tessellated body most triangles never reach a pixel (tools/raster_stats.py: 1 000 - 1 300 of 20 950 on
// the probe scene; the rest face away or lie outside the crop), and the survivors' boxes hold ~55 000 pixels in all:
// little work, spread thin.  focused_setup_kernel (grid: slices x renderers) does the set-up and APPENDS the survivors
// to a list; focused_resolve_kernel (a workgroup per band of rows of every renderer) clears its band of the z-buffer in
// LDS, rasterises the list into it with LDS atomics, and writes the depth and id images: no clear launch, no global
// atomics, no unpack launch.  The words and their minimum are those of the three-launch form.
struct RasterSurvivor {
  RasterTriangle tri;
  uint32_t low_bits, pad;
};
static_assert(sizeof(RasterSurvivor) == M3T_SURVIVOR_BYTES, "M3T_SURVIVOR_BYTES");

// which: pairs {renderer, twin or -1}.  A twin is a second renderer of the same camera, geometry, referenced bodies,
// depth range and image size (a FocusedBasicDepthRenderer and a FocusedSilhouetteRenderer of one camera, say): its
// rendering is this one -- the z-buffer word orders by depth and draw order, the id byte follows from the draw order --
// so one set-up and one rasterisation serve both; the resolve kernel writes the twin's images with the twin's ids.
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
focused_setup_kernel(const RendererDev* renderers, const int* which, const CameraDev* cams, const float* body_poses) {
  const RendererDev& r = renderers[which[2 * blockIdx.y]];
  const int twin = which[2 * blockIdx.y + 1];
  const CameraDev& cam = cams[r.camera];
  const FocusedProjection f = focused_projection(r, cam, body_poses);
  if (threadIdx.x == 0 && blockIdx.x == 0) {  // the crop, for the modalities that read the rendering
    for (int w = 0; w < (twin >= 0 ? 2 : 1); ++w) {
      float* state = w == 0 ? r.state : renderers[twin].state;
      state[RS_CORNER_U] = f.corner_u;
      state[RS_CORNER_V] = f.corner_v;
      state[RS_SCALE] = f.scale;
      state[RS_TERM_A] = r.z_max * r.z_min * 65535.0f / (r.z_max - r.z_min);  // renderer.cpp:567-570
      state[RS_TERM_B] = r.z_max * 65535.0f / (r.z_max - r.z_min);
      state[RS_N_VISIBLE] = (float)f.n_visible;
      for (int k = 0; k < M3T_MAX_RENDERER_BODIES; ++k)
        state[RS_VISIBLE0 + k] = (f.visible_mask >> k & 1u) ? 1.0f : 0.0f;
    }
  }
  if (f.n_visible == 0) return;  // block-uniform
  const int S = r.image_size;
  RasterSurvivor* list = static_cast<RasterSurvivor*>(r.survivors);
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & (kWave - 1);
  for (int order = 0; order < r.n_bodies; ++order) {
    const M44 trans = mul44(f.P, mul44(load44(cam.world2camera),
                                       mul44(load44(body_poses + 16 * r.body[order]), load44(r.geometry2body[order]))));
    const uint32_t low_bits = ((uint32_t)order << 8) | (r.silhouette ? (uint32_t)r.id[order] : 0u);
    const float* vertices = r.vertices[order];
    const int* triangles = r.triangles[order];
    const bool culling = r.culling[order] != 0;
    const int per_slice = (r.n_triangles[order] + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per_slice;
    const int t_end = min(t_begin + per_slice, r.n_triangles[order]);
    // A slice is tens of trips long at 128 pairs and a trip is two dependent loads (indices, then the vertices they
    // name -- the meshes of 64 objects do not stay in L2) in front of ~400 instructions, at two waves per SIMD: the
    // loads run two trips ahead -- the indices of trip i + 2 and the vertices of trip i + 1 are on their way while
    // trip i is set up (round 5, 128 pairs x 2 slices: 49.5 -> 45 us with the indices alone -> 41.4 us).  Measured
    // and not kept, all with identical images: the body's vertices snapped once into an LDS table and the triangles
    // set up from it (a third of the instructions, 41.9 us: the trips are chains of dependent f64 operations at two
    // waves per SIMD, not instruction issue), two triangles per thread and trip on top of that (41.8), the list
    // append's atomic answered one trip later (57: registers), a 128-VGPR build with two workgroups per CU (46.9)
    int idx1[3] = {0, 0, 0}, idx2[3] = {0, 0, 0};
    float xyz1[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) xyz1[k] = 0.0f;
    auto load_indices = [&](int t, int (&index)[3]) {
      if (t < t_end) {
#pragma unroll
        for (int k = 0; k < 3; ++k) index[k] = triangles[t * 3 + k];
      }
    };
    auto load_vertices = [&](int t, const int (&index)[3], float (&xyz)[9]) {
      if (t < t_end) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float* p = vertices + (size_t)index[k] * 3;
          xyz[3 * k] = p[0]; xyz[3 * k + 1] = p[1]; xyz[3 * k + 2] = p[2];
        }
      }
    };
    load_indices(t_begin + tid, idx1);
    load_indices(t_begin + nt + tid, idx2);
    load_vertices(t_begin + tid, idx1, xyz1);
    for (int base = t_begin; base < t_end; base += nt) {
      const int t = base + tid;
      float xyz[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) xyz[k] = xyz1[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) idx1[k] = idx2[k];
      load_vertices(t + nt, idx1, xyz1);
      load_indices(t + 2 * nt, idx2);
      RasterSurvivor sv;
      const bool ok = t < t_end && raster_setup_vertices(trans, xyz, culling, S, sv.tri);
      // one atomic per wave: the lanes with a survivor take consecutive entries
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(ok);
      if (mask == 0) continue;  // wave-uniform
      int first = 0;
      if (lane == 0) first = atomicAdd(r.n_survivors, __builtin_popcountll(mask));
      first = __builtin_amdgcn_readfirstlane(first);
      if (ok) {
        const int at = first + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (at < r.survivor_capacity) {
          sv.low_bits = low_bits;
          sv.pad = 0;
          list[at] = sv;
        }
      }
    }
  }
}

// grid: (bands, renderers).  A workgroup owns a band of image rows: its z-buffer band lives in LDS, it looks at every
// survivor and rasterises the rows that fall into its band (round 4: ONE workgroup per renderer took 166 us for the
// probe scene's ~1 200 survivors, the three-launch form 98 us).  The workgroup that finishes last resets the counters.
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
focused_resolve_kernel(const RendererDev* renderers, const int* which) {
  extern __shared__ uint32_t lds_z[];  // [band rows * S] packed words, then first_item[threads + 1], wave_total[16]
  // Every survivor's rows inside the band are cut into pieces of kPiece pixels; the pieces of ALL survivors of a trip
  // are numbered through (block-wide prefix sum) and dealt out evenly: a covered pixel costs ~30 f64 operations, and a
  // thread that finished a 100-pixel box by itself kept its whole wave waiting (measured: 54 us per resolve).
  constexpr int kPiece = 8, kPer = 4;  // survivors a thread looks at per trip: one trip up to 2048 survivors
  // grid: (renderer pairs, bands) -- workgroup b runs on XCD b mod 8, so with the pair as the fast index the bands of a
  // pair share an XCD (whenever the number of pairs is a multiple of 8) and its survivor list is fetched into ONE L2:
  // round 5, 128 pairs x 8 bands -- with the band as the fast index each of the eight L2s read all 12 MB of lists
  const RendererDev& r = renderers[which[2 * blockIdx.x]];
  const int twin = which[2 * blockIdx.x + 1];  // a renderer whose rendering is this one (focused_setup_kernel), or -1
  const int S = r.image_size;
  const int n_bands = (int)gridDim.y;
  const int band_rows = (S + n_bands - 1) / n_bands;
  const int row_lo = (int)blockIdx.y * band_rows, row_hi = min(row_lo + band_rows, S) - 1;  // inclusive
  const int n_px = band_rows * S;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & (kWave - 1), wave = tid / kWave;
  int* first_item = reinterpret_cast<int*>(lds_z + n_px);  // [nt + 1]: pieces before thread t's survivors
  int* wave_total = first_item + nt + 1;                   // [16]
  int* own_count = wave_total + 16;                        // [kPer][nt]: pieces of thread t's j-th survivor
  for (int i = tid; i < n_px; i += nt) lds_z[i] = 0xffffffffu;
  const int n = min(__hip_atomic_load(r.n_survivors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), r.survivor_capacity);
  const RasterSurvivor* list = static_cast<const RasterSurvivor*>(r.survivors);
  auto sink = [S, row_lo](int px, int py, uint32_t word) { atomicMin(&lds_z[(py - row_lo) * S + px], word); };
  for (int base = 0; base < n && row_lo <= row_hi; base += nt * kPer) {  // block-uniform trip count
    int mine = 0, cnt[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int i = base + j * nt + tid;
      cnt[j] = 0;
      if (i < n) {
        const RasterTriangle& tri = list[i].tri;
        const int ya = max(tri.y0, row_lo), yb = min(tri.y1, row_hi);
        if (ya <= yb) cnt[j] = ((tri.x1 - tri.x0 + kPiece) / kPiece) * (yb - ya + 1);
      }
      mine += cnt[j];
    }
    // inclusive prefix sum over the wave (DPP: row_shr 1 2 4 8, row_bcast 15 / 31), then over the waves
    int incl = mine;
    incl += dpp_zero_i<0x111, 0xf>(incl);
    incl += dpp_zero_i<0x112, 0xf>(incl);
    incl += dpp_zero_i<0x114, 0xf>(incl);
    incl += dpp_zero_i<0x118, 0xf>(incl);
    incl += dpp_zero_i<0x142, 0xa>(incl);
    incl += dpp_zero_i<0x143, 0xc>(incl);
    __syncthreads();  // (the previous trip's pieces are done with the tables; the first time: the cleared z-buffer)
    if (lane == kWave - 1) wave_total[wave] = incl;
#pragma unroll
    for (int j = 0; j < kPer; ++j) own_count[j * nt + tid] = cnt[j];
    __syncthreads();
    int before = 0;
    for (int wv = 0; wv < wave; ++wv) before += wave_total[wv];
    first_item[tid] = before + incl - mine;
    if (tid == nt - 1) first_item[nt] = before + incl;
    __syncthreads();
    const int total = first_item[nt];
    for (int k = tid; k < total; k += nt) {
      int lo = 0, hi = nt - 1;  // the last thread whose first piece is <= k (threads without pieces repeat the value:
      while (lo < hi) {         // the last of equals is the one that owns the piece)
        const int mid = (lo + hi + 1) >> 1;
        if (first_item[mid] <= k) lo = mid; else hi = mid - 1;
      }
      int local = k - first_item[lo], j = 0;
      while (j < kPer - 1 && local >= own_count[j * nt + lo]) { local -= own_count[j * nt + lo]; ++j; }
      const RasterSurvivor& sv = list[base + j * nt + lo];
      const int ya = max(sv.tri.y0, row_lo);
      const int pieces = (sv.tri.x1 - sv.tri.x0 + kPiece) / kPiece;
      const int row = local / pieces, xa = sv.tri.x0 + (local - row * pieces) * kPiece;
      raster_row(sv.tri, ya + row, xa, min(xa + kPiece - 1, sv.tri.x1), sv.low_bits, sink);
    }
  }
  __syncthreads();
  const int n_out = (row_hi - row_lo + 1) * S;
  // four pixels per thread and store where the band allows it (its first pixel and its length multiples of four: the
  // images come from hipMalloc): one 8-byte and one 4-byte store instead of four 2-byte and four 1-byte ones -- the
  // output of 128 pairs cost 10.7 of the launch's 69 us (round 5, probe builds)
  const RendererDev* t = twin >= 0 ? &renderers[twin] : nullptr;  // the same rendering with the twin's id byte: the
  const size_t first = (size_t)row_lo * S;                        // winner's draw order sits in bits 8..15
  auto depth_of = [](uint32_t v) { return v == 0xffffffffu ? (uint32_t)65535 : v >> 16; };
  auto id_of = [](uint32_t v) { return v == 0xffffffffu ? 0u : (v & 0xffu); };
  auto twin_id_of = [t](uint32_t v) { return (v == 0xffffffffu || !t->silhouette) ? 0u : (uint32_t)(uint8_t)t->id[(v >> 8) & 0xffu]; };
  if ((first & 3) == 0 && (n_out & 3) == 0) {
    for (int i = tid * 4; i < n_out; i += nt * 4) {
      const uint32_t v0 = lds_z[i], v1 = lds_z[i + 1], v2 = lds_z[i + 2], v3 = lds_z[i + 3];
      const uint2 d = make_uint2(depth_of(v0) | depth_of(v1) << 16, depth_of(v2) | depth_of(v3) << 16);
      *reinterpret_cast<uint2*>(r.depth_image + first + i) = d;
      *reinterpret_cast<uint32_t*>(r.silhouette_image + first + i) =
          id_of(v0) | id_of(v1) << 8 | id_of(v2) << 16 | id_of(v3) << 24;
      if (t) {
        *reinterpret_cast<uint2*>(t->depth_image + first + i) = d;
        *reinterpret_cast<uint32_t*>(t->silhouette_image + first + i) =
            twin_id_of(v0) | twin_id_of(v1) << 8 | twin_id_of(v2) << 16 | twin_id_of(v3) << 24;
      }
    }
  } else {
    for (int i = tid; i < n_out; i += nt) {
      const uint32_t v = lds_z[i];
      r.depth_image[first + i] = (uint16_t)depth_of(v);
      r.silhouette_image[first + i] = (uint8_t)id_of(v);
      if (t) {
        t->depth_image[first + i] = (uint16_t)depth_of(v);
        t->silhouette_image[first + i] = (uint8_t)twin_id_of(v);
      }
    }
  }
  // every band has read the count by now once it says it is done: the last one clears the list for the next rendering
  if (tid == 0) {
    if (atomicAdd(r.n_survivors + 1, 1) == n_bands - 1) {
      __hip_atomic_store(r.n_survivors, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(r.n_survivors + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
