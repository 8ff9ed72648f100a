// m3t_ingest.hip — ROI ingest: instead of whole frames, only the rectangle of every
// camera frame that the trackers can read crosses PCIe.
//   roi_rect_kernel   every camera's rectangle from the bodies' poses (m3t_roi.h: projected box around the model's
//                     points + the modality's reach + the caller's margin for the motion until the frame is used)
//   roi_pull_kernel   ONE launch per batch-frame on the copy stream: the rectangles' rows straight from the mapped,
//                     page-locked host block into the ring slot, 16 bytes per thread and trip (measured: 33-51 GB/s,
//                     against 2.8-6.4 GB/s for one 2-D DMA per camera: tools/ubench_ingest.hip)
//   roi_check_kernel  after a tracking step: the rectangle the step really needed -- the union over the poses its
//                     searches ran at (the tracking kernels store them) -- against the rectangle that was in the
//                     slot; a body whose needs stick out is reported (m3t_hip_roi_get_status), its pose of this step
//                     is not to be trusted
// Included by m3t_hip_api.hip after m3t_kernels.hip.
#ifndef M3T_INGEST_HIP_
#define M3T_INGEST_HIP_

#include "m3t_roi.h"

namespace {

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

}  // namespace

extern "C" {

// one thread per camera of the batch: cam_ids[i] = camera id; item_first[camera id .. + 1] = its readers in `items`;
// rects: the rectangle table of the slot, by camera id
__global__ void __launch_bounds__(64)
roi_rect_kernel(const RoiItemDev* items, const int* item_first, const int* cam_ids, int n_batch, const CameraDev* cams,
                const float* body_poses, float margin_px, m3t_roi_rect* rects) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_batch) return;
  const int cam_id = cam_ids[i];
  const CameraDev& cam = cams[cam_id];
  const m3t_intrinsics k = roi_intrinsics(cam);
  m3t_roi_rect r = m3t_roi_empty();
  for (int j = item_first[cam_id]; j < item_first[cam_id + 1]; ++j) {
    const RoiItemDev& it = items[j];
    float b2c[16];
    roi_body2camera(cam.world2camera, body_poses + 16 * it.body, b2c);
    r = m3t_roi_union(r, m3t_roi_body(b2c, it.box_min, it.box_max, &k, it.reach_px + margin_px, it.reach_m));
  }
  rects[cam_id] = r;
}

// grid: (ceil(height / 8), cameras of the batch).  src / dst: camera blockIdx.y of the batch at + blockIdx.y * stride
// (one host block, one slab).  At most 32 VGPRs: a wave of this kernel fits next to the two 240-VGPR waves per SIMD of
// tracking_step_split_kernel, so the rectangles of frame k + 1 CAN cross PCIe while step k runs on all CUs -- measured
// (profiles/r04_roi_trace.txt): the step kernel then runs 2.3-3 x longer (the CUs' memory pipelines hold the ~2 us PCIe
// reads), and the loop takes what step + pull take one after the other.
__global__ void __launch_bounds__(256)
roi_pull_kernel(const int* cam_ids, const m3t_roi_rect* rects /* of this slot, by camera id */, const uint8_t* src0,
                size_t src_camera_stride, uint32_t src_row_step, uint8_t* dst0, size_t dst_camera_stride,
                uint32_t dst_pitch, int bytes_per_pixel) {
  const m3t_roi_rect r = rects[cam_ids[blockIdx.y]];
  if (r.x1 < r.x0) return;
  const int row0 = r.y0 + (int)blockIdx.x * 8;
  if (row0 > r.y1) return;
  // the span widened to 16-byte boundaries of the row (rows start 16-byte aligned on both sides: checked by the host)
  const int b0 = (r.x0 * bytes_per_pixel) & ~15, b1 = ((r.x1 + 1) * bytes_per_pixel + 15) & ~15;
  const int chunks = (b1 - b0) >> 4;
  const uint8_t* src = src0 + (size_t)blockIdx.y * src_camera_stride + b0;
  uint8_t* dst = dst0 + (size_t)blockIdx.y * dst_camera_stride + b0;
  const int rows = min(8, r.y1 - row0 + 1), total = rows * chunks;
  for (int i = threadIdx.x; i < total; i += 512) {  // two reads over PCIe in flight per thread
    const int j = i + 256;
    const int dr0 = i / chunks, c0 = i - dr0 * chunks;
    const int dr1 = j / chunks, c1 = j - dr1 * chunks;
    const uint4 v0 = *reinterpret_cast<const uint4*>(src + (size_t)(row0 + dr0) * src_row_step + ((size_t)c0 << 4));
    uint4 v1 = v0;
    if (j < total) v1 = *reinterpret_cast<const uint4*>(src + (size_t)(row0 + dr1) * src_row_step + ((size_t)c1 << 4));
    *reinterpret_cast<uint4*>(dst + (size_t)(row0 + dr0) * dst_pitch + ((size_t)c0 << 4)) = v0;
    if (j < total) *reinterpret_cast<uint4*>(dst + (size_t)(row0 + dr1) * dst_pitch + ((size_t)c1 << 4)) = v1;
  }
}

// one thread per reader: n_poses poses per object ([0] the pose at the start of the step, [1 .. n_corr] the poses of
// the searches, [n_corr + 1] the final pose: the histogram lines); rects: [slot][camera id], the camera table says
// which slot the step read; misses[0] = count, misses[1 ..] = body ids
__global__ void __launch_bounds__(64)
roi_check_kernel(const RoiItemDev* items, int n_items, const CameraDev* cams, int n_cams, const RigidOptDev* opts,
                 int n_poses, const m3t_roi_rect* rects, int n_rect_slots, int* misses, int capacity) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_items) return;
  const RoiItemDev& it = items[i];
  if (it.opt < 0) return;
  const float* poses = opts[it.opt].search_poses;
  if (!poses) return;
  const CameraDev& cam = cams[it.camera];
  if (cam.slot >= n_rect_slots) return;  // (a ring slot added since the tables were built: whole frames only)
  const m3t_roi_rect have = rects[(size_t)cam.slot * n_cams + it.camera];
  if (have.x0 <= 0 && have.y0 <= 0 && have.x1 >= cam.width - 1 && have.y1 >= cam.height - 1) return;  // a whole frame
  const m3t_intrinsics k = roi_intrinsics(cam);
  m3t_roi_rect need = m3t_roi_empty();
  for (int j = 0; j < n_poses; ++j) {
    float b2c[16];
    roi_body2camera(cam.world2camera, poses + 16 * j, b2c);
    need = m3t_roi_union(need, m3t_roi_body(b2c, it.box_min, it.box_max, &k, it.reach_px, it.reach_m));
  }
  if (!m3t_roi_contains(have, need)) {
    const int at = atomicAdd(&misses[0], 1);
    if (at < capacity) misses[1 + at] = it.body;
  }
}

// a whole frame went into (slot, camera): its rectangle is the frame
__global__ void roi_set_rect_kernel(m3t_roi_rect* rect, int x0, int y0, int x1, int y1) {
  rect->x0 = x0; rect->y0 = y0; rect->x1 = x1; rect->y1 = y1;
}

}  // extern "C"
#endif  // M3T_INGEST_HIP_
