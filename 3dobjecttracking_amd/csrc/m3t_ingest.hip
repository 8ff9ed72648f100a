// m3t_ingest.hip — ROI ingest: instead of whole frames, only the rectangle of every
// camera frame that the trackers can read crosses PCIe.
//   roi_rect_kernel    every camera's rectangle from the bodies' poses (m3t_roi.h: projected box around the model's
//                      points + the modality's reach + a margin for the motion until the frame is used: the caller's,
//                      or -- adaptive -- from what the body's rectangle moved over the last steps, at most the caller's)
//   roi_pull_kernel    ONE launch per batch-frame on the copy stream: the rectangles' rows straight from the mapped,
//                      page-locked host block into the ring slot, 16 bytes per thread and trip (measured: 33-51 GB/s,
//                      against 2.8-6.4 GB/s for one 2-D DMA per camera: tools/ubench_ingest.hip)
//   roi_repair_kernel  after a guarded tracking step (tracking_step_*_guard_kernel, m3t_kernels.hip: a step that needs
//                      pixels outside its rectangles is not committed, its object is flagged): the WHOLE frames of the
//                      cameras the flagged objects read, from the same host block; the step is then repeated for the
//                      flagged objects.  Nothing to do, and next to no time, when no object is flagged
// Included by m3t_hip_api.hip after m3t_kernels.hip (roi_body2camera, roi_intrinsics and the guard live there).
#ifndef M3T_INGEST_HIP_
#define M3T_INGEST_HIP_

#include "m3t_roi.h"

extern "C" {

// one thread per camera of the batch: cam_ids[i] = camera id; item_first[camera id .. + 1] = its readers in `items`;
// rects: the rectangle table of the slot, by camera id.  Adaptive margins (prev_poses and motion_peak not null): the
// frame is read two steps after body_poses, so a reader's rectangle has to hold two steps of the body's motion.  With
// m = what the rectangle's edges moved between prev_poses (one step earlier) and body_poses, and peak = the largest m
// of the recent past (motion_peak[reader], decaying by 2 % per step), the margin is max(2 m, 1.25 peak) + 3 pixels --
// twice the last step for a body that keeps its velocity, the recent extreme for one that jitters -- at least
// min_margin_px, at most margin_px.  A body that outruns that is caught by the tracking kernels' guard and repeated.
__global__ void __launch_bounds__(64)
roi_rect_kernel(const RoiItemDev* items, const int* item_first, const int* cam_ids, int n_batch, const CameraDev* cams,
                const float* body_poses, const float* prev_poses, float* motion_peak, float margin_px, float min_margin_px,
                m3t_roi_rect* rects) {
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
    float margin = margin_px;
    if (prev_poses && motion_peak) {
      const m3t_roi_rect now = m3t_roi_body(b2c, it.box_min, it.box_max, &k, it.reach_px, it.reach_m, it.rho);
      float b2c_prev[16];
      roi_body2camera(cam.world2camera, prev_poses + 16 * it.body, b2c_prev);
      const m3t_roi_rect before = m3t_roi_body(b2c_prev, it.box_min, it.box_max, &k, it.reach_px, it.reach_m, it.rho);
      const float moved = (float)max(max(abs(now.x0 - before.x0), abs(now.x1 - before.x1)),
                                     max(abs(now.y0 - before.y0), abs(now.y1 - before.y1)));
      const float known = motion_peak[j];  // < 0: no motion seen yet -- the first rectangle takes the whole margin
      const float peak = known < 0.0f ? moved : fmaxf(moved, 0.98f * known);
      motion_peak[j] = peak;  // (this camera's thread is the only one that touches reader j)
      if (!(known < 0.0f)) margin = fminf(margin_px, fmaxf(min_margin_px, fmaxf(2.0f * moved, 1.25f * peak) + 3.0f));
    }
    r = m3t_roi_union(r, m3t_roi_body(b2c, it.box_min, it.box_max, &k, it.reach_px + margin, it.reach_m, it.rho));
  }
  rects[cam_id] = r;
}

// grid: (ceil(height / 8), cameras of the batch).  src / dst: camera blockIdx.y of the batch at + blockIdx.y * stride
// (one host block, one slab).  At most 32 VGPRs: a wave of this kernel fits next to the two 240-VGPR waves per SIMD of
// tracking_step_split_kernel, so the rectangles of frame k + 1 CAN cross PCIe while step k runs on all CUs -- measured
// (profiles/r04_roi_trace.txt): the step kernel then runs 2.3-3 x longer (the CUs' memory pipelines hold the ~2 us PCIe
// reads), and the loop takes what step + pull take one after the other (hence m3t_hip_reserve_ingest_cus).
// Round 6 (tools/ubench_ingest.hip, profiles/r06_ubench_ingest.txt): a kernel that streams the whole host block reaches
// the link's 57 GB/s (as the DMA engines do); rectangles of the tracked size come at 43 GB/s in this shape and at 47 GB/s
// with 32 rows per workgroup, four loads in flight and 64-byte-aligned spans ON AN IDLE GPU -- but beside the step, on
// 32 / 64 reserved CUs, that shape gave 168 k / 150 k pose-updates/s against this one's 180 k / 185 k: not taken.
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

// The repair after a guarded step, same grid and same source / destination as the pull of the batch: a camera one of
// whose readers is flagged (RoiGuardDev::flag of the reader's optimizer, behind its search poses) gets its whole frame;
// the first workgroup of the camera notes that in the rectangle table.  (The rectangle is read by the repeated step
// only, which is launched after this kernel.)
__global__ void __launch_bounds__(256)
roi_repair_kernel(const int* cam_ids, const RoiItemDev* items, const int* item_first, const RigidOptDev* opts, int n_poses,
                  m3t_roi_rect* rects, const uint8_t* src0, size_t src_camera_stride, uint32_t src_row_step, uint8_t* dst0,
                  size_t dst_camera_stride, uint32_t dst_pitch, int width, int height, int bytes_per_pixel) {
  const int cam_id = cam_ids[blockIdx.y];
  bool flagged = false;
  for (int j = item_first[cam_id]; j < item_first[cam_id + 1] && !flagged; ++j) {
    const int opt = items[j].opt;
    if (opt >= 0 && opts[opt].search_poses)
      flagged = reinterpret_cast<const RoiGuardDev*>(opts[opt].search_poses + 16 * n_poses)->flag != 0;
  }
  if (!flagged) return;
  const int row0 = (int)blockIdx.x * 8;
  if (row0 >= height) return;
  const int chunks = (width * bytes_per_pixel + 15) >> 4;
  const uint8_t* src = src0 + (size_t)blockIdx.y * src_camera_stride;
  uint8_t* dst = dst0 + (size_t)blockIdx.y * dst_camera_stride;
  const int rows = min(8, height - row0), total = rows * chunks;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int dr = i / chunks, c = i - dr * chunks;
    *reinterpret_cast<uint4*>(dst + (size_t)(row0 + dr) * dst_pitch + ((size_t)c << 4)) =
        *reinterpret_cast<const uint4*>(src + (size_t)(row0 + dr) * src_row_step + ((size_t)c << 4));
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) rects[cam_id] = m3t_roi_rect{0, 0, width - 1, height - 1};
}

// whole frames went into the slot of the cameras first_id .. first_id + n - 1 (equal geometry): their rectangles are
// the frames
__global__ void __launch_bounds__(64)
roi_set_rects_kernel(m3t_roi_rect* rects /* of the slot, by camera id */, int first_id, int n, int width, int height) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rects[first_id + i] = m3t_roi_rect{0, 0, width - 1, height - 1};
}

}  // extern "C"
#endif  // M3T_INGEST_HIP_
