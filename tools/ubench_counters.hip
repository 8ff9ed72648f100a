// tools/ubench_counters.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts,
// in the access patterns of the tracking kernels (MI355X_MICROARCH.md §HBM: "FETCH_SIZE reports exactly 1/2 of the bytes
// of a wide coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Every kernel below touches every byte of its buffer exactly once (the
// gathers: every request its own 256-byte block), the buffers are 1-2 GiB (>> 32 MB of L2 + 256 MB of Infinity Cache),
// and the program prints the byte count per kernel; tools/counter_calibration.py divides what rocprofv3 reports by it.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/ubench_counters tools/ubench_counters.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -- tools/bin/ubench_counters
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -- tools/bin/ubench_counters
//
// Patterns:
//   cal_read_16B     16 bytes per lane, lanes consecutive                (the guide's calibrated case: expect 0.5)
//   cal_read_4B      4 bytes per lane, lanes consecutive                 (model rows, histogram tables read in order)
//   cal_read_pixels  a lane per image ROW, unaligned 4-byte loads 3 bytes apart along the row: the pixel walk of the
//                    correspondence lines (lanes of a wave on different rows, one BGR pixel per load); a 640 x 512 BGR8
//                    frame per workgroup, 2048 frames
//   cal_read_u16     the same for 16-bit depth pixels (the depth window scans), aligned 2-byte loads
//   cal_gather_4B    one 4-byte load per 256-byte block, blocks in a pseudo-random order: every request misses in
//                    every cache and moves one fill granule (what does the counter tally per sparse request?)
//   cal_write_16B / cal_write_4B   consecutive stores (histogram blend, line state)
//   cal_write_8B_scattered         one 8-byte store per 256-byte block ({tag, value} granules, poses)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct __attribute__((packed)) PackedU32 { uint32_t v; };

__global__ void __launch_bounds__(256) cal_read_16B(const uint4* p, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 v = p[i];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) cal_read_4B(const unsigned* p, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 0x12345678u) out[0] = acc;
}
// workgroup = frame, thread = row: 640 pixels of 3 bytes, 8 loads in flight
__global__ void __launch_bounds__(512) cal_read_pixels(const uint8_t* frames, int pitch, int width, size_t frame_bytes,
                                                       unsigned* out) {
  const uint8_t* row = frames + (size_t)blockIdx.x * frame_bytes + (size_t)threadIdx.x * pitch;
  unsigned acc = 0;
  for (int x = 0; x < width; x += 8) {
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const PackedU32*>(row + (x + j) * 3)->v;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j] & 0xffffffu;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(512) cal_read_u16(const uint8_t* frames, int pitch, int width, size_t frame_bytes,
                                                    unsigned* out) {
  const uint8_t* row = frames + (size_t)blockIdx.x * frame_bytes + (size_t)threadIdx.x * pitch;
  unsigned acc = 0;
  for (int x = 0; x < width; x += 8) {
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const unsigned short*>(row + (x + j) * 2);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// block b of 256 bytes is visited exactly once: index -> (index * odd) mod 2^k is a permutation
__global__ void __launch_bounds__(256) cal_gather_4B(const unsigned* p, size_t n_blocks, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_blocks; i += (size_t)gridDim.x * 256) {
    const size_t b = (i * 2654435761ull) & (n_blocks - 1);
    acc += p[b * 64 + (b & 63)];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) cal_write_16B(uint4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    p[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void __launch_bounds__(256) cal_write_4B(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (unsigned)i;
}
__global__ void __launch_bounds__(256) cal_write_8B_scattered(unsigned long long* p, size_t n_blocks) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_blocks; i += (size_t)gridDim.x * 256) {
    const size_t b = (i * 2654435761ull) & (n_blocks - 1);
    p[b * 32 + (b & 31)] = i;
  }
}

int main() {
  const size_t bytes = 2ull << 30;
  uint8_t* buffer;
  unsigned* out;
  CHECK(hipMalloc(&buffer, bytes + 4096));
  CHECK(hipMemset(buffer, 1, bytes + 4096));
  CHECK(hipMalloc(&out, 4));
  CHECK(hipDeviceSynchronize());
  const int grid = 256 * 8;
  const size_t stream = 1ull << 30;
  const int pitch = 1920, width = 640, rows = 512;
  const size_t frame_bytes = (size_t)pitch * rows;
  const int frames = 2048;  // 1.875 GiB
  const int dpitch = 1280;  // 640 u16
  const size_t dframe = (size_t)dpitch * rows;
  const size_t n_blocks = 1ull << 22;  // x 256 B = 1 GiB
  // every pattern twice in a row on DIFFERENT halves where it fits; the kernels of one name are summed by the script
  hipLaunchKernelGGL(cal_read_16B, dim3(grid), dim3(256), 0, 0, (const uint4*)buffer, stream / 16, out);
  hipLaunchKernelGGL(cal_read_4B, dim3(grid), dim3(256), 0, 0, (const unsigned*)(buffer + stream), stream / 4, out);
  hipLaunchKernelGGL(cal_read_pixels, dim3(frames), dim3(rows), 0, 0, buffer, pitch, width, frame_bytes, out);
  hipLaunchKernelGGL(cal_read_u16, dim3(frames), dim3(rows), 0, 0, buffer, dpitch, width, dframe, out);
  hipLaunchKernelGGL(cal_gather_4B, dim3(grid), dim3(256), 0, 0, (const unsigned*)buffer, n_blocks, out);
  hipLaunchKernelGGL(cal_write_16B, dim3(grid), dim3(256), 0, 0, (uint4*)buffer, stream / 16);
  hipLaunchKernelGGL(cal_write_4B, dim3(grid), dim3(256), 0, 0, (unsigned*)(buffer + stream), stream / 4);
  hipLaunchKernelGGL(cal_write_8B_scattered, dim3(grid), dim3(256), 0, 0, (unsigned long long*)buffer, n_blocks);
  CHECK(hipDeviceSynchronize());
  // kernel, bytes the kernel asks for (each byte once), bytes at a 64-byte / 128-byte fill granule for the sparse ones
  printf("cal_read_16B %zu %zu %zu\n", stream, stream, stream);
  printf("cal_read_4B %zu %zu %zu\n", stream, stream, stream);
  printf("cal_read_pixels %zu %zu %zu\n", (size_t)frames * rows * width * 3, (size_t)frames * frame_bytes, (size_t)frames * frame_bytes);
  printf("cal_read_u16 %zu %zu %zu\n", (size_t)frames * rows * width * 2, (size_t)frames * dframe, (size_t)frames * dframe);
  printf("cal_gather_4B %zu %zu %zu\n", n_blocks * 4, n_blocks * 64, n_blocks * 128);
  printf("cal_write_16B %zu %zu %zu\n", stream, stream, stream);
  printf("cal_write_4B %zu %zu %zu\n", stream, stream, stream);
  printf("cal_write_8B_scattered %zu %zu %zu\n", n_blocks * 8, n_blocks * 64, n_blocks * 128);
  return 0;
}
