// tools/ubench_gather.hip — developer micro-benchmark: how many single-dword gathers per second gfx950 sustains when
// every lane of a wave reads its own address (the access pattern of the correspondence lines' pixel walk: one BGR
// pixel per lane and load, lanes on different image rows), by the level that serves them:
//   L1   every workgroup gathers inside its own 8 KB                 -> texture addresser / L1 rate (lane-requests/s)
//   L2   inside 16 MB shared by all workgroups of an XCD ... 32 MB   -> L1-miss / L2-hit rate
//   HBM  every workgroup inside its own 1 MB "frame" of a 4 GB buffer (1024 resident workgroups x 1 MB >> L2 + MALL)
//   HBM* anywhere in the 4 GB buffer
// Launch shape of tracking_step_compact_kernel at 4096 objects: 4096 workgroups x 256 threads, 4 resident per CU,
// 8 loads in flight per lane.  Compare with the kernel's own rates (profiles/r03_pmc_rbot4096.json: lane-requests,
// TCP_TCC_READ_REQ, FETCH_SIZE per launch / its duration).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_gather tools/ubench_gather.hip && ./ubench_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// region_dwords: size of the window a workgroup gathers in; regions: how many distinct windows the buffer holds
// (workgroup b uses window b % regions); rounds x 8 loads per lane
__global__ void __launch_bounds__(256, 4)
gather_kernel(const unsigned* __restrict__ buffer, unsigned long long region_dwords, unsigned regions, int rounds,
              unsigned* out) {
  const unsigned long long base = (unsigned long long)(blockIdx.x % regions) * region_dwords;
  unsigned state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  unsigned acc = 0;
  const unsigned long long mask = region_dwords - 1;  // power of two
  for (int r = 0; r < rounds; ++r) {
    unsigned v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      state = state * 1664525u + 1013904223u;
      const unsigned long long at = (((unsigned long long)state * 2654435761ull) >> 20) & mask;
      v[i] = buffer[base + at];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
  }
  if (acc == 0x12345678u) out[0] = acc;  // (keeps the loads)
}

int main() {
  const size_t bytes = 4ull << 30;
  unsigned* buffer;
  unsigned* out;
  CHECK(hipMalloc(&buffer, bytes));
  CHECK(hipMemset(buffer, 1, bytes));
  CHECK(hipMalloc(&out, 4));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  struct Case { const char* name; unsigned long long region_bytes; unsigned regions; int rounds; };
  const Case cases[] = {
      {"L1   8 KB window per workgroup (4096 windows)", 8ull << 10, 4096, 64},
      {"L2   64 KB window per workgroup, 256 windows = 16 MB in all", 64ull << 10, 256, 32},
      {"L2   1 MB window per workgroup, 32 windows = 32 MB in all", 1ull << 20, 32, 32},
      {"MALL 1 MB window per workgroup, 128 windows = 128 MB in all", 1ull << 20, 128, 16},
      {"HBM  1 MB window per workgroup, 4096 windows = 4 GB", 1ull << 20, 4096, 16},
      {"HBM* one 4 GB window", 4ull << 30, 1, 16},
  };
  for (const Case& c : cases) {
    const int grid = 4096;
    for (int rep = 0; rep < 2; ++rep) {  // the second run is the measured one
      CHECK(hipEventRecord(a));
      hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(256), 0, 0, buffer, c.region_bytes / 4, c.regions, c.rounds, out);
      CHECK(hipEventRecord(b));
      CHECK(hipEventSynchronize(b));
    }
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    const double loads = (double)grid * 256 * c.rounds * 8;
    printf("%-64s %8.3f ms  %7.1f G lane-requests/s  (%6.1f GB/s if every request were its own 64-byte line)\n", c.name, ms,
           loads / ms * 1e-6, loads * 64 / ms * 1e-6);
  }
  return 0;
}
