// tools/ubench_walk.hip — developer micro-benchmark: the dependency structure of the correspondence lines' pixel walk
// (compact_walk in m3t_compact.hip) with synthetic addresses, to see how its throughput moves with the number of
// resident waves per SIMD and with the number of loads a lane keeps in flight -- i.e. whether a kernel rebuilt for
// fewer registers per thread (more waves) would get closer to the gather rates of tools/ubench_gather.hip.
//
// Per round and lane: IN_FLIGHT unaligned dword loads at random byte offsets of the workgroup's own 1 MB window
// ("pixels", HBM-resident: 4096 windows), IN_FLIGHT dependent 8-byte gathers from a 256 KB table indexed by the loaded
// value ("histogram pair", L2-resident), then per load ~48 VALU operations that consume the pair (the 8-term
// distribution product of a segment).  256-thread workgroups = one wave per SIMD each; the dynamic LDS size caps the
// workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ubench_walk tools/ubench_walk.hip && ./ubench_walk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct __attribute__((packed, aligned(1))) PackedU32 { unsigned v; };

template <int IN_FLIGHT, int VALU_TERMS>
__global__ void __launch_bounds__(256)
walk_kernel(const unsigned char* __restrict__ frames, const float2* __restrict__ table, int rounds, float* out) {
  extern __shared__ float lds[];
  const unsigned char* window = frames + (size_t)blockIdx.x * (1u << 20);
  unsigned state = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  float lf[8], lb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { lf[k] = 0.5f + 0.01f * k; lb[k] = 0.5f - 0.01f * k; }
  float acc = 1.0f;
  if (threadIdx.x == 1000) lds[0] = 0.0f;
  for (int r = 0; r < rounds; ++r) {
    unsigned px[IN_FLIGHT];
#pragma unroll
    for (int i = 0; i < IN_FLIGHT; ++i) {
      state = state * 1664525u + 1013904223u;
      const unsigned off = (state >> 12) & ((1u << 20) - 1u);
      px[i] = reinterpret_cast<const PackedU32*>(window + off)->v;
    }
    float2 h[IN_FLIGHT];
#pragma unroll
    for (int i = 0; i < IN_FLIGHT; ++i) h[i] = table[(px[i] ^ (px[i] >> 9)) & 32767u];
#pragma unroll
    for (int i = 0; i < IN_FLIGHT; ++i) {
      float value = h[i].x - h[i].y;
#pragma unroll
      for (int k = 0; k < VALU_TERMS; ++k) value *= h[i].x * lf[k & 7] + h[i].y * lb[k & 7];
      acc += value;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

__global__ void fill_kernel(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned v = (unsigned)i * 2654435761u;
    v ^= v >> 15;
    p[i] = v * 2246822519u;
  }
}

template <int IN_FLIGHT, int VALU_TERMS>
void run(const unsigned char* frames, const float2* table, float* out, int waves_per_simd) {
  const int grid = 4096, rounds = 48 * 8 / IN_FLIGHT;  // the same number of loads for every IN_FLIGHT
  const size_t lds = (size_t)(160 * 1024 / waves_per_simd) / 1024 * 1024 - (waves_per_simd > 2 ? 1024 : 0);
  auto kernel = walk_kernel<IN_FLIGHT, VALU_TERMS>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int resident = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, kernel, 256, lds));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  float ms = 0.0f;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, 0, frames, table, rounds, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    CHECK(hipEventElapsedTime(&ms, a, b));
  }
  const double loads = (double)grid * 256 * rounds * IN_FLIGHT;
  printf("in flight %2d  VALU terms %2d  workgroups/CU %d (asked %d)  %8.3f ms  %6.1f G pixel loads/s\n", IN_FLIGHT, VALU_TERMS,
         resident, waves_per_simd, ms, loads / ms * 1e-6);
}

int main() {
  unsigned char* frames;
  float2* table;
  float* out;
  CHECK(hipMalloc(&frames, ((size_t)4096 << 20) + 64));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<unsigned*>(frames), ((size_t)4096 << 20) / 4 + 16);
  CHECK(hipMalloc(&table, 32768 * sizeof(float2)));
  hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, reinterpret_cast<unsigned*>(table), (size_t)32768 * 2);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMalloc(&out, 4));
  for (int w : {1, 2, 3, 4, 5, 6, 8}) run<8, 8>(frames, table, out, w);
  for (int w : {2, 4, 8}) run<16, 8>(frames, table, out, w);
  for (int w : {2, 4, 8}) run<4, 8>(frames, table, out, w);
  for (int w : {2, 4, 8}) run<8, 24>(frames, table, out, w);  // three times the VALU work per pixel (scale > 1 searches)
  for (int w : {2, 4, 8}) run<8, 0>(frames, table, out, w);   // no VALU work: the two dependent round trips alone
  return 0;
}
