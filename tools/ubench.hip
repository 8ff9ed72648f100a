// tools/ubench.hip — developer micro-benchmarks for gfx950: what a DEPENDENT instruction of a lone wave costs, and
// how that changes with more waves per SIMD.  The tracking kernels' serial sections (reference-order summation
// chains, the 6x6 solve) are priced by these numbers.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o ubench tools/ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 4096;

template <int KIND>
__global__ void chain_kernel(float* out, unsigned long long* cycles, const float* in, int n_iter) {
  __shared__ float lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += blockDim.x) lds[i] = (float)((i * 7 + 1) & 1023);
  __syncthreads();
  float a0 = in[0], a1 = in[1], a2 = in[2], a3 = in[3];
  float s = in[4] + (float)lane;
  float t = in[5];
  double d = (double)in[6];
  int idx = lane;
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
  unsigned long long t0 = __builtin_readcyclecounter();
  t0 = clock64();
  for (int it = 0; it < n_iter; ++it) {
#pragma unroll 16
    for (int i = 0; i < 64; ++i) {
      if (KIND == 0) {          // dependent v_sub_f32
        s -= (i & 1) ? a0 : a1;
      } else if (KIND == 1) {   // two interleaved dependent chains
        s -= a0; t -= a1;
      } else if (KIND == 2) {   // dependent correctly rounded divide
        s = s / a2;
      } else if (KIND == 3) {   // readlane -> VALU
        s = s * a3 + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 3));
      } else if (KIND == 4) {   // dependent LDS read (pointer chase)
        idx = (int)lds[idx & 1023];
      } else if (KIND == 5) {   // dependent global load (L2 / L1 hit)
        idx = (int)in[8 + (idx & 1023)];
      } else if (KIND == 6) {   // dependent f64 fma
        d = d * (double)a3 + (double)a2;
      } else if (KIND == 7) {   // DPP row_shr dependent
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x111, 0xf, 0xf, false));
      } else if (KIND == 8) {   // sub fed by two independent multiplies (the J-row chain step)
        s -= (a0 * t) * a1; t += 1.0f;
      } else if (KIND == 9) {   // independent subs (issue rate)
        s -= a0; t -= a1; a2 -= a0; a3 -= a1;
      } else if (KIND == 10) {  // sqrt chain
        s = sqrtf(s + a2);
      } else if (KIND == 11) {  // barrier
        __syncthreads();
      }
    }
  }
  unsigned long long t1 = clock64();
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + lane] = s + t + (float)d + (float)idx + a2 + a3;
}

template <int KIND>
void run(const char* name, float* d_out, unsigned long long* d_cyc, float* d_in, int per_iter_ops) {
  const int configs[][2] = {{1, 64}, {1, 256}, {1, 512}, {1, 1024}, {256, 256}, {512, 256}, {1024, 256}, {2048, 256}};
  printf("%-34s", name);
  for (auto& c : configs) {
    const int n_iter = N / 64;
    hipLaunchKernelGGL(chain_kernel<KIND>, dim3(c[0]), dim3(c[1]), 0, 0, d_out, d_cyc, d_in, n_iter);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(chain_kernel<KIND>, dim3(c[0]), dim3(c[1]), 0, 0, d_out, d_cyc, d_in, n_iter);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(c[0]);
    CHECK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * c[0], hipMemcpyDeviceToHost));
    double sum = 0;
    for (auto v : h) sum += (double)v;
    printf(" %7.2f", sum / c[0] / (double)(N * per_iter_ops));
  }
  printf("\n");
}

int main() {
  float* d_out; unsigned long long* d_cyc; float* d_in;
  CHECK(hipMalloc(&d_out, sizeof(float) * 2048 * 1024));
  CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 4096));
  std::vector<float> in(8 + 1024);
  in[0] = 1.25f; in[1] = -0.75f; in[2] = 1.0000001f; in[3] = 0.9999999f; in[4] = 3.0f; in[5] = 5.0f; in[6] = 1.5f;
  for (int i = 0; i < 1024; ++i) in[8 + i] = (float)((i * 13 + 5) & 1023);
  CHECK(hipMalloc(&d_in, sizeof(float) * in.size()));
  CHECK(hipMemcpy(d_in, in.data(), sizeof(float) * in.size(), hipMemcpyHostToDevice));
  printf("cycles (s_memtime) per operation; columns = grid x block: 1x64 1x256 1x512 1x1024 256x256 512x256 1024x256 2048x256\n");
  run<0>("dependent v_sub_f32", d_out, d_cyc, d_in, 1);
  run<1>("2 interleaved dependent subs (per pair)", d_out, d_cyc, d_in, 1);
  run<9>("4 independent subs (per group)", d_out, d_cyc, d_in, 1);
  run<8>("sub fed by 2 muls (per step)", d_out, d_cyc, d_in, 1);
  run<2>("dependent f32 divide", d_out, d_cyc, d_in, 1);
  run<10>("dependent sqrtf(add)", d_out, d_cyc, d_in, 1);
  run<3>("readlane -> fma", d_out, d_cyc, d_in, 1);
  run<7>("dpp row_shr -> add", d_out, d_cyc, d_in, 1);
  run<6>("dependent f64 fma", d_out, d_cyc, d_in, 1);
  run<4>("dependent ds_read (+cvt)", d_out, d_cyc, d_in, 1);
  run<5>("dependent global_load (+cvt)", d_out, d_cyc, d_in, 1);
  run<11>("__syncthreads", d_out, d_cyc, d_in, 1);
  // clock rate of s_memtime vs wall
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(chain_kernel<0>, dim3(1), dim3(64), 0, 0, d_out, d_cyc, d_in, 4096);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc = 0; CHECK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
  printf("long chain: %llu ticks in %.3f ms -> %.1f MHz tick rate (if the kernel dominates the interval)\n", cyc, ms, cyc / ms / 1e3);
  return 0;
}
