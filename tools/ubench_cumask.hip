// tools/ubench_cumask.hip -- developer probe for DESIGN 9 item 4 (ROI ingest): can the PCIe pull kernel run on a few
// CUs of its own (hipExtStreamCreateWithCUMask) while a latency-bound kernel runs on the others, without slowing it?
//   1. which CUs does a masked stream get: bits 0..239 vs bits 240..255 of the 256-bit mask, counted per XCD
//   2. the pull kernel's rate from mapped host memory on 16 CUs (and on all)
//   3. a latency-bound victim kernel (dependent L2 loads + VALU, one 512-thread workgroup per CU) alone, beside the
//      pull on the SAME CUs (unmasked streams), and beside the pull on the 16 reserved CUs
//   hipcc --offload-arch=gfx950 -O3 -o ubench_cumask tools/ubench_cumask.hip && ./ubench_cumask
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>
#define CHECK(x)                                                                            \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; }      \
  } while (0)

__global__ void __launch_bounds__(64) where(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  float v = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = (hwid >> 8) & 0xff; }
  if (v == 123.456f) out[0] = 0;
}

// rows of 64 rectangles of side x side BGR pixels out of 64 frames of 640 x 512 (pitch 1920) in mapped host memory
__global__ void __launch_bounds__(256) pull(const unsigned char* src, unsigned char* dst, int side) {
  const size_t frame = 1920 * 512;
  const int cam = blockIdx.y, row0 = (cam * 53) % (512 - side) + blockIdx.x * 8;
  const int x0 = ((cam * 37) % (640 - side)) * 3 & ~15, chunks = (side * 3 + 15) / 16;
  if ((int)blockIdx.x * 8 >= side) return;
  const int total = 8 * chunks;
  for (int i = threadIdx.x; i < total; i += 512) {
    const int j = i + 256;
    const int r0 = i / chunks, c0 = i - r0 * chunks, r1 = j / chunks, c1 = j - r1 * chunks;
    const size_t a0 = cam * frame + (size_t)(row0 + r0) * 1920 + x0 + c0 * 16;
    const size_t a1 = cam * frame + (size_t)(row0 + r1) * 1920 + x0 + c1 * 16;
    const uint4 v0 = *reinterpret_cast<const uint4*>(src + a0);
    uint4 v1 = v0;
    if (j < total) v1 = *reinterpret_cast<const uint4*>(src + a1);
    *reinterpret_cast<uint4*>(dst + a0) = v0;
    if (j < total) *reinterpret_cast<uint4*>(dst + a1) = v1;
  }
}

// latency-bound like the tracking step: every thread chases indices through an L2-resident table, some VALU between
__global__ void __launch_bounds__(512) victim(const unsigned* table, unsigned mask, int rounds, unsigned* out) {
  extern __shared__ float lds[];
  unsigned at = (blockIdx.x * 512 + threadIdx.x) & mask;
  float v = 0.0f;
  for (int i = 0; i < rounds; ++i) {
    at = table[at] & mask;
    v = v * 1.0001f + (float)(at & 7);
    lds[threadIdx.x] = v;
    __syncthreads();
    v += lds[(threadIdx.x + 1) & 511];
  }
  if (v == 123.456f) out[0] = at;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, reserve = argc > 1 ? atoi(argv[1]) : 16;
  printf("%s: %d CUs, %d reserved for the pull (the highest mask bits)\n", prop.name, cus, reserve);
  std::vector<uint32_t> mask_a((cus + 31) / 32, 0), mask_b((cus + 31) / 32, 0);
  for (int i = 0; i < cus; ++i) (i < cus - reserve ? mask_a : mask_b)[i / 32] |= 1u << (i % 32);
  hipStream_t sa, sb, ua, ub;
  CHECK(hipExtStreamCreateWithCUMask(&sa, (uint32_t)mask_a.size(), mask_a.data()));
  CHECK(hipExtStreamCreateWithCUMask(&sb, (uint32_t)mask_b.size(), mask_b.data()));
  CHECK(hipStreamCreateWithFlags(&ua, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&ub, hipStreamNonBlocking));
  unsigned* d_where;
  const int n_where = 4096;
  CHECK(hipMalloc(&d_where, n_where * 8));
  bool round_robin = true;  // bit i of the mask -> XCD i mod 8 (amdkfd's mqd_symmetrically_map_cu_mask)?
  for (int which = 0; which < 3; ++which) {
    if (which == 1 && !round_robin) {
      printf("the mask is not dealt round-robin over the XCDs: bits 240..255 would leave XCDs without a CU -- stopping\n");
      return 0;
    }
    hipStream_t s = which == 0 ? sa : (which == 1 ? sb : ua);
    hipLaunchKernelGGL(where, dim3(n_where), dim3(64), 0, s, d_where, 4000);
    CHECK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * n_where);
    CHECK(hipMemcpy(h.data(), d_where, n_where * 8, hipMemcpyDeviceToHost));
    std::set<unsigned> per_xcc[16];
    for (int b = 0; b < n_where; ++b) per_xcc[h[2 * b]].insert(h[2 * b + 1]);
    printf("%-34s distinct CUs per XCD:", which == 0 ? "compute mask (low bits)" : (which == 1 ? "ingest mask (high bits)" : "no mask"));
    int total = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", per_xcc[x].size()); total += (int)per_xcc[x].size(); }
    printf("  (%d)\n", total);
    if (which == 0)
      for (int x = 0; x < 8; ++x) round_robin = round_robin && (int)per_xcc[x].size() == (cus - reserve) / 8;
  }
  // ---- pull rate ----
  const size_t frame = 1920 * 512, N = 64;
  unsigned char *host, *host_dev, *ring;
  CHECK(hipHostMalloc(reinterpret_cast<void**>(&host), N * frame, hipHostMallocMapped));
  memset(host, 7, N * frame);
  CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&host_dev), host, 0));
  CHECK(hipMalloc(&ring, N * frame));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
  const int side = 296;
  const double pull_bytes = (double)N * side * ((side * 3 + 15) / 16 * 16);
  for (int which = 0; which < 2; ++which) {
    hipStream_t s = which == 0 ? ub : sb;
    hipLaunchKernelGGL(pull, dim3(64, N), dim3(256), 0, s, host_dev, ring, side);
    CHECK(hipStreamSynchronize(s));
    auto t = now();
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(pull, dim3(64, N), dim3(256), 0, s, host_dev, ring, side);
    CHECK(hipStreamSynchronize(s));
    const double ms = ms_since(t) / 20;
    printf("pull of 64 x %d^2 px (%.1f MB) on %-12s %7.3f ms  %5.1f GB/s\n", side, pull_bytes * 1e-6, which == 0 ? "all CUs:" : "reserved:", ms,
           pull_bytes / ms * 1e-6);
  }
  // ---- the victim alone, beside the pull on its own CUs, beside the pull on reserved CUs ----
  const unsigned tmask = (1u << 20) - 1;  // 4 MB table: L2-resident
  std::vector<unsigned> table(tmask + 1);
  for (unsigned i = 0; i <= tmask; ++i) table[i] = i * 2654435761u + 12345u;
  unsigned *d_table, *d_out;
  CHECK(hipMalloc(&d_table, table.size() * 4));
  CHECK(hipMalloc(&d_out, 64));
  CHECK(hipMemcpy(d_table, table.data(), table.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(victim), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int rounds = 400;
  for (int mode = 0; mode < 4; ++mode) {
    // 0: victim alone on all CUs; 1: victim on all CUs + pull on all CUs; 2: victim alone on 240 CUs (240 workgroups);
    // 3: victim on 240 CUs + pull on the 16 reserved CUs
    hipStream_t sv = mode < 2 ? ua : sa, sp = mode == 1 ? ub : sb;
    const int wgs = mode < 2 ? cus : cus - reserve;
    const bool with_pull = mode == 1 || mode == 3;
    float best = 1e9f, sum = 0.0f;
    for (int rep = 0; rep < 12; ++rep) {
      if (with_pull)
        for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(pull, dim3(64, N), dim3(256), 0, sp, host_dev, ring, side);
      CHECK(hipEventRecord(e0, sv));
      hipLaunchKernelGGL(victim, dim3(wgs), dim3(512), 100 * 1024, sv, d_table, tmask, rounds, d_out);
      CHECK(hipEventRecord(e1, sv));
      CHECK(hipStreamSynchronize(sv));
      CHECK(hipStreamSynchronize(sp));
      float ms = 0.0f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    static const char* what[4] = {"victim alone, 256 workgroups, no mask", "victim + pull, both on all CUs",
                                  "victim alone on the compute mask, one workgroup per CU", "victim on the compute mask + pull on the reserved CUs"};
    printf("%-52s %7.3f ms (min %7.3f)\n", what[mode], sum / 10, best);
  }
  // ---- is the masked stream's extra time a constant per launch or a factor?  (rounds 100 / 400 / 1600) ----
  for (int r : {100, 400, 1600}) {
    float t[3] = {0, 0, 0};
    for (int which = 0; which < 3; ++which) {  // 0: no mask, 256 workgroups; 1: no mask, 240 workgroups; 2: 240-CU mask, 240 workgroups
      hipStream_t sv = which == 2 ? sa : ua;
      const int wgs = which == 0 ? cus : cus - reserve;
      float sum = 0.0f;
      for (int rep = 0; rep < 8; ++rep) {
        CHECK(hipEventRecord(e0, sv));
        hipLaunchKernelGGL(victim, dim3(wgs), dim3(512), 100 * 1024, sv, d_table, tmask, r, d_out);
        CHECK(hipEventRecord(e1, sv));
        CHECK(hipStreamSynchronize(sv));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) sum += ms;
      }
      t[which] = sum / 6;
    }
    printf("victim, %4d rounds: no mask 256 wgs %7.3f ms | no mask, fewer wgs %7.3f ms | compute mask %7.3f ms\n", r, t[0], t[1], t[2]);
  }
  // ---- variations: a mask that masks nothing; the victim on all CUs beside a pull confined to 16 of them ----
  {
    std::vector<uint32_t> mask_all((cus + 31) / 32, 0xffffffffu);
    hipStream_t sall;
    CHECK(hipExtStreamCreateWithCUMask(&sall, (uint32_t)mask_all.size(), mask_all.data()));
    for (int mode = 0; mode < 3; ++mode) {
      // 0: victim on the all-ones mask; 1: victim unmasked + pull on the 16-CU stream; 2: victim unmasked, 240 workgroups + pull on 16 CUs
      hipStream_t sv = mode == 0 ? sall : ua;
      const int wgs = mode == 2 ? cus - reserve : cus;
      float sum = 0.0f;
      for (int rep = 0; rep < 10; ++rep) {
        if (mode >= 1)
          for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(pull, dim3(64, N), dim3(256), 0, sb, host_dev, ring, side);
        CHECK(hipEventRecord(e0, sv));
        hipLaunchKernelGGL(victim, dim3(wgs), dim3(512), 100 * 1024, sv, d_table, tmask, rounds, d_out);
        CHECK(hipEventRecord(e1, sv));
        CHECK(hipStreamSynchronize(sv));
        CHECK(hipStreamSynchronize(sb));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) sum += ms;
      }
      static const char* what[3] = {"victim on a stream whose mask has all 256 bits set", "victim unmasked (all CUs) + pull confined to the reserved CUs",
                                    "victim unmasked (fewer wgs) + pull confined to the reserved CUs"};
      printf("%-52s %7.3f ms\n", what[mode], sum / 8);
    }
  }
  return 0;
}
