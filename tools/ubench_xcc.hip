// tools/ubench_xcc.hip -- developer probe: on which XCD (HW_REG_XCC_ID) and CU does workgroup b of a 256-workgroup,
// one-workgroup-per-CU launch run?  The split kernels place the workgroups of one object at b = x + 8 j, assuming the
// dispatcher deals workgroups round-robin over the 8 XCDs (so that they share one L2).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_xcc tools/ubench_xcc.hip && ./ubench_xcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(512) probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
  // stay resident for a while so that all workgroups of the grid are placed at once
  float v = lds[threadIdx.x % 16];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) out[0] = 0;
}
int main() {
  const int n = 256;
  unsigned* d;
  hipMalloc(&d, n * 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(512), 100 * 1024, 0, d, 20000);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int b = 0; b < n; ++b) ok += (h[2 * b] & 0xf) == (unsigned)((b + (h[0] & 0xf)) % 8);
    printf("launch %d: XCC_ID of blocks 0..15:", rep);
    for (int b = 0; b < 16; ++b) printf(" %u", h[2 * b] & 0xf);
    printf("   blocks with xcc == (b + xcc0) mod 8: %d / %d\n", ok, n);
  }
  return 0;
}
