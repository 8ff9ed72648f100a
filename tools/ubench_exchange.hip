// tools/ubench_exchange.hip — developer micro-benchmark: what one hand-over between two workgroups costs on gfx950,
// for the {tag, value} granule exchange of tracking_step_split_kernel / tracking_step_tree_kernel, and whether the
// cheaper forms (through the XCD's L2 instead of through memory) are VISIBLE to the partner at all.
//
// Two workgroups play ping-pong over two 8-byte granules: A publishes (round, x), B waits for the tag, answers
// (round, x + 1), A waits for the answer.  A's s_memtime over N rounds / N = one round trip = two hand-overs.
// Placement: workgroup b runs on XCD b % 8 (round-robin dispatch), so (0, 8) share an XCD and (0, 1) do not.
//
// Variants (store side / poll side):
//   0  agent-scope relaxed atomic store / agent-scope relaxed atomic load        (what the kernels do today)
//   1  plain store + s_waitcnt        / L2 atomic (fetch_or 0) at workgroup scope (executes in the XCD's L2)
//   2  L2 atomic exchange, wg scope   / L2 atomic (fetch_or 0) at workgroup scope
//   3  agent-scope atomic store       / L2 atomic (fetch_or 0) at workgroup scope
//   4  plain store                    / buffer_inv sc1 + plain load
//   5  L2 atomic exchange, agent scope / L2 atomic fetch_or 0, agent scope
//   6  L2 atomic exchange             / global_atomic_or_x2 ... sc0 with data 0, written in asm
//   7  agent-scope atomic store       / the same
//   8  plain store + s_waitcnt        / the same
// (hipcc folds an atomic fetch_or with 0 into a LOAD with the scope's sc bits -- global_load_dwordx2 ... sc0 at
// workgroup scope, which may hit in the polling CU's own L1 for ever; variants 1-3 and 5 therefore measure loads, and
// only 6-8 poll with a read-modify-write that has to execute in the L2.)
// A variant whose polls never see the partner's store within the spin limit reports FAILED.
// Measured (profiles/r03_ubench_exchange.txt): every working form costs ~2 000 ticks per round trip, same XCD or not
// (1 960 vs 2 120 for variant 0) -- agent-scope stores, loads and L2 read-modify-writes all travel to the memory side;
// the workgroup-scope polls (1-3) never see the partner, on the same XCD either.  There is no cheaper hand-over.
//
//   hipcc --offload-arch=gfx950 -O3 -o ubench_exchange tools/ubench_exchange.hip && ./ubench_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64* GlobalU64;

__device__ __forceinline__ u64 l2_atomic_peek(u64* p) {
  u64 out;
  const u64 zero = 0ull;
  asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(out) : "v"(p), "v"(zero) : "memory");
  return out;
}

template <int V>
__device__ __forceinline__ void publish(u64* p, u64 v) {
  if (V == 0 || V == 3 || V == 7) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (V == 1 || V == 4 || V == 8) {
    *(volatile u64*)p = v;
    __builtin_amdgcn_s_waitcnt(0);
  } else if (V == 2) {
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <int V>
__device__ __forceinline__ u64 peek(u64* p) {
  if (V == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (V == 1 || V == 2 || V == 3) return __hip_atomic_fetch_or(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (V == 4) {
    asm volatile("buffer_inv sc1" ::: "memory");
    return *(volatile u64*)p;
  }
  if (V >= 6) return l2_atomic_peek(p);
  return __hip_atomic_fetch_or(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lanes: how many lanes of the first wave take part (each with its own pair of granules, 8 bytes apart: the real
// exchange moves up to 8 granules per thread)
template <int V>
__global__ void __launch_bounds__(64)
pingpong_kernel(u64* granules, int block_a, int block_b, int rounds, int lanes, unsigned tag0, u64* cycles, int* xcc,
                int* failed) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b != block_a && b != block_b) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (lane == 0) xcc[b == block_a ? 0 : 1] = (int)(id & 0xf);
  if (lane >= lanes) return;
  u64* ping = granules + lane;
  u64* pong = granules + 64 + lane;
  const u64 t0 = __builtin_amdgcn_s_memtime();
  bool lost = false;
  for (int r = 0; r < rounds && !lost; ++r) {
    const u64 tag = (u64)(tag0 + (unsigned)r + 1u) << 32;
    if (b == block_a) {
      publish<V>(ping, tag | (unsigned)r);
      unsigned spins = 0;
      u64 v = peek<V>(pong);
      while ((v >> 32) != (tag >> 32)) {
        if (++spins > (1u << 16)) { lost = true; break; }
        v = peek<V>(pong);
      }
      if (!lost && (unsigned)v != (unsigned)r + 1u) lost = true;
    } else {
      unsigned spins = 0;
      u64 v = peek<V>(ping);
      while ((v >> 32) != (tag >> 32)) {
        if (++spins > (1u << 16)) { lost = true; break; }
        v = peek<V>(ping);
      }
      if (!lost) publish<V>(pong, tag | ((unsigned)v + 1u));
    }
  }
  const u64 t1 = __builtin_amdgcn_s_memtime();
  if (lost) atomicAdd(failed, 1);
  if (b == block_a && lane == 0) cycles[0] = t1 - t0;
}

template <int V>
void run(const char* what, u64* granules, u64* cycles, int* xcc, int* failed) {
  static unsigned tag0 = 0;
  for (int partner : {8, 1}) {
    for (int lanes : {1, 64}) {
      const int rounds = 2000;
      CHECK(hipMemset(cycles, 0, 8));
      CHECK(hipMemset(failed, 0, 4));
      hipLaunchKernelGGL(pingpong_kernel<V>, dim3(16), dim3(64), 0, 0, granules, 0, partner, rounds, lanes, tag0, cycles, xcc,
                         failed);
      CHECK(hipDeviceSynchronize());
      tag0 += rounds + 8;
      u64 c = 0;
      int x[2] = {0, 0}, f = 0;
      CHECK(hipMemcpy(&c, cycles, 8, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
      // (ticks of s_memtime: the unit of tools/phase_timing.py, 2.4 GHz-equivalent per tools/ubench.hip)
      printf("variant %d %-58s blocks (0,%d) XCC (%d,%d) lanes %2d : %s %8.1f ticks / round trip\n", V, what, partner, x[0], x[1],
             lanes, f ? "FAILED" : "ok    ", (double)c / rounds);
    }
  }
}

int main() {
  u64 *granules, *cycles;
  int *xcc, *failed;
  CHECK(hipMalloc(&granules, 128 * 8));
  CHECK(hipMemset(granules, 0, 128 * 8));
  CHECK(hipMalloc(&cycles, 8));
  CHECK(hipMalloc(&xcc, 8));
  CHECK(hipMalloc(&failed, 4));
  run<0>("agent store / agent load", granules, cycles, xcc, failed);
  run<1>("plain store / wg-scope L2 atomic", granules, cycles, xcc, failed);
  run<2>("wg-scope L2 xchg / wg-scope L2 atomic", granules, cycles, xcc, failed);
  run<3>("agent store / wg-scope L2 atomic", granules, cycles, xcc, failed);
  run<4>("plain store / buffer_inv sc1 + load", granules, cycles, xcc, failed);
  run<5>("agent L2 xchg / agent L2 atomic", granules, cycles, xcc, failed);
  run<6>("L2 xchg / asm global_atomic_or_x2 sc0", granules, cycles, xcc, failed);
  run<7>("agent store / asm global_atomic_or_x2 sc0", granules, cycles, xcc, failed);
  run<8>("plain store / asm global_atomic_or_x2 sc0", granules, cycles, xcc, failed);
  return 0;
}
