// tools/ubench_ingest.hip — developer micro-benchmark for ROI ingest: three ways to bring the part of
// 64 camera frames (640 x 512 BGR8, one page-locked block) that the trackers read into the frame ring:
//   full     one hipMemcpyAsync of the whole block (what m3t_hip_cameras_upload_batch_async does today)
//   2d       one hipMemcpy2DAsync per camera for its rectangle
//   pull     ONE kernel that reads the rectangles straight from the mapped host block over PCIe and writes the ring
//   2d/8     the per-camera 2-D copies spread over 8 streams (do several SDMA queues run them side by side?)
//   graph    the 64 2-D copies as memcpy nodes of ONE hipGraph without edges between them
// for rectangles of 128^2 ... 512^2 pixels.  Prints ms per batch-frame and the frames/s x 64 objects they allow.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_ingest tools/ubench_ingest.hip && ./ubench_ingest
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int W = 640, H = 512, BPP = 3, PITCH = W * BPP, N = 64;
constexpr size_t FRAME = (size_t)PITCH * H;

struct Rect { int x0, y0, x1, y1; };

// grid: (rows of 16-byte chunks) -- one workgroup per (camera, 8 rows); a thread moves 16 bytes at a time
__global__ void __launch_bounds__(256)
pull_kernel(const unsigned char* __restrict__ host_block, unsigned char* __restrict__ ring, const Rect* rects) {
  const int cam = blockIdx.y;
  const Rect r = rects[cam];
  const int row0 = r.y0 + blockIdx.x * 8;
  // widen the span to 16-byte boundaries of the row (rows start 16-byte aligned: 1920 = 120 x 16)
  const int b0 = (r.x0 * BPP) & ~15, b1 = (r.x1 * BPP + 15) & ~15;
  const int chunks = (b1 - b0) >> 4;
  for (int i = threadIdx.x; i < 8 * chunks; i += 256) {
    const int row = row0 + i / chunks, c = i - (i / chunks) * chunks;
    if (row >= r.y1) break;
    const size_t at = (size_t)cam * FRAME + (size_t)row * PITCH + b0 + ((size_t)c << 4);
    *reinterpret_cast<uint4*>(ring + at) = *reinterpret_cast<const uint4*>(host_block + at);
  }
}

// pull variants (round 6): ROWS rows per workgroup, K 16-byte loads in flight per thread, spans widened to ALIGN bytes
template <int ROWS, int K, int ALIGN>
__global__ void __launch_bounds__(256)
pull_variant(const unsigned char* __restrict__ host_block, unsigned char* __restrict__ ring, const Rect* rects) {
  const int cam = blockIdx.y;
  const Rect r = rects[cam];
  const int row0 = r.y0 + blockIdx.x * ROWS;
  if (row0 >= r.y1) return;
  const int b0 = (r.x0 * BPP) & ~(ALIGN - 1);
  int b1 = (r.x1 * BPP + ALIGN - 1) & ~(ALIGN - 1);
  if (b1 > PITCH) b1 = PITCH;
  const int chunks = (b1 - b0) >> 4;
  const int rows = (r.y1 - row0) < ROWS ? (r.y1 - row0) : ROWS;
  const int total = rows * chunks;
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * K) {
    uint4 v[K];
    size_t at[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = i0 + k * 256;
      const int row = i / chunks, c = i - row * chunks;
      at[k] = (size_t)cam * FRAME + (size_t)(row0 + row) * PITCH + b0 + ((size_t)c << 4);
      if (i < total) v[k] = *reinterpret_cast<const uint4*>(host_block + at[k]);
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (i0 + k * 256 < total) *reinterpret_cast<uint4*>(ring + at[k]) = v[k];
  }
}

// The ceiling of a kernel's reads from mapped host memory (round 6): the whole block streamed by `wgs` workgroups, every
// thread K 16-byte loads in flight (a wave asks for K x 1 KB of consecutive host memory at a time), consecutive or with
// the loads of a thread 128 bytes apart (WIDE: two 64-byte halves of a 128-byte line per pair of loads)
template <int K>
__global__ void __launch_bounds__(256)
stream_kernel(const uint4* __restrict__ host_block, uint4* __restrict__ ring, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * K;
  for (size_t base = (size_t)blockIdx.x * 256 * K + threadIdx.x; base < n16; base += stride) {
    uint4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = base + (size_t)k * 256 < n16 ? host_block[base + (size_t)k * 256] : uint4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < K; ++k)
      if (base + (size_t)k * 256 < n16) ring[base + (size_t)k * 256] = v[k];
  }
}

int main() {
  unsigned char *host, *host_dev, *ring;
  Rect* d_rects;
  CHECK(hipHostMalloc(reinterpret_cast<void**>(&host), N * FRAME, hipHostMallocMapped));
  memset(host, 7, N * FRAME);
  CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&host_dev), host, 0));
  CHECK(hipMalloc(&ring, N * FRAME));
  CHECK(hipMalloc(&d_rects, N * sizeof(Rect)));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms_since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(now() - t).count();
  };
  const int reps = 20;
  {
    CHECK(hipMemcpyAsync(ring, host, N * FRAME, hipMemcpyHostToDevice, s));
    CHECK(hipStreamSynchronize(s));
    auto t = now();
    for (int k = 0; k < reps; ++k) CHECK(hipMemcpyAsync(ring, host, N * FRAME, hipMemcpyHostToDevice, s));
    CHECK(hipStreamSynchronize(s));
    const double ms = ms_since(t) / reps;
    printf("full frames             %7.3f ms per batch-frame  %6.1f GB/s  -> %8.0f pose-updates/s\n", ms, N * FRAME / ms * 1e-6,
           N / ms * 1e3);
  }
  {  // what the link gives a KERNEL: the same 62.9 MB streamed from the mapped block, by workgroups x loads in flight
    const size_t n16 = N * FRAME / 16;
    for (int wgs : {64, 256, 1024, 4096}) {
      double ms[4];
      int col = 0;
#define STREAM(K)                                                                                                   \
      {                                                                                                             \
        hipLaunchKernelGGL(stream_kernel<K>, dim3(wgs), dim3(256), 0, s, (const uint4*)host_dev, (uint4*)ring, n16); \
        CHECK(hipStreamSynchronize(s));                                                                             \
        auto t = now();                                                                                             \
        for (int k = 0; k < reps; ++k)                                                                              \
          hipLaunchKernelGGL(stream_kernel<K>, dim3(wgs), dim3(256), 0, s, (const uint4*)host_dev, (uint4*)ring, n16); \
        CHECK(hipStreamSynchronize(s));                                                                             \
        ms[col++] = ms_since(t) / reps;                                                                             \
      }
      STREAM(1) STREAM(2) STREAM(4) STREAM(8)
#undef STREAM
      printf("kernel stream, %4d workgroups x 256 threads: 16 B x {1, 2, 4, 8} loads in flight per thread: %5.1f / %5.1f / %5.1f / %5.1f GB/s\n",
             wgs, N * FRAME / ms[0] * 1e-6, N * FRAME / ms[1] * 1e-6, N * FRAME / ms[2] * 1e-6, N * FRAME / ms[3] * 1e-6);
    }
  }
  for (int side : {128, 192, 256, 320, 384, 512}) {
    Rect rects[N];
    for (int i = 0; i < N; ++i) {
      const int x0 = (i * 37) % (W - (side < W ? side : W) + 1), y0 = (i * 53) % (H - (side < H ? side : H) + 1);
      rects[i] = {x0, y0, x0 + (side < W ? side : W), y0 + (side < H ? side : H)};
    }
    CHECK(hipMemcpy(d_rects, rects, sizeof rects, hipMemcpyHostToDevice));
    size_t bytes = 0;
    for (int i = 0; i < N; ++i) bytes += (size_t)(rects[i].x1 - rects[i].x0) * BPP * (rects[i].y1 - rects[i].y0);
    double ms2d, mspull;
    {
      auto run = [&] {
        for (int i = 0; i < N; ++i) {
          const Rect& r = rects[i];
          const size_t at = (size_t)i * FRAME + (size_t)r.y0 * PITCH + (size_t)r.x0 * BPP;
          CHECK(hipMemcpy2DAsync(ring + at, PITCH, host + at, PITCH, (size_t)(r.x1 - r.x0) * BPP, r.y1 - r.y0,
                                 hipMemcpyHostToDevice, s));
        }
      };
      run();
      CHECK(hipStreamSynchronize(s));
      auto t = now();
      for (int k = 0; k < reps; ++k) run();
      CHECK(hipStreamSynchronize(s));
      ms2d = ms_since(t) / reps;
    }
    double ms2d8 = 0.0, msgraph = 0.0;
    {
      static hipStream_t many[8];
      static bool made = false;
      if (!made) { for (auto& st : many) CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); made = true; }
      auto run = [&] {
        for (int i = 0; i < N; ++i) {
          const Rect& r = rects[i];
          const size_t at = (size_t)i * FRAME + (size_t)r.y0 * PITCH + (size_t)r.x0 * BPP;
          CHECK(hipMemcpy2DAsync(ring + at, PITCH, host + at, PITCH, (size_t)(r.x1 - r.x0) * BPP, r.y1 - r.y0,
                                 hipMemcpyHostToDevice, many[i & 7]));
        }
      };
      run();
      for (auto& st : many) CHECK(hipStreamSynchronize(st));
      auto t = now();
      for (int k = 0; k < reps; ++k) run();
      for (auto& st : many) CHECK(hipStreamSynchronize(st));
      ms2d8 = ms_since(t) / reps;
    }
    {
      hipGraph_t graph;
      CHECK(hipGraphCreate(&graph, 0));
      for (int i = 0; i < N; ++i) {
        const Rect& r = rects[i];
        const size_t at = (size_t)i * FRAME + (size_t)r.y0 * PITCH + (size_t)r.x0 * BPP;
        hipMemcpy3DParms p{};
        p.srcPtr = make_hipPitchedPtr(host + at, PITCH, W, H);
        p.dstPtr = make_hipPitchedPtr(ring + at, PITCH, W, H);
        p.extent = make_hipExtent((size_t)(r.x1 - r.x0) * BPP, r.y1 - r.y0, 1);
        p.kind = hipMemcpyHostToDevice;
        hipGraphNode_t node;
        CHECK(hipGraphAddMemcpyNode(&node, graph, nullptr, 0, &p));
      }
      hipGraphExec_t exec;
      CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      CHECK(hipGraphLaunch(exec, s));
      CHECK(hipStreamSynchronize(s));
      auto t = now();
      for (int k = 0; k < reps; ++k) CHECK(hipGraphLaunch(exec, s));
      CHECK(hipStreamSynchronize(s));
      msgraph = ms_since(t) / reps;
      CHECK(hipGraphExecDestroy(exec));
      CHECK(hipGraphDestroy(graph));
    }
    {
      const int rows = rects[0].y1 - rects[0].y0;
      auto run = [&] { hipLaunchKernelGGL(pull_kernel, dim3((rows + 7) / 8, N), dim3(256), 0, s, host_dev, ring, d_rects); };
      run();
      CHECK(hipStreamSynchronize(s));
      auto t = now();
      for (int k = 0; k < reps; ++k) run();
      CHECK(hipStreamSynchronize(s));
      mspull = ms_since(t) / reps;
    }
    {
      const int rows = rects[0].y1 - rects[0].y0;
      double v[6];
      int col = 0;
#define VARIANT(ROWS, K, ALIGN)                                                                                        \
      {                                                                                                                \
        auto run = [&] {                                                                                               \
          hipLaunchKernelGGL((pull_variant<ROWS, K, ALIGN>), dim3((rows + ROWS - 1) / ROWS, N), dim3(256), 0, s, host_dev, ring, d_rects); \
        };                                                                                                             \
        run();                                                                                                         \
        CHECK(hipStreamSynchronize(s));                                                                                \
        auto t = now();                                                                                                \
        for (int k = 0; k < reps; ++k) run();                                                                          \
        CHECK(hipStreamSynchronize(s));                                                                                \
        v[col++] = bytes / (ms_since(t) / reps) * 1e-6;                                                                \
      }
      VARIANT(8, 2, 16) VARIANT(8, 2, 64) VARIANT(16, 4, 64) VARIANT(32, 4, 64) VARIANT(32, 8, 64) VARIANT(64, 8, 128)
#undef VARIANT
      printf("               pull variants (rows per workgroup, loads in flight, span alignment), GB/s of rectangle bytes: "
             "8/2/16 %5.1f  8/2/64 %5.1f  16/4/64 %5.1f  32/4/64 %5.1f  32/8/64 %5.1f  64/8/128 %5.1f\n", v[0], v[1], v[2], v[3], v[4], v[5]);
    }
    printf("rect %3d^2 (%5.1f %% of the frames)  2d x 64: %7.3f ms (%5.1f GB/s, %8.0f pose-updates/s)   pull kernel: %7.3f ms (%5.1f GB/s, "
           "%8.0f pose-updates/s)\n", side, 100.0 * bytes / (N * FRAME), ms2d, bytes / ms2d * 1e-6, N / ms2d * 1e3, mspull,
           bytes / mspull * 1e-6, N / mspull * 1e3);
    printf("               2d x 64 over 8 streams: %7.3f ms (%5.1f GB/s)   64 memcpy nodes in one graph: %7.3f ms (%5.1f GB/s)\n",
           ms2d8, bytes / ms2d8 * 1e-6, msgraph, bytes / msgraph * 1e-6);
  }
  return 0;
}
