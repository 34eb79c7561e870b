// What does the HBM give a streaming kernel whose traffic is shaped like the carrier expansion's -- 41 % reads, 59 %
// writes (1.05 GB in, 1.49 GB out per launch by the counters), non-temporal 16-byte stores?  Reference points for
// roofline.frac: pure read, pure write, copy (1:1) and the 2:3 read:write mix, each streaming through 2 - 6 GB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// per thread and iteration: R 16-byte loads (consecutive across the wave), W 16-byte stores
template <int R, int W, bool NT>
__global__ void __launch_bounds__(256) mix(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t iters, u32x4* sink) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
  u32x4 acc = {0, 0, 0, 0};
  for (size_t i = tid; i < iters; i += nt) {
    u32x4 v[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = src[(size_t)r * iters + i];
#pragma unroll
    for (int r = 0; r < R; ++r) acc += v[r];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const u32x4 o = acc + (unsigned)w;
      if (NT) __builtin_nontemporal_store(o, &dst[(size_t)w * iters + i]); else dst[(size_t)w * iters + i] = o;
    }
  }
  if (W == 0 && acc.x == 0x12345u) *sink = acc;   // keep the loads of the read-only form
}

template <int R, int W, bool NT>
static int run(const char* name, const u32x4* src, u32x4* dst, u32x4* sink, size_t iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {2048, 8192, 32768}) {
    auto f = [&] { hipLaunchKernelGGL((mix<R, W, NT>), dim3(blocks), dim3(256), 0, 0, src, dst, iters, sink); };
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double bytes = (double)iters * 16 * (R + W);
    printf("%-34s grid %6d: %7.3f ms  %7.1f GB/s (%.2f GB read, %.2f GB written)\n", name, blocks, ms, bytes / 1e9 / (ms * 1e-3),
           (double)iters * 16 * R / 1e9, (double)iters * 16 * W / 1e9);
  }
  return 0;
}

int main() {
  const size_t iters = 64ull << 20;   // 1 GiB per stream of 16-byte elements
  u32x4 *src, *dst, *sink;
  CK(hipMalloc(&src, iters * 16 * 3)); CK(hipMalloc(&dst, iters * 16 * 3)); CK(hipMalloc(&sink, 16));
  CK(hipMemset(src, 1, iters * 16 * 3));
  if (run<1, 0, false>("read only", src, dst, sink, iters)) return 1;
  if (run<0, 1, false>("write only", src, dst, sink, iters)) return 1;
  if (run<0, 1, true>("write only, non-temporal", src, dst, sink, iters)) return 1;
  if (run<1, 1, false>("copy 1:1", src, dst, sink, iters)) return 1;
  if (run<1, 1, true>("copy 1:1, non-temporal stores", src, dst, sink, iters)) return 1;
  if (run<2, 3, false>("mix 2:3 (the expansion's shape)", src, dst, sink, iters)) return 1;
  if (run<2, 3, true>("mix 2:3, non-temporal stores", src, dst, sink, iters)) return 1;
  return 0;
}
