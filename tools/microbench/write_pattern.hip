// How does the order in which waves cover an output arena affect write bandwidth?  Every wave writes `run`
// consecutive 1 KiB blocks (16 B per lane per block) starting at a pseudo-random run index -- the shape of
// k_fill_carriers' output (each wave walks its own ~35 KB window) -- against the streaming order of a memset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void wruns(uint4* p, size_t nruns, int run, int scramble, const uint4* src) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (size_t r = wave; r < nruns; r += nw) {
    size_t rr = r;
    if (scramble) rr = (r * 2654435761ull + 12345) % nruns;   // nruns is a power of two: odd multiplier = bijection
    uint4 v = uint4{(uint32_t)r, 1, 2, 3};
    if (src) v = src[(rr * run) * 64 / 8 + lane];              // optional read stream, 1/8 of the written bytes
    for (int b = 0; b < run; ++b) p[(rr * run + b) * 64 + lane] = v;
  }
}

int main() {
  const size_t bytes = 8ull << 30;
  void *a, *s;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&s, bytes / 8 + (1 << 20)));
  CK(hipMemset(s, 1, bytes / 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int with_read = 0; with_read < 2; ++with_read)
    for (int scramble = 0; scramble < 2; ++scramble)
      for (int run : {1, 2, 4, 8, 16, 32, 33, 35, 64, 100}) {
        size_t nruns = bytes / 1024 / run; if (scramble) { size_t p2 = 1; while (p2 * 2 <= nruns) p2 *= 2; nruns = p2; }
        auto f = [&] { hipLaunchKernelGGL(wruns, dim3(8192), dim3(256), 0, 0, (uint4*)a, nruns, run, scramble, with_read ? (const uint4*)s : nullptr); };
        f(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 3; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
        printf("read=%d scramble=%d run=%3d KiB per wave: %7.3f ms  %7.1f GB/s written\n", with_read, scramble, run, ms, nruns * run * 1024 / 1e9 / (ms * 1e-3));
      }
  return 0;
}
