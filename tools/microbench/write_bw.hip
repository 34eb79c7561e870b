// Write-bandwidth calibration for the carrier-expansion kernel: how fast can MI355X
// absorb a pure store stream (4 B / 16 B per lane), and a 1:4 read:write mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void w4(uint32_t* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = (uint32_t)i;
}
__global__ void w16(uint4* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = uint4{(uint32_t)i, 1, 2, 3};
}
// each wave writes one contiguous 4*len-byte segment per iteration with dword stores (like the dense path)
__global__ void wseg(uint32_t* p, size_t nseg, int len) {
  size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  int lane = threadIdx.x & 63;
  for (size_t s = wave; s < nseg; s += nw)
    if (lane < len) p[s * len + lane] = (uint32_t)s;
}
__global__ void copy16(const uint4* a, uint4* b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) b[i] = a[i];
}

int main() {
  size_t bytes = 12ull << 30;
  void *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto f, const char* name, double gb) {
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-28s %8.3f ms  %8.1f GB/s\n", name, ms, gb / (ms * 1e-3));
    return 0;
  };
  double gb = bytes / 1e9;
  time([&] { hipLaunchKernelGGL(w4, dim3(8192), dim3(256), 0, 0, (uint32_t*)a, bytes / 4); }, "store dword/lane", gb);
  time([&] { hipLaunchKernelGGL(w16, dim3(8192), dim3(256), 0, 0, (uint4*)a, bytes / 16); }, "store dwordx4/lane", gb);
  for (int len : {8, 16, 36, 64})
    time([&] { hipLaunchKernelGGL(wseg, dim3(8192), dim3(256), 0, 0, (uint32_t*)a, bytes / 4 / len, len); }, len == 8 ? "seg 8 lanes x4B" : len == 16 ? "seg 16 lanes x4B" : len == 36 ? "seg 36 lanes x4B" : "seg 64 lanes x4B", gb);
  time([&] { hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16); }, "copy dwordx4 (r+w bytes)", 2 * gb);
  time([&] { (void)hipMemsetAsync(a, 0, bytes, 0); }, "hipMemsetAsync", gb);
  return 0;
}
