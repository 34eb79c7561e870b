// What does an LDS store cost by width and alignment?  Every wave of a full machine (32 waves per CU) issues `iters` stores
// per lane into its own 4 KiB of LDS; lane l writes at byte offset (l * stride + mis + 8 * (i & 7)):
//   b16 a        ds_write_b16, lanes 20 bytes apart (the peel loop's shape: one id per set bit)
//   b64 aligned  ds_write_b64 at 8-byte-aligned addresses, lanes 24 bytes apart
//   b64 +2       the same addresses + 2 (2-byte-aligned only: what a nibble-table peel would store)
//   b64 +4       the same + 4
//   b32 +2       ds_write_b32 at 2-byte alignment
// and checks that a misaligned store puts its bytes where a byte-wise copy would.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k_store(uint32_t* out, int iters, int stride, int mis) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint8_t* base = lds + wv * 4096;
  uint32_t a = lane * stride + mis;
  uint64_t v = ((uint64_t)(lane * 0x01010101u) << 32) | (uint32_t)(blockIdx.x + 7);
  for (int i = 0; i < iters; ++i) {
    uint8_t* p = base + a + 8 * (i & 7);
    if (MODE == 0) { uint16_t x = (uint16_t)v; __builtin_memcpy(p, &x, 2); }
    else if (MODE == 1) { __builtin_memcpy(p, &v, 8); }
    else { uint32_t x = (uint32_t)v; __builtin_memcpy(p, &x, 4); }
    v += 0x0001000100010001ull;
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = reinterpret_cast<uint32_t*>(lds)[threadIdx.x];
}

// correctness: one wave, lane l stores the 8 bytes {l, l+1, ..} at offset l * 24 + mis over a zeroed block; the host checks the bytes
__global__ void k_check(uint8_t* out, int mis) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[2048];
  const int lane = threadIdx.x;
  for (int i = lane; i < 2048 / 4; i += 64) reinterpret_cast<uint32_t*>(lds)[i] = 0;
  __syncthreads();
  uint64_t v = 0;
  for (int b = 0; b < 8; ++b) v |= (uint64_t)((lane + b + 1) & 0xFF) << (8 * b);
  __builtin_memcpy(lds + lane * 24 + mis, &v, 8);
  __syncthreads();
  for (int i = lane; i < 2048; i += 64) out[i] = lds[i];
}

int main() {
  uint32_t* out; CK(hipMalloc(&out, 8192ull * 256 * 4));
  uint8_t* chk; CK(hipMalloc(&chk, 2048));
  for (int mis : {0, 2, 4, 6, 1}) {
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, chk, mis);
    std::vector<uint8_t> h(2048);
    CK(hipMemcpy(h.data(), chk, 2048, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int b = 0; b < 8; ++b) if (h[l * 24 + mis + b] != ((l + b + 1) & 0xFF)) ++bad;
    printf("misalignment %d: %s (%d wrong bytes)\n", mis, bad ? "WRONG" : "bytes land where a byte copy would put them", bad);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4096;
  auto run = [&](const char* name, int mode, int stride, int mis) {
    auto f = [&] {
      if (mode == 0) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_store<0>), dim3(2048), dim3(256), 16384, 0, out, iters, stride, mis);
      else if (mode == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_store<1>), dim3(2048), dim3(256), 16384, 0, out, iters, stride, mis);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_store<2>), dim3(2048), dim3(256), 16384, 0, out, iters, stride, mis);
    };
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // 2048 blocks x 4 waves = 8192 waves = 32 per CU, one round; per CU: 32 waves x iters wave-instructions
    const double per_cu_instr = 32.0 * iters, cycles = ms * 1e-3 * 2.4e9;
    printf("%-22s stride %2d mis %d: %7.3f ms  = %6.1f cycles (at 2.4 GHz) per wave-instruction per CU\n", name, stride, mis, ms, cycles / per_cu_instr);
    return 0;
  };
  run("ds_write_b16", 0, 20, 0);
  run("ds_write_b16", 0, 2, 0);
  run("ds_write_b64 aligned", 1, 24, 0);
  run("ds_write_b64 aligned", 1, 8, 0);
  run("ds_write_b64 +2", 1, 24, 2);
  run("ds_write_b64 +4", 1, 24, 4);
  run("ds_write_b64 +6", 1, 24, 6);
  run("ds_write_b64 +2 dense", 1, 8, 2);
  run("ds_write_b32 aligned", 2, 20, 0);
  run("ds_write_b32 +2", 2, 20, 2);
  return 0;
}
