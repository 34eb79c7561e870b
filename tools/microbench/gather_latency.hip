// Latency of a DEPENDENT 64-lane gather against the spread of its addresses: every lane chases its own chain; the chains
// of a wave live in a window of `spread` bytes that moves randomly through a large buffer from step to step.
// Tells apart "cold line" cost from address-translation cost (lanes in one page vs in 64 pages).
// hipcc --offload-arch=gfx950 -O3 gather_latency.hip -o gather_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// buf[i] holds a random 64-bit value; lane address(step) = window_base(step, wave) + lane_offset; window_base depends on the loaded value (dependency)
__global__ void gather(const uint64_t* buf, uint64_t n_words, uint64_t spread_words, int steps, uint64_t* out) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint64_t v = mix(wave + 1);
  const uint64_t t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) {
    const uint64_t base = (mix(v) % (n_words - spread_words));      // wave-uniform? no: v differs per lane after step 0 -> use lane 0's
    const uint64_t b0 = __shfl(base, 0, 64);
    const uint64_t idx = b0 + (mix(v + lane) % spread_words);
    v = buf[idx] + s;
  }
  const uint64_t t1 = wall_clock64();
  if (lane == 0) out[wave] = t1 - t0;
  if (v == 0x1234567) out[0] = v;
}
int main() {
  const uint64_t gb = 12;
  const uint64_t n = gb * (1ull << 30) / 8;
  uint64_t* d; hipMalloc(&d, n * 8); hipMemset(d, 0x5a, n * 8);
  uint64_t* out; hipMalloc(&out, 8 * 65536);
  const int steps = 2000;
  for (int waves : {1, 1024, 8192}) {
    for (uint64_t spread : {64ull, 4096ull, 65536ull, 2ull << 20, 64ull << 20, 4096ull << 20}) {
      hipMemset(out, 0, 8 * 65536);
      hipLaunchKernelGGL(gather, dim3(waves), dim3(64), 0, 0, d, n, spread / 8, steps, out);
      hipDeviceSynchronize();
      static uint64_t h[65536];
      hipMemcpy(h, out, 8 * waves, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < waves; ++i) s += h[i];
      printf("waves %5d  window %10llu B: %.0f ns per dependent 64-lane gather\n", waves, (unsigned long long)spread, s / waves * 10.0 / steps);
    }
  }
  return 0;
}
