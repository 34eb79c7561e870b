// Dependent random-load latency against footprint (one wave, one lane chasing): what a cold access costs once the
// working set outgrows the L2 / MALL / the TLBs' reach.  hipcc --offload-arch=gfx950 -O3 chase_latency.hip -o chase_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
__global__ void chase(const uint64_t* buf, uint64_t start, int steps, uint64_t* out) {
  uint64_t i = start;
  const uint64_t t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) i = buf[i];
  const uint64_t t1 = wall_clock64();
  out[0] = t1 - t0; out[1] = i;
}
// many independent chasers at once (lanes x waves): latency under load
__global__ void chase_many(const uint64_t* buf, uint64_t n, int steps, uint64_t* out) {
  uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ULL % n;
  const uint64_t t0 = wall_clock64();
  for (int s = 0; s < steps; ++s) i = buf[i];
  const uint64_t t1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (i == 0xFFFFFFFFFFFFull) out[1] = i;
}
int main() {
  uint64_t* out; hipMalloc(&out, 16);
  for (uint64_t mb : {16ull, 256ull, 1024ull, 4096ull, 12288ull, 32768ull}) {
    const uint64_t n = mb * (1ull << 20) / 8;
    uint64_t* d; if (hipMalloc(&d, n * 8) != hipSuccess) { printf("alloc %llu MB failed\n", (unsigned long long)mb); continue; }
    // a random cyclic permutation with stride >= one page between successive elements: i -> (i * a + c) mod n is not a single cycle in
    // general, so build the chain on the host over a subsample of 1M nodes spread over the whole buffer
    const uint64_t nodes = 1 << 20;
    std::vector<uint64_t> pos(nodes);
    std::mt19937_64 rng(mb);
    for (auto& p : pos) p = rng() % n;
    std::vector<uint64_t> h(n > (1ull << 27) ? 0 : 0);
    hipMemset(d, 0, n * 8);
    // write only the chain's nodes
    for (uint64_t k = 0; k < nodes; ++k) { uint64_t nxt = pos[(k + 1) % nodes]; hipMemcpy(d + pos[k], &nxt, 8, hipMemcpyHostToDevice); if (k > 20000) break; }
    // (20k nodes are enough: the chase below takes 10k steps)
    uint64_t last = pos[0]; hipMemcpy(d + pos[20001 % nodes], &last, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, pos[0], 10000, out);
    uint64_t r[2]; hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, pos[0], 10000, out);
    uint64_t r2[2]; hipMemcpy(r2, out, 16, hipMemcpyDeviceToHost);
    printf("%6llu MB: single chaser %.0f ns per dependent load (second pass over the same 10k nodes: %.0f ns)\n", (unsigned long long)mb, r[0] * 10.0 / 10000, r2[0] * 10.0 / 10000);
    hipFree(d);
  }
  return 0;
}
