// What one kernel launch costs on this system, host clock, submit -> a flag in mapped host memory:
//   (a) an empty kernel that only posts the flag,
//   (b) a kernel that first walks a chain of k dependent global loads over a 1 GiB table (cold lines) -- the shape
//       of the region-bounds arithmetic (rank block -> node list -> slot table -> prefix arrays),
//   (c) the same with 1 KiB of kernel arguments.
// hipcc --offload-arch=gfx950 -O3 -o latency_floor latency_floor.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
struct Big { uint64_t v[128]; };
__global__ void k_post(volatile uint64_t* flag, uint64_t seq) {
  if (threadIdx.x == 0) { __threadfence_system(); *flag = seq; }
}
__global__ void k_chain(const uint32_t* tab, uint32_t start, int k, volatile uint64_t* flag, uint64_t seq, uint32_t* sink) {
  uint32_t i = start;
  for (int j = 0; j < k; ++j) i = tab[i];
  if (threadIdx.x == 0) { if (i == 0xFFFFFFFFu) *sink = i; __threadfence_system(); *flag = seq; }
}
// the same chain with one independent chain per lane (vector loads, 64 different lines per hop)
__global__ void k_chain_vec(const uint32_t* tab, uint32_t start, int k, volatile uint64_t* flag, uint64_t seq, uint32_t* sink) {
  uint32_t i = (start + threadIdx.x * 40503u) & ((1u << 28) - 1);
  for (int j = 0; j < k; ++j) i = tab[i];
  if (i == 0xFFFFFFFFu) *sink = i;
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); *flag = seq; }
}
// 800 bytes of arguments, every word of them used (as the query kernels use their image descriptor) ...
struct Desc { uint64_t v[100]; };
__global__ void k_args_used(Desc d, volatile uint64_t* flag, uint64_t seq) {
  uint64_t acc = 0;
  for (int i = 0; i < 100; ++i) acc += d.v[i];
  if (threadIdx.x == 0) { __threadfence_system(); *flag = seq + (acc == 0x12345 ? 1 : 0); }
}
// ... against the same descriptor resident in device memory behind one pointer
__global__ void k_args_indirect(const Desc* d, volatile uint64_t* flag, uint64_t seq) {
  uint64_t acc = 0;
  for (int i = 0; i < 100; ++i) acc += d->v[i];
  if (threadIdx.x == 0) { __threadfence_system(); *flag = seq + (acc == 0x12345 ? 1 : 0); }
}
__global__ void k_big(Big b, volatile uint64_t* flag, uint64_t seq) {
  if (threadIdx.x == 0) { __threadfence_system(); *flag = seq + (b.v[5] & 0); }
}
template <typename F> static void run(const char* name, F launch, volatile uint64_t* flag) {
  std::vector<double> us;
  uint64_t seq = *flag;
  for (int i = 0; i < 700; ++i) {
    ++seq;
    auto t0 = std::chrono::steady_clock::now();
    launch(seq, i);
    while (*flag != seq) __builtin_ia32_pause();
    auto t1 = std::chrono::steady_clock::now();
    if (i >= 100) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
  }
  std::sort(us.begin(), us.end());
  printf("%-44s p50 %6.1f us  p10 %6.1f  p90 %6.1f\n", name, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
}
int main() {
  uint64_t* flag = nullptr;
  hipHostMalloc((void**)&flag, 64, hipHostMallocCoherent | hipHostMallocMapped);
  *flag = 0;
  hipStream_t st;
  hipStreamCreate(&st);
  const size_t N = 1u << 28;   // 1 GiB of uint32
  uint32_t *tab = nullptr, *sink = nullptr;
  hipMalloc((void**)&tab, N * 4);
  hipMalloc((void**)&sink, 4);
  {  // a permutation-ish chain: i -> (i * 2654435761 + 12345) mod N
    std::vector<uint32_t> h(N);
    for (size_t i = 0; i < N; ++i) h[i] = (uint32_t)((i * 2654435761ull + 12345ull) & (N - 1));
    hipMemcpy(tab, h.data(), N * 4, hipMemcpyHostToDevice);
  }
  volatile uint64_t* vf = flag;
  run("empty kernel, 1 block x 64", [&](uint64_t s, int) { hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, vf, s); }, vf);
  run("empty kernel, 8 blocks x 256", [&](uint64_t s, int) { hipLaunchKernelGGL(k_post, dim3(8), dim3(256), 0, st, vf, s); }, vf);
  run("1 KiB of kernel arguments", [&](uint64_t s, int) { Big b{}; hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, st, b, vf, s); }, vf);
  {
    Desc hd{};
    Desc* dd = nullptr;
    hipMalloc((void**)&dd, sizeof(Desc));
    hipMemcpy(dd, &hd, sizeof(Desc), hipMemcpyHostToDevice);
    run("800 B of arguments, all read", [&](uint64_t s, int) { hipLaunchKernelGGL(k_args_used, dim3(1), dim3(64), 0, st, hd, vf, s); }, vf);
    run("800 B descriptor in device memory", [&](uint64_t s, int) { hipLaunchKernelGGL(k_args_indirect, dim3(1), dim3(64), 0, st, (const Desc*)dd, vf, s); }, vf);
  }
  for (int k : {1, 2, 4, 8, 12})
    run(("chain of " + std::to_string(k) + " dependent loads").c_str(),
        [&](uint64_t s, int i) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, st, (const uint32_t*)tab, (uint32_t)((i * 7919u * 4099u) & ((1u << 28) - 1)), k, vf, s, sink); }, vf);
  for (int k : {1, 4, 8})
    run(("per-lane chain of " + std::to_string(k) + " dependent loads").c_str(),
        [&](uint64_t s, int i) { hipLaunchKernelGGL(k_chain_vec, dim3(1), dim3(64), 0, st, (const uint32_t*)tab, (uint32_t)((i * 7919u * 4099u) & ((1u << 28) - 1)), k, vf, s, sink); }, vf);
  run("two dependent empty kernels", [&](uint64_t s, int) {
    hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, vf, s - 1);
    hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, st, vf, s); }, vf);
  return 0;
}
