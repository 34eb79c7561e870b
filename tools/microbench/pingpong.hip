// Round trip host -> resident kernel -> host without a launch in between: the host posts a sequence number, one
// polling wave answers by posting it back into mapped host memory.  Two request mailboxes are compared:
//   (a) mapped HOST memory (the GPU polls across PCIe),
//   (b) fine-grained DEVICE memory the CPU writes through the PCIe BAR (the GPU polls its own HBM).
// Every device loop is bounded by the device clock (s_memrealtime, 100 MHz): the kernel leaves after `budget_ticks`
// whatever the host does.   hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_server(uint64_t* req, uint64_t* resp, uint64_t budget_ticks) {
  const uint64_t t0 = wall_clock64();
  uint64_t seen = 0;
  while (wall_clock64() - t0 < budget_ticks) {
    const uint64_t s = __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (s == ~0ULL) break;                       // exit request
    if (s != seen) {
      seen = s;
      __hip_atomic_store(resp, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __hip_atomic_store(resp, ~0ULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // gone
}
static int run(const char* name, uint64_t* req_host_view, uint64_t* req_dev_view, uint64_t* resp) {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *(volatile uint64_t*)req_host_view = 0;
  *(volatile uint64_t*)resp = 0;
  hipLaunchKernelGGL(k_server, dim3(1), dim3(64), 0, st, req_dev_view, resp, (uint64_t)(2.0 * 100e6));   // 2 s budget
  CK(hipGetLastError());
  std::vector<double> us;
  bool ok = true;
  for (uint64_t s = 1; s <= 3000 && ok; ++s) {
    auto t0 = std::chrono::steady_clock::now();
    *(volatile uint64_t*)req_host_view = s;
    while (*(volatile uint64_t*)resp != s) {
      __builtin_ia32_pause();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { ok = false; break; }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (s > 500) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
  }
  *(volatile uint64_t*)req_host_view = ~0ULL;
  CK(hipStreamSynchronize(st));
  if (!ok) { printf("%-40s no answer within 200 ms\n", name); return 0; }
  std::sort(us.begin(), us.end());
  printf("%-40s round trip p50 %5.2f us  p10 %5.2f  p90 %5.2f\n", name, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
  return 0;
}
int main() {
  uint64_t *h_req = nullptr, *h_resp = nullptr, *d_req = nullptr;
  CK(hipHostMalloc((void**)&h_req, 64, hipHostMallocCoherent | hipHostMallocMapped));
  CK(hipHostMalloc((void**)&h_resp, 64, hipHostMallocCoherent | hipHostMallocMapped));
  if (run("request in mapped host memory", h_req, h_req, h_resp)) return 1;
  hipError_t e = hipExtMallocWithFlags((void**)&d_req, 64, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { printf("fine-grained device memory: %s\n", hipGetErrorString(e)); return 0; }
  hipPointerAttribute_t at{};
  (void)hipPointerGetAttributes(&at, d_req);
  printf("fine-grained device buffer %p, host view %p\n", (void*)d_req, at.hostPointer);
  // is it CPU-writable at all?  probe from a child process-free way: hipMemcpy a pattern, then read through the pointer
  uint64_t pat = 0x1234;
  CK(hipMemcpy(d_req, &pat, 8, hipMemcpyHostToDevice));
  FILE* self = fopen("/proc/self/maps", "r");
  bool mapped = false;
  if (self) {
    char line[512];
    while (fgets(line, sizeof line, self)) {
      unsigned long lo, hi;
      if (sscanf(line, "%lx-%lx", &lo, &hi) == 2 && (unsigned long)d_req >= lo && (unsigned long)d_req < hi) mapped = true;
    }
    fclose(self);
  }
  printf("the buffer is %smapped into this process's address space\n", mapped ? "" : "NOT ");
  if (mapped) return run("request in fine-grained device memory", d_req, d_req, h_resp);
  return 0;
}
