// Issue cost of the integer VALU operations the expansion's dense path is made of, per wave64 instruction and SIMD:
// 8192 waves (8 per SIMD) run `iters` x 64 operations of one kind on eight independent register streams (or one dependent
// chain); cycles per instruction per SIMD = launch time x clock x 1024 SIMDs / wave-instructions issued.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)

template <int KIND>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters, uint32_t seed) {
  uint32_t r[8];
  for (int i = 0; i < 8; ++i) r[i] = seed * (threadIdx.x + 1 + i) | 1u;
  uint32_t s = seed | 3u;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
      REP64(OP)
#undef OP
    } else if (KIND == 1) {
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
      REP64(OP)
#undef OP
    } else if (KIND == 2) {
#define OP(i) asm volatile("v_ffbl_b32 %0, %0" : "+v"(r[i]));
      REP64(OP)
#undef OP
    } else if (KIND == 3) {
#define OP(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(s));
      REP64(OP)
#undef OP
    } else if (KIND == 4) {
#define OP(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r[i]) : "v"(s));
      REP64(OP)
#undef OP
    } else if (KIND == 5) {   // one dependent chain
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[0]) : "v"(s));
      REP64(OP)
#undef OP
    } else if (KIND == 6) {
#define OP(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
      REP64(OP)
#undef OP
    } else if (KIND == 7) {
#define OP(i) asm volatile("v_and_or_b32 %0, %0, %1, %0" : "+v"(r[i]) : "s"(s));
      REP64(OP)
#undef OP
    } else if (KIND == 8) {
#define OP(i) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(r[i]), "v"(s) : "vcc");
      REP64(OP)
#undef OP
    } else if (KIND == 9) {
#define OP(i) asm volatile("v_alignbit_b32 %0, %0, %1, 3" : "+v"(r[i]) : "v"(s));
      REP64(OP)
#undef OP
    } else if (KIND == 10) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[i]));
      REP64(OP)
#undef OP
    } else if (KIND == 11) {   // scalar: s_add_u32
      uint32_t t = s;
#define OP(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(t) : : "scc");
      REP64(OP)
#undef OP
      s = t;
    } else if (KIND == 12) {   // v_readlane
      uint32_t t;
#define OP(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(t) : "v"(r[i]));
      REP64(OP)
#undef OP
      s ^= t;
    } else if (KIND == 13) {   // v_lshl_add_u64 (address arithmetic)
      uint64_t a = ((uint64_t)r[1] << 32) | r[0];
#define OP(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %0" : "+v"(a));
      REP64(OP)
#undef OP
      r[0] = (uint32_t)a; r[1] = (uint32_t)(a >> 32);
    }
  }
  uint32_t x = s;
  for (int i = 0; i < 8; ++i) x ^= r[i];
  if (x == 0x12345678u) out[threadIdx.x] = x;
}

template <int KIND>
static float run(uint32_t* out, int iters, int waves_per_simd) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * waves_per_simd;   // 4 waves per block: one per SIMD
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rate<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_rate<KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  uint32_t* out; CK(hipMalloc(&out, 4096));
  const int iters = 2000;
  const char* names[] = {"v_add_u32 (8 streams)", "v_and_b32", "v_ffbl_b32", "v_bcnt_u32_b32", "v_lshl_add_u32", "v_add_u32 dependent chain", "v_add_u32_dpp row_shr:1",
                         "v_and_or_b32 (sgpr mask)", "v_cmp_eq_u32", "v_alignbit_b32", "v_lshlrev_b32", "s_add_u32", "v_readlane_b32", "v_lshl_add_u64"};
  for (int wps : {8, 4, 1}) {
    printf("-- %d wave(s) per SIMD\n", wps);
    float ms[14];
    ms[0] = run<0>(out, iters, wps); ms[1] = run<1>(out, iters, wps); ms[2] = run<2>(out, iters, wps); ms[3] = run<3>(out, iters, wps);
    ms[4] = run<4>(out, iters, wps); ms[5] = run<5>(out, iters, wps); ms[6] = run<6>(out, iters, wps); ms[7] = run<7>(out, iters, wps);
    ms[8] = run<8>(out, iters, wps); ms[9] = run<9>(out, iters, wps); ms[10] = run<10>(out, iters, wps); ms[11] = run<11>(out, iters, wps);
    ms[12] = run<12>(out, iters, wps); ms[13] = run<13>(out, iters, wps);
    for (int k = 0; k < 14; ++k) {
      const double instr_per_simd = (double)wps * iters * 64;
      printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", names[k], ms[k], ms[k] * 1e-3 * 2.4e9 / instr_per_simd);
    }
  }
  return 0;
}
