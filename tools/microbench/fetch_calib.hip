// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, per access pattern
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ...
// other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern").
// The expansion kernel's reads are not one pattern: 16-byte list groups and 8-byte nibble windows per lane (consecutive
// lanes mostly consecutive), 320-byte class rows, 32-byte site rows.  Each kernel below moves a known number of bytes in
// one of those shapes; run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and compare (tools/pmc_summary.py).
//   kernels: calib_stream_read (16 B/lane coalesced), calib_stream_write_nt, calib_seg<BYTES> (random BYTES-byte aligned
//   segments read by BYTES/16 adjacent lanes: 16, 32, 64, 128, 320), calib_seg8 (random 8-byte words, one per lane)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

__global__ void calib_stream_read(const u32x4* __restrict__ src, size_t n, u32x4* sink) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32x4 v = src[i];
  if (v.x == 0x12345u) *sink = v;
}
__global__ void calib_stream_write_nt(u32x4* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  __builtin_nontemporal_store(u32x4{(unsigned)i, 1, 2, 3}, &dst[i]);
}
__global__ void calib_stream_write(u32x4* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dst[i] = u32x4{(unsigned)i, 1, 2, 3};
}
// n lanes; lanes [k * L, (k + 1) * L) (L = BYTES / 16) read segment hash(k) of the buffer, 16 bytes each
template <int BYTES>
__global__ void calib_seg(const u32x4* __restrict__ src, size_t n, size_t nseg_in_buffer, u32x4* sink) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr int L = BYTES / 16;
  const size_t k = i / L, l = i % L;
  const size_t seg = mix(k * 2654435761ULL + 12345) % nseg_in_buffer;
  const u32x4 v = src[seg * L + l];
  if (v.x == 0x12345u) *sink = v;
}
__global__ void calib_seg8(const uint64_t* __restrict__ src, size_t n, size_t nwords, uint64_t* sink) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t v = src[mix(i * 2654435761ULL + 777) % nwords];
  if (v == 0x12345u) *sink = v;
}

int main() {
  const size_t buf_bytes = 8ull << 30, n16 = buf_bytes / 16;
  u32x4 *src, *dst, *sink;
  CK(hipMalloc(&src, buf_bytes)); CK(hipMalloc(&dst, buf_bytes)); CK(hipMalloc(&sink, 16));
  CK(hipMemset(src, 1, buf_bytes));
  CK(hipDeviceSynchronize());
  const size_t n = 64ull << 20;   // lanes per launch: 1 GiB of 16-byte accesses
  const unsigned blocks = (unsigned)(n / 256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_stream_read, dim3(blocks * 4), dim3(256), 0, 0, src, n * 4, sink);          // 4 GiB read
    hipLaunchKernelGGL(calib_stream_write_nt, dim3(blocks * 4), dim3(256), 0, 0, dst, n * 4);            // 4 GiB written
    hipLaunchKernelGGL(calib_stream_write, dim3(blocks * 4), dim3(256), 0, 0, dst, n * 4);               // 4 GiB written
    hipLaunchKernelGGL(calib_seg<16>, dim3(blocks), dim3(256), 0, 0, src, n, n16, sink);                 // 1 GiB useful
    hipLaunchKernelGGL(calib_seg<32>, dim3(blocks), dim3(256), 0, 0, src, n, n16 / 2, sink);
    hipLaunchKernelGGL(calib_seg<64>, dim3(blocks), dim3(256), 0, 0, src, n, n16 / 4, sink);
    hipLaunchKernelGGL(calib_seg<128>, dim3(blocks), dim3(256), 0, 0, src, n, n16 / 8, sink);
    hipLaunchKernelGGL(calib_seg<320>, dim3(blocks), dim3(256), 0, 0, src, n, n16 / 20, sink);
    hipLaunchKernelGGL(calib_seg8, dim3(blocks), dim3(256), 0, 0, (const uint64_t*)src, n, buf_bytes / 8, (uint64_t*)sink);   // 0.5 GiB useful
    CK(hipDeviceSynchronize());
  }
  printf("known bytes per launch: calib_stream_read 4294967296 read; calib_stream_write(_nt) 4294967296 written; calib_seg<N> 1073741824 useful read "
         "(64 Mi lanes x 16 B, random N-byte segments of an 8 GiB buffer); calib_seg8 536870912 useful read\n");
  return 0;
}
