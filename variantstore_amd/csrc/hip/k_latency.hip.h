// k_latency.hip.h -- latency path: a batch of at most 64 regions in one launch; the resident query server.
// Part of kernels.hip.h (the kernel index with reference file:line is there).
#pragma once
#include "k_expand.hip.h"

namespace vsamd {

// ---------------------------------------------------------------------------
// Latency path: a batch of at most 64 regions in ONE launch.  Every wave works the region bounds out for itself (lane
// q takes region q: the same dozen loads in every wave, L2 hits after the first), a wave prefix sum lays the slot,
// arena and task offsets out, and the wave then takes 4-slot tasks straight from the site table: it writes their
// variant headers and expands their carriers (expand_task) -- no header kernel, no kernel-to-kernel dependency, no
// host round trip.  The regions travel in the kernel arguments.  The result buffers were sized on the host from the
// same arithmetic (engine.hip: host_region_size); should the device ever need more it writes nothing and says so.
// The last block to finish applies the literal "already seen" rule to the regions that need it and posts the
// completion mailbox.
// ---------------------------------------------------------------------------
template <int NMAX>
struct SmallRegions { uint64_t xy[2 * NMAX]; };

__device__ __forceinline__ uint64_t wave_inclusive_scan64(uint64_t v, uint32_t lane) {
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v += t;
  }
  return v;
}

// The result of a small batch lives in ONE slab; host and device lay it out with the same arithmetic (the resident
// server gets only the slab address and the capacities with a request).
__host__ __device__ inline size_t small_result_layout(DevResult& d, uint8_t* slab, uint64_t n, uint64_t capA, uint64_t capS, uint32_t car_width) {
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) & ~(size_t)255; return at; };
  const size_t o_flags = take(n), o_g0 = take(n * 4), o_nvar = take(n * 8), o_ncar = take(n * 8), o_vb = take((n + 1) * 8),
               o_cb = take((n + 1) * 8), o_vc = take(n * 8), o_rows = take(capA * sizeof(VariantRow)),
               o_car = take(capS * car_width + 16);
  d.Q = n; d.A = capA; d.S = capS;
  d.regions = nullptr;
  d.q_flags = slab + o_flags; d.q_g0 = (uint32_t*)(slab + o_g0); d.q_nvar = (uint64_t*)(slab + o_nvar);
  d.q_ncar = (uint64_t*)(slab + o_ncar); d.var_begin = (uint64_t*)(slab + o_vb); d.car_base = (uint64_t*)(slab + o_cb);
  d.var_count = (uint64_t*)(slab + o_vc);
  d.rows = (VariantRow*)(slab + o_rows); d.r_class = nullptr; d.r_gt0 = nullptr; d.q_car_len = nullptr; d.carriers = slab + o_car;
  d.car_width = car_width; d.pad3_ = 0;
  return off;
}

// Bounds, offsets, per-region arrays and tasks of a small batch for ONE wave (see k_query_small).  `wave`/`nwaves`:
// this wave's place among the waves sharing the batch.  Returns any-slow | over << 1.
template <bool WIDE>
__device__ __forceinline__ uint32_t small_batch_wave(const DevImage& im, const DevResult& r, uint64_t x, uint64_t y, uint32_t n, uint32_t wave,
                                                     uint32_t nwaves, uint32_t* lds_wave, uint32_t gt_words, uint64_t cap_slots,
                                                     uint64_t cap_carriers, bool stamps) {
  constexpr uint32_t CH = kFillChunkSmall;
  const uint32_t lane = threadIdx.x & 63;
  // ---- bounds of region `lane`, offsets of all regions ----
  RegionBounds b{0, 0, 0, 0, 0, 0};
  if (lane < n) b = region_bounds_of(im, x, y);
  const uint64_t pre0 = b.pre0, npad = b.npad, nkept = b.nkept;
  const uint32_t nv = b.g1 - b.g0, ntask = (nv + CH - 1) / CH;
  const uint64_t vend = wave_inclusive_scan64(nv, lane), cend = wave_inclusive_scan64(npad, lane);
  const uint32_t tend = wave_inclusive_scan(ntask);
  const uint64_t A = wave_bcast64(vend, 63), S = wave_bcast64(cend, 63);
  const uint32_t T = __builtin_amdgcn_readlane(tend, 63);
  const bool any_slow = __ballot(b.flags & kRegionSlow) != 0;
  const bool over = A > cap_slots || S > cap_carriers;
  if (stamps && wave == 0 && lane == 0) __hip_atomic_store(&r.done_counter[2], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bounds and offsets done
  if (wave == 0 && !over) {   // the per-region arrays of the result
    if (lane < n) {
      r.q_flags[lane] = b.flags; r.q_g0[lane] = b.g0; r.q_nvar[lane] = nv;
      r.var_begin[lane] = vend - nv; r.car_base[lane] = cend - npad;
      if (!(b.flags & kRegionSlow)) { r.var_count[lane] = nv; r.q_ncar[lane] = nkept; }
    }
    if (lane == 0) { r.var_begin[n] = A; r.car_base[n] = S; }
  }
  // ---- tasks: 8 consecutive sites of one region ----
  if (!over) {
    for (uint32_t c = wave; c < T; c += nwaves) {
      // the region of task c: the last one whose first task is <= c (regions without tasks share their successor's offset)
      const uint32_t q = (uint32_t)__popcll(__ballot(lane < n && tend - ntask <= c)) - 1u;
      const uint32_t g0_q = __builtin_amdgcn_readlane(b.g0, q), nv_q = __builtin_amdgcn_readlane(nv, q);
      const uint32_t t0_q = __builtin_amdgcn_readlane(tend - ntask, q);
      const uint64_t a0_q = wave_bcast64(vend - nv, q), cb_q = wave_bcast64(cend - npad, q) - wave_bcast64(pre0, q);
      const uint32_t j = (c - t0_q) * CH + lane;
      uint32_t cnt = 0, cls = 0;
      uint64_t gt0 = 0, cb = 0;
      if (lane < CH && j < nv_q) {
        const uint32_t g = g0_q + j;
        const uint64_t a = a0_q + j;
        const uint32_t fl = im.s_flags[g];
        cnt = im.s_ncar[g];
        cls = im.s_class[g];
        gt0 = im.s_gt0[g];
        cb = cb_q + im.s_carpre[g];
        // the variant row (building std::vector<Variant>, query.h:736-771)
        row_store(r.rows, a, im.s_pos[g], im.s_ref_off[g], im.s_ref_len[g], im.s_alt_off[g], im.s_alt_len[g], cnt,
                  (fl & kSiteAlwaysDrop) != 0, cb);
      }
      expand_task<WIDE, true>(im, r.carriers, lds_wave, lane, cnt, cls, gt0, cb, 0u, gt_words);
    }
  }
  return (any_slow ? 1u : 0u) | (over ? 2u : 0u);
}

template <bool WIDE, int NMAX>
__global__ void __launch_bounds__(256) k_query_small(DevImage im, DevResult r, SmallRegions<NMAX> regs, uint32_t gt_words,
                                                     uint64_t cap_slots, uint64_t cap_carriers) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const uint32_t n = (uint32_t)r.Q;
  const bool stamps = r.host_totals != nullptr;   // VS_LAT_DEBUG: device clock, 100 MHz
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&r.done_counter[1], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_blk[];
  const uint32_t lds_words_per_wave = WIDE ? gt_words + kRingWords : slice_lds_words(im.num_samples);
  uint64_t x = 0, y = 0;
  if (lane < n) { x = regs.xy[2 * lane]; y = regs.xy[2 * lane + 1]; }
  const uint32_t st = small_batch_wave<WIDE>(im, r, x, y, n, wave, nwaves, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], gt_words,
                                             cap_slots, cap_carriers, stamps);
  const bool any_slow = st & 1u, over = st & 2u;
  // ---- completion: the last block applies the literal dedup rule where needed, then posts the mailbox.  The
  //      flag word carries everything the host does not know yet: sequence number | any-slow << 62 | over << 63 (the
  //      sizes are the host's own).  Without dedup work nobody reads another block's data inside this launch, so a
  //      block only waits until its own stores are acknowledged (no L2 write-back) before it counts itself done. ----
  __shared__ uint32_t s_last;
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&r.done_counter[3], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // block 0's first wave is through
  if (any_slow) __threadfence();
  else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(r.done_counter, 1ULL) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (s_last) {
    if (any_slow && !over) {
      __threadfence();   // the headers other blocks wrote
      for (uint32_t q = threadIdx.x; q < n; q += blockDim.x)
        if (r.q_flags[q] & kRegionSlow) dedup_region(im, r, q);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      *r.done_counter = 0;   // re-armed for the next launch on this stream
      if (stamps) {   // device-clock durations in 10 ns ticks {kernel, bounds + offsets, block 0's tasks}
        const uint64_t t_end = wall_clock64();   // (the stamps come from another block, possibly another XCD: agent-scope loads)
        const uint64_t t1 = __hip_atomic_load(&r.done_counter[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t2 = __hip_atomic_load(&r.done_counter[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t t3 = __hip_atomic_load(&r.done_counter[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r.host_totals[0] = t_end - t1;
        r.host_totals[1] = t2 - t1;
        r.host_totals[2] = t3 - t2;
        __threadfence_system();
      }
      __hip_atomic_store(const_cast<uint64_t*>(r.done_flag), r.done_seq | (any_slow ? 1ULL << 62 : 0ULL) | (over ? 1ULL << 63 : 0ULL),
                         __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---------------------------------------------------------------------------
// Resident query server: the same work as k_query_small without a launch per query.  A small grid stays on the GPU for
// a bounded time and polls a request line in mapped host memory; the host posts {sequence number, slab, capacities,
// regions}, every block picks the request up by itself (no device-side broadcast), the waves share the tasks, each
// block pushes its results out with a system-scope release and counts itself done, the last one applies the dedup
// rule where needed and posts the sequence number back.  A request costs one PCIe round trip instead of a kernel
// dispatch (2.5 us against 6 us + the launch call, tools/microbench/pingpong.hip / latency_floor.hip).
// Every loop is bounded by the device clock: a block leaves `life_ticks` after its start or `idle_ticks` after the last
// request whatever the host does (so a device-wide synchronisation elsewhere in the process waits a millisecond at most), no
// block ever waits for another one, and a request caught by a block's exit simply is not answered -- the host then
// falls back to the launch path (engine.hip).
// ---------------------------------------------------------------------------
struct ServerRequest {        // mapped host memory, 64-byte aligned; the host writes the body first, then tail, then head
  uint64_t head;              // sequence number; ~0 = leave
  uint64_t slab;              // device address of the result slab (small_result_layout)
  uint64_t cap_slots, cap_carriers;
  uint64_t n_and_width;       // regions | carrier width << 32
  uint64_t x0, y0;            // the first region (a single-region request is this one line)
  uint64_t tail;              // server_request_tail(head, body): seals the six words above
  uint64_t xy[128];           // all regions
};
static_assert(sizeof(ServerRequest) == 64 + 1024, "request layout");
// The tail word seals the line: sequence number mixed with a checksum of the six body words.  The device accepts a
// line only when head == the expected sequence number AND tail matches the body it read, so the hand-off does not
// depend on the eight 8-byte loads of the poll being served as one 64-byte transaction (a torn read -- new head and
// tail, old body -- fails the checksum and is simply polled again).
__host__ __device__ inline uint64_t server_request_tail(uint64_t seq, const uint64_t body[6]) {
  uint64_t h = seq * 0x9E3779B97F4A7C15ULL;
  for (int i = 0; i < 6; ++i) { h = (h ^ body[i]) * 0xff51afd7ed558ccdULL; h ^= h >> 29; }
  return h;
}

template <bool WIDE>
__global__ void __launch_bounds__(256) k_query_server(DevImage im, const ServerRequest* req, unsigned long long* done_counter,
                                                      volatile uint64_t* done_flag, uint64_t first_seq, uint32_t gt_words,
                                                      uint64_t life_ticks, uint64_t idle_ticks) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_blk[];
  const uint32_t lds_words_per_wave = WIDE ? gt_words + kRingWords : slice_lds_words(im.num_samples);
  __shared__ uint64_t s_req[8];
  __shared__ uint64_t s_xy[128];
  __shared__ uint32_t s_state;   // 0 idle, 1 request in s_req, 2 leave
  __shared__ uint32_t s_last;
  const uint64_t t_start = wall_clock64();
  uint64_t t_last = t_start;       // (every block sees the same requests: the idle clocks agree to within microseconds)
  uint64_t expect = first_seq;
  const uint64_t* reqw = reinterpret_cast<const uint64_t*>(req);
  while (true) {
    // ---- wave 0 polls the request line: lanes 0..7 read its eight words in one access ----
    if (threadIdx.x < 64) {
      uint32_t state = 0;
      while (state == 0) {
        uint64_t w = 0;
        if (lane < 8) w = __hip_atomic_load(reqw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint64_t head = wave_bcast64(w, 0), tail = wave_bcast64(w, 7);
        const uint64_t now = wall_clock64();
        bool sealed = false;
        if (head == expect) {
          const uint64_t body[6] = {wave_bcast64(w, 1), wave_bcast64(w, 2), wave_bcast64(w, 3), wave_bcast64(w, 4), wave_bcast64(w, 5), wave_bcast64(w, 6)};
          sealed = tail == server_request_tail(expect, body);
        }
        if (head == ~0ULL || now - t_start > life_ticks || now - t_last > idle_ticks) state = 2;
        else if (sealed) {
          if (lane < 8) s_req[lane] = w;
          const uint32_t n = (uint32_t)wave_bcast64(w, 4) & 0xFFFFu;
          if (n > 1) {   // the other regions: one more round trip
            s_xy[2 * lane] = lane < n ? __hip_atomic_load(reqw + 8 + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
            s_xy[2 * lane + 1] = lane < n ? __hip_atomic_load(reqw + 9 + 2 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
          }
          state = 1;
        } else __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) s_state = state;
    }
    __syncthreads();
    if (s_state == 2) break;
    // ---- the request ----
    const uint64_t seq = s_req[0];
    const uint32_t n = (uint32_t)s_req[4] & 0xFFFFu, car_width = (uint32_t)(s_req[4] >> 32);
    const bool stamps = (s_req[4] >> 16) & 1;   // VS_LAT_DEBUG: device-clock stamps into done_flag[1..4] (block 0, last block)
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) done_flag[1] = wall_clock64();
    DevResult r{};
    small_result_layout(r, reinterpret_cast<uint8_t*>(s_req[1]), n, s_req[2], s_req[3], car_width);
    r.done_counter = done_counter;
    uint64_t x = 0, y = 0;
    if (lane < n) { x = n > 1 ? s_xy[2 * lane] : s_req[5]; y = n > 1 ? s_xy[2 * lane + 1] : s_req[6]; }
    const uint32_t st = small_batch_wave<WIDE>(im, r, x, y, n, wave, nwaves, &lds_blk[(threadIdx.x >> 6) * lds_words_per_wave], gt_words,
                                               s_req[2], s_req[3], false);
    const bool any_slow = st & 1u, over = st & 2u;
    // ---- completion: results must be out of this XCD's L2 before anybody is told (the kernel does not end here) ----
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) done_flag[2] = wall_clock64();
    __threadfence_system();
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) done_flag[3] = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(done_counter, 1ULL) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (s_last) {
      if (any_slow && !over) {
        __threadfence();
        for (uint32_t q = threadIdx.x; q < n; q += blockDim.x)
          if (r.q_flags[q] & kRegionSlow) dedup_region(im, r, q);
        __threadfence_system();
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        *done_counter = 0;
        __threadfence();
        if (stamps) { done_flag[4] = wall_clock64(); __threadfence_system(); }
        __hip_atomic_store(const_cast<uint64_t*>(done_flag), seq | (any_slow ? 1ULL << 62 : 0ULL) | (over ? 1ULL << 63 : 0ULL),
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    expect = seq + 1;
    t_last = wall_clock64();
    __syncthreads();   // s_req / s_xy are rewritten by the next poll
  }
}

// Hit-list records for a collective: 4 x uint64 per reported row of every region, regions back to back
//   {pos | dropped << 63, ref_off | ref_len << 32, alt_off | alt_len << 32, region | car_count << 32}
// (one wave per region; slot_begin = exclusive scan of the regions' row counts)
__global__ void __launch_bounds__(256) k_pack_headers(DevResult r, uint64_t* dst, const uint64_t* slot_begin, uint64_t region_base) {
  const uint64_t q = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (q >= r.Q) return;
  const uint64_t n = r.q_nvar[q], a0 = r.var_begin[q], o0 = slot_begin[q];
  for (uint64_t j = threadIdx.x & 63; j < n; j += 64) {
    const VariantRow v = row_load(r.rows, a0 + j);
    uint64_t* d = dst + 4 * (o0 + j);
    d[0] = (uint64_t)v.pos | (row_dropped(v) ? (1ULL << 63) : 0ULL);
    d[1] = (uint64_t)v.ref_off | ((uint64_t)v.ref_len << 32);
    d[2] = (uint64_t)v.alt_off | ((uint64_t)v.alt_len << 32);
    d[3] = (region_base + q) | ((uint64_t)row_count(v) << 32);
  }
}

}  // namespace vsamd
