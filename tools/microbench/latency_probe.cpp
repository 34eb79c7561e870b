// Host-side latency of a single-region type-6 call through the C ABI (no Python in the loop).
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "variantstore_hip.h"
int main() {
  vs_synth_params p{};
  p.ref_length = 20000000; p.num_variants = 400000; p.num_samples = 2504; p.seed = 1; p.first_pos = 1000;
  p.frac_ins = 0.05; p.frac_del = 0.05; p.frac_multi = 0.01; p.max_indel = 6; p.af_exponent = 11.0;
  vs_index* idx = nullptr;
  if (vs_index_synthetic(&p, 0, nullptr, &idx) != VS_OK) { printf("open failed: %s\n", vs_last_error()); return 1; }
  std::vector<double> us, prep, launch, wait, gpu;
  for (int i = 0; i < 600; ++i) {
    vs_region r{(uint64_t)(1000 + (i * 7919ull * 31) % 19900000), 0};
    r.y = r.x + 10000;
    auto t0 = std::chrono::steady_clock::now();
    vs_result* res = nullptr;
    if (vs_query_var_in_ref(idx, &r, 1, &res) != VS_OK) { printf("query failed: %s\n", vs_last_error()); return 1; }
    auto t1 = std::chrono::steady_clock::now();
    vs_result_free(res);
    vs_timing tm{};
    vs_index_last_timing(idx, &tm);
    if (i >= 100) {
      us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
      prep.push_back(tm.ms_bounds * 1e3); launch.push_back(tm.ms_scan * 1e3); wait.push_back(tm.ms_emit * 1e3); gpu.push_back(tm.ms_fill * 1e3);
    }
  }
  std::sort(us.begin(), us.end());
  printf("single-region vs_query_var_in_ref: p50 %.1f us  p10 %.1f  p90 %.1f\n", us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10]);
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("  medians: host sizing + slab %.1f us, launch call %.1f us, mailbox wait %.1f us (kernel by the device clock %.1f us)\n",
         med(prep), med(launch), med(wait), med(gpu));
  vs_index_close(idx);
  return 0;
}
