#!/usr/bin/env python3
"""Batches of the sample-coordinate query types (2, 3, 5) on the bench's `sc` index, with the engine's phase timings --
the thing to profile when their kernels are the subject."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from variantstore_amd import VariantStore

w = dict(bench.WORKLOADS[os.environ.get("VS_BENCH_WORKLOAD", "chr1-2504")])
kw = bench.synth_kwargs(w)
kw["ref_length"] = max(200_000, w["ref_length"] * 2 // 25)
kw["num_variants"] = max(1000, w["num_variants"] * 2 // 25)
kw["first_pos"] = min(w["first_pos"], kw["ref_length"] // 10)
vs = VariantStore.synthetic(device=0, sample_coordinates=True, **kw)
vs.set_option("phase_events", 1)   # the phase times of walking batches are read below
if os.environ.get("VS_SC_GROUP"):
    vs.set_option("sc_group", int(os.environ["VS_SC_GROUP"]))   # (tuning builds)
n = int(os.environ.get("VS_SC_REGIONS", "100000"))
rng = np.random.default_rng(23)
st = rng.integers(max(1, kw["first_pos"]), kw["ref_length"] - w["region_len"], size=n, dtype=np.int64)
reg = np.stack([st, st + w["region_len"]], axis=1).astype(np.uint64)
ns = vs.info().num_samples
per = np.array([1 + ((i % 16) * 157) % (ns - 1) for i in range(n)], dtype=np.uint32)
forms = [int(v) for v in os.environ.get("VS_T4_WALK", "2").split(",")]   # walk forms to time: 2 cooperative, 1 one lane per region
seen = {}
for form in forms:
  vs.set_option("t4_walk", form)
  for name, call in (("type2",   lambda: vs.query_sample_seq(reg, per, sample_coordinates=False)),
                     ("type3", lambda: vs.query_sample_seq(reg, per, sample_coordinates=True)),
                     ("type5", lambda: vs.get_sample_var_in_sample(reg, per))):
      call().close()
      torch.cuda.synchronize()
      a = time.perf_counter()
      for _ in range(3):
          r = call()
          t = vs.last_timing()
          r.close()
      torch.cuda.synchronize()
      dt = (time.perf_counter() - a) / 3
      r = call()
      key = (r.totals(), hash(r.sequences()[0].tobytes()), hash(tuple(r.sequences()[1][:2000]))) if name != "type5" else (r.totals(), r.digest())
      r.close()
      same = "" if name not in seen else ("  == first form" if seen[name] == key else "  DIFFERS from the first form")
      seen.setdefault(name, key)
      print(f"walk form {form} {name}: {n / dt / 1e6:.1f} M regions/s, {dt * 1e3:.3f} ms per batch | phases ms: total {t.ms_total:.3f} bounds/walk {t.ms_bounds:.3f} "
            f"scan {t.ms_scan:.3f} emit {t.ms_emit:.3f} fill {t.ms_fill:.3f}{same}", flush=True)
