"""A/B of the shared type-6 batch forms on the bench workload: per-phase times by HIP events and wall per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from variantstore_amd import VariantStore
w = bench.WORKLOADS[os.environ.get("WL", "chr1-2504")]
vs = VariantStore.synthetic(device=0, **bench.synth_kwargs(w))
nreg = int(os.environ.get("NREG", w["regions"]))
regions = bench.make_regions(w, 0, nreg)
if os.environ.get("SHARD"):   # K/N: rank K's shard of the 1 M-region strong batch (bench.py --scaling strong --emulate-shard)
    from variantstore_amd.parallel import shard_bounds
    k, n = (int(v) for v in os.environ["SHARD"].split("/"))
    whole = bench.make_regions(dict(w, region_seed=3), 0, 1_000_000)
    lo, hi = shard_bounds(1_000_000, k, n)
    regions = np.ascontiguousarray(whole[lo:hi])
    nreg = hi - lo
dev = torch.from_numpy(regions.astype(np.int64)).cuda()
def run(label, opts, steps=30):
    for k, v in opts.items():
        vs.set_option(k, v)
    for _ in range(3):
        vs.get_var_in_ref_device(dev.data_ptr(), nreg).close()
    torch.cuda.synchronize()
    acc = np.zeros(5)
    prev = None
    fills = []
    t0 = time.perf_counter()
    for _ in range(steps):     # as bench.py's loop: a result is closed one step late, nothing waits for the batch inside the loop
        r = vs.get_var_in_ref_device(dev.data_ptr(), nreg)
        if prev is not None:
            fills.append(prev.fill_ms())
            prev.close()
        prev = r
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    prev.close()
    label = f"{label} [fill in the loop {np.mean(fills):.4f}]"
    if os.environ.get("HOSTTIME"):   # where the host's time goes in a loop that reads nothing back: the call, the close one step late
        prev = None; tc = tcl = 0.0
        t0 = time.perf_counter()
        for _ in range(steps):
            ta = time.perf_counter()
            r = vs.get_var_in_ref_device(dev.data_ptr(), nreg)
            tb = time.perf_counter()
            if prev is not None: prev.close()
            tcl += time.perf_counter() - tb; tc += tb - ta
            prev = r
        torch.cuda.synchronize()
        dtl = (time.perf_counter() - t0) / steps * 1e3
        prev.close()
        label += f" [lean loop {dtl:.4f} ms: call {tc / steps * 1e3:.4f}, close {tcl / steps * 1e3:.4f}]"
    for _ in range(steps):     # phases by the handle's events (reading them waits for the batch)
        r = vs.get_var_in_ref_device(dev.data_ptr(), nreg)
        t = vs.last_timing()
        acc += (t.ms_total, t.ms_bounds, t.ms_scan, t.ms_emit, t.ms_fill)
        r.close()
    r = vs.get_var_in_ref_device(dev.data_ptr(), nreg)
    dg = r.digest()
    r.close()
    a = acc / steps
    print(f"{label:28s} wall {dt:.4f} ms  stream {a[0]:.4f} = plan {a[1]:.4f} + host {a[2]:.4f} + rows {a[3]:.4f} + fill {a[4]:.4f}   {nreg/dt/1e3:.1f} M/s  digest {dg:016x}", flush=True)
configs = os.environ.get("CONFIGS")
if configs:   # e.g. CONFIGS="a:fill_chunk=32;b:fill_chunk=16"
    for c in configs.split(";"):
        label, _, rest = c.partition(":")
        run(label, {k: int(v) for k, v in (kv.split("=") for kv in rest.split(",") if kv)})
    sys.exit(0)
for label, opts in [("fused16", dict(fill_fused=1, fill_chunk=16)), ("unfused16", dict(fill_fused=0, fill_chunk=16)),
                    ("fused32", dict(fill_fused=1, fill_chunk=32)), ("fused64", dict(fill_fused=1, fill_chunk=64)),
                    ("fused8", dict(fill_fused=1, fill_chunk=8)), ("unfused16 again", dict(fill_fused=0, fill_chunk=16)), ("fused16 again", dict(fill_fused=1, fill_chunk=16))]:
    run(label, opts)
