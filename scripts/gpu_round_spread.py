"""How evenly the headline batch's searches end: per-history expanded configs (one iteration of the narrow kernel each) -- the launch
deals every group of lanes exactly one history at 32,768 histories, so the kernel lasts as long as its LONGEST search."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jepsen_tigerbeetle_amd import _native as N, core, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
OPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
hists = synth.register_ops_many(range(B), n_ops=OPS, n_procs=64, busy=0.1, info=0.0)
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
with core.Batch(hists, gm, core.make_opts(time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=8)) as b:
    for it in range(2):
        t = time.time(); b.run(); dt = time.time() - t
    tm = b.timing_ns()
    ex = np.array([b._res[i].counters.backtracks for i in range(B)], np.int64)
    pr = np.array([b._res[i].counters.probes for i in range(B)], np.int64)
    print(f"B {B} ops {OPS} GB {b.device_bytes()/1e9:.1f} search us per longest iteration {tm['search']/1e3/ex.max():.2f} run {dt*1e3:.1f} ms pack {tm['pack']/1e6:.1f} search {tm['search']/1e6:.1f}")
    for name, x in (("expanded", ex), ("probes", pr)):
        q = np.percentile(x, [0, 1, 25, 50, 75, 99, 99.9, 100])
        print(name, "min/p1/p25/p50/p75/p99/p99.9/max", [int(v) for v in q], "mean", round(float(x.mean()), 1), "max/mean", round(float(x.max() / x.mean()), 3))
    w = ex.reshape(-1, 8).max(axis=1)
    print("per wavefront of 8 consecutive histories: mean of max", round(float(w.mean()), 1), "max", int(w.max()))
