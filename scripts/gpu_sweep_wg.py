"""One history through tbc_check (the level sweep) with K6 (TBC_SWEEP_WG unset) or K6w (TBC_SWEEP_WG=4 / 8: a workgroup of wavefronts per
segment, jit_sweep_wg.hip): time to verdict and the sweep's own counters, which must not depend on the kernel.
usage: [TBC_SWEEP_WG=8] [TBC_SWEEP_SEG=16] python scripts/gpu_sweep_wg.py"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jepsen_tigerbeetle_amd  # noqa
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
o = core.make_opts(device=0, want_witness=False, algorithm=N.ALG_COMPETITION)
hs = synth.register_ops_many(range(12), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
bad = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=12345, busy=0.1, info=0.0, corrupt=0.5))
core.check_ops(hs[0], model, o); core.check_ops(hs[0], model, o)
tt, sig = [], []
for h in hs:
    best = 1e9
    for _ in range(4):
        t = time.perf_counter(); r = core.check_ops(h, model, o); best = min(best, (time.perf_counter() - t) * 1e3)
    tt.append(best); sig.append((r["valid"], r["analyzer"], r["steps"], r["visited"], r["probes"]))
t = time.perf_counter(); rb = core.check_ops(bad, model, o); tb = (time.perf_counter() - t) * 1e3
print("TBC_SWEEP_WG", os.environ.get("TBC_SWEEP_WG"), "TBC_SWEEP_SEG", os.environ.get("TBC_SWEEP_SEG"),
      "| best-of-4 ms per history: median %.3f min %.3f max %.3f" % (statistics.median(tt), min(tt), max(tt)),
      "| invalid example %.3f ms verdict %d fail_op %d" % (tb, rb["valid"], rb["fail_op"]))
print("  counters", sig[:4], "sum of probes", sum(s[4] for s in sig), flush=True)
