"""A batch of the headline workload with crashed calls (count form, a wavefront per history): B histories, info rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
INFO = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
VPO = int(sys.argv[3]) if len(sys.argv) > 3 else 8
LANES = int(sys.argv[4]) if len(sys.argv) > 4 else 0
hists = synth.register_ops_many(range(30_000_000, 30_000_000 + B), n_ops=10000, n_procs=64, busy=0.1, info=INFO)
print("generated", flush=True)
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
with core.Batch(hists, gm, core.make_opts(time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=VPO, lanes_per_history=LANES)) as b:
    print(f"created: device GB {b.device_bytes() / 1e9:.1f} width {b.search_width()} lanes {b.lanes_per_history()}", flush=True)
    for it in range(2):
        t = time.perf_counter(); b.run(); dt = time.perf_counter() - t
        v = b.verdicts(); c = b.counters(); tm = b.timing_ns()
        print(f"  run{it}: {dt * 1e3:.1f} ms  hist/s={B / dt:.0f}  valid={int((v == 1).sum())} unknown={int((v == -1).sum())} invalid={int((v == 0).sum())} "
              f"ms={ {k: round(x / 1e6, 2) for k, x in tm.items()} } probes={c['probes']} visited={c['visited']}", flush=True)
