"""A/B on the bench workload: one history per wavefront (lanes 64, width 2) against 8 / 16 / 32 lanes per history."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
BUSY = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
LANES = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [64, 8, 16]
VPO = int(sys.argv[4]) if len(sys.argv) > 4 else 8
RUNS = int(sys.argv[5]) if len(sys.argv) > 5 else 3
t = time.time()
hists = synth.register_ops_many(range(B), n_ops=10000, n_procs=64, busy=BUSY, info=0.0)
print(f"gen {time.time()-t:.2f}s", flush=True)
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
ref = None
for L in LANES:
    t = time.time()
    with core.Batch(hists, gm, core.make_opts(time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=VPO, lanes_per_history=L)) as b:
        print(f"lanes {L}: create {time.time()-t:.2f}s device GB {b.device_bytes()/1e9:.1f} width {b.search_width()} lanes {b.lanes_per_history()}", flush=True)
        for it in range(RUNS):
            t = time.time(); b.run(); dt = time.time() - t
            v = b.verdicts(); c = b.counters(); tm = b.timing_ns()
            print(f"  run{it}: {dt*1e3:.1f} ms  hist/s={B/dt:.0f}  valid={int((v==1).sum())} unknown={int((v==-1).sum())}  ms={ {k: round(x/1e6,2) for k,x in tm.items()} }  probes={c['probes']} visited={c['visited']} expanded={c['backtracks']}", flush=True)
        if ref is None:
            ref = v.copy()
        else:
            assert np.array_equal(ref, v), "verdicts differ between the schedules"
