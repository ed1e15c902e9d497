"""Quick throughput probe: B valid 10k-op/64-proc cas-register histories, batch mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_ops = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
uniq = 64
W = int(sys.argv[3]) if len(sys.argv) > 3 else 0
VPO = int(sys.argv[4]) if len(sys.argv) > 4 else 0
BUSY = float(sys.argv[5]) if len(sys.argv) > 5 else 0.1
RULES = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True
RUNS = int(sys.argv[7]) if len(sys.argv) > 7 else 3
t = time.time()
base = [columns.pair_events(synth.register_events(n_ops=n_ops, n_procs=64, seed=s, busy=BUSY, info=0.0)) for s in range(uniq)]
hists = [base[i % uniq] for i in range(B)]
print(f"gen {time.time()-t:.2f}s ops/hist={len(base[0])}", flush=True)
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
with core.Batch(hists, gm, core.make_opts(time_limit_ms=120000, want_witness=False, search_width=W, algorithm=N.ALG_COMPETITION, visited_per_op=VPO, eager_reads=RULES, twin_rule=RULES)) as b:
    print("device MB", b.device_bytes() / 1e6, flush=True)
    for it in range(RUNS):
        t = time.time(); b.run(); dt = time.time() - t
        v = b.verdicts(); c = b.counters(); tm = b.timing_ns()
        print(f"B={B} W={W} busy={BUSY} rules={RULES} run{it}: {dt*1e3:.1f} ms  hist/s={B/dt:.0f}  valid={int((v==1).sum())}/{B}  timing_ms={ {k: round(x/1e6,3) for k,x in tm.items()} }  steps={c['steps']} visited={c['visited']} backtracks={c['backtracks']}", flush=True)
