"""The pack's time under a build variant of the front walk (TBC_LIB_PATH=.../libtbcheck_walkexpN.so: groups of its stores left out, so the
tables are garbage and the search is only bounded by its time limit) -- what the walk's stores cost.  usage: python scripts/gpu_walk_exp.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, core, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
hists = synth.register_ops_many(range(B), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
with core.Batch(hists, gm, core.make_opts(time_limit_ms=1500, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=4)) as b:
    for it in range(3):
        t = time.time(); b.run(); dt = time.time() - t
        tm = b.timing_ns()
        print(f"{os.path.basename(os.environ.get('TBC_LIB_PATH', 'default'))} run{it}: {dt*1e3:.1f} ms  " + " ".join(f"{k}={x/1e6:.2f}" for k, x in tm.items()), flush=True)
