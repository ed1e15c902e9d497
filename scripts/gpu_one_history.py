"""One 10k-op / 64-process history through tbc_check a few times (profiling target)."""
import sys
sys.path.insert(0, ".")
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
alg = {"linear": N.ALG_LINEAR, "competition": N.ALG_COMPETITION, "wgl": N.ALG_WGL}[sys.argv[1] if len(sys.argv) > 1 else "linear"]
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
h = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=seed, busy=0.1))
for i in range(5):
    r = core.check_ops(h, gm, core.make_opts(algorithm=alg, want_witness=False))
    print(i, r["valid"], r["analyzer"], round(r["ns_total"] / 1e6, 3), round(r["ns_pack"] / 1e6, 3), round(r["ns_search"] / 1e6, 3))
