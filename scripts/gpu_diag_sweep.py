"""Diagnostics: the level sweep vs oracle/sweep_ref.c on one history, several segment lengths."""
import os, sys
sys.path.insert(0, ".")
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import wgl
CAS = {"kind": 1, "init": N.NIL}
cases = [dict(n_ops=200, n_procs=8, seed=0, busy=0.5), dict(n_ops=1000, n_procs=16, seed=0, busy=0.3), dict(n_ops=3000, n_procs=64, seed=0, busy=0.1)]
for c in cases:
    h = columns.pair_events(synth.register_events(**c))
    for T in (0, 16, 32, 64):
        os.environ["TBC_SWEEP_SEG"] = str(T)
        with core.Batch([h], core.make_model(N.MODEL_CAS_REGISTER, N.NIL), core.make_opts(algorithm=N.ALG_LINEAR, want_witness=False)) as b:
            got = b.run().results()[0]; info = b.sweep_info()
        exp = wgl.check_sweep(h.as_dict(), CAS, seg_target=T)
        print(c, "T", T, info, "got", got["valid"], got["analyzer"], (got["visited"], got["probes"], got["backtracks"], got["max_depth"]),
              "exp", exp["valid"], (exp["configs_total"], exp["probes"], exp["subrounds"], exp["max_level"]), "nseg", exp["n_segments"], flush=True)
