"""BASELINE config 4 (100k-op multi-register history, 256 processes) through the wide kernel under its two rules: the library's time
beside the CPU oracle's on the same history."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import multi_register_history
from jepsen_tigerbeetle_amd import _native as N, core
from jepsen_tigerbeetle_amd.knossos import _analysis, model as M
from oracle import wgl as oracle

busy = float(sys.argv[1]) if len(sys.argv) > 1 else 0.03
n_ops = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
hist = multi_register_history(n_ops, 256, 7, n_keys=8, n_values=5, busy=busy, info=0.0)
e = _analysis.Encoded(M.multi_register({}), hist)
om = {"kind": 4, "init": 0, "pool": e.ops.pool}
for width in (8, 16, 4):
    t = time.time()
    got = core.check_ops(e.ops, e.native_model, core.make_opts(time_limit_ms=600000, algorithm=N.ALG_COMPETITION, search_width=width, want_witness=False))
    dt = time.time() - t
    t = time.time()
    exp = oracle.check_beam(e.ops.as_dict(), om, width, want_witness=False)
    dto = time.time() - t
    print(f"busy {busy} ops {n_ops} width {width}: device {dt:.2f} s (search {got['ns_search']/1e9:.2f} s) valid {got['valid']} probes {got['probes']} visited {got['visited']}; oracle {dto:.2f} s probes {exp['probes']}", flush=True)
