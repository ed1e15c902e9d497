"""Time-to-verdict of ONE 10k-op / 64-process history through tbc_check, per algorithm, and where the time goes."""
import os, sys, time, statistics
sys.path.insert(0, ".")
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import wgl
CAS = {"kind": 1, "init": N.NIL}
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
busy = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
hs = [columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=s, busy=busy)) for s in range(6)]
bad = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=77, busy=busy, corrupt=0.5))
core.check_ops(hs[0], gm, core.make_opts(algorithm=N.ALG_LINEAR, want_witness=False))   # warm-up (module load)
for T in sys.argv[2:] or ["auto"]:
    if T == "auto": os.environ.pop("TBC_SWEEP_SEG", None)
    else: os.environ["TBC_SWEEP_SEG"] = T
    for name, alg, ww in (("linear", N.ALG_LINEAR, False), ("competition+witness(K5)", N.ALG_COMPETITION, True)):
        if name != "linear" and T != (sys.argv[2:] or ["auto"])[0]: continue
        tot, pk, se, wall = [], [], [], []
        for h in hs:
            t = time.perf_counter()
            r = core.check_ops(h, gm, core.make_opts(algorithm=alg, want_witness=ww))
            wall.append((time.perf_counter() - t) * 1e3)
            tot.append(r["ns_total"] / 1e6); pk.append(r["ns_pack"] / 1e6); se.append(r["ns_search"] / 1e6)
            assert r["valid"] == 1
        t = time.perf_counter(); rb = core.check_ops(bad, gm, core.make_opts(algorithm=alg, want_witness=ww)); tb = (time.perf_counter() - t) * 1e3
        print(f"busy {busy} T={T} {name}: tbc_check wall ms median {statistics.median(wall):.3f} min {min(wall):.3f} | in-call total {statistics.median(tot):.3f} pack {statistics.median(pk):.3f} search {statistics.median(se):.3f} | invalid: {tb:.3f} ms verdict {rb['valid']} analyzer {rb['analyzer']}", flush=True)
# CPU port on the same histories
t = time.perf_counter()
for h in hs: wgl.check(h.as_dict(), CAS, "window", want_witness=False)
print(f"cpu port (wgl_window.c, 1 thread): {(time.perf_counter()-t)/len(hs)*1e3:.3f} ms per history")
t = time.perf_counter()
for h in hs: wgl.check_sweep(h.as_dict(), CAS)
print(f"cpu sweep_ref.c (1 thread): {(time.perf_counter()-t)/len(hs)*1e3:.3f} ms per history")
