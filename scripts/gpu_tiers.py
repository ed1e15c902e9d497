"""BASELINE.md section 3 tiers: 10k-op / 64-process cas-register histories, crashed-op (:info) rate 0 / 1 / 5 %,
each as generated (valid) and with ONE injected violation (a completed read near the middle made to return a
value nobody writes); seeds 0..9.  Time-to-verdict = tbc_check, host columns in -> verdict out (H2D + kernels
+ D2H), against the CPU restatement (oracle/wgl_window.c, 1 thread) on the same host.  Limits: 5 s on the GPU,
5e6 steps on the CPU (a run that hits one counts as unknown)."""
import statistics, sys, time
sys.path.insert(0, ".")
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import wgl

gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
om = {"kind": 1, "init": N.NIL}
opts = core.make_opts(time_limit_ms=5000, want_witness=True, algorithm=N.ALG_COMPETITION)
seeds = range(int(sys.argv[1]) if len(sys.argv) > 1 else 10)
print(__doc__)
print("%-5s %-9s | %-34s | %-34s" % ("info", "history", "GPU tbc_check ms (median / max), verdicts", "CPU port ms (median / max), verdicts"))
tiers = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0.0, 0.01, 0.05]
for info in tiers:
    for inject in (False, True):
        g_ms, c_ms, g_v, c_v, agree = [], [], [], [], 0
        for s in seeds:
            ops = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=s, busy=0.1, info=info))
            if inject:
                reads = np.flatnonzero((ops.f == N.F_READ) & (ops.ret_pos != N.POS_CRASHED) & (ops.a != N.NIL))
                k = reads[len(reads) // 2]
                ops.a[k] = 12
            if s == 0:
                core.check_ops(ops, gm, opts)        # warm (first call of a shape pays module load)
            r = core.check_ops(ops, gm, opts)
            g_ms.append(r["ns_total"] / 1e6); g_v.append(r["valid"])
            t = time.perf_counter()
            c = wgl.check(ops.as_dict(), om, "window", max_steps=5_000_000, want_witness=False)
            c_ms.append((time.perf_counter() - t) * 1e3); c_v.append(c["valid"])
            if r["valid"] != N.UNKNOWN and c["valid"] != -1:
                assert r["valid"] == c["valid"], (info, inject, s)
                if r["valid"] == 0:
                    assert r["fail_op"] == c["fail_op"], (info, inject, s)
                agree += 1
        def vs(v): return "valid %d invalid %d unknown %d" % (sum(x == 1 for x in v), sum(x == 0 for x in v), sum(x == -1 for x in v))
        print("%-5s %-9s | %8.1f / %8.1f  %-14s | %8.1f / %8.1f  %-14s | both decided & agree: %d" % (
            f"{info:.0%}", "1 bad read" if inject else "as is", statistics.median(g_ms), max(g_ms), vs(g_v),
            statistics.median(c_ms), max(c_ms), vs(c_v), agree), flush=True)
