"""Writes tests/golden/edn/*.edn (Jepsen history.edn format, one op map per line) and expected.json.

Purpose: nobody can run stock Knossos in this environment, so parity with it
is unpinned.  These files are what an outside JVM needs to pin it: scripts/knossos_crosscheck.clj runs
knossos.wgl/analysis and knossos.linear/analysis on every file and prints the same JSON shape as
expected.json.  Until someone does, expected.json carries `"provenance": "oracle"` (our CPU restatement +
hand-derived KATs); replacing it with stock-Knossos output changes that field to "stock-knossos".
The -m gpu test tests/test_edn_golden.py takes every file through the EDN reader and tbc_check.

Run from the repo root:  python tests/golden/make_edn_golden.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import jepsen_tigerbeetle_amd  # noqa: E402,F401
from jepsen_tigerbeetle_amd import _native as N, columns, synth  # noqa: E402
from jepsen_tigerbeetle_amd.jepsen import edn  # noqa: E402
from oracle import wgl  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "edn")
TYPES = {N.INVOKE: "invoke", N.OK: "ok", N.FAIL: "fail", N.INFO: "info"}
FS = {N.F_READ: "read", N.F_WRITE: "write", N.F_CAS: "cas"}
SYNTH = [dict(n_ops=40, n_procs=4, seed=0, busy=0.5, info=0.05, corrupt=0.0), dict(n_ops=40, n_procs=4, seed=1, busy=0.5, info=0.0, corrupt=0.5),
         dict(n_ops=300, n_procs=8, seed=0, busy=0.4, info=0.02, corrupt=0.0), dict(n_ops=300, n_procs=8, seed=2, busy=0.3, info=0.0, corrupt=0.6),
         dict(n_ops=1000, n_procs=16, seed=0, busy=0.3, info=0.01, corrupt=0.0), dict(n_ops=1000, n_procs=16, seed=1, busy=0.2, info=0.0, corrupt=0.6),
         dict(n_ops=2000, n_procs=64, seed=0, busy=0.1, info=0.0, corrupt=0.0), dict(n_ops=2000, n_procs=64, seed=3, busy=0.3, info=0.0, corrupt=0.0),
         # crash-heavy (14 - 29 crashed calls: the library's count form; small enough for stock Knossos to finish the invalid ones)
         dict(n_ops=200, n_procs=6, seed=11, busy=0.4, info=0.06, corrupt=0.0), dict(n_ops=200, n_procs=6, seed=12, busy=0.4, info=0.06, corrupt=0.6),
         dict(n_ops=500, n_procs=8, seed=13, busy=0.3, info=0.03, corrupt=0.0), dict(n_ops=500, n_procs=8, seed=14, busy=0.3, info=0.03, corrupt=0.5),
         dict(n_ops=300, n_procs=4, seed=15, busy=0.6, info=0.08, corrupt=0.0), dict(n_ops=300, n_procs=4, seed=16, busy=0.6, info=0.08, corrupt=0.4)]


def event_maps(ev):
    """EventColumns of the seeded generator -> Jepsen op maps (cas-register shapes, README.md:41-50 style)."""
    out = []
    for r in range(len(ev.type)):
        f, a, b = int(ev.f[r]), int(ev.a[r]), int(ev.b[r])
        v = None if (f == N.F_READ and a == N.NIL) else (a if f != N.F_CAS else [a, b])
        out.append({"type": TYPES[int(ev.type[r])], "f": FS[f], "value": v, "process": int(ev.process[r]), "index": r})
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    expected = []
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_histories.json")))["cases"]
    for c in kats:
        hist = [{"type": t, "f": f, "value": v, "process": p, "index": i} for i, (t, f, v, p) in enumerate(c["history"])]
        name = f"kat_{c['name']}.edn"
        edn.write_history(os.path.join(OUT, name), hist)
        expected.append({"file": name, "model": c["model"], "valid?": c["valid"], "op-index": c.get("fail_index"),
                         "provenance": "hand-derived", "why": c.get("why")})
    for c in SYNTH:
        ev = synth.register_events(**c)
        hist = event_maps(ev)
        ops = columns.pair_events(ev)
        r = wgl.check(ops.as_dict(), {"kind": 1, "init": N.NIL}, "window")
        name = "synth_n{n_ops}_p{n_procs}_s{seed}_b{busy}_i{info}_c{corrupt}.edn".format(**c)
        edn.write_history(os.path.join(OUT, name), hist)
        expected.append({"file": name, "model": "cas-register", "valid?": r["valid"] == 1,
                         "op-index": None if r["valid"] == 1 else int(ops.ret_pos[r["fail_op"]]), "provenance": "oracle",
                         "why": "seeded generator (include/tbsynth.h) " + json.dumps(c)})
    with open(os.path.join(OUT, "expected.json"), "w") as fh:
        json.dump({"_comment": "valid? / op-index (:index of the completion that cannot be linearized) per file; provenance says who "
                               "produced the expectation: hand-derived KAT, our CPU oracle, or (once someone runs "
                               "scripts/knossos_crosscheck.clj) stock-knossos",
                   "cases": expected}, fh, indent=1)
    print(len(expected), "files")


if __name__ == "__main__":
    main()
