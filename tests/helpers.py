"""Shared test helpers: KAT loading, history -> op tuples for the brute-force oracle."""
import json
import os

from jepsen_tigerbeetle_amd import _native as N
from jepsen_tigerbeetle_amd.knossos import model as M

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MODELS = {"cas-register": M.cas_register, "register": M.register, "mutex": M.mutex}
ORACLE_KIND = {"cas-register": 1, "register": 0, "mutex": 2}


def load_kats():
    with open(os.path.join(GOLDEN, "kat_histories.json")) as fh:
        cases = json.load(fh)["cases"]
    out = []
    for c in cases:
        hist = [{"type": t, "f": f, "value": v, "process": p} for (t, f, v, p) in c["history"]]
        out.append((c["name"], c["model"], hist, c["valid"], c.get("fail_index")))
    return out


def oracle_model(name):
    kind = ORACLE_KIND[name]
    return {"kind": kind, "init": 0 if name == "mutex" else N.NIL}


def op_tuples(ops):
    return [(int(ops.f[i]), int(ops.a[i]), int(ops.b[i]), int(ops.inv_pos[i]), int(ops.ret_pos[i]))
            for i in range(len(ops))]
