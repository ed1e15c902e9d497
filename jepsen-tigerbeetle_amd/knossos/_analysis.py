"""Shared implementation of (knossos.wgl/analysis model history) et al.: encode
the Jepsen history into columns, call the C-ABI, rebuild the Knossos-shaped
result map.  The search runs in the HIP kernels only (core.check_ops)."""
from __future__ import annotations

import numpy as np

from .. import _native as N
from .. import core
from ..columns import EventColumns, pair_events
from . import history as H
from . import memo as memo_ns
from . import model as M

_TYPE = {"invoke": N.INVOKE, "ok": N.OK, "fail": N.FAIL, "info": N.INFO}
_ALG = {None: N.ALG_COMPETITION, "competition": N.ALG_COMPETITION, "wgl": N.ALG_WGL, "linear": N.ALG_LINEAR}
_ALG_NAME = {N.ALG_WGL: "wgl", N.ALG_LINEAR: "linear", N.ALG_COMPETITION: "competition"}
_CAUSE = {N.CAUSE_TIME_LIMIT: "time-limit", N.CAUSE_STEP_LIMIT: "step-limit", N.CAUSE_VISITED_FULL: "memory"}


class _Interner:
    """Values cross the ABI as int32.  Ints pass through when every value is a
    small int; anything else (strings, tuples ...) is interned densely.  Only
    equality matters to the register family, so this preserves the verdict."""

    def __init__(self, values):
        vals = [v for v in values if v is not None]
        self.identity = all(isinstance(v, int) and not isinstance(v, bool) and -(2 ** 31) < v < 2 ** 31 for v in vals)
        self.ids, self.back = {}, []
        if not self.identity:
            for v in vals:
                k = memo_ns._freeze(v)
                if k not in self.ids:
                    self.ids[k] = len(self.back)
                    self.back.append(v)

    def enc(self, v):
        if v is None:
            return N.NIL
        return int(v) if self.identity else self.ids[memo_ns._freeze(v)]

    def dec(self, x):
        if x == N.NIL:
            return None
        return x if self.identity else self.back[x]


class Encoded:
    """A history ready for the C-ABI plus what is needed to decode the verdict."""

    def __init__(self, model, history):
        self.model = model
        hist = [op for op in H.index(list(history)) if H.client_op(op)]
        self.rows = hist                       # row r of the event columns == hist[r]
        n = len(hist)
        typ = np.zeros(n, np.uint8)
        proc = np.zeros(n, np.int32)
        f = np.zeros(n, np.uint8)
        a = np.zeros(n, np.int32)
        b = np.zeros(n, np.int32)
        self.table_info = None
        for r, op in enumerate(hist):
            typ[r] = _TYPE[op["type"]]
            proc[r] = op["process"]
        if isinstance(model, (M.Register, M.CASRegister)):
            vals = [model.value]
            for op in hist:
                v = op.get("value")
                if op["f"] == "cas" and v is not None:
                    vals.extend(v)
                else:
                    vals.append(v)
            self.intern = _Interner(vals)
            kind = N.MODEL_CAS_REGISTER if isinstance(model, M.CASRegister) else N.MODEL_REGISTER
            fmap = {"read": N.F_READ, "write": N.F_WRITE, "cas": N.F_CAS}
            for r, op in enumerate(hist):
                fc = fmap.get(op["f"])
                if fc is None or (fc == N.F_CAS and kind == N.MODEL_REGISTER):
                    raise ValueError(f"op {op['f']!r} is not understood by {model!r}")
                f[r] = fc
                v = op.get("value")
                if fc == N.F_CAS:
                    a[r], b[r] = (self.intern.enc(v[0]), self.intern.enc(v[1])) if v is not None else (N.NIL, N.NIL)
                else:
                    a[r] = self.intern.enc(v)
            self.native_model = core.make_model(kind, self.intern.enc(model.value))
        elif isinstance(model, M.Mutex):
            self.intern = None
            fmap = {"acquire": N.F_ACQUIRE, "release": N.F_RELEASE}
            for r, op in enumerate(hist):
                if op["f"] not in fmap:
                    raise ValueError(f"op {op['f']!r} is not understood by {model!r}")
                f[r] = fmap[op["f"]]
            self.native_model = core.make_model(N.MODEL_MUTEX, 1 if model.locked else 0)
        elif isinstance(model, M.MultiRegister) and self._multi_register_direct(model, hist, typ, f, a, b):
            pass
        else:
            # any other Model: knossos.model.memo -> transition table
            self.intern = None
            completed = H.complete(hist)
            info = memo_ns.memo(model, [op for op in completed if op["type"] in ("invoke", "ok")])
            self.table_info = info
            for r, op in enumerate(hist):
                f[r] = N.F_CLASS
                a[r] = info["classes"].get(memo_ns.op_class_key(op), 0)
            self.native_model = core.make_model(N.MODEL_TABLE, 0, info["table"])
        self.events = EventColumns(typ, proc, f, a, b)
        self.ops = pair_events(self.events)
        if getattr(self, "pool", None) is not None:
            self.ops.pool = self.pool

    def _multi_register_direct(self, model, hist, typ, f, a, b):
        """Device multi-register: <= 8 keys, <= 14 distinct values; txn micro-ops go to the value pool
        as {f, key, value} triples (op a = offset, b = count).  Returns False to fall back to the memo table."""
        keys, vals = {}, [v for _, v in model.values]
        for k, _ in model.values:
            keys.setdefault(k, len(keys))
        for op in hist:
            if op["f"] != "txn":
                raise ValueError(f"op {op['f']!r} is not understood by {model!r}")
            for mf, k, v in (op.get("value") or []):
                if mf not in ("r", "read", "w", "write"):
                    raise ValueError(f"unknown micro-op {mf!r}")
                keys.setdefault(k, len(keys))
                vals.append(v)
        self.intern = _Interner(vals)
        if not self.intern.identity:
            dense = self.intern
        else:   # small ints may still be out of the 0..13 range: intern them densely as well
            dense = _Interner([("v", v) for v in vals if v is not None])
            dense.enc0, dense.dec0 = dense.enc, dense.dec
            dense.enc = lambda v: N.NIL if v is None else dense.enc0(("v", v))
            dense.dec = lambda x: None if x == N.NIL else dense.dec0(x)[1]
            self.intern = dense
        if len(keys) > 8 or len(dense.back) > 14:
            return False
        pool = []
        for r, op in enumerate(hist):
            f[r] = N.F_TXN
            a[r] = len(pool)
            mops = op.get("value") or []
            b[r] = len(mops)
            for mf, k, v in mops:
                pool += [0 if mf in ("r", "read") else 1, keys[k], dense.enc(v)]
        self.pool = np.array(pool, np.int32)
        self.mr_keys = keys
        init = 0
        for k, v in model.values:
            init |= (dense.enc(v) + 1) << (4 * keys[k]) if v is not None else 0
        self.native_model = core.make_model(N.MODEL_MULTI_REGISTER, init, n_keys=max(1, len(keys)))
        return True

    # ---- decoding helpers
    def op_invocation(self, i):
        return self.rows[int(self.ops.inv_pos[i])]

    def op_completion(self, i):
        r = int(self.ops.ret_pos[i])
        return None if r == N.POS_CRASHED else self.rows[r]

    def model_of_state(self, s):
        if self.table_info is not None:
            return self.table_info["states"][s]
        if isinstance(self.model, M.Mutex):
            return M.Mutex(bool(s))
        if isinstance(self.model, M.MultiRegister):
            vals = {}
            for k, i in self.mr_keys.items():
                nib = (s >> (4 * i)) & 15
                if nib:
                    vals[k] = self.intern.dec(nib - 1)
            return M.MultiRegister(tuple(sorted(vals.items())))
        return type(self.model)(self.intern.dec(s))


def result_map(enc: Encoded, res: dict, algorithm):
    out = {"analyzer": "wgl" if res["analyzer"] == N.ALG_WGL else "linear",
           "configs": [], "final-paths": []}
    v = res["valid"]
    if v == N.VALID:
        out["valid?"] = True
        pending = [enc.op_invocation(i) for i in range(len(enc.ops))
                   if enc.ops.ret_pos[i] == N.POS_CRASHED]
        wit = res.get("witness")
        lin = set(int(x) for x in wit) if wit is not None else set()
        out["configs"] = [{"model": enc.model_of_state(res["final_state"]),
                           "last-op": enc.op_invocation(int(wit[-1])) if wit is not None and len(wit) else None,
                           "pending": [enc.op_invocation(i) for i in range(len(enc.ops))
                                       if enc.ops.ret_pos[i] == N.POS_CRASHED and i not in lin]}]
        out["witness"] = [enc.op_invocation(int(i)) for i in wit] if wit is not None else None
        del pending
    elif v == N.INVALID:
        out["valid?"] = False
        out["op"] = enc.op_completion(res["fail_op"])
        out["previous-ok"] = enc.op_completion(res["prev_ok_op"]) if res["prev_ok_op"] is not None else None
        for c in res.get("configs", []):
            out["configs"].append({"model": enc.model_of_state(c["state"]),
                                   "last-op": enc.op_invocation(c["last_op"]) if c["last_op"] is not None else None,
                                   "pending": [enc.op_invocation(i) for i in c["pending"]]})
    else:
        out["valid?"] = "unknown"
        out["cause"] = _CAUSE.get(res["cause"], "unknown")
    out["stats"] = {k: res[k] for k in ("steps", "visited", "probes", "backtracks", "max_depth",
                                        "table_slots", "ns_pack", "ns_search", "ns_total")}
    return out


def analysis(model, history, algorithm="wgl", **opts):
    enc = Encoded(model, history)
    o = core.make_opts(algorithm=_ALG[algorithm], device=opts.get("device", 0),
                       time_limit_ms=int(opts.get("time-limit", opts.get("time_limit", 0)) or 0),
                       max_steps=opts.get("max-steps", opts.get("max_steps", 0)) or 0,
                       max_visited_bytes=opts.get("max-visited-bytes", opts.get("max_visited_bytes", 0)) or 0,
                       want_witness=True)
    res = core.check_ops(enc.ops, enc.native_model, o)
    return result_map(enc, res, algorithm)
