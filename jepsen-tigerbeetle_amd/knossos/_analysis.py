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

    def __init__(self, model, history, shared_values=None):
        """shared_values: values every history of one batch must agree on (jepsen.independent checks all keys in
        one launch with ONE model struct, so the register values of all keys are interned together)."""
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
            vals = [model.value] + list(shared_values or [])
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
        elif isinstance(model, M.SetModel) and self._set_direct(model, hist, typ, f, a, b):
            pass
        elif isinstance(model, M.Bank) and self._bank_direct(model, hist, typ, f, a, b):
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

    @staticmethod
    def _pairs(hist):
        """(invoke row, completion row or None, completion type) per client op, in invocation order."""
        open_by_proc, out = {}, []
        for r, op in enumerate(hist):
            p = op["process"]
            if op["type"] == "invoke":
                open_by_proc[p] = len(out)
                out.append([r, None, None])
            elif p in open_by_proc:
                i = open_by_proc.pop(p)
                out[i][1], out[i][2] = r, op["type"]
        return out

    def _set_direct(self, model, hist, typ, f, a, b):
        """knossos.model/set as a COMMUTATIVE device model (TBC_MODEL_SET): the state is the set of
        linearized adds, so configs carry no state; a read is checked against the adds completed
        before the front plus the open adds already linearized.  Needs unique elements and an
        empty initial set; otherwise the memo table (small histories) is used."""
        if model.s:
            return False
        pairs = self._pairs(hist)
        adds = [(i, pr) for i, pr in enumerate(pairs) if hist[pr[0]]["f"] == "add"]
        if any(hist[pr[0]]["f"] not in ("add", "read") for pr in pairs):
            raise ValueError(f"op not understood by {model!r}")
        elems = [memo_ns._freeze(hist[pr[0]]["value"]) for _, pr in adds if pr[2] != "fail"]
        if len(set(elems)) != len(elems):
            return False
        # adds in completion order (crashed ones last, by invocation); failed adds never happened
        live = [(pr[1], pr) for _, pr in adds if pr[2] == "ok"]
        crashed = [pr for _, pr in adds if pr[2] not in ("ok", "fail")]
        order = [pr for _, pr in sorted(live, key=lambda t: t[0])] + crashed
        j_of = {memo_ns._freeze(hist[pr[0]]["value"]): j for j, pr in enumerate(order)}
        n_adds = len(order)
        nwords = max(1, (n_adds + 31) // 32)
        # completions (of ops that are kept) in row order -> nadds_before[F]
        comp_rows = sorted((pr[1], hist[pr[0]]["f"] == "add") for pr in pairs if pr[2] == "ok")
        nb = [0]
        for _, is_add in comp_rows:
            nb.append(nb[-1] + (1 if is_add else 0))
        chunks, pool_len = [np.array(nb, np.int32)], len(nb)
        ints = all(isinstance(k, int) and not isinstance(k, bool) for k in j_of)
        if ints and j_of:
            elem_arr = np.array(sorted(j_of), np.int64)
            j_arr = np.array([j_of[int(x)] for x in elem_arr], np.int64)
        nbits = nwords * 32
        for pr in pairs:
            r0, r1, ctype = pr
            op0 = hist[r0]
            if op0["f"] == "add":
                j = j_of.get(memo_ns._freeze(op0["value"]), 0)
                for r in (r0, r1):
                    if r is not None:
                        f[r] = N.F_ADD; a[r] = j
            else:
                f[r0] = N.F_READ; a[r0] = N.NIL
                if r1 is not None:
                    f[r1] = N.F_READ
                    v = hist[r1].get("value") if ctype == "ok" else None
                    if v is None:
                        a[r1] = N.NIL
                        continue
                    v = list(v)
                    bits = np.zeros(nbits, bool)
                    if ints and j_of and all(isinstance(x, int) and not isinstance(x, bool) for x in v):
                        arr = np.asarray(v, np.int64) if v else np.zeros(0, np.int64)
                        pos = np.searchsorted(elem_arr, arr)
                        pos_c = np.minimum(pos, len(elem_arr) - 1)
                        hit = elem_arr[pos_c] == arr
                        ok = bool(hit.all())
                        bits[j_arr[pos_c[hit]]] = True
                        n_distinct = len(np.unique(arr))
                    else:
                        ok = True
                        for e in v:
                            j = j_of.get(memo_ns._freeze(e))
                            if j is None:
                                ok = False
                            else:
                                bits[j] = True
                        n_distinct = len(set(memo_ns._freeze(e) for e in v))
                    head = bits[:n_adds]
                    lead = int(n_adds if head.all() else np.argmin(head))
                    words = np.packbits(bits, bitorder="little").view(np.uint32).astype(np.int64)
                    words = np.where(words >= (1 << 31), words - (1 << 32), words).astype(np.int32)
                    a[r1] = pool_len
                    rec = np.concatenate([np.array([n_distinct if ok else -1, lead], np.int32), words])
                    chunks.append(rec)
                    pool_len += len(rec)
        pool = np.concatenate(chunks)
        self.intern = None
        self.pool = np.array(pool, np.int32)
        self.set_order = [hist[pr[0]]["value"] for pr in order]
        self.n_adds = n_adds
        self.native_model = core.make_model(N.MODEL_SET, 0)
        return True

    def _bank_direct(self, model, hist, typ, f, a, b):
        """The bank model (new: Knossos has none; reference mapping tests/ledger.clj:89-114) as a
        COMMUTATIVE device model (TBC_MODEL_BANK) when negative balances are allowed: transfers
        commute, so configs carry no state and a read is checked against the balances after the
        transfers completed before the front plus the open transfers already linearized."""
        if not model.neg or any(v != 0 for _, v in model.balances):
            return False
        accts = {acct: i for i, (acct, _) in enumerate(model.balances)}
        A = len(accts)
        if A > 16:
            return False
        pairs = self._pairs(hist)
        comp = sorted((pr[1], pr) for pr in pairs if pr[2] == "ok")
        bal = [0] * A
        pool = list(bal)                                # bal_before[0]
        for _, pr in comp:
            op0 = hist[pr[0]]
            if op0["f"] == "transfer":
                v = op0["value"]
                if v["debit-acct"] not in accts or v["credit-acct"] not in accts:
                    raise ValueError(f"unknown account in {v}")
                bal[accts[v["debit-acct"]]] -= v["amount"]
                bal[accts[v["credit-acct"]]] += v["amount"]
            pool += bal                                 # bal_before[F + 1]
        for pr in pairs:
            r0, r1, ctype = pr
            op0 = hist[r0]
            if op0["f"] == "transfer":
                v = op0["value"]
                if v["debit-acct"] not in accts or v["credit-acct"] not in accts:
                    raise ValueError(f"unknown account in {v}")
                off = len(pool)
                pool += [accts[v["debit-acct"]], accts[v["credit-acct"]], int(v["amount"])]
                for r in (r0, r1):
                    if r is not None:
                        f[r] = N.F_TRANSFER; a[r] = off
            elif op0["f"] == "read":
                f[r0] = N.F_READ; a[r0] = N.NIL
                if r1 is not None:
                    f[r1] = N.F_READ
                    v = hist[r1].get("value") if ctype == "ok" else None
                    if v is None:
                        a[r1] = N.NIL
                    else:
                        v = dict(v)
                        if set(v) != set(accts):
                            raise ValueError(f"read of accounts {sorted(v)} but the model has {sorted(accts)}")
                        a[r1] = len(pool)
                        pool += [int(v[acct]) for acct in accts]
            else:
                raise ValueError(f"op {op0['f']!r} is not understood by {model!r}")
        self.intern = None
        self.pool = np.array(pool, np.int32)
        self.bank_accts = list(accts)
        self.native_model = core.make_model(N.MODEL_BANK, 0, n_keys=A)
        return True

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
        if isinstance(self.model, (M.SetModel, M.Bank)):
            return self.model          # commutative device models carry no state in the config
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
        out["final-paths"] = final_paths(enc, res)
    else:
        out["valid?"] = "unknown"
        out["cause"] = _CAUSE.get(res["cause"], "unknown")
    out["stats"] = {k: res[k] for k in ("steps", "visited", "probes", "backtracks", "max_depth",
                                        "table_slots", "ns_pack", "ns_search", "ns_total")}
    return out


def final_paths(enc: Encoded, res: dict, limit=10):
    """knossos.linear's :final-paths of an invalid verdict (recalled shape; the Jepsen tutorial prints
    `[{:op {... :f :write, :value 4} :model {:value 4}} {:op {... :f :read, :value 3} :model {:msg "can't read 3 from
    register 4"}}]`): for each config stuck in front of the failing completion, the ways its last steps end in an
    inconsistent model -- the failing op applied directly, or after one more pending call the model accepts.
    Built on the host from the <= 10 configs the device returns and the Python model's own step (so :msg is the
    model's wording).  The state-free device models (set, bank) carry no state in their configs: no paths."""
    if isinstance(enc.model, (M.SetModel, M.Bank)) or res.get("fail_op") is None:
        return []
    x = enc.op_completion(res["fail_op"])
    paths = []
    for c in res.get("configs", []):
        m = enc.model_of_state(c["state"])
        prefix = []
        if c["last_op"] is not None:
            prefix = [{"op": enc.op_completion(c["last_op"]) or enc.op_invocation(c["last_op"]), "model": m}]
        mx = m.step(x)
        if M.inconsistent_p(mx):
            paths.append(prefix + [{"op": x, "model": mx}])
        for k, i in enumerate(c["pending"]):
            if len(paths) >= limit:
                break
            if (c["linearized_mask"] >> k) & 1 or i == res["fail_op"]:
                continue
            y = enc.op_completion(i) or enc.op_invocation(i)
            m1 = m.step(y)
            if M.inconsistent_p(m1):
                continue
            m2 = m1.step(x)
            if M.inconsistent_p(m2):
                paths.append(prefix + [{"op": y, "model": m1}, {"op": x, "model": m2}])
        if len(paths) >= limit:
            break
    return paths[:limit]


def register_values(history):
    """Every value a register-family history mentions (for Encoded's shared_values)."""
    out = []
    for op in history:
        v = op.get("value")
        if op.get("f") == "cas" and v is not None:
            out.extend(v)
        else:
            out.append(v)
    return out


def make_opts_from(algorithm, opts, want_witness):
    return core.make_opts(algorithm=_ALG[algorithm], device=opts.get("device", 0),
                          time_limit_ms=int(opts.get("time-limit", opts.get("time_limit", 0)) or 0),
                          max_steps=opts.get("max-steps", opts.get("max_steps", 0)) or 0,
                          max_visited_bytes=opts.get("max-visited-bytes", opts.get("max_visited_bytes", 0)) or 0,
                          want_witness=want_witness)


def analysis(model, history, algorithm="wgl", **opts):
    enc = Encoded(model, history)
    o = core.make_opts(algorithm=_ALG[algorithm], device=opts.get("device", 0),
                       time_limit_ms=int(opts.get("time-limit", opts.get("time_limit", 0)) or 0),
                       max_steps=opts.get("max-steps", opts.get("max_steps", 0)) or 0,
                       max_visited_bytes=opts.get("max-visited-bytes", opts.get("max_visited_bytes", 0)) or 0,
                       # knossos.linear / competition return configs, not a linearization: asking for one
                       # ("witness": True) keeps the answer with the depth-first search
                       want_witness=bool(opts.get("witness", algorithm == "wgl")))
    res = core.check_ops(enc.ops, enc.native_model, o)
    return result_map(enc, res, algorithm)
