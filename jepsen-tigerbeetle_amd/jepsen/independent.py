"""jepsen.independent -- tuple values and the per-key checker.

The reference wraps its set-full checkers this way
(workloads/set_full.clj:29-31,42-45 build [k v] tuples; :155 wraps the
checker).  Keys are independent objects, so each key's sub-history is
checked on its own (P-compositionality) -- and when the wrapped checker is the
GPU linearizability checker, ALL keys go to the device in one batch launch
(tbc_batch_*), one history per wavefront.
"""
from __future__ import annotations

from .. import _native as N
from .. import core
from ..knossos import _analysis
from . import checker as jc


class Tuple(tuple):
    """(independent/tuple k v)"""

    def __new__(cls, k, v):
        return super().__new__(cls, (k, v))

    key = property(lambda self: self[0])
    value = property(lambda self: self[1])


def tuple_(k, v):
    return Tuple(k, v)


def tuple_p(x):
    return isinstance(x, Tuple)


def history_keys(history):
    keys = []
    seen = set()
    for op in history:
        v = op.get("value")
        if tuple_p(v) and v.key not in seen:
            seen.add(v.key)
            keys.append(v.key)
    return keys


def subhistory(k, history):
    """Ops of key k with the tuple stripped; non-tuple ops (nemesis ...) are kept for every key."""
    out = []
    for op in history:
        v = op.get("value")
        if tuple_p(v):
            if v.key == k:
                out.append(dict(op, value=v.value))
        else:
            out.append(op)
    return out


class IndependentChecker(jc.Checker):
    def __init__(self, inner):
        self.inner = inner

    def check(self, test, history, opts=None):
        keys = history_keys(history)
        subs = {k: subhistory(k, history) for k in keys}
        if isinstance(self.inner, jc.Linearizable) and keys:
            results = self._check_batched(keys, subs)
        else:
            results = {k: self.inner.check(test, subs[k], opts) for k in keys}
        failures = [k for k in keys if results[k].get("valid?") is False]
        return {"valid?": jc.merge_valid(r.get("valid?") for r in results.values()),
                "results": results, "failures": failures}

    def _check_batched(self, keys, subs):
        lin = self.inner
        shared = None
        if isinstance(lin.model, (_analysis.M.Register, _analysis.M.CASRegister)):
            # ONE model struct serves the whole batch: intern the values of all keys together, so that the encoded
            # initial value (and every other value) means the same thing in every key's columns
            shared = [v for k in keys for v in _analysis.register_values(subs[k])]
        encs = [_analysis.Encoded(lin.model, subs[k], shared_values=shared) for k in keys]
        kinds = {e.native_model[0].kind for e in encs}
        if (len(kinds) != 1 or N.MODEL_TABLE in kinds or len({e.native_model[0].n_keys for e in encs}) != 1
                or len({e.native_model[0].init for e in encs}) != 1 and N.MODEL_SET not in kinds and N.MODEL_BANK not in kinds):
            # table models have one table per key, and keys that do not agree on the encoded model: one by one
            return {k: lin.check(None, subs[k], None) for k in keys}
        want_witness = bool(lin.opts.get("witness", lin.algorithm == "wgl"))
        o = _analysis.make_opts_from(lin.algorithm, lin.opts, want_witness)
        with core.Batch([e.ops for e in encs], encs[0].native_model, o) as b:
            b.run(tolerate_bad_histories=True)
            res = b.results()
        out = {}
        for k, e, r in zip(keys, encs, res):
            if r["valid"] == N.UNKNOWN and r["cause"] == N.CAUSE_NONE:
                # rejected by the device-side validation (malformed rows): this key only, not the batch
                out[k] = {"valid?": "unknown", "cause": "malformed-history", "analyzer": "wgl", "configs": [], "final-paths": []}
                continue
            a = _analysis.result_map(e, r, lin.algorithm)
            a["final-paths"] = a["final-paths"][:10]
            a["configs"] = a["configs"][:10]
            out[k] = a
        return out


def checker(inner):
    return IndependentChecker(inner)
