"""jepsen.checker -- the Checker protocol, `linearizable` and `compose`.

(check checker test history opts) -> map with :valid? in true | false |
:unknown; shape visible in the reference at workloads/set_full.clj:54-55 and
tests/ledger.clj:159-160, :unknown at tests/ledger.clj:216,219; compose used at
core.clj:139-146, set_full.clj:156-158, tests/ledger.clj:363-367.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

from ..knossos import competition, linear, wgl


class Checker:
    def check(self, test, history, opts=None):  # pragma: no cover
        raise NotImplementedError


def merge_valid(valids):
    """jepsen.checker/merge-valid: false beats :unknown beats true."""
    out = True
    for v in valids:
        if v is False:
            return False
        if v == "unknown":
            out = "unknown"
    return out


class Linearizable(Checker):
    """(checker/linearizable {:model m :algorithm :linear | :wgl | nil})."""

    def __init__(self, opts):
        self.model = opts["model"]
        self.algorithm = opts.get("algorithm")
        self.opts = {k: v for k, v in opts.items() if k not in ("model", "algorithm")}

    def check(self, test, history, opts=None):
        fn = {"linear": linear.analysis, "wgl": wgl.analysis}.get(self.algorithm, competition.analysis)
        a = fn(self.model, history, self.opts)
        # jepsen truncates these so results.edn stays readable
        a["final-paths"] = list(a.get("final-paths", []))[:10]
        a["configs"] = list(a.get("configs", []))[:10]
        return a


def linearizable(opts):
    return Linearizable(opts)


class SetFull(Checker):
    """(checker/set-full {:linearizable? bool}) -- per-element timing analysis of a grow-only set; the scan of the
    reads x elements matrix runs on the GPU (jepsen/set_full.py, csrc/set_full.hip).  The reference's own use:
    workloads/set_full.clj:157."""

    def __init__(self, opts=None):
        self.linearizable = bool((opts or {}).get("linearizable?", False))

    def check(self, test, history, opts=None):
        from . import set_full as sf
        return sf.check(history, self.linearizable, device=(opts or {}).get("device", 0))


def set_full(opts=None):
    return SetFull(opts)


class Compose(Checker):
    def __init__(self, checkers):
        self.checkers = dict(checkers)

    def check(self, test, history, opts=None):
        # jepsen evaluates sub-checkers concurrently; libtbcheck is re-entrant
        with ThreadPoolExecutor(max_workers=max(1, len(self.checkers))) as ex:
            futs = {k: ex.submit(c.check, test, history, opts) for k, c in self.checkers.items()}
            res = {k: f.result() for k, f in futs.items()}
        res["valid?"] = merge_valid(r.get("valid?") for k, r in res.items() if isinstance(r, dict))
        return res


def compose(checkers):
    return Compose(checkers)
