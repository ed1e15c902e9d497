"""jepsen.checker/set-full on the MI355X -- the checker the reference runs for its set-full workload
(/root/reference/src/tigerbeetle/workloads/set_full.clj:155-158:
`(independent/checker (checker/compose {:set-full (checker/set-full {:linearizable? true}) ...}))`).

Host side: flatten the history into the reads x elements membership matrix and four index columns, call
`tbc_setfull_*` (csrc/set_full.hip scans the matrix for known / last-present / last-absent per element), turn the
three indices into jepsen's result map (:valid? :attempt-count :stable-count :lost :never-read :stale :worst-stale
...).  The semantics are recalled from jepsen.checker (jepsen is not in /root/reference and cannot run here) and
restated independently in oracle/set_full.py, which the tests compare with.  No CPU fallback: without a GPU
`check` raises NoDeviceError."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _native as N
from ..columns import _p
from ..knossos import history as H

NONE = 0xFFFFFFFF


class Encoded:
    """reads x elements bit matrix + index columns of one history (one key).

    Recalled jepsen.checker/set-full details kept here: an :add INVOCATION creates the element's state, so adding an
    element again starts it afresh (its column is numbered by its LAST add invocation and only reads completing after
    that count); an element that occurs more than once in one read's value is a duplicate (`duplicated`: element ->
    greatest multiplicity seen), which makes the result invalid."""

    def __init__(self, history):
        hist = [op for op in H.index(list(history)) if H.client_op(op)]
        self.has_time = all("time" in op for op in hist) and bool(hist)
        self.times = {op["index"]: op.get("time", op["index"]) for op in hist}
        last_invoke, add_ok, reads, open_reads = {}, {}, [], {}
        self.duplicated = {}
        for op in hist:
            i = op["index"]
            if op["f"] == "add":
                v = _freeze(op["value"])
                if op["type"] == "invoke":
                    last_invoke[v] = (i, op["value"])
                    add_ok.pop(v, None)                        # a fresh element state: nothing known yet
                elif op["type"] == "ok" and v in last_invoke and v not in add_ok:
                    add_ok[v] = i
            elif op["f"] == "read":
                if op["type"] == "invoke":
                    open_reads[op["process"]] = i
                elif op["type"] == "fail":
                    open_reads.pop(op["process"], None)
                elif op["type"] == "ok":
                    inv = open_reads.pop(op["process"], None)
                    if inv is not None and op.get("value") is not None:
                        reads.append((inv, i, op["value"]))
        reads.sort(key=lambda r: r[0])
        by_invoke = sorted(last_invoke.items(), key=lambda kv: kv[1][0])
        order = [val for _, (_, val) in by_invoke]
        elems = {k: (n, inv) for n, (k, (inv, _)) in enumerate(by_invoke)}
        self.elements = order
        E, R = len(order), len(reads)
        self.E, self.R = E, R
        self.wpr = max(1, (E + 31) // 32)
        self.add_invoke = np.array([elems[_freeze(v)][1] for v in order], np.uint32)
        self.add_ok = np.array([add_ok.get(_freeze(v), NONE) for v in order], np.uint32)
        self.read_invoke = np.array([r[0] for r in reads], np.uint32)
        self.read_ok = np.array([r[1] for r in reads], np.uint32)
        # the reads in COMPACT form (tbc_setfull_rows): top[r] = every element numbered below it is in read r, except the listed
        # exceptions -- a listed element below top is absent, one at or above it present.  A read of a grow-only set is a prefix of
        # the elements in add-invocation order with a few holes, so top = greatest element read + 1 and the holes are the list.
        self.top = np.zeros(max(R, 1), np.uint32)
        self.exc_off = np.zeros(R + 1, np.uint64)
        exc_parts = []
        cols_of = []                                       # per read: the column numbers it contains (sorted, unique)
        ints = all(isinstance(v, int) and not isinstance(v, bool) for v in order)
        if ints and E:
            keys = np.array(order, np.int64)
            srt = np.argsort(keys)
            ks = keys[srt]
        for r, (_, _, val) in enumerate(reads):
            vals = list(val)
            if not vals:
                continue
            if ints and E and all(isinstance(x, int) and not isinstance(x, bool) for x in vals):
                arr = np.asarray(vals, np.int64)
                if len(arr) > 1:
                    u, c = np.unique(arr, return_counts=True)
                    for x, n in zip(u[c > 1].tolist(), c[c > 1].tolist()):
                        self.duplicated[x] = max(self.duplicated.get(x, 0), n)
                pos = np.minimum(np.searchsorted(ks, arr), E - 1)
                hit = ks[pos] == arr                       # values nobody added are not columns: jepsen ignores them here too
                cols_of.append((r, np.unique(srt[pos[hit]])))
            else:
                seen = {}
                for x in vals:
                    fx = _freeze(x)
                    seen[fx] = seen.get(fx, 0) + 1
                cols_of.append((r, np.unique(np.array([elems[fx][0] for fx in seen if fx in elems], np.int64))))
                for fx, n in seen.items():
                    if n > 1:
                        self.duplicated[fx] = max(self.duplicated.get(fx, 0), n)
        counts = np.zeros(R, np.int64)
        for r, cols in cols_of:
            if len(cols) == 0:
                continue
            t = int(cols[-1]) + 1
            self.top[r] = t
            holes = np.setdiff1d(np.arange(t, dtype=np.int64), cols, assume_unique=True)
            counts[r] = len(holes)
            if len(holes):
                exc_parts.append(holes.astype(np.uint32))
        self.exc_off[1:] = np.cumsum(counts)
        self.exc = np.concatenate(exc_parts) if exc_parts else np.zeros(0, np.uint32)
        self._cols_of = cols_of

    @property
    def present(self):
        """The dense reads x elements bit matrix (tbc_setfull_in.present), built on demand: the checker itself hands the compact form
        to the device, which builds the matrix there; tests and the dense entry point take this one."""
        bits = np.zeros((max(self.R, 1), self.wpr * 32), bool)
        for r, cols in self._cols_of:
            bits[r, cols] = True
        return np.ascontiguousarray(np.packbits(bits, axis=1, bitorder="little").view(np.uint32))


def _freeze(v):
    return tuple(v) if isinstance(v, list) else v


class Scan:
    """tbc_setfull_*: the matrix resident in HBM, `run()` scans it."""

    def __init__(self, enc_or_arrays, device=0, rows=None):
        """rows: True = hand the reads over in compact form (top / exc_off / exc: tbc_setfull_create_rows, the matrix is built on
        the device), False = the dense bit matrix (`present`); None = compact where the argument has it."""
        a = enc_or_arrays
        self._keep = a
        if rows is None:
            rows = hasattr(a, "top") and hasattr(a, "exc_off")
        self.E = a.E
        self._h = C.c_void_p()
        if rows:
            top = np.ascontiguousarray(a.top, np.uint32)
            off = np.ascontiguousarray(a.exc_off, np.uint64)
            exc = np.ascontiguousarray(a.exc if len(a.exc) else np.zeros(1, np.uint32), np.uint32)
            self._keep = (a, top, off, exc)
            s = N.SetFullRows()
            s.n_elements, s.n_reads, s.device, s.reserved0 = a.E, a.R, device, 0
            s.add_invoke, s.add_ok = _p(a.add_invoke, C.c_uint32), _p(a.add_ok, C.c_uint32)
            s.read_invoke, s.read_ok = _p(a.read_invoke, C.c_uint32), _p(a.read_ok, C.c_uint32)
            s.top, s.exc_off, s.exc = _p(top, C.c_uint32), _p(off, C.c_uint64), _p(exc, C.c_uint32)
            N.check_status(N.lib().tbc_setfull_create_rows(C.byref(s), C.byref(self._h)))
            return
        present = a.present
        self._keep = (a, present)
        s = N.SetFullIn()
        s.n_elements, s.n_reads, s.words_per_row, s.device = a.E, a.R, a.wpr, device
        s.add_invoke, s.add_ok = _p(a.add_invoke, C.c_uint32), _p(a.add_ok, C.c_uint32)
        s.read_invoke, s.read_ok = _p(a.read_invoke, C.c_uint32), _p(a.read_ok, C.c_uint32)
        s.present = _p(present, C.c_uint32)
        N.check_status(N.lib().tbc_setfull_create(C.byref(s), C.byref(self._h)))

    def run(self):
        n = max(1, self.E)
        known, lp, la = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        o = N.SetFullOut()
        o.known, o.last_present, o.last_absent = _p(known, C.c_uint32), _p(lp, C.c_uint32), _p(la, C.c_uint32)
        N.check_status(N.lib().tbc_setfull_run(self._h, C.byref(o)))
        return {"known": known[:self.E], "last_present": lp[:self.E], "last_absent": la[:self.E],
                "ns_scan": o.ns_scan, "bytes_scanned": o.bytes_scanned, "bytes_matrix": o.bytes_matrix}

    def close(self):
        if self._h:
            N.lib().tbc_setfull_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def frequency_distribution(points, xs):
    """jepsen.checker's latency summary (recalled): {point: the value at floor(n * point) of the sorted sample, clamped}."""
    srt = sorted(xs)
    n = len(srt)
    return {p: srt[min(n - 1, int(n * p))] for p in points} if n else None


def result_map(enc: Encoded, st: dict, linearizable: bool):
    """jepsen.checker/set-full's result from the three indices per element (vectorised: a few operations each).
    Latencies as jepsen computes them: max(0, dt) in nanoseconds -> whole milliseconds (util/nanos->ms, then long)
    when the ops carry :time; histories without :time (hand-written ones) keep the index difference.  So a gap
    below one millisecond is not a stale read, exactly as in jepsen."""
    k = st["known"].astype(np.int64)
    lp = np.where(st["last_present"] == NONE, -1, st["last_present"].astype(np.int64))
    la = np.where(st["last_absent"] == NONE, -1, st["last_absent"].astype(np.int64))
    known = st["known"] != NONE
    stable = (lp >= 0) & (la < lp)
    lost = known & (la >= 0) & (lp < la) & (k < la)
    never = ~(stable | lost)
    t = enc.times
    tk = np.array([t[int(x)] if kn else 0 for x, kn in zip(k, known)], np.int64)
    t_la = np.array([t[int(x)] + 1 if x >= 0 else 0 for x in la], np.int64)
    t_lp = np.array([t[int(x)] + 1 if x >= 0 else 0 for x in lp], np.int64)
    unit = 1_000_000 if getattr(enc, "has_time", False) else 1
    stable_lat = np.maximum(0, t_la - tk) // unit
    lost_lat = np.maximum(0, t_lp - tk) // unit
    el = enc.elements
    idx = lambda m: [el[i] for i in np.nonzero(m)[0]]
    stale = stable & (stable_lat > 0)
    worst = sorted(np.nonzero(stale)[0], key=lambda i: -int(stable_lat[i]))[:8]
    dups = dict(sorted(getattr(enc, "duplicated", {}).items(), key=lambda kv: repr(kv[0])))
    valid = False if lost.any() else ("unknown" if not stable.any() else (False if (linearizable and stale.any()) else True))
    if dups:
        valid = False                                   # (and (empty? dups) (:valid? results)): nil / :unknown -> falsey
    points = (0, 0.5, 0.95, 0.99, 1)
    out = {"valid?": valid, "attempt-count": enc.E, "stable-count": int(stable.sum()), "lost-count": int(lost.sum()),
           "lost": _sorted(idx(lost)), "never-read-count": int(never.sum()), "never-read": _sorted(idx(never)),
           "stale-count": int(stale.sum()), "stale": _sorted(idx(stale)),
           "worst-stale": [{"element": el[i], "outcome": "stable", "stable-latency": int(stable_lat[i]), "lost-latency": None,
                            "known": int(k[i]), "last-absent": int(la[i]) if la[i] >= 0 else None} for i in worst],
           "duplicated-count": len(dups), "duplicated": dups}
    if stable.any():
        out["stable-latencies"] = frequency_distribution(points, [int(x) for x in stable_lat[stable]])
    if lost.any():
        out["lost-latencies"] = frequency_distribution(points, [int(x) for x in lost_lat[lost]])
    return out


def _sorted(xs):
    try:
        return sorted(xs)
    except TypeError:
        return xs


def check(history, linearizable=False, device=0):
    enc = Encoded(history)
    with Scan(enc, device) as s:
        st = s.run()
    return result_map(enc, st, linearizable)
