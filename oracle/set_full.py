"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle_model.h).

Plain-Python restatement of jepsen.checker/set-full, the checker the reference actually runs for its set-full
workload (/root/reference/src/tigerbeetle/workloads/set_full.clj:157: `(checker/set-full {:linearizable? true})`).
jepsen is not in /root/reference (it arrives through project.clj:8) and cannot run here, so this follows the
algorithm as RECALLED from jepsen.checker (MED confidence; PARITY UNPINNED like everything else):

  a fold over the history keeps, per element (created at its :add invocation), three ops --
    known         the first op that proved the element exists: its add's :ok, or the :ok of a read that contains it
    last-present  the INVOCATION of the latest-invoked :ok read that contained it
    last-absent   the INVOCATION of the latest-invoked :ok read that did not (only reads that complete after the
                  element's add was invoked count, for both)
  and per element then:
    stable?  last-present exists and was invoked after last-absent
    lost?    known, last-absent exists, invoked after last-present and after known
    never-read otherwise
    stable-latency = max(0, (time just after last-absent, or 0) - time known) in WHOLE MILLISECONDS (nanos->ms, long);
    lost-latency likewise from last-present
  an element is (re)created by every :add invocation (adding it again starts its state afresh); an element occurring
  more than once in one read's value is a duplicate: :duplicated {element count}, :duplicated-count, and :valid? false
  result: :valid? false if anything was lost; :unknown if nothing is stable; false if :linearizable? and some
  stable element has latency > 0 ms (a stale read); else true; false whenever something was duplicated.
  :stable-latencies / :lost-latencies are quantile maps {0 .5 .95 .99 1}.  Times are op :time (ns) when every client op
  carries one, else the :index (hand-written histories: no ms conversion).

`element_states(history)` returns, per element in order of first :add invocation, the three op indices (the part
the GPU kernel computes: csrc/set_full.hip); `check(history, linearizable)` the whole result map.
"""
NONE = 0xFFFFFFFF


def _client(op):
    return isinstance(op.get("process"), int)


def fold(history):
    """The fold of jepsen.checker/set-full as recalled: an :add invocation (re)creates the element's state; an :add :ok
    makes it known if nothing did before; an :ok read updates EVERY element created so far (present / absent, by the
    read's invocation) and reports elements occurring more than once in its value.  Returns (states in order of the
    creating invocation, {element: greatest multiplicity seen in one read})."""
    elems = {}
    open_reads, dups = {}, {}
    for idx, op in enumerate(history):
        if not _client(op):
            continue
        i = op.get("index", idx)
        if op["f"] == "add":
            v = op["value"]
            if op["type"] == "invoke":
                elems[v] = {"element": v, "add_invoke": i, "known": NONE, "last_present": NONE, "last_absent": NONE}
            elif op["type"] == "ok" and v in elems:
                if elems[v]["known"] == NONE:
                    elems[v]["known"] = i
        elif op["f"] == "read":
            if op["type"] == "invoke":
                open_reads[op["process"]] = i
            elif op["type"] == "fail":
                open_reads.pop(op["process"], None)
            elif op["type"] == "ok":
                inv = open_reads.pop(op["process"], None)
                if inv is None or op.get("value") is None:
                    continue
                counts = {}
                for x in op["value"]:
                    counts[x] = counts.get(x, 0) + 1
                for x, n in counts.items():
                    if n > 1 and n > dups.get(x, 0):
                        dups[x] = n
                for v, e in elems.items():          # every element created so far (its add was invoked before this completion)
                    if v in counts:
                        if e["known"] == NONE:
                            e["known"] = i
                        if e["last_present"] == NONE or e["last_present"] < inv:
                            e["last_present"] = inv
                    elif e["last_absent"] == NONE or e["last_absent"] < inv:
                        e["last_absent"] = inv
    return sorted(elems.values(), key=lambda e: e["add_invoke"]), dups


def element_states(history):
    return fold(history)[0]


def _nanos_to_ms(dt, has_time):
    """jepsen: (-> t (- known-time) (max 0) util/nanos->ms long); histories without :time keep the index difference"""
    dt = max(0, dt)
    return int(dt // 1_000_000) if has_time else dt


def outcomes(states, time_of, has_time=False):
    """per-element outcome maps from the three indices; time_of(index) -> time."""
    out = []
    for e in states:
        k, lp, la = e["known"], e["last_present"], e["last_absent"]
        lp_i = -1 if lp == NONE else lp
        la_i = -1 if la == NONE else la
        stable = lp != NONE and la_i < lp_i
        lost = k != NONE and la != NONE and lp_i < la_i and k < la_i
        r = {"element": e["element"], "outcome": "stable" if stable else ("lost" if lost else "never-read"),
             "stable-latency": None, "lost-latency": None, "known": None if k == NONE else k,
             "last-absent": None if la == NONE else la}
        if stable:
            t = (time_of(la) + 1) if la != NONE else 0
            r["stable-latency"] = _nanos_to_ms(t - time_of(k), has_time)
        if lost:
            t = (time_of(lp) + 1) if lp != NONE else 0
            r["lost-latency"] = _nanos_to_ms(t - time_of(k), has_time)
        out.append(r)
    return out


def _quantiles(xs):
    srt = sorted(xs)
    n = len(srt)
    return {p: srt[min(n - 1, int(n * p))] for p in (0, 0.5, 0.95, 0.99, 1)}


def check(history, linearizable=False):
    hist = [dict(op, index=op.get("index", i)) for i, op in enumerate(history)]
    clients = [op for op in hist if _client(op)]
    has_time = bool(clients) and all("time" in op for op in clients)
    times = {op["index"]: op.get("time", op["index"]) for op in hist}
    states, dups = fold(hist)
    rs = outcomes(states, lambda i: times[i], has_time)
    stable = [r for r in rs if r["outcome"] == "stable"]
    lost = [r for r in rs if r["outcome"] == "lost"]
    never = [r for r in rs if r["outcome"] == "never-read"]
    stale = [r for r in stable if r["stable-latency"] > 0]
    valid = False if lost else ("unknown" if not stable else (False if (linearizable and stale) else True))
    if dups:
        valid = False
    out = {"valid?": valid, "attempt-count": len(rs), "stable-count": len(stable), "lost-count": len(lost),
           "lost": sorted(r["element"] for r in lost), "never-read-count": len(never),
           "never-read": sorted(r["element"] for r in never), "stale-count": len(stale),
           "stale": sorted(r["element"] for r in stale),
           "worst-stale": sorted(stale, key=lambda r: -r["stable-latency"])[:8],
           "duplicated-count": len(dups), "duplicated": dict(sorted(dups.items(), key=lambda kv: repr(kv[0])))}
    if stable:
        out["stable-latencies"] = _quantiles([r["stable-latency"] for r in stable])
    if lost:
        out["lost-latencies"] = _quantiles([r["lost-latency"] for r in lost])
    return out
