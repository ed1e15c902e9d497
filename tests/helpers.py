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


def multi_register_history(n_ops, n_procs, seed, n_keys=4, n_values=4, busy=0.4, info=0.0, corrupt=False):
    """Seeded Jepsen-shaped multi-register history (:f :txn, value [[f k v] ...]) from a simulated
    atomic store: each txn takes effect at a random instant inside its interval (valid by
    construction unless `corrupt` plants an impossible read)."""
    import heapq
    import random
    rng = random.Random(seed)
    state = {}
    heap, seq, hist = [], 0, []
    think = (1.0 - busy) / busy
    pid = list(range(n_procs))
    next_pid = n_procs
    for w in range(n_procs):
        heapq.heappush(heap, (rng.expovariate(1.0 / (think + 0.05)), seq, w, "inv")); seq += 1
    cur = {}
    issued = 0
    while heap:
        t, _, w, kind = heapq.heappop(heap)
        if kind == "inv":
            if issued >= n_ops:
                continue
            issued += 1
            mops = []
            for _ in range(rng.randint(1, 3)):
                k = "k%d" % rng.randrange(n_keys)
                mops.append(["r", k, None] if rng.random() < 0.5 else ["w", k, rng.randrange(n_values)])
            crashed = rng.random() < info
            cur[w] = {"mops": mops, "crashed": crashed, "effect": (rng.random() < 0.5) if crashed else True, "res": None}
            hist.append({"type": "invoke", "f": "txn", "value": [list(m) for m in mops], "process": pid[w]})
            L = 0.02 + rng.expovariate(1.0)
            heapq.heappush(heap, (t + rng.random() * L, seq, w, "eff")); seq += 1
            heapq.heappush(heap, (t + L, seq, w, "ret")); seq += 1
        elif kind == "eff":
            c = cur[w]
            if c["effect"]:
                res = []
                for f, k, v in c["mops"]:
                    if f == "r":
                        res.append(["r", k, state.get(k)])
                    else:
                        state[k] = v
                        res.append(["w", k, v])
                c["res"] = res
        else:
            c = cur[w]
            if c["crashed"]:
                hist.append({"type": "info", "f": "txn", "value": [list(m) for m in c["mops"]], "process": pid[w]})
                pid[w] = next_pid
                next_pid += 1
            else:
                hist.append({"type": "ok", "f": "txn", "value": c["res"], "process": pid[w]})
            heapq.heappush(heap, (t + rng.expovariate(1.0 / think) + 1e-9, seq, w, "inv")); seq += 1
    if corrupt:
        oks = [i for i, o in enumerate(hist) if o["type"] == "ok" and any(m[0] == "r" for m in o["value"])]
        if oks:
            i = oks[len(oks) * 2 // 3]
            for m in hist[i]["value"]:
                if m[0] == "r":
                    m[2] = n_values + 3     # a value nobody writes
                    break
    return hist


def _simulate(n_ops, n_procs, seed, busy, info, make_op, apply_op):
    """Generic discrete-event simulation of one atomic object (see multi_register_history)."""
    import heapq
    import random
    rng = random.Random(seed)
    heap, seq, hist = [], 0, []
    think = (1.0 - busy) / busy
    pid = list(range(n_procs))
    next_pid = n_procs
    for w in range(n_procs):
        heapq.heappush(heap, (rng.expovariate(1.0 / (think + 0.05)), seq, w, "inv")); seq += 1
    cur, issued = {}, 0
    while heap:
        t, _, w, kind = heapq.heappop(heap)
        if kind == "inv":
            if issued >= n_ops:
                continue
            issued += 1
            f, value = make_op(rng, issued)
            crashed = rng.random() < (info(issued) if callable(info) else info)
            cur[w] = {"f": f, "value": value, "crashed": crashed, "effect": (rng.random() < 0.5) if crashed else True, "res": value}
            hist.append({"type": "invoke", "f": f, "value": value, "process": pid[w]})
            L = 0.02 + rng.expovariate(1.0)
            heapq.heappush(heap, (t + rng.random() * L, seq, w, "eff")); seq += 1
            heapq.heappush(heap, (t + L, seq, w, "ret")); seq += 1
        elif kind == "eff":
            c = cur[w]
            if c["effect"]:
                c["res"] = apply_op(c["f"], c["value"])
        else:
            c = cur[w]
            if c["crashed"]:
                hist.append({"type": "info", "f": c["f"], "value": c["value"], "process": pid[w], "error": "timeout"})
                pid[w] = next_pid
                next_pid += 1
            else:
                hist.append({"type": "ok", "f": c["f"], "value": c["res"], "process": pid[w]})
            heapq.heappush(heap, (t + rng.expovariate(1.0 / think) + 1e-9, seq, w, "inv")); seq += 1
    return hist


def partition_windows(windows, rate):
    """:info probability as a function of the op number: `rate` inside the given [lo, hi) windows (a partition:
    the reference's clients time out and report :info, set_full.clj:107-110), 0 outside."""
    return lambda issued: rate if any(lo <= issued < hi for lo, hi in windows) else 0.0


def set_history(n_ops, n_procs, seed, busy=0.3, info=0.0, corrupt=None):
    """Grow-only set, the reference's set-full shapes (set_full.clj:29-31,42-45,113-116,128-134):
    :add of globally increasing ids (from 9: set_full.clj:159), :read returns the whole sorted set;
    timeouts are :info (`info` = a rate, or a function of the op number: partition_windows).
    corrupt = "lost" drops an element from a late read, "phantom" adds one."""
    state = set()
    nxt = [9]

    def make_op(rng, _):
        if rng.random() < 0.5:
            nxt[0] += 1
            return "add", nxt[0] - 1
        return "read", None

    def apply_op(f, v):
        if f == "add":
            state.add(v)
            return v
        return sorted(state)

    hist = _simulate(n_ops, n_procs, seed, busy, info, make_op, apply_op)
    if corrupt:
        reads = [o for o in hist if o["type"] == "ok" and o["f"] == "read" and o["value"]]
        if reads:
            o = reads[len(reads) * 2 // 3]
            o["value"] = o["value"][:-1] if corrupt == "lost" else o["value"] + [10 ** 6]
    return hist


def bank_history(n_ops, n_procs, seed, accounts=range(1, 9), busy=0.3, info=0.0, corrupt=False):
    """Bank ops as the reference's ledger->bank mapping produces them (tests/ledger.clj:89-114;
    README.md:41-50): :transfer {:debit-acct :credit-acct :amount 1..5}, :read -> {acct balance}."""
    accounts = list(accounts)
    bal = {a: 0 for a in accounts}

    def make_op(rng, _):
        if rng.random() < 0.5:
            d, c = rng.sample(accounts, 2)
            return "transfer", {"debit-acct": d, "credit-acct": c, "amount": rng.randint(1, 5)}
        return "read", None

    def apply_op(f, v):
        if f == "transfer":
            bal[v["debit-acct"]] -= v["amount"]
            bal[v["credit-acct"]] += v["amount"]
            return v
        return dict(bal)

    hist = _simulate(n_ops, n_procs, seed, busy, info, make_op, apply_op)
    if corrupt:
        reads = [o for o in hist if o["type"] == "ok" and o["f"] == "read" and o["value"]]
        if reads:
            o = reads[len(reads) * 2 // 3]
            o["value"] = dict(o["value"])
            o["value"][accounts[0]] += 1          # the sum is off by one: no interleaving explains it
    return hist
