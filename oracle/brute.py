"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.

brute.py -- linearizability straight from the DEFINITION (Herlihy & Wing 1990),
for tiny histories only.  It shares no code and no search strategy with the
Wing-Gong/Lowe restatements (wgl_ref.c, wgl_window.c) or the HIP kernel: it is
the independent ground truth for `:valid?` and for "the first completion whose
prefix has no linearization" (what knossos.linear reports as :op -- SURVEY.md
section 8a, `Result map`).  PARITY UNPINNED: see oracle_model.h.

A history (op level: f, a, b, inv_pos, ret_pos; ret_pos = CRASHED for :info) is
linearizable iff there is a subset C of the crashed ops and a total order of
(completed ops + C) that (1) respects real time -- x before y whenever
ret_pos[x] < inv_pos[y] -- and (2) is a legal run of the sequential model.
"""
from __future__ import annotations

from itertools import permutations

NIL = -(2 ** 31)
CRASHED = 0xFFFFFFFF
READ, WRITE, CAS, ACQUIRE, RELEASE, CLASS = 0, 1, 2, 3, 4, 8
REGISTER, CAS_REGISTER, MUTEX, TABLE, MULTI_REGISTER, SET, BANK = 0, 1, 2, 3, 4, 5, 6
ADD, TXN, TRANSFER = 5, 6, 7


def step(model, state, f, a, b):
    """knossos.model/step; returns next state or None (inconsistent)."""
    kind = model["kind"]
    if kind in (REGISTER, CAS_REGISTER):
        if f == WRITE:
            return a
        if f == READ:
            return state if (a == NIL or a == state) else None
        if f == CAS and kind == CAS_REGISTER:
            return b if a == state else None
        return None
    if kind == MUTEX:
        if f == ACQUIRE:
            return 1 if state == 0 else None
        if f == RELEASE:
            return 0 if state == 1 else None
        return None
    if kind == MULTI_REGISTER:
        if f != TXN:
            return None
        s, pool, ok = state, model["pool"], True
        for i in range(b):
            mf, k, v = (int(x) for x in pool[a + 3 * i: a + 3 * i + 3])
            cur = (s >> (4 * k)) & 15
            if mf == 0:
                ok = ok and (v == NIL or cur == v + 1)
            else:
                s = (s & ~(15 << (4 * k))) | ((v + 1) << (4 * k))
        return s if ok else None
    if kind == SET:      # ordinary stateful semantics (state = set of add indices): independent of the
        pool = model["pool"]   # commutativity trick the kernels use
        state = frozenset() if state == 0 else state
        if f == ADD:
            return state | {a}
        if f == READ:
            if a == NIL:
                return state
            nR, nwords = int(pool[a]), (model["n_adds"] + 31) // 32
            if nR < 0:
                return None
            R = frozenset(j for j in range(model["n_adds"]) if int(pool[a + 2 + (j >> 5)]) >> (j & 31) & 1)
            del nwords
            return state if R == state else None
        return None
    if kind == BANK:
        pool, A = model["pool"], model["n_accounts"]
        state = tuple([0] * A) if state == 0 else state
        if f == TRANSFER:
            d, c, amt = (int(x) for x in pool[a:a + 3])
            s = list(state); s[d] -= amt; s[c] += amt
            return tuple(s)
        if f == READ:
            if a == NIL:
                return state
            return state if tuple(int(x) for x in pool[a:a + A]) == state else None
        return None
    if kind == TABLE:
        t = model["table"][state][a]
        return None if t == 0xFFFF else t
    raise ValueError(kind)


def _legal_order(model, ops, order):
    s = model["init"]
    for i in order:
        s = step(model, s, ops[i][0], ops[i][1], ops[i][2])
        if s is None:
            return False
    return True


def _respects_real_time(ops, order):
    # x must precede y whenever x returned before y was invoked; equivalently no
    # op may come after one that was invoked later than it returned: O(n).
    max_inv = -1
    for x in order:
        if ops[x][4] != CRASHED and ops[x][4] < max_inv:
            return False
        if ops[x][3] > max_inv:
            max_inv = ops[x][3]
    return True


def linearizable(model, ops):
    """ops: list of (f, a, b, inv_pos, ret_pos).  Exhaustive; use for <= ~8 ops."""
    done = [i for i, o in enumerate(ops) if o[4] != CRASHED]
    crashed = [i for i, o in enumerate(ops) if o[4] == CRASHED]
    for mask in range(1 << len(crashed)):
        chosen = done + [c for j, c in enumerate(crashed) if mask >> j & 1]
        for order in permutations(chosen):
            if _respects_real_time(ops, order) and _legal_order(model, ops, order):
                return True
    return False


def first_bad_completion(model, ops):
    """None if linearizable, else the index of the op whose completion is the
    first one at which the history prefix stops being linearizable."""
    rets = sorted((o[4], i) for i, o in enumerate(ops) if o[4] != CRASHED)
    for pos, i in rets:
        prefix = []
        for o in ops:
            if o[3] < pos:   # invoked before this completion
                prefix.append((o[0], o[1], o[2], o[3], o[4] if o[4] <= pos else CRASHED))
        if not linearizable(model, prefix):
            return i
    return None


def check_witness(model, ops, witness):
    """A witness is a legal, real-time-respecting order containing every
    completed op (crashed ops optional).  Returns the final state or raises."""
    done = {i for i, o in enumerate(ops) if o[4] != CRASHED}
    assert len(set(witness)) == len(witness), "duplicate op in witness"
    assert done <= set(witness), "witness misses completed ops"
    assert _respects_real_time(ops, list(witness)), "witness violates real-time order"
    s = model["init"]
    for i in witness:
        s = step(model, s, ops[i][0], ops[i][1], ops[i][2])
        assert s is not None, f"witness step {i} inconsistent"
    return s
