"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle_model.h).

ctypes front-end of liboracle.so (wgl_ref.c = published DLL/bitset form,
wgl_window.c = windowed-key form).  `build()` compiles it with gcc.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

NIL = -(2 ** 31)
CRASHED = 0xFFFFFFFF


class OracleModel(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("init", C.c_int32),
                ("table", C.POINTER(C.c_uint16)),
                ("n_states", C.c_uint32), ("n_classes", C.c_uint32), ("pool", C.POINTER(C.c_int32))]


class OracleResult(C.Structure):
    _fields_ = [("valid", C.c_int32), ("fail_op", C.c_uint32), ("prev_ok_op", C.c_uint32),
                ("final_state", C.c_int32), ("n_witness", C.c_uint32),
                ("steps", C.c_uint64), ("visited", C.c_uint64), ("probes", C.c_uint64),
                ("backtracks", C.c_uint64), ("max_depth", C.c_uint64)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("wgl_ref.c", "wgl_window.c", "linear_ref.c", "wgl_beam.c", "wgl_count.c", "sweep_ref.c", "many.c", "oracle_model.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for fn in (_LIB.wgl_ref_check, _LIB.wgl_window_check, _LIB.linear_ref_check, _LIB.wgl_beam_check, _LIB.wgl_beam_check_rp, _LIB.sweep_ref_check, _LIB.wgl_count_check):
            fn.restype = C.c_int
    return _LIB


def _model(model):
    m = OracleModel()
    m.kind = model["kind"]
    m.init = model.get("init", 0)
    keep = None
    if model.get("table") is not None:
        t = np.ascontiguousarray(model["table"], dtype=np.uint16)
        m.n_states, m.n_classes = t.shape
        m.table = t.ctypes.data_as(C.POINTER(C.c_uint16))
        keep = t
    if model.get("n_accounts"):
        m.n_states = int(model["n_accounts"])
    if model.get("pool") is not None:
        p = np.ascontiguousarray(model["pool"], dtype=np.int32)
        m.pool = p.ctypes.data_as(C.POINTER(C.c_int32))
        keep = (keep, p)
    return m, keep


def _p(arr, ct):
    return arr.ctypes.data_as(C.POINTER(ct))


def check(ops, model, algorithm="window", max_steps=0, want_witness=True):
    """ops: dict of numpy columns f,a,b,process,inv_pos,ret_pos (+ n_process).
    Returns dict(valid, fail_op, prev_ok_op, final_state, witness, counters...)."""
    n = len(ops["f"])
    f = np.ascontiguousarray(ops["f"], np.uint8)
    a = np.ascontiguousarray(ops["a"], np.int32)
    b = np.ascontiguousarray(ops["b"], np.int32)
    inv = np.ascontiguousarray(ops["inv_pos"], np.uint32)
    ret = np.ascontiguousarray(ops["ret_pos"], np.uint32)
    m, keep = _model(model)
    res = OracleResult()
    wit = np.zeros(max(n, 1), np.uint32)
    if algorithm == "ref":
        rc = lib().wgl_ref_check(C.c_uint32(n), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32),
                                 _p(inv, C.c_uint32), _p(ret, C.c_uint32), C.byref(m),
                                 C.c_uint64(max_steps), _p(wit, C.c_uint32), C.byref(res))
    else:
        proc = np.ascontiguousarray(ops["process"], np.int32)
        rc = lib().wgl_window_check(C.c_uint32(n), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32),
                                    _p(proc, C.c_int32), C.c_uint32(int(ops["n_process"])),
                                    _p(inv, C.c_uint32), _p(ret, C.c_uint32), C.byref(m),
                                    C.c_uint64(max_steps), _p(wit, C.c_uint32), C.byref(res))
    if rc != 0:
        raise ValueError(f"oracle rejected history (rc={rc})")
    out = {k: getattr(res, k) for k, _ in OracleResult._fields_}
    out["witness"] = wit[:res.n_witness].copy() if (res.valid == 1 and want_witness) else None
    return out


class LinearStats(C.Structure):
    _fields_ = [("configs_total", C.c_uint64), ("max_configs", C.c_uint64), ("probes", C.c_uint64),
                ("expanded", C.c_uint64), ("levels", C.c_uint64)]


def check_linear(ops, model, max_configs=0, max_final=4096):
    """knossos.linear restatement (linear_ref.c).  Returns verdict, failing op, the sorted final
    config set as an array of rows [state, mask_word0, ...] and the sweep statistics."""
    n = len(ops["f"])
    f = np.ascontiguousarray(ops["f"], np.uint8)
    a = np.ascontiguousarray(ops["a"], np.int32)
    b = np.ascontiguousarray(ops["b"], np.int32)
    inv = np.ascontiguousarray(ops["inv_pos"], np.uint32)
    ret = np.ascontiguousarray(ops["ret_pos"], np.uint32)
    proc = np.ascontiguousarray(ops["process"], np.int32)
    W = max(1, int(ops["n_process"]))
    kw = 1 + (W + 63) // 64
    m, keep = _model(model)
    res, st = OracleResult(), LinearStats()
    fin = np.zeros((max_final, kw), np.uint64)
    nf = C.c_uint32(0)
    rc = lib().linear_ref_check(C.c_uint32(n), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32),
                                _p(proc, C.c_int32), C.c_uint32(int(ops["n_process"])),
                                _p(inv, C.c_uint32), _p(ret, C.c_uint32), C.byref(m), C.c_uint64(max_configs),
                                _p(fin, C.c_uint64), C.c_uint32(max_final), C.byref(nf), C.byref(res), C.byref(st))
    if rc != 0:
        raise ValueError(f"oracle rejected history (rc={rc})")
    out = {"valid": res.valid, "fail_op": res.fail_op, "prev_ok_op": res.prev_ok_op, "final_state": res.final_state,
           "n_configs": nf.value}
    k = min(nf.value, max_final)
    cfg = fin[:k].copy()
    cfg[:, 0] -= 1          # state word is stored +1
    out["configs"] = cfg
    for name, _ in LinearStats._fields_:
        out[name] = getattr(st, name)
    return out


class BeamStats(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("probes", C.c_uint64), ("visited", C.c_uint64),
                ("expanded", C.c_uint64), ("max_stack", C.c_uint64), ("rounds", C.c_uint64)]


def rules_apply(ops, model, round_pairs=64):
    """Where the library applies the dominance rules (eager reads, twin rule) by default: register /
    cas-register under the single-wavefront wide schedule, every register value in 0..30."""
    if round_pairs != 64 or model["kind"] not in (0, 1):
        return False
    init = model.get("init", NIL)
    vals = [np.asarray(ops["a"], np.int64)]
    f = np.asarray(ops["f"])
    vals.append(np.asarray(ops["b"], np.int64)[f == 2])
    v = np.concatenate(vals)
    v = v[v != NIL]
    if init != NIL:
        v = np.append(v, init)
    return bool(len(v) == 0 or (v.min() >= 0 and v.max() <= 30))


class _ListOrders(dict):       # csrc PackOpenArgs.list_order (TBC_NARROW_ORDER) -> wgl_beam_set_list_order; 16 + W is 16 + W on both sides
    def __missing__(self, k):
        if k >= 16:
            return k
        raise KeyError(k)


ORACLE_LIST_ORDER = _ListOrders({0: 0, 1: 1, 2: 4})


def check_beam(ops, model, width=16, max_probes=0, want_witness=True, round_pairs=64, widen_after=0, lookahead=None, eager_reads=None, twin_rule=None,
               twin_selfcheck=False, rules_at_any_round_size=False, branch_lists=False, lazy_commuting=None, eager_txns=None, txn_independence=None, look_two=False, list_order=0, lazy_look=False, defer=False):
    """The wide (K configs per iteration) schedule of the same search: wgl_beam.c.

    lookahead: None = what the library does by default (on for register / cas-register under the
    single-wavefront wide schedule, i.e. round_pairs == 64).  Configs it finds dead are set aside and
    expanded only if the search would otherwise end INVALID, so results are exact either way."""
    if lookahead is None and not rules_at_any_round_size:
        lookahead = round_pairs == 64 and model["kind"] in (0, 1)
    # eager_reads / twin_rule: None = what the library does by default (tbc_opts.dominance = 0)
    # rules_at_any_round_size: the library applies the dominance rules only under its 64-pair rounds; the schedules with
    # narrower rounds (several histories per wavefront) are specified with them all the same
    if rules_at_any_round_size:
        ok = rules_apply(ops, model, 64)
        eager_reads = ok if eager_reads is None else (eager_reads and ok)
        twin_rule = ok if twin_rule is None else (twin_rule and ok)
        if lookahead is None:
            lookahead = model["kind"] in (0, 1)
    elif eager_reads is None or twin_rule is None:
        dflt = rules_apply(ops, model, round_pairs)
        eager_reads = dflt if eager_reads is None else eager_reads
        twin_rule = dflt if twin_rule is None else twin_rule
    elif (eager_reads or twin_rule) and not rules_apply(ops, model, round_pairs):
        eager_reads = twin_rule = False
    lib().wgl_beam_set_eager_reads(C.c_uint32(1 if eager_reads else 0))
    lib().wgl_beam_set_twin_rule(C.c_uint32(1 if twin_rule else 0))
    # lazy_commuting: the lazy rule of the commutative models (set, bank); None = what the library does by default (on)
    lib().wgl_beam_set_lazy_commuting(C.c_uint32(1 if ((model["kind"] in (5, 6)) if lazy_commuting is None else lazy_commuting) else 0))
    # twin_selfcheck: evaluate every twin test of a crashed candidate also by the previous-crashed-twin shortcut
    # (wgl_beam.c) and report how many were compared / disagreed as twin_checked / twin_mismatch
    # branch_lists: candidate lists hold the live :write / :cas calls only and the root starts in normal form (the narrow
    # kernel's lists under the eager rule); without eager reads the switch does nothing
    lib().wgl_beam_set_branch_lists(C.c_uint32(1 if (branch_lists and eager_reads) else 0))
    lib().wgl_beam_set_twin_selfcheck(C.c_uint32(1 if twin_selfcheck else 0))
    # eager_txns / txn_independence (multi-register; csrc kRuleTxnEager / kRuleTxnIndep, tbcheck.h TBC_DOM_NO_EAGER_TXNS / _NO_TXN_INDEPENDENCE):
    # an open txn of micro-reads only that the state allows is linearized at once; the candidates of a config are the closure of the call
    # completing at its front under "conflicts with" (wgl_beam.c).  None = what the library does by default under its wide schedule: on
    eager_txns = (width > 1) if eager_txns is None else eager_txns
    txn_independence = (width > 1) if txn_independence is None else txn_independence
    lib().wgl_beam_set_eager_txns(C.c_uint32(1 if (eager_txns and model["kind"] == 4) else 0))
    lib().wgl_beam_set_txn_independence(C.c_uint32(1 if (txn_independence and model["kind"] == 4) else 0))
    # look_two: the lean lookahead record's reading of three or more open producers (wgl_beam.c, g_look_two; csrc kLeanLook)
    lib().wgl_beam_set_look_two(C.c_uint32(1 if look_two else 0))
    # list_order: 0 = a front's open calls in process-slot order, 1 = in order of completion (csrc PackOpenArgs.list_order); study knobs of
    # wgl_beam.c beyond those: 2 = in order of invocation, 3 = latest completion first, 4 = in order of completion with the :write calls
    # last (PackOpenArgs.list_order = 2: ORACLE_LIST_ORDER maps the library's numbers to these), 5 = ... with the :cas calls last, 6 / 7 = writes last by latest completion / by invocation (both worse), 16 + W = in order of completion with a
    # :write as if it completed W ranks later (PackOpenArgs.list_order = 16 + W)
    lib().wgl_beam_set_list_order(C.c_uint32(list_order))
    # lazy_look: DESIGN STUDY (no kernel counterpart): the lookahead at once only for the config that will be popped next (wgl_beam.c)
    lib().wgl_beam_set_lazy_look(C.c_uint32(1 if lazy_look else 0))
    # defer: DESIGN STUDY: one child at a time -- a round's other viable pairs wait on the stack as (parent, pair) markers (wgl_beam.c)
    lib().wgl_beam_set_defer(C.c_uint32(1 if defer else 0))
    try:
        r = _check_beam(ops, model, width, max_probes, want_witness, round_pairs, widen_after, bool(lookahead))
        if twin_selfcheck:
            lib().wgl_beam_twin_checked.restype = C.c_uint64
            lib().wgl_beam_twin_mismatch.restype = C.c_uint64
            r["twin_checked"], r["twin_mismatch"] = int(lib().wgl_beam_twin_checked()), int(lib().wgl_beam_twin_mismatch())
        return r
    finally:
        lib().wgl_beam_set_eager_reads(C.c_uint32(0))
        lib().wgl_beam_set_twin_rule(C.c_uint32(0))
        lib().wgl_beam_set_twin_selfcheck(C.c_uint32(0))
        lib().wgl_beam_set_branch_lists(C.c_uint32(0))
        lib().wgl_beam_set_lazy_commuting(C.c_uint32(0))
        lib().wgl_beam_set_eager_txns(C.c_uint32(0))
        lib().wgl_beam_set_txn_independence(C.c_uint32(0))
        lib().wgl_beam_set_look_two(C.c_uint32(0))
        lib().wgl_beam_set_list_order(C.c_uint32(0))
        lib().wgl_beam_set_lazy_look(C.c_uint32(0))
        lib().wgl_beam_set_defer(C.c_uint32(0))


def _check_beam(ops, model, width, max_probes, want_witness, round_pairs, widen_after, lookahead):
    n = len(ops["f"])
    f = np.ascontiguousarray(ops["f"], np.uint8)
    a = np.ascontiguousarray(ops["a"], np.int32)
    b = np.ascontiguousarray(ops["b"], np.int32)
    inv = np.ascontiguousarray(ops["inv_pos"], np.uint32)
    ret = np.ascontiguousarray(ops["ret_pos"], np.uint32)
    proc = np.ascontiguousarray(ops["process"], np.int32)
    m, keep = _model(model)
    res, st = OracleResult(), BeamStats()
    wit = np.zeros(max(n, 1), np.uint32)
    lib().wgl_beam_set_widen_after(C.c_uint32(widen_after))
    lib().wgl_beam_set_lookahead(C.c_uint32(1 if lookahead else 0))
    rc = lib().wgl_beam_check_rp(C.c_uint32(n), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32),
                                 _p(proc, C.c_int32), C.c_uint32(int(ops["n_process"])), _p(inv, C.c_uint32),
                                 _p(ret, C.c_uint32), C.byref(m), C.c_uint32(width), C.c_uint32(round_pairs),
                                 C.c_uint64(max_probes), _p(wit, C.c_uint32), C.byref(res), C.byref(st))
    if rc != 0:
        raise ValueError(f"oracle rejected history (rc={rc})")
    out = {k: getattr(res, k) for k, _ in OracleResult._fields_}
    out["witness"] = wit[:res.n_witness].copy() if (res.valid == 1 and want_witness) else None
    for name, _ in BeamStats._fields_:
        out[name] = getattr(st, name)
    return out


def last_configs(which="window", max_rows=4096):
    """(total, rows) of the configs stuck at the failing completion in the LAST invalid run of
    check(..., "window") / check_beam: rows = [state, mask_word0, ...] sorted by (state, mask)."""
    fn = lib().wgl_window_last_configs if which == "window" else lib().wgl_beam_last_configs
    fn.restype = C.c_uint32
    buf = np.zeros((max_rows, 17), np.uint64)
    flat = np.zeros(max_rows * 17, np.uint64)
    kw = C.c_uint32(0)
    total = fn(_p(flat, C.c_uint64), C.c_uint32(max_rows), C.byref(kw))
    k = kw.value
    rows = flat[: min(total, max_rows) * k].reshape(-1, k).copy() if k else np.zeros((0, 1), np.uint64)
    out = []
    for r in rows:
        st = int(r[0] >> np.uint64(32))
        if st >= 2 ** 31:
            st -= 2 ** 32
        out.append([st] + [int(x) for x in r[1:]])
    del buf
    return total, out


class SweepStats(C.Structure):
    _fields_ = [("configs_total", C.c_uint64), ("max_level", C.c_uint64), ("probes", C.c_uint64),
                ("subrounds", C.c_uint64), ("levels", C.c_uint64), ("n_segments", C.c_uint64),
                ("max_origins", C.c_uint64), ("max_pending", C.c_uint64), ("longest_segment", C.c_uint64),
                ("n_waves", C.c_uint64), ("max_segment_probes", C.c_uint64),
                ("critical_steps", C.c_uint64), ("critical_levels", C.c_uint64), ("total_steps", C.c_uint64)]


def check_sweep(ops, model, eager_reads=True, twin_rule=True, seg_target=0, max_cut_open=4, max_level=0, want_levels=False, n_dom=0, relaxed=False):
    """The segmented level sweep (sweep_ref.c): knossos.linear's just-in-time linearization with the
    dominance rules, cut into independently swept segments that are composed afterwards."""
    n = len(ops["f"])
    f = np.ascontiguousarray(ops["f"], np.uint8)
    a = np.ascontiguousarray(ops["a"], np.int32)
    b = np.ascontiguousarray(ops["b"], np.int32)
    inv = np.ascontiguousarray(ops["inv_pos"], np.uint32)
    ret = np.ascontiguousarray(ops["ret_pos"], np.uint32)
    proc = np.ascontiguousarray(ops["process"], np.int32)
    m, keep = _model(model)
    res, st = OracleResult(), SweepStats()
    lv = np.zeros(max(n, 1), np.uint32)
    L = lib()
    L.sweep_set_rules(C.c_uint32(1 if eager_reads else 0), C.c_uint32(1 if twin_rule else 0))
    L.sweep_set_segments(C.c_uint32(seg_target), C.c_uint32(max_cut_open))
    L.sweep_set_domain(C.c_uint32(n_dom))
    # relaxed: crashed calls as classes of effects in unlimited supply (sweep_ref.c, sweep_set_relaxed): INVALID proves invalid at or before
    # the completion it names, VALID proves nothing -- the refutation pass of the count form as a level sweep
    L.sweep_set_relaxed(C.c_uint32(1 if relaxed else 0))
    try:
        rc = _sweep_call(L, n, f, a, b, proc, ops, inv, ret, m, max_level, lv, want_levels, res, st)
    finally:
        L.sweep_set_relaxed(C.c_uint32(0))
    return _sweep_out(rc, res, st, lv, want_levels)


def _sweep_call(L, n, f, a, b, proc, ops, inv, ret, m, max_level, lv, want_levels, res, st):
    return L.sweep_ref_check(C.c_uint32(n), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32),
                           _p(proc, C.c_int32), C.c_uint32(int(ops["n_process"])), _p(inv, C.c_uint32),
                           _p(ret, C.c_uint32), C.byref(m), C.c_uint64(max_level),
                           _p(lv, C.c_uint32) if want_levels else None, C.byref(res), C.byref(st))


def _sweep_out(rc, res, st, lv, want_levels):
    if rc != 0:
        raise ValueError(f"oracle rejected history (rc={rc})")
    out = {k: getattr(res, k) for k, _ in OracleResult._fields_}
    for name, _ in SweepStats._fields_:
        out[name] = getattr(st, name)
    if want_levels:
        out["level_sizes"] = lv
    return out


def check_many(ops_list, model, n_threads, max_steps=0, beam_width=0, round_pairs=64, list_order=0, branch_lists=False):
    """Many histories on a pthread pool (many.c): verdicts as an int32 array.  beam_width = 0: wgl_window_check (the
    sequential knossos.wgl restatement); > 0: wgl_beam.c at that many configs per round with the library's default
    rules (lookahead, eager reads, twin rule where they apply to the FIRST history's model and values); round_pairs = 8 / 16 / 32 with
    beam_width 1 is the narrow kernel's schedule (several histories per wavefront)."""
    nh = len(ops_list)
    keep = []

    def col(name, dt, ct):
        arrs = [np.ascontiguousarray(o[name], dt) for o in ops_list]
        keep.append(arrs)
        return (C.POINTER(ct) * nh)(*[_p(x, ct) for x in arrs])

    n = np.array([len(o["f"]) for o in ops_list], np.uint32)
    npr = np.array([int(o["n_process"]) for o in ops_list], np.uint32)
    f, a, b = col("f", np.uint8, C.c_uint8), col("a", np.int32, C.c_int32), col("b", np.int32, C.c_int32)
    pr = col("process", np.int32, C.c_int32)
    inv, ret = col("inv_pos", np.uint32, C.c_uint32), col("ret_pos", np.uint32, C.c_uint32)
    m, keep_m = _model(model)
    valid = np.zeros(nh, np.int32)
    fn = lib().wgl_check_many
    fn.restype = C.c_int
    rules = bool(beam_width) and nh > 0 and rules_apply(ops_list[0], model, 64)
    if beam_width:
        lib().wgl_beam_set_widen_after(C.c_uint32(0))
        lib().wgl_beam_set_lookahead(C.c_uint32(1 if model["kind"] in (0, 1) else 0))
        lib().wgl_beam_set_eager_reads(C.c_uint32(1 if rules else 0))
        lib().wgl_beam_set_twin_rule(C.c_uint32(1 if rules else 0))
        lib().wgl_beam_set_list_order(C.c_uint32(list_order))          # (as check_beam's: the order of a front's list of open calls)
        lib().wgl_beam_set_branch_lists(C.c_uint32(1 if (branch_lists and rules) else 0))
    try:
        started = fn(C.c_uint32(nh), _p(n, C.c_uint32), _p(npr, C.c_uint32), f, a, b, pr, inv, ret, C.byref(m),
                     C.c_uint64(max_steps), C.c_uint32(n_threads), C.c_uint32(beam_width), C.c_uint32(round_pairs), _p(valid, C.c_int32))
    finally:
        if beam_width:
            lib().wgl_beam_set_eager_reads(C.c_uint32(0))
            lib().wgl_beam_set_twin_rule(C.c_uint32(0))
            lib().wgl_beam_set_list_order(C.c_uint32(0))
            lib().wgl_beam_set_branch_lists(C.c_uint32(0))
    del keep, keep_m
    return valid, started


def sweep_relations(ops, model, seg_target, n_dom, max_segs, rank=0, world=1, rel_bytes=688):
    """What rank `rank` of `world` GPUs leaves in its relation table after tbc_batch_sweep_partial, computed by
    the CPU restatement: max_segs * 4 tbc_sweep_rel records as a uint8 array (records of other ranks all zero)."""
    buf = np.zeros(max_segs * 4 * rel_bytes, np.uint8)
    L = lib()
    L.sweep_set_export(buf.ctypes.data_as(C.c_void_p), C.c_uint32(max_segs), C.c_uint32(rank), C.c_uint32(world))
    try:
        check_sweep(ops, model, seg_target=seg_target, n_dom=n_dom)
    finally:
        L.sweep_set_export(None, C.c_uint32(0), C.c_uint32(0), C.c_uint32(1))
    return buf


class CountStats(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("probes", C.c_uint64), ("visited", C.c_uint64),
                ("expanded", C.c_uint64), ("max_stack", C.c_uint64), ("rounds", C.c_uint64),
                ("dominated", C.c_uint64), ("hot", C.c_uint64), ("class_steps", C.c_uint64),
                ("n_slots", C.c_uint32), ("n_classes", C.c_uint32), ("count_bits", C.c_uint32), ("n_crashed", C.c_uint32)]


def check_count(ops, model, width=4, max_probes=0, want_witness=True, round_pairs=64, lookahead=True, want_slots=False, relaxed=False, target=0):
    """The COUNT FORM of the wide schedule (wgl_count.c): crashed calls as a count per effect class, process slots
    re-used, the lazy ("hot") rule and the Pareto rule.  Register / cas-register with values 0..30 only; returns None
    when the form does not apply (other models, values out of range, more than 128 bits of counts)."""
    n = len(ops["f"])
    f = np.ascontiguousarray(ops["f"], np.uint8)
    a = np.ascontiguousarray(ops["a"], np.int32)
    b = np.ascontiguousarray(ops["b"], np.int32)
    inv = np.ascontiguousarray(ops["inv_pos"], np.uint32)
    ret = np.ascontiguousarray(ops["ret_pos"], np.uint32)
    proc = np.ascontiguousarray(ops["process"], np.int32)
    m, keep = _model(model)
    res, st = OracleResult(), CountStats()
    wit = np.zeros(max(n, 1), np.uint32)
    slots = np.zeros(max(n, 1), np.uint32)
    L = lib()
    L.wgl_count_set_slots_out(_p(slots, C.c_uint32) if want_slots else None)
    L.wgl_count_set_limit_per_round(C.c_uint32(1 if round_pairs != 64 else 0))      # (the narrow kernel looks at its step limit after every round)
    try:
        rc = L.wgl_count_check(C.c_uint32(n), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32),
                               _p(proc, C.c_int32), C.c_uint32(int(ops["n_process"])), _p(inv, C.c_uint32),
                               _p(ret, C.c_uint32), C.byref(m), C.c_uint32(width), C.c_uint32(round_pairs),
                               C.c_uint64(max_probes), C.c_uint32(1 if lookahead else 0), C.c_uint32(1 if relaxed else 0), C.c_uint32(target),
                               _p(wit, C.c_uint32) if want_witness else None, C.byref(res), C.byref(st))
    finally:
        L.wgl_count_set_slots_out(None)
    if rc == 3:
        return None
    if rc != 0:
        raise ValueError(f"oracle rejected history (rc={rc})")
    out = {k: getattr(res, k) for k, _ in OracleResult._fields_}
    out["witness"] = wit[:res.n_witness].copy() if (res.valid == 1 and want_witness) else None
    for name, _ in CountStats._fields_:
        out[name] = getattr(st, name)
    if want_slots:
        out["slots"] = slots[:n].copy()
    return out


def completion_rank(ops, op):
    """rank of op's completion among the completions (crashed calls hold 0xFFFFFFFF: never below a completion)"""
    ret = np.asarray(ops["ret_pos"]).astype(np.int64)
    return int((ret < ret[op]).sum())


# the list orders the library's ORDER RESTARTS go through after the default one, as PackOpenArgs.list_order numbers (csrc/batch_run.hip kRestartOrders)
RESTART_ORDERS = (16 + 48, 2, 1, 16 + 8, 0)
DEFAULT_ORDER = 16 + 24


def check_restart_pipeline(ops, model, width=4, budget=None, want_witness=False):
    """What the library does with a hard history of the wide depth-first search (csrc/batch_run.hip, order_restarts), pass by pass over
    wgl_beam.c: the search in the library's default list order (16 + 24) under a budget of probes (the library: 32 per op of the batch's
    longest history); a history that has not ended by then is searched AGAIN FROM SCRATCH in the next list order under the same budget
    -- at high concurrency the cost of a depth-first search is heavy-tailed in the order its candidates are tried and nearly independent
    between orders, so the first order that is lucky ends it -- and when no order ended it, in the default order without a budget.
    Returns (valid, fail_op, result of the last pass, counters summed over the passes, which pass answered)."""
    n = len(ops["f"])
    budget = 32 * n if budget is None else budget
    tot = {"probes": 0, "visited": 0, "expanded": 0, "max_stack": 0}

    def add(r):
        for k in ("probes", "visited", "expanded"):
            tot[k] += r[k]
        tot["max_stack"] = max(tot["max_stack"], r["max_stack"])
        return r
    seq = [(DEFAULT_ORDER, budget)]
    for o in RESTART_ORDERS:          # PackOpenArgs numbers -> the oracle's (0 slot, 1 completion, 2 writes last = the oracle's 4)
        seq.append(((4 if o == 2 else o), budget))
    seq.append((DEFAULT_ORDER, 0))
    for k, (order, cap) in enumerate(seq):
        r = add(check_beam(ops, model, width, want_witness=want_witness, list_order=order, max_probes=cap))
        if r["valid"] != -1:
            return r["valid"], r["fail_op"], r, tot, f"pass {k} (order {order}, budget {cap})"
    return -1, r["fail_op"], r, tot, "no pass ended"


def check_count_pipeline(ops, model, width=4, budget=None, want_witness=False, round_pairs=64, relaxed_sweep=False):
    """What the library does with a history in count form (tbc_api.hip, batch_run_impl), pass by pass over wgl_count.c: the exact
    search under a budget of probes (the library: 32 per op of the batch's longest history); past it the RELAXED search (every
    class an unlimited supply: a superset of the linearizations), whose INVALID verdict bounds the failing completion from above;
    then the exact search of the PREFIX before that completion -- a linearization of it pins the failing op without exhausting
    the exact config space.  Returns (valid, fail_op, result of the last pass, counters summed over the passes, which passes ran),
    or None where the count form does not apply."""
    n = len(ops["f"])
    budget = 32 * n if budget is None else budget
    tot = {"probes": 0, "visited": 0, "expanded": 0}

    def add(r):
        for k in tot:
            tot[k] += r[k]
        return r
    # relaxed_sweep (the library for a handful of histories, nobody asking for a witness or a schedule: tbc_batch::rsweep): the RELAXED
    # LEVEL SWEEP in front -- a refuted history goes straight to the prefix search (it is never searched exactly under a budget, nor
    # relaxed depth-first), one it finds valid under the relaxation skips the relaxed depth-first search; the sweep's own probes are not
    # in the sums (the library reports the depth-first passes' counters)
    swept_valid = False
    if relaxed_sweep:
        if check_count(ops, model, width=width, want_witness=False, max_probes=1, round_pairs=round_pairs) is None:
            return None
        sw = check_sweep(ops, model, relaxed=True, seg_target=32)
        if sw["valid"] == 0:
            t = completion_rank(ops, sw["fail_op"])
            if t == 0:
                return 0, sw["fail_op"], sw, tot, "relaxed sweep"
            g = add(check_count(ops, model, width=width, want_witness=False, target=t, round_pairs=round_pairs))
            if g["valid"] == 1:
                return 0, sw["fail_op"], g, tot, "relaxed sweep, prefix"
            return g["valid"], g["fail_op"], g, tot, "relaxed sweep, prefix exhausted"
        swept_valid = sw["valid"] == 1
    g = check_count(ops, model, width=width, want_witness=want_witness, max_probes=budget, round_pairs=round_pairs)
    if g is None:
        return None
    add(g)
    if g["valid"] != -1:
        return g["valid"], g["fail_op"], g, tot, "exact"
    if swept_valid:
        g = add(check_count(ops, model, width=width, want_witness=want_witness, round_pairs=round_pairs))
        return g["valid"], g["fail_op"], g, tot, "exact, no budget"
    r = add(check_count(ops, model, width=width, want_witness=False, relaxed=True, round_pairs=round_pairs))
    if r["valid"] == 1:
        g = add(check_count(ops, model, width=width, want_witness=want_witness, round_pairs=round_pairs))
        return g["valid"], g["fail_op"], g, tot, "exact, no budget"
    t = completion_rank(ops, r["fail_op"])
    if t == 0:
        return 0, r["fail_op"], r, tot, "relaxed"
    g = add(check_count(ops, model, width=width, want_witness=False, target=t, round_pairs=round_pairs))
    if g["valid"] == 1:
        return 0, r["fail_op"], g, tot, "prefix"
    return g["valid"], g["fail_op"], g, tot, "prefix exhausted"
