"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  The wavefront emulator: builds tests/emu/_build/libemu_narrow.so with g++ from
emu_narrow.cpp + the product's own kernel body (jepsen-tigerbeetle_amd/csrc/wgl_narrow_impl.h compiled with TBC_EMU) and
runs it through ctypes.  Nothing in the product imports this package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_CSRC = os.path.join(_ROOT, "jepsen-tigerbeetle_amd", "csrc")
# TBC_EMU_ASAN=1: the same library under AddressSanitizer (every table is a std::vector of exactly the device's size, so a read or a
# write of the kernel body past any of them stops the test); run as  LD_PRELOAD=$(gcc -print-file-name=libasan.so)
# ASAN_OPTIONS=detect_leaks=0 TBC_EMU_ASAN=1 python -m pytest tests/test_narrow_emu.py
_ASAN = os.environ.get("TBC_EMU_ASAN") == "1"
# TBC_EMU_DEFS="-DTBC_NARROW_EB=1 ...": the kernel body's build-time forms (wgl_narrow_impl.h) under the emulator, in a library of their own
_DEFS = os.environ.get("TBC_EMU_DEFS", "").split()
_TAG = "".join(c if c.isalnum() else "_" for c in "".join(_DEFS))
_SO = os.path.join(_HERE, "_build", ("libemu_narrow_asan" if _ASAN else "libemu_narrow") + (("_" + _TAG) if _TAG else "") + ".so")
_LIB = None


class DevResult(C.Structure):
    _fields_ = [("valid", C.c_int32), ("cause", C.c_int32), ("max_front", C.c_uint32), ("depth", C.c_uint32),
                ("final_state", C.c_int32), ("n_configs", C.c_uint32), ("fail_op", C.c_uint32), ("prev_ok_op", C.c_uint32),
                ("tab_log2", C.c_uint32), ("pad", C.c_uint32), ("steps", C.c_uint64), ("visited", C.c_uint64),
                ("probes", C.c_uint64), ("backtracks", C.c_uint64), ("max_depth", C.c_uint64), ("bucket_reads", C.c_uint64)]


def build(force=False):
    srcs = [os.path.join(_HERE, "emu_narrow.cpp"), os.path.join(_HERE, "wave_env_emu.h"), os.path.join(_HERE, "host_tables.h"),
            os.path.join(_CSRC, "wgl_narrow_impl.h"), os.path.join(_CSRC, "tbc_internal.h"), os.path.join(_CSRC, "wave_env.h"),
            os.path.join(_CSRC, "witness_expand.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]
                              + (["-fsanitize=address", "-fno-omit-frame-pointer"] if _ASAN else [])
                              + _DEFS + ["-I", _HERE, "-I", _CSRC, "-o", _SO, srcs[0]])
    return _SO


_SO_WALK = os.path.join(_HERE, "_build", "libemu_walk_asan.so" if _ASAN else "libemu_walk.so")
_LIB_WALK = None


def build_walk(force=False):
    srcs = [os.path.join(_HERE, "emu_walk.cpp"), os.path.join(_HERE, "wave_env_emu.h"), os.path.join(_HERE, "host_tables.h"),
            os.path.join(_CSRC, "open_walk_impl.h"), os.path.join(_CSRC, "tbc_internal.h"), os.path.join(_CSRC, "wave_env.h")]
    if force or not os.path.exists(_SO_WALK) or any(os.path.getmtime(s) > os.path.getmtime(_SO_WALK) for s in srcs):
        os.makedirs(os.path.dirname(_SO_WALK), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]
                              + (["-fsanitize=address", "-fno-omit-frame-pointer"] if _ASAN else [])
                              + ["-I", _HERE, "-I", _CSRC, "-o", _SO_WALK, srcs[0]])
    return _SO_WALK


def walk_check(hists, vpad, twin=True, look=True, branch=False, front="plain", by_ret=False):
    """Run the front walk with lane = front (csrc/open_walk_impl.h) under the emulator on these histories (at most 64 process
    slots each) and compare every word it writes with the host-built tables.  front: "plain" rows, "wide" or "compact" front
    records.  Returns None when all agree, else (what, history, front, index, got, want)."""
    global _LIB_WALK
    if _LIB_WALK is None:
        _LIB_WALK = C.CDLL(build_walk())
        _LIB_WALK.emu_walk_check.restype = C.c_int
    ds = [h if isinstance(h, dict) else h.as_dict() for h in hists]
    nh = len(ds)
    op_off = np.zeros(nh + 1, np.uint64)
    for i, d in enumerate(ds):
        op_off[i + 1] = op_off[i] + len(d["f"])
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(d[k], dt) for d in ds]), dt)
    f, a, b = cat("f", np.uint8), cat("a", np.int32), cat("b", np.int32)
    pr, inv, ret = cat("process", np.int32), cat("inv_pos", np.uint32), cat("ret_pos", np.uint32)
    npr = np.array([int(d["n_process"]) for d in ds], np.uint32)
    assert int(npr.max()) <= 64
    flags = (1 if twin else 0) | (2 if look else 0) | (4 if branch else 0) | {"plain": 0, "wide": 8, "compact": 24}[front] | ((int(by_ret) << 16) if int(by_ret) >= 16 else 128 if by_ret == 2 else 64 if by_ret else 0)      # lean: csrc kLeanCands | kLeanLook; by_ret: list_order 1
    diag = np.zeros(8, np.uint64)
    rc = _LIB_WALK.emu_walk_check(C.c_uint32(nh), _p(op_off, C.c_uint64), _p(npr, C.c_uint32), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32),
                                  _p(pr, C.c_int32), _p(inv, C.c_uint32), _p(ret, C.c_uint32), C.c_uint32(vpad), C.c_uint32(flags), _p(diag, C.c_uint64))
    if rc == 0:
        return None
    what = {1: "lst", 2: "twn", 3: "rdm", 4: "look word 0", 5: "look mask", 6: "tmp", 9: "bad input"}.get(rc, str(rc))
    return (what, int(diag[0]), int(diag[1]), int(diag[2]), hex(int(diag[3])), hex(int(diag[4])))


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.emu_narrow_run.restype = C.c_int
    return _LIB


def expand_witness(d, chain, init, branch, by_completion=False):
    """The product's host-side replay of a chain of branching calls into the whole witness (csrc/witness_expand.h)."""
    f = np.ascontiguousarray(d["f"], np.uint8); a = np.ascontiguousarray(d["a"], np.int32); b = np.ascontiguousarray(d["b"], np.int32)
    proc = np.ascontiguousarray(d["process"], np.int32)
    inv = np.ascontiguousarray(d["inv_pos"], np.uint32); ret = np.ascontiguousarray(d["ret_pos"], np.uint32)
    ch = np.ascontiguousarray(chain, np.uint32)
    out = np.zeros(len(f) + 1, np.uint32)
    fn = lib().emu_expand_witness
    fn.restype = C.c_int64
    n = fn(C.c_uint32(len(f)), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32), _p(proc, C.c_int32), _p(inv, C.c_uint32), _p(ret, C.c_uint32),
           C.c_uint32(int(proc.max()) + 1 if len(proc) else 1), C.c_int32(init), C.c_uint32(bool(branch)), C.c_uint32(bool(by_completion)),
           _p(ch, C.c_uint32), C.c_uint32(len(ch)), _p(out, C.c_uint32))
    return None if n < 0 else [int(x) for x in out[:n]]


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def rules_for(hists, model_kind, init, nil=-(2 ** 31)):
    """(rules, vpad) as libtbcheck chooses them for a register-family batch (tbc_api.hip): both rules when every value is 0..30."""
    if model_kind not in (0, 1):
        return 0, 0
    vmax = -1 if init == nil else init
    ok = init == nil or init >= 0
    for h in hists:
        d = h if isinstance(h, dict) else h.as_dict()
        a = np.asarray(d["a"], np.int64); f = np.asarray(d["f"]); b = np.asarray(d["b"], np.int64)
        av = a[a != nil]
        if len(av):
            ok = ok and av.min() >= 0
            vmax = max(vmax, int(av.max()))
        bv = b[f == 2]
        if len(bv):
            ok = ok and bv.min() >= 0
            vmax = max(vmax, int(bv.max()))
    if not ok or vmax > 30:
        return 0, 0
    vpad = 2
    while vpad < vmax + 2:
        vpad <<= 1
    return 3, vpad


def run(hists, model_kind, init, L, rules=None, lookahead=True, entries_per_op=8, max_steps=0, pool_words=0, want_witness=True, max_waves=0, branch_lists=True, compact=True, epochs=0, count=False, relaxed=False, targets=None, mw=None, by_ret=False, stall=0):
    """hists: list of op-column dicts (f,a,b,process,inv_pos,ret_pos,n_process).  Returns one result dict per history."""
    ds = [h if isinstance(h, dict) else h.as_dict() for h in hists]
    nh = len(ds)
    op_off = np.zeros(nh + 1, np.uint64)
    for i, d in enumerate(ds):
        op_off[i + 1] = op_off[i] + len(d["f"])
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(d[k], dt) for d in ds]) if nh else np.zeros(0, dt), dt)
    f, a, b = cat("f", np.uint8), cat("a", np.int32), cat("b", np.int32)
    pr, inv, ret = cat("process", np.int32), cat("inv_pos", np.uint32), cat("ret_pos", np.uint32)
    npr = np.array([int(d["n_process"]) for d in ds], np.uint32)
    if mw is None:
        mw = max(1, (int(npr.max()) + 63) // 64)
        mw = 1 if mw <= 1 else 2 if mw <= 2 else 4
    r, vpad = rules_for(ds, model_kind, init)
    if rules is not None:
        r &= rules
    if (r & 1) and branch_lists and not count:
        r |= 4                 # kRuleBranch: what libtbcheck sets for the narrow kernel whenever the eager rule is on
    look = bool(lookahead) and model_kind in (0, 1)
    res = (DevResult * nh)()
    total = int(op_off[-1])
    wit = np.zeros(total + 1, np.uint32)
    cfg = np.zeros(nh * 256 * (2 + mw) + 1, np.uint64)
    rc = lib().emu_narrow_run(C.c_uint32(nh), _p(op_off, C.c_uint64), _p(npr, C.c_uint32), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32),
                              _p(pr, C.c_int32), _p(inv, C.c_uint32), _p(ret, C.c_uint32), C.c_uint32(model_kind), C.c_int32(init),
                              C.c_uint32(L), C.c_uint32(mw), C.c_uint32(r), C.c_uint32(vpad), C.c_uint32(1 if look else 0),
                              C.c_uint32(entries_per_op), C.c_uint64(max_steps), C.c_uint64(pool_words), C.c_uint32(1 if want_witness else 0), C.c_uint32(max_waves), C.c_uint32((1 if compact else 0) | (int(stall) << 8) | ((int(by_ret) << 16) if int(by_ret) >= 16 else 16 if by_ret == 2 else 4 if by_ret else 0)), C.c_uint32(epochs), C.c_uint32(1 if count else 0), C.c_uint32(1 if relaxed else 0),
                              _p(np.ascontiguousarray(targets, np.uint32), C.c_uint32) if targets is not None else None,
                              res, _p(wit, C.c_uint32), _p(cfg, C.c_uint64))
    if rc != 0:
        raise RuntimeError(f"emu_narrow_run rc={rc}")
    out = []
    for i in range(nh):
        d = {k: getattr(res[i], k) for k, _ in DevResult._fields_}
        o = int(op_off[i])
        d["chain"] = wit[o:o + d["depth"]].copy() if (d["valid"] == 1 and want_witness) else None
        rec = cfg[i * 256 * (2 + mw):(i + 1) * 256 * (2 + mw)].reshape(256, 2 + mw)
        d["cfg"] = rec[:min(d["n_configs"], 256)].copy()
        d["rules"], d["vpad"], d["mw"] = r, vpad, mw
        out.append(d)
    return out


# ---- K6w: the level sweep by workgroups (csrc/jit_sweep_wg_impl.h) under the workgroup emulator
_SO_SWEEP = os.path.join(_HERE, "_build", "libemu_sweep_asan.so" if _ASAN else "libemu_sweep.so")
_LIB_SWEEP = None


def build_sweep(force=False):
    srcs = [os.path.join(_HERE, "emu_sweep.cpp"), os.path.join(_HERE, "wave_env_emu.h"), os.path.join(_HERE, "wave_env_wg_emu.h"), os.path.join(_HERE, "host_tables.h"),
            os.path.join(_CSRC, "jit_sweep_wg_impl.h"), os.path.join(_CSRC, "tbc_internal.h"), os.path.join(_CSRC, "wave_env.h"), os.path.join(_CSRC, "wave_env_wg.h"),
            os.path.join(_CSRC, "reach_table.h")]
    if force or not os.path.exists(_SO_SWEEP) or any(os.path.getmtime(s) > os.path.getmtime(_SO_SWEEP) for s in srcs):
        os.makedirs(os.path.dirname(_SO_SWEEP), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]
                              + (["-fsanitize=address", "-fno-omit-frame-pointer"] if _ASAN else [])
                              + ["-I", _HERE, "-I", _CSRC, "-I", os.path.join(_ROOT, "include"), "-o", _SO_SWEEP, srcs[0]])
    return _SO_SWEEP


def sweep_wg(ops, model_kind, init, seg_target, n_dom, max_segs, waves=8, cap=1024, rules=None, seed=1, rel_bytes=688, again=None, first=None, compact=False, relaxed=False):
    """ONE history through K6w: max_segs * 4 tbc_sweep_rel records as a uint8 array (as oracle.wgl.sweep_relations returns them).
    again = [(segment, slice), ...] with first = the first pass's records: the second pass over those segments only."""
    global _LIB_SWEEP
    if _LIB_SWEEP is None:
        _LIB_SWEEP = C.CDLL(build_sweep())
        _LIB_SWEEP.emu_sweep_wg_run.restype = C.c_int
    d = ops if isinstance(ops, dict) else ops.as_dict()
    f, a, b = (np.ascontiguousarray(d[k], t) for k, t in (("f", np.uint8), ("a", np.int32), ("b", np.int32)))
    pr, inv, ret = (np.ascontiguousarray(d[k], t) for k, t in (("process", np.int32), ("inv_pos", np.uint32), ("ret_pos", np.uint32)))
    r, vpad = rules_for([d], model_kind, init)
    if rules is not None:
        r &= rules
    buf = np.zeros(max_segs * 4 * rel_bytes, np.uint8) if again is None else np.array(first, np.uint8, copy=True)
    sl = None if again is None else np.ascontiguousarray([x for (k, j) in again for x in (0, k, j)], np.uint32)
    rc = _LIB_SWEEP.emu_sweep_wg_run(C.c_uint32(len(f)), C.c_uint32(int(d["n_process"])), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32), _p(pr, C.c_int32),
                                     _p(inv, C.c_uint32), _p(ret, C.c_uint32), C.c_uint32(model_kind), C.c_int32(init), C.c_uint32(vpad), C.c_uint32(r),
                                     C.c_uint32(n_dom), C.c_uint32(seg_target), C.c_uint32(max_segs), C.c_uint32(waves), C.c_uint32(cap), C.c_uint32((4 if compact else 0) | (8 if compact == 2 else 0) | (16 if relaxed else 0)), C.c_uint64(seed),      # compact=2: + solo passes
                                     None if sl is None else _p(sl, C.c_uint32), C.c_uint32(0 if again is None else len(again)), buf.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(f"emu_sweep_wg_run rc={rc}")
    return buf


# ---- K1 for one history: pack by a workgroup's sixteen wavefronts (csrc/pack_one_impl.h) under the workgroup emulator
_SO_PACK = os.path.join(_HERE, "_build", "libemu_pack_asan.so" if _ASAN else "libemu_pack.so")
_LIB_PACK = None


def build_pack(force=False):
    srcs = [os.path.join(_HERE, "emu_pack.cpp"), os.path.join(_HERE, "wave_env_emu.h"), os.path.join(_HERE, "wave_env_wg_emu.h"), os.path.join(_HERE, "host_tables.h"),
            os.path.join(_CSRC, "pack_one_impl.h"), os.path.join(_CSRC, "tbc_internal.h"), os.path.join(_CSRC, "wave_env.h"), os.path.join(_CSRC, "wave_env_wg.h")]
    if force or not os.path.exists(_SO_PACK) or any(os.path.getmtime(s) > os.path.getmtime(_SO_PACK) for s in srcs):
        os.makedirs(os.path.dirname(_SO_PACK), exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas"]
                              + (["-fsanitize=address", "-fno-omit-frame-pointer"] if _ASAN else [])
                              + ["-I", _HERE, "-I", _CSRC, "-I", os.path.join(_ROOT, "include"), "-o", _SO_PACK, srcs[0]])
    return _SO_PACK


def pack_one_check(hists, model_kind=1, n_classes=0, count=False, per_launch=0, seed=1, n_events=None):
    """Run pack_one (csrc/pack_one_impl.h) under the workgroup emulator on these histories (op-column dicts or OpColumns), one
    workgroup of sixteen wavefronts each, and compare every word it leaves -- records, list starts, completion tables, the scratch
    arena, n_ret, status -- with the host restatement in emu_pack.cpp.  n_events: rows of each history (default: last position + 1).
    Returns None when all agree, else (what, history, index, got, want)."""
    global _LIB_PACK
    if _LIB_PACK is None:
        _LIB_PACK = C.CDLL(build_pack())
        _LIB_PACK.emu_pack_one_check.restype = C.c_int
    ds = [h if isinstance(h, dict) else h.as_dict() for h in hists]
    nh = len(ds)
    op_off = np.zeros(nh + 1, np.uint64)
    for i, d in enumerate(ds):
        op_off[i + 1] = op_off[i] + len(d["f"])
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(d[k], dt) for d in ds]), dt)
    f, a, b = cat("f", np.uint8), cat("a", np.int32), cat("b", np.int32)
    pr, inv, ret = cat("process", np.int32), cat("inv_pos", np.uint32), cat("ret_pos", np.uint32)
    npr = np.array([int(d["n_process"]) for d in ds], np.uint32)
    if n_events is None:
        n_events = []
        for h, d in zip(hists, ds):
            if getattr(h, "n_events", None) is not None:
                n_events.append(int(h.n_events))
                continue
            r = np.asarray(d["ret_pos"], np.uint64); i = np.asarray(d["inv_pos"], np.uint64)
            live = r[r != 0xFFFFFFFF]
            n_events.append(int(max(int(i.max()) if len(i) else 0, int(live.max()) if len(live) else 0)) + 1)
    ne = np.ascontiguousarray(n_events, np.uint32)
    diag = np.zeros(8, np.uint64)
    rc = _LIB_PACK.emu_pack_one_check(C.c_uint32(nh), _p(op_off, C.c_uint64), _p(npr, C.c_uint32), _p(ne, C.c_uint32), _p(f, C.c_uint8), _p(a, C.c_int32),
                                      _p(b, C.c_int32), _p(pr, C.c_int32), _p(inv, C.c_uint32), _p(ret, C.c_uint32), C.c_uint32(model_kind),
                                      C.c_uint32(n_classes), C.c_uint32(1 if count else 0), C.c_uint32(per_launch), C.c_uint64(seed), _p(diag, C.c_uint64))
    if rc == 0:
        return None
    what = {1: "status", 2: "n_ret", 3: "seg", 4: "ret_slot", 5: "ret_op", 6: "scratch", 7: "rec word", 90: "does not fit"}.get(rc, str(rc))
    return (what, int(diag[0]), int(diag[1]), int(diag[2]), int(diag[3]))


def pack_wg_check(hists, model_kind=1, n_classes=0, vpad=8, count=False, branch=False, look=True, rk8=True, lst_cap=0, per_launch=0, seed=1, n_events=None, one=False, slots64=False):
    """The batch form (csrc/pack_one_impl.h, BatchGeo: four wavefronts per history, pack + open counts in one pass) under the workgroup
    emulator: every word pack_kernel AND open_counts_kernel would leave -- as pack_one_check, plus off[], ncr[], slot8, rk8, the
    crashed-call list, the lookahead records past the last rank, BeamHist.status / n_crashed / lst_need -- against the restatements in
    emu_pack.cpp and host_tables.h.  Returns None when all agree, else (what, history, index, got, want)."""
    global _LIB_PACK
    if _LIB_PACK is None:
        _LIB_PACK = C.CDLL(build_pack())
        _LIB_PACK.emu_pack_one_check.restype = C.c_int
    _LIB_PACK.emu_pack_wg_check.restype = C.c_int
    ds = [h if isinstance(h, dict) else h.as_dict() for h in hists]
    nh = len(ds)
    op_off = np.zeros(nh + 1, np.uint64)
    for i, d in enumerate(ds):
        op_off[i + 1] = op_off[i] + len(d["f"])
    cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(d[k], dt) for d in ds]), dt)
    f, a, b = cat("f", np.uint8), cat("a", np.int32), cat("b", np.int32)
    pr, inv, ret = cat("process", np.int32), cat("inv_pos", np.uint32), cat("ret_pos", np.uint32)
    npr = np.array([int(d["n_process"]) for d in ds], np.uint32)
    if n_events is None:
        n_events = []
        for h, d in zip(hists, ds):
            if getattr(h, "n_events", None) is not None:
                n_events.append(int(h.n_events))
                continue
            r = np.asarray(d["ret_pos"], np.uint64); i = np.asarray(d["inv_pos"], np.uint64)
            live = r[r != 0xFFFFFFFF]
            n_events.append(int(max(int(i.max()) if len(i) else 0, int(live.max()) if len(live) else 0)) + 1)
    ne = np.ascontiguousarray(n_events, np.uint32)
    diag = np.zeros(8, np.uint64)
    flags = (1 if count else 0) | (2 if branch else 0) | (4 if look else 0) | (8 if rk8 else 0) | (16 if one else 0) | (64 if slots64 else 0)      # one: sixteen wavefronts (OneCountsGeo); slots64: Batch64Geo (at most 64 process slots, 19 KB)
    rc = _LIB_PACK.emu_pack_wg_check(C.c_uint32(nh), _p(op_off, C.c_uint64), _p(npr, C.c_uint32), _p(ne, C.c_uint32), _p(f, C.c_uint8), _p(a, C.c_int32),
                                     _p(b, C.c_int32), _p(pr, C.c_int32), _p(inv, C.c_uint32), _p(ret, C.c_uint32), C.c_uint32(model_kind),
                                     C.c_uint32(n_classes), C.c_uint32(vpad), C.c_uint32(flags), C.c_uint32(lst_cap), C.c_uint32(per_launch), C.c_uint64(seed), _p(diag, C.c_uint64))
    if rc == 0:
        return None
    what = {1: "status", 2: "n_ret", 3: "seg", 4: "ret_slot", 5: "ret_op", 6: "scratch", 7: "rec word", 10: "off", 11: "ncr", 12: "slot8", 13: "rk8", 14: "crashed",
            15: "BeamHist.status", 16: "n_crashed", 17: "lst_need", 18: "look", 90: "does not fit", 91: "host tables"}.get(rc, str(rc))
    return (what, int(diag[0]), int(diag[1]), int(diag[2]), int(diag[3]))
