"""ctypes binding of libtbcheck.so (include/tbcheck.h, include/tbsynth.h).

This is the Python stand-in for the JNA binding a Clojure caller would use
(INTEGRATION.md): same entry points, same structs.  The library is built
in-tree by `build()` (hipcc, gfx950) and there is NO fallback: if the shared
object is missing or a GPU entry point reports TBC_ERR_NO_DEVICE the caller
gets an exception, never a CPU answer.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# TBC_LIB_PATH: load another build of the SAME library (A/B runs of kernel variants on the GPU box)
LIB_PATH = os.environ.get("TBC_LIB_PATH") or os.path.join(CSRC, "libtbcheck.so")

NIL = -(2 ** 31)
POS_CRASHED = 0xFFFFFFFF
NO_OP = 0xFFFFFFFF
MAX_FINAL_CONFIGS = 10

# :type
INVOKE, OK, FAIL, INFO = 0, 1, 2, 3
# :f
F_READ, F_WRITE, F_CAS, F_ACQUIRE, F_RELEASE, F_ADD, F_TXN, F_TRANSFER, F_CLASS = range(9)
# models
MODEL_REGISTER, MODEL_CAS_REGISTER, MODEL_MUTEX, MODEL_TABLE, MODEL_MULTI_REGISTER, MODEL_SET, MODEL_BANK = range(7)
# algorithms
ALG_COMPETITION, ALG_WGL, ALG_LINEAR = 0, 1, 2
# verdicts / causes
VALID, INVALID, UNKNOWN = 1, 0, -1
CAUSE_NONE, CAUSE_TIME_LIMIT, CAUSE_STEP_LIMIT, CAUSE_VISITED_FULL = 0, 1, 2, 3
DOM_NO_EAGER_READS, DOM_NO_TWIN_RULE, DOM_NO_COUNT_FORM, DOM_NO_LAZY_COMMUTING, DOM_STALL_HANDOVER, DOM_NO_ORDER_RESTARTS = 1, 2, 4, 8, 16, 32
DOM_NO_EAGER_TXNS, DOM_NO_TXN_INDEPENDENCE = 64, 128
# tbc_opts.list_order (16 + W: completion order, a :write as if it completed W ranks later; the default where it applies is 16 + 24)
ORDER_DEFAULT, ORDER_SLOT, ORDER_COMPLETION, ORDER_WRITES_LAST, ORDER_WRITE_DELAY = 0, 1, 2, 3, 16
# status
(OK_STATUS, ERR_INVALID_ARG, ERR_BAD_HISTORY, ERR_NO_DEVICE, ERR_OOM, ERR_WINDOW_TOO_WIDE,
 ERR_MODEL, ERR_HIP, ERR_UNSUPPORTED) = range(9)


class TbcError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__(f"libtbcheck status {status}: {detail}")
        self.status = status
        self.detail = detail


class NoDeviceError(TbcError):
    pass


class Events(C.Structure):
    _fields_ = [("n", C.c_uint32), ("type", C.POINTER(C.c_uint8)), ("process", C.POINTER(C.c_int32)),
                ("f", C.POINTER(C.c_uint8)), ("a", C.POINTER(C.c_int32)), ("b", C.POINTER(C.c_int32))]


class Ops(C.Structure):
    _fields_ = [("n", C.c_uint32), ("n_events", C.c_uint32), ("f", C.POINTER(C.c_uint8)),
                ("a", C.POINTER(C.c_int32)), ("b", C.POINTER(C.c_int32)), ("process", C.POINTER(C.c_int32)),
                ("inv_pos", C.POINTER(C.c_uint32)), ("ret_pos", C.POINTER(C.c_uint32)),
                ("pool", C.POINTER(C.c_int32)), ("pool_len", C.c_uint32), ("n_process", C.c_uint32)]


class Model(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("init", C.c_int32), ("table", C.POINTER(C.c_uint16)),
                ("n_states", C.c_uint32), ("n_classes", C.c_uint32), ("n_keys", C.c_uint32),
                ("flags", C.c_uint32)]


class Opts(C.Structure):
    _fields_ = [("algorithm", C.c_uint32), ("device", C.c_uint32), ("time_limit_ms", C.c_uint64),
                ("max_steps", C.c_uint64), ("max_visited_bytes", C.c_uint64),
                ("want_witness", C.c_uint32), ("visited_per_op", C.c_uint32),
                ("search_width", C.c_uint32), ("round_budget", C.c_uint32),
                ("lookahead", C.c_uint32), ("dominance", C.c_uint32), ("lanes_per_history", C.c_uint32), ("list_order", C.c_uint32)]


class Config(C.Structure):
    _fields_ = [("state", C.c_int32), ("last_op", C.c_uint32), ("n_pending", C.c_uint32),
                ("n_linearized", C.c_uint32), ("pending", C.c_uint32 * 16), ("linearized_mask", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("steps", C.c_uint64), ("visited", C.c_uint64), ("probes", C.c_uint64),
                ("backtracks", C.c_uint64), ("max_depth", C.c_uint64), ("table_slots", C.c_uint64),
                ("ns_pack", C.c_uint64), ("ns_search", C.c_uint64), ("ns_total", C.c_uint64)]


class Result(C.Structure):
    _fields_ = [("valid", C.c_int32), ("cause", C.c_int32), ("analyzer", C.c_uint32),
                ("fail_op", C.c_uint32), ("prev_ok_op", C.c_uint32), ("final_state", C.c_int32),
                ("n_witness", C.c_uint32), ("search_width", C.c_uint32), ("witness", C.POINTER(C.c_uint32)), ("n_configs", C.c_uint32),
                ("configs", Config * MAX_FINAL_CONFIGS), ("counters", Counters)]


class BatchDesc(C.Structure):
    _fields_ = [("n_hist", C.c_uint32), ("op_off", C.POINTER(C.c_uint64)), ("n_events", C.POINTER(C.c_uint32)),
                ("n_process", C.POINTER(C.c_uint32)), ("cols", Ops), ("model_aux", C.POINTER(C.c_int32))]


class SweepInfo(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("seg_target", C.c_uint32), ("max_segs", C.c_uint32),
                ("cut_open", C.c_uint32), ("n_dom", C.c_uint32), ("n_segments", C.c_uint32), ("n_fallback", C.c_uint32)]


class SweepRel(C.Structure):
    """tbc_sweep_rel: what one wavefront of the level sweep hands on (exchanged between ranks as raw bytes)."""
    _fields_ = [("status", C.c_uint32), ("F0", C.c_uint32), ("F1", C.c_uint32), ("n_org", C.c_uint32),
                ("max_level", C.c_uint32), ("subrounds", C.c_uint32), ("n_end", C.c_uint32), ("end_state", C.c_uint32),
                ("configs_total", C.c_uint64), ("probes", C.c_uint64),
                ("M", (C.c_uint32 * 4) * 32), ("last_level", C.c_uint32 * 32)]


class SweepVerdict(C.Structure):
    _fields_ = [("valid", C.c_int32), ("fail_level", C.c_uint32), ("fail_seg", C.c_uint32), ("live_in", C.c_uint32 * 4),
                ("final_bits", C.c_uint32), ("end_state", C.c_uint32), ("n_wavefronts", C.c_uint32),
                ("probes", C.c_uint64), ("configs_total", C.c_uint64), ("subrounds", C.c_uint64), ("max_level", C.c_uint64)]


SWEEP_SLICES = 4


class SetFullIn(C.Structure):
    _fields_ = [("n_elements", C.c_uint32), ("n_reads", C.c_uint32), ("words_per_row", C.c_uint32), ("device", C.c_uint32),
                ("add_invoke", C.POINTER(C.c_uint32)), ("add_ok", C.POINTER(C.c_uint32)), ("read_invoke", C.POINTER(C.c_uint32)),
                ("read_ok", C.POINTER(C.c_uint32)), ("present", C.POINTER(C.c_uint32))]


class SetFullRows(C.Structure):
    _fields_ = [("n_elements", C.c_uint32), ("n_reads", C.c_uint32), ("device", C.c_uint32), ("reserved0", C.c_uint32),
                ("add_invoke", C.POINTER(C.c_uint32)), ("add_ok", C.POINTER(C.c_uint32)), ("read_invoke", C.POINTER(C.c_uint32)),
                ("read_ok", C.POINTER(C.c_uint32)), ("top", C.POINTER(C.c_uint32)), ("exc_off", C.POINTER(C.c_uint64)),
                ("exc", C.POINTER(C.c_uint32))]


class SetFullOut(C.Structure):
    _fields_ = [("known", C.POINTER(C.c_uint32)), ("last_present", C.POINTER(C.c_uint32)), ("last_absent", C.POINTER(C.c_uint32)),
                ("ns_scan", C.c_uint64), ("bytes_scanned", C.c_uint64), ("bytes_matrix", C.c_uint64)]


class BatchInput(C.Structure):
    """tbc_batch_input: pointers into one pinned slot of a batch (tbc_batch_map_input)."""
    _fields_ = [("n_hist_cap", C.c_uint32), ("reserved0", C.c_uint32), ("ops_cap", C.c_uint64),
                ("op_off", C.POINTER(C.c_uint64)), ("n_events", C.POINTER(C.c_uint32)), ("n_process", C.POINTER(C.c_uint32)),
                ("word", C.POINTER(C.c_uint32)), ("inv_pos", C.POINTER(C.c_uint32)), ("ret_pos", C.POINTER(C.c_uint32))]


class Progress(C.Structure):
    """tbc_progress: a run that is out, seen from another thread (include/tbcheck.h tbc_batch_progress)."""
    _fields_ = [("n_histories", C.c_uint32), ("n_decided", C.c_uint32), ("phase", C.c_uint32), ("running", C.c_uint32), ("elapsed_ns", C.c_uint64)]


PHASE_IDLE, PHASE_PACK, PHASE_RETRIES = 0, 1, 2


class InputInfo(C.Structure):
    _fields_ = [("n_hist", C.c_uint32), ("pending", C.c_uint32), ("total_ops", C.c_uint64), ("bytes_copied", C.c_uint64),
                ("ns_copy", C.c_uint64), ("inputs_consumed", C.c_uint64), ("lists_regrown", C.c_uint32), ("n_hist_cap", C.c_uint32),
                ("ops_cap", C.c_uint64)]


WIRE_NIL = 0xFF


class SynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_ops", C.c_uint32), ("n_procs", C.c_uint32), ("n_values", C.c_uint32),
                ("busy_permille", C.c_uint32), ("info_permille", C.c_uint32), ("read_permille", C.c_uint32),
                ("write_permille", C.c_uint32), ("corrupt_permille", C.c_uint32)]


STEP_FN = C.CFUNCTYPE(C.c_int64, C.c_int64, C.c_uint32, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)      # tbc_allgather_fn(user, send, recv, bytes)
COMM_ID_BYTES = 128

# every symbol the headers declare: (name, restype, argtypes)
SYMBOLS = {
    "tbc_pair_events": (C.c_int, [C.POINTER(Events), C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "tbc_check": (C.c_int, [C.POINTER(Ops), C.POINTER(Model), C.POINTER(Opts), C.POINTER(Result)]),
    "tbc_result_free": (None, [C.POINTER(Result)]),
    "tbc_batch_create": (C.c_int, [C.POINTER(BatchDesc), C.POINTER(Model), C.POINTER(Opts), C.POINTER(C.c_void_p)]),
    "tbc_batch_run": (C.c_int, [C.c_void_p, C.POINTER(Result)]),
    "tbc_batch_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "tbc_batch_last_turn_wait": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "tbc_batch_last_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "tbc_batch_device_bytes": (C.c_uint64, [C.c_void_p]),
    "tbc_batch_search_width": (C.c_uint32, [C.c_void_p]),
    "tbc_batch_lanes_per_history": (C.c_uint32, [C.c_void_p]),
    "tbc_batch_list_order": (C.c_uint32, [C.c_void_p]),
    "tbc_batch_last_raced": (C.c_uint32, [C.c_void_p]),
    "tbc_batch_sweep_info": (C.c_int, [C.c_void_p, C.POINTER(SweepInfo)]),
    "tbc_sweep_compose": (C.c_int, [C.POINTER(SweepRel), C.c_uint32, C.c_uint32, C.POINTER(SweepVerdict)]),
    "tbc_batch_set_shard": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "tbc_batch_sweep_partial": (C.c_int, [C.c_void_p]),
    "tbc_batch_sweep_table": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
    "tbc_batch_sweep_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Result)]),
    "tbc_batch_sweep_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(Result)]),
    "tbc_setfull_create": (C.c_int, [C.POINTER(SetFullIn), C.POINTER(C.c_void_p)]),
    "tbc_setfull_create_rows": (C.c_int, [C.POINTER(SetFullRows), C.POINTER(C.c_void_p)]),
    "tbc_setfull_run": (C.c_int, [C.c_void_p, C.POINTER(SetFullOut)]),
    "tbc_setfull_destroy": (None, [C.c_void_p]),
    "tbc_batch_destroy": (None, [C.c_void_p]),
    "tbc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "tbc_comm_init": (C.c_int, [C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "tbc_comm_init_host": (C.c_int, [C.c_uint32, C.c_uint32, ALLGATHER_FN, C.c_void_p, C.POINTER(C.c_void_p)]),
    "tbc_comm_rank": (C.c_uint32, [C.c_void_p]),
    "tbc_comm_world": (C.c_uint32, [C.c_void_p]),
    "tbc_comm_destroy": (None, [C.c_void_p]),
    "tbc_batch_sweep_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Result)]),
    "tbc_batch_map_input": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(BatchInput)]),
    "tbc_batch_submit_input": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "tbc_batch_reload": (C.c_int, [C.c_void_p, C.POINTER(BatchDesc)]),
    "tbc_batch_input_info": (C.c_int, [C.c_void_p, C.POINTER(InputInfo)]),
    "tbc_batch_progress": (C.c_int, [C.c_void_p, C.POINTER(Progress)]),
    "tbc_memo_build": (C.c_int, [C.c_int64, C.c_uint32, STEP_FN, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint16),
                                 C.POINTER(C.c_int64), C.POINTER(C.c_uint32)]),
    "tbc_version": (C.c_uint32, []),
    "tbc_strerror": (C.c_char_p, [C.c_int]),
    "tbc_last_error": (C.c_char_p, []),
    "tbc_device_count": (C.c_int32, []),
    "tbc_debug_peek": (C.c_int, [C.POINTER(C.c_uint32), C.c_uint32]),
    "tbs_gen_register": (C.c_int, [C.POINTER(SynthParams), C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                   C.POINTER(C.c_uint32)]),
}

_LIB = None
ABI_VERSION = 2          # include/tbcheck.h TBC_ABI_VERSION


def build(force: bool = False) -> str:
    """Compile libtbcheck.so for gfx950 with hipcc (in-tree, so it travels to the GPU box)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h"))]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("tbcheck.h", "tbsynth.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j8", "libtbcheck.so"])
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is missing: run __graft_entry__.build() (hipcc). There is no CPU fallback.")
        try:  # share torch's HIP runtime (same SONAME) when torch is in the process
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
        _LIB = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(_LIB, name)
            fn.restype = res
            fn.argtypes = args
        if _LIB.tbc_version() != ABI_VERSION:      # the structs above are this version's: another layout would be read past its end
            v, _LIB = _LIB.tbc_version(), None
            raise ImportError(f"{LIB_PATH} has ABI version {v}, this binding is written for {ABI_VERSION}: rebuild (__graft_entry__.build())")
    return _LIB


def check_status(status: int):
    if status == OK_STATUS:
        return
    detail = lib().tbc_last_error().decode() or lib().tbc_strerror(status).decode()
    if status == ERR_NO_DEVICE:
        raise NoDeviceError(status, detail)
    raise TbcError(status, detail)
