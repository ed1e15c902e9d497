"""Column (SoA) forms of a history and the host-side pairing step.

`EventColumns` is one row per Jepsen op map (:type :process :f :value);
`OpColumns` is the history after knossos.history/complete + without-failures
+ pairing (tbc_pair_events in the library), which is what crosses the C-ABI.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _native as N


def _p(arr, ct):
    return arr.ctypes.data_as(C.POINTER(ct))


@dataclass
class EventColumns:
    type: np.ndarray      # uint8
    process: np.ndarray   # int32
    f: np.ndarray         # uint8
    a: np.ndarray         # int32
    b: np.ndarray         # int32

    def __len__(self):
        return len(self.type)

    def struct(self) -> N.Events:
        e = N.Events()
        e.n = len(self.type)
        e.type = _p(self.type, C.c_uint8)
        e.process = _p(self.process, C.c_int32)
        e.f = _p(self.f, C.c_uint8)
        e.a = _p(self.a, C.c_int32)
        e.b = _p(self.b, C.c_int32)
        return e


@dataclass
class OpColumns:
    f: np.ndarray         # uint8
    a: np.ndarray         # int32
    b: np.ndarray         # int32
    process: np.ndarray   # int32, dense
    inv_pos: np.ndarray   # uint32, ascending
    ret_pos: np.ndarray   # uint32, POS_CRASHED for :info
    n_events: int
    n_process: int
    pool: np.ndarray = None   # int32 values of wide ops (multi-register micro-ops), or None

    def __len__(self):
        return len(self.f)

    def struct(self) -> N.Ops:
        o = N.Ops()
        o.n = len(self.f)
        o.n_events = self.n_events
        o.f = _p(self.f, C.c_uint8)
        o.a = _p(self.a, C.c_int32)
        o.b = _p(self.b, C.c_int32)
        o.process = _p(self.process, C.c_int32)
        o.inv_pos = _p(self.inv_pos, C.c_uint32)
        o.ret_pos = _p(self.ret_pos, C.c_uint32)
        if self.pool is not None and len(self.pool):
            o.pool = _p(self.pool, C.c_int32)
            o.pool_len = len(self.pool)
        else:
            o.pool = None
            o.pool_len = 0
        o.n_process = self.n_process
        return o

    def as_dict(self):
        return {"f": self.f, "a": self.a, "b": self.b, "process": self.process,
                "inv_pos": self.inv_pos, "ret_pos": self.ret_pos, "n_process": self.n_process, "pool": self.pool}


def pair_events(ev: EventColumns) -> OpColumns:
    """knossos.history/complete + without-failures + pairing (host code in the library)."""
    n = len(ev)
    f = np.zeros(max(n, 1), np.uint8)
    a = np.zeros(max(n, 1), np.int32)
    b = np.zeros(max(n, 1), np.int32)
    proc = np.zeros(max(n, 1), np.int32)
    inv = np.zeros(max(n, 1), np.uint32)
    ret = np.zeros(max(n, 1), np.uint32)
    n_ops = C.c_uint32(0)
    n_proc = C.c_uint32(0)
    es = ev.struct()
    st = N.lib().tbc_pair_events(C.byref(es), _p(f, C.c_uint8), _p(a, C.c_int32), _p(b, C.c_int32),
                                 _p(proc, C.c_int32), _p(inv, C.c_uint32), _p(ret, C.c_uint32),
                                 C.byref(n_ops), C.byref(n_proc))
    N.check_status(st)
    k = n_ops.value
    return OpColumns(f[:k].copy(), a[:k].copy(), b[:k].copy(), proc[:k].copy(), inv[:k].copy(), ret[:k].copy(),
                     n_events=n, n_process=n_proc.value)
