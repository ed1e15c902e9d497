"""knossos.model.memo/memo -- turn any Model into a dense transition table
(states x op classes) so the search kernel steps it with one lookup.

The closure enumeration itself is tbc_memo_build in the library (C-ABI, host
code); this module supplies the Python `Model.step` as its callback and hands
the table to the TBC_MODEL_TABLE device path.  Like Knossos's memo it gives up
past a size cap (MemoTooLarge) -- the direct device models (register family,
mutex) do not need it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _native as N
from . import model as M


class MemoTooLarge(ValueError):
    pass


def _freeze(v):
    if isinstance(v, dict):
        return ("#map", tuple(sorted((k, _freeze(x)) for k, x in v.items())))
    if isinstance(v, (list, tuple)):
        return tuple(_freeze(x) for x in v)
    if isinstance(v, (set, frozenset)):
        return ("#set", tuple(sorted(_freeze(x) for x in v)))
    return v


def op_class_key(op):
    return (op["f"], _freeze(op.get("value")))


def memo(model, ops, max_states=4096):
    """ops: iterable of op maps (after `complete`).  Returns dict with
    table (n_states x n_classes uint16), classes {key: id}, class_ops [op],
    states [Model] (state id -> model)."""
    classes, class_ops = {}, []
    for op in ops:
        k = op_class_key(op)
        if k not in classes:
            classes[k] = len(class_ops)
            class_ops.append({"f": op["f"], "value": op.get("value")})
    n_classes = max(1, len(class_ops))
    states = [model]
    ids = {model: 0}

    def step(handle, cls, _user):
        if cls >= len(class_ops):
            return -1
        nxt = states[handle].step(class_ops[cls])
        if M.inconsistent_p(nxt):
            return -1
        h = ids.get(nxt)
        if h is None:
            h = len(states)
            ids[nxt] = h
            states.append(nxt)
        return h

    cb = N.STEP_FN(step)
    table = np.zeros((max_states, n_classes), np.uint16)
    handles = np.zeros(max_states, np.int64)
    n_states = C.c_uint32(0)
    st = N.lib().tbc_memo_build(0, n_classes, cb, None, max_states,
                                table.ctypes.data_as(C.POINTER(C.c_uint16)),
                                handles.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n_states))
    if st == N.ERR_MODEL:
        raise MemoTooLarge(N.lib().tbc_last_error().decode())
    N.check_status(st)
    ns = n_states.value
    # handles are our own state indices in discovery order == table row order
    assert list(handles[:ns]) == list(range(ns))
    return {"table": table[:ns].copy(), "classes": classes, "class_ops": class_ops, "states": states[:ns]}
