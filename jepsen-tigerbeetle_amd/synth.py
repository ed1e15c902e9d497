"""Seeded synthetic histories (include/tbsynth.h) -- the shapes SURVEY.md section 8d
prescribes; the reference itself holds no recorded histories."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .columns import EventColumns, _p


def register_events(n_ops=1000, n_procs=16, seed=0, n_values=5, busy=0.5, info=0.0,
                    read=1 / 3, write=1 / 3, corrupt=0.0) -> EventColumns:
    """cas-register history (plain register when read + write == 1)."""
    p = N.SynthParams()
    p.seed = seed
    p.n_ops = n_ops
    p.n_procs = n_procs
    p.n_values = n_values
    p.busy_permille = int(round(busy * 1000))
    p.info_permille = int(round(info * 1000))
    p.read_permille = int(round(read * 1000))
    p.write_permille = int(round(write * 1000))
    p.corrupt_permille = int(round(corrupt * 1000))
    cap = 2 * n_ops + 2
    typ = np.zeros(cap, np.uint8)
    proc = np.zeros(cap, np.int32)
    f = np.zeros(cap, np.uint8)
    a = np.zeros(cap, np.int32)
    b = np.zeros(cap, np.int32)
    rows = C.c_uint32(0)
    rc = N.lib().tbs_gen_register(C.byref(p), _p(typ, C.c_uint8), _p(proc, C.c_int32), _p(f, C.c_uint8),
                                  _p(a, C.c_int32), _p(b, C.c_int32), C.byref(rows))
    if rc != 0:
        raise ValueError(f"tbs_gen_register rc={rc}")
    k = rows.value
    return EventColumns(typ[:k].copy(), proc[:k].copy(), f[:k].copy(), a[:k].copy(), b[:k].copy())


def register_ops_many(seeds, workers=None, **kw):
    """pair_events(register_events(seed=s, **kw)) for every seed.  The generator and the pairing
    step are C code called through ctypes (the GIL is released), so a thread pool scales."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    from .columns import pair_events
    seeds = [int(s) for s in seeds]
    workers = workers or min(32, os.cpu_count() or 1)

    def one(seed):
        return pair_events(register_events(seed=seed, **kw))

    if workers <= 1 or len(seeds) < 64:
        return [one(s) for s in seeds]
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(one, seeds))
