"""Thin Python over the C-ABI compute entry points: `check_ops` (tbc_check) and
`Batch` (tbc_batch_*).  Nothing here computes a verdict: it marshals columns to
the library and verdict structs back.  No GPU => NoDeviceError."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from . import _native as N
from .columns import OpColumns, _p


def make_model(kind, init=N.NIL, table=None, n_keys=0):
    """-> (N.Model, keepalive).  `table`: (n_states, n_classes) uint16 for MODEL_TABLE."""
    m = N.Model()
    m.kind = kind
    m.init = init
    m.n_keys = n_keys
    keep = None
    if table is not None:
        t = np.ascontiguousarray(table, np.uint16)
        m.table = _p(t, C.c_uint16)
        m.n_states, m.n_classes = t.shape
        keep = t
    return m, keep


# tbc_opts.dominance, TBC_DOM_NO_COUNT_FORM: the library's default is the count form (crashed calls as counts per effect class)
# wherever it applies; tests that pin the mask-form schedules against their oracles switch this off (tests/conftest.py)
DEFAULT_COUNT_FORM = True
# tbc_opts.list_order when make_opts() is not told: 0 = the library's choice (tests/conftest.py pins slot order for the tests that compare with the slot-order oracle)
DEFAULT_LIST_ORDER = 0


def make_opts(algorithm=N.ALG_WGL, device=0, time_limit_ms=0, max_steps=0, max_visited_bytes=0,
              want_witness=True, visited_per_op=0, search_width=0, round_budget=0, lookahead=True,
              eager_reads=True, twin_rule=True, lanes_per_history=0, count_form=None, lazy_commuting=True, list_order=None, stall_handover=False, order_restarts=True, eager_txns=True, txn_independence=True):
    o = N.Opts()
    o.algorithm = algorithm
    o.device = device
    o.time_limit_ms = int(time_limit_ms)
    o.max_steps = int(max_steps)
    o.max_visited_bytes = int(max_visited_bytes)
    o.want_witness = 1 if want_witness else 0
    o.visited_per_op = int(visited_per_op)
    o.search_width = int(search_width)
    o.round_budget = int(round_budget)
    o.lookahead = 0 if lookahead else 1     # C-ABI: 0 = on (default), 1 = off
    o.dominance = (0 if eager_reads else N.DOM_NO_EAGER_READS) | (0 if twin_rule else N.DOM_NO_TWIN_RULE) | (0 if (DEFAULT_COUNT_FORM if count_form is None else count_form) else N.DOM_NO_COUNT_FORM) | (0 if lazy_commuting else N.DOM_NO_LAZY_COMMUTING) | (N.DOM_STALL_HANDOVER if stall_handover else 0) | (0 if order_restarts else N.DOM_NO_ORDER_RESTARTS) | (0 if eager_txns else N.DOM_NO_EAGER_TXNS) | (0 if txn_independence else N.DOM_NO_TXN_INDEPENDENCE)      # (stall_handover: the one opt-in bit, tbcheck.h)
    o.lanes_per_history = int(lanes_per_history)     # 8 / 16 / 32: several histories per wavefront; 64: one; 0: the library's choice
    o.list_order = int(DEFAULT_LIST_ORDER if list_order is None else list_order)      # N.ORDER_*: 0 = the library's choice (completion order, a :write 24 ranks later, where it applies)
    return o


def _result_dict(r: N.Result, copy_witness=True):
    d = {
        "valid": r.valid, "cause": r.cause, "analyzer": r.analyzer,
        "fail_op": None if r.fail_op == N.NO_OP else r.fail_op,
        "prev_ok_op": None if r.prev_ok_op == N.NO_OP else r.prev_ok_op,
        "final_state": r.final_state, "n_witness": r.n_witness, "witness": None, "search_width": r.search_width,
        "configs": [],
    }
    if r.valid == N.VALID and bool(r.witness) and copy_witness:
        d["witness"] = np.ctypeslib.as_array(r.witness, shape=(max(r.n_witness, 1),))[:r.n_witness].copy()
    for i in range(r.n_configs):
        c = r.configs[i]
        d["configs"].append({"state": c.state, "last_op": None if c.last_op == N.NO_OP else c.last_op,
                             "pending": list(c.pending[:min(c.n_pending, 16)]),
                             "n_pending": c.n_pending, "linearized_mask": c.linearized_mask})
    for k, _ in N.Counters._fields_:
        d[k] = getattr(r.counters, k)
    return d


def check_ops(ops: OpColumns, model, opts=None):
    """tbc_check: one history, host columns in, verdict dict out."""
    m, keep = model if isinstance(model, tuple) else (model, None)
    o = opts or make_opts()
    s = ops.struct()
    r = N.Result()
    st = N.lib().tbc_check(C.byref(s), C.byref(m), C.byref(o), C.byref(r))
    N.check_status(st)
    try:
        return _result_dict(r)
    finally:
        N.lib().tbc_result_free(C.byref(r))
        del keep


class Batch:
    """tbc_batch_*: many independent histories resident in HBM, one launch per run."""

    def __init__(self, histories: Sequence[OpColumns], model, opts=None):
        self._m, self._keep = model if isinstance(model, tuple) else (model, None)
        self._o = opts or make_opts()
        nh = len(histories)
        self.n_hist = nh
        self.op_off = np.zeros(nh + 1, np.uint64)
        for i, h in enumerate(histories):
            self.op_off[i + 1] = self.op_off[i] + len(h)
        self.n_events = np.array([h.n_events for h in histories], np.uint32)
        self.n_process = np.array([h.n_process for h in histories], np.uint32)
        cat = lambda name, dt: (np.concatenate([getattr(h, name) for h in histories]).astype(dt, copy=False)
                                if nh else np.zeros(0, dt))
        spec = (("f", np.uint8), ("a", np.int32), ("b", np.int32), ("process", np.int32), ("inv_pos", np.uint32), ("ret_pos", np.uint32))
        if nh >= 1024:          # a big batch: the six columns side by side (numpy copies outside the GIL; 32,768 histories: 0.6 s -> a quarter)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(6) as ex:
                cols = list(ex.map(lambda nd: cat(*nd), spec))
        else:
            cols = [cat(*nd) for nd in spec]
        self._cols = OpColumns(*cols, n_events=0, n_process=0)
        self._aux = None
        if any(h.pool is not None and len(h.pool) for h in histories):
            # one pool for the batch: shift the pool offsets held in column `a` (txn micro-ops; set / bank
            # reads and transfers) and give every history the offset of its own per-front table
            kind = self._m.kind
            pools, shift, a_cols, aux = [], 0, [], []
            for h in histories:
                p = h.pool if h.pool is not None else np.zeros(0, np.int32)
                if kind == N.MODEL_MULTI_REGISTER:
                    in_pool = h.f == N.F_TXN
                elif kind == N.MODEL_SET:
                    in_pool = (h.f == N.F_READ) & (h.a != N.NIL)
                elif kind == N.MODEL_BANK:
                    in_pool = (h.f == N.F_TRANSFER) | ((h.f == N.F_READ) & (h.a != N.NIL))
                else:
                    in_pool = np.zeros(len(h), bool)
                a_cols.append(np.where(in_pool, h.a + shift, h.a).astype(np.int32))
                aux.append(shift + int(self._m.init) if kind in (N.MODEL_SET, N.MODEL_BANK) else int(self._m.init))
                pools.append(np.asarray(p, np.int32))
                shift += len(p)
            self._cols.a = np.concatenate(a_cols)
            self._cols.pool = np.ascontiguousarray(np.concatenate(pools))
            self._aux = np.array(aux, np.int32)
        for name in ("f", "a", "b", "process", "inv_pos", "ret_pos"):
            setattr(self._cols, name, np.ascontiguousarray(getattr(self._cols, name)))
        d = N.BatchDesc()
        d.n_hist = nh
        d.op_off = _p(self.op_off, C.c_uint64)
        d.n_events = _p(self.n_events, C.c_uint32)
        d.n_process = _p(self.n_process, C.c_uint32)
        d.cols = self._cols.struct()
        d.model_aux = _p(self._aux, C.c_int32) if self._aux is not None else None
        self._h = C.c_void_p()
        import time
        t0 = time.perf_counter()
        st = N.lib().tbc_batch_create(C.byref(d), C.byref(self._m), C.byref(self._o), C.byref(self._h))
        self.create_s = time.perf_counter() - t0          # tbc_batch_create alone: arenas, H2D of the op columns, list sizing
        N.check_status(st)
        self._res = (N.Result * nh)()

    @property
    def total_ops(self):
        return int(self.op_off[-1])

    def run(self, want_results=True, tolerate_bad_histories=False):
        """tolerate_bad_histories: a history the device-side validation rejects (TBC_ERR_BAD_HISTORY / TBC_ERR_MODEL)
        comes back as :unknown with cause 0 instead of failing the whole batch.
        A submitted input that is waiting (submit_input / reload) is consumed by this run: the results are ITS histories'."""
        pend = getattr(self, "_pending_n", None)
        if pend:
            self.n_hist = pend.pop(0)
        st = N.lib().tbc_batch_run(self._h, self._res if want_results else None)
        if not (tolerate_bad_histories and st in (N.ERR_BAD_HISTORY, N.ERR_MODEL)):
            N.check_status(st)
        return self

    # ---- fresh inputs: the same batch, new histories (include/tbcheck.h "streaming"; csrc/batch_stream.hip)
    def map_input(self, slot):
        """Slot `slot` of the batch's pinned host memory as numpy views the caller fills IN PLACE:
        {"op_off" u64[n_hist_cap + 1], "n_events", "n_process" u32[n_hist_cap], "word", "inv_pos", "ret_pos" u32[ops_cap]}.
        Waits until the slot's previous input has left for the device."""
        i = N.BatchInput()
        N.check_status(N.lib().tbc_batch_map_input(self._h, slot, C.byref(i)))
        nh, no = int(i.n_hist_cap), int(i.ops_cap)
        view = lambda p, n: np.ctypeslib.as_array(p, shape=(n,))
        return {"n_hist_cap": nh, "ops_cap": no, "op_off": view(i.op_off, nh + 1), "n_events": view(i.n_events, nh), "n_process": view(i.n_process, nh),
                "word": view(i.word, no), "inv_pos": view(i.inv_pos, no), "ret_pos": view(i.ret_pos, no)}

    @staticmethod
    def wire_words(f, a, b, process):
        """TBC_WIRE_WORD over columns: f | a << 4 | b << 12 | process << 20 (nil = 0xFF); values must be 0..254, processes 0..4095."""
        a8 = np.where(a == N.NIL, N.WIRE_NIL, a).astype(np.uint32)
        b8 = np.where(b == N.NIL, N.WIRE_NIL, np.where(f == N.F_CAS, b, 0)).astype(np.uint32)
        if len(a8) and (int(a8.max()) > 255 or int(b8.max()) > 255 or int(process.max()) > 4095 or int(process.min()) < 0):
            raise ValueError("the wire format holds values 0..254 (or nil) and processes 0..4095")
        return f.astype(np.uint32) | (a8 << np.uint32(4)) | (b8 << np.uint32(12)) | (process.astype(np.uint32) << np.uint32(20))

    def fill_input(self, slot, histories: Sequence[OpColumns]):
        """Write `histories` into slot `slot` in wire format (what a caller's own encoder would do in place).  -> n_hist"""
        m = self.map_input(slot)
        nh = len(histories)
        if nh > m["n_hist_cap"]:
            raise ValueError(f"{nh} histories, the batch's slots hold {m['n_hist_cap']}")
        lens = np.fromiter((len(h) for h in histories), np.uint64, nh)
        T = int(lens.sum())
        if T > m["ops_cap"]:
            raise ValueError(f"{T} ops, the batch's slots hold {m['ops_cap']}")
        m["op_off"][0] = 0
        np.cumsum(lens, out=m["op_off"][1:nh + 1])
        m["n_events"][:nh] = np.fromiter((h.n_events for h in histories), np.uint32, nh)
        m["n_process"][:nh] = np.fromiter((h.n_process for h in histories), np.uint32, nh)
        cat = lambda name: np.concatenate([getattr(h, name) for h in histories]) if nh else np.zeros(0, np.int32)
        m["word"][:T] = self.wire_words(cat("f"), cat("a"), cat("b"), cat("process"))
        m["inv_pos"][:T] = cat("inv_pos")
        m["ret_pos"][:T] = cat("ret_pos")
        return nh

    def submit_input(self, slot, n_hist):
        """Queue the slot's copy to the device (asynchronous); the next run() that finds it waiting consumes it."""
        N.check_status(N.lib().tbc_batch_submit_input(self._h, slot, n_hist))
        if not hasattr(self, "_pending_n"):
            self._pending_n = []
        self._pending_n.append(int(n_hist))
        return self

    def reload(self, histories: Sequence[OpColumns]):
        """tbc_batch_reload: the histories' six columns -> wire format in the batch's next slot -> submitted."""
        nh = len(histories)
        op_off = np.zeros(nh + 1, np.uint64)
        for i, h in enumerate(histories):
            op_off[i + 1] = op_off[i] + len(h)
        n_events = np.array([h.n_events for h in histories], np.uint32)
        n_process = np.array([h.n_process for h in histories], np.uint32)
        spec = (("f", np.uint8), ("a", np.int32), ("b", np.int32), ("process", np.int32), ("inv_pos", np.uint32), ("ret_pos", np.uint32))
        cols = OpColumns(*[np.ascontiguousarray(np.concatenate([getattr(h, name) for h in histories]).astype(dt, copy=False)) for name, dt in spec],
                         n_events=0, n_process=0)
        d = N.BatchDesc()
        d.n_hist = nh
        d.op_off = _p(op_off, C.c_uint64)
        d.n_events = _p(n_events, C.c_uint32)
        d.n_process = _p(n_process, C.c_uint32)
        d.cols = cols.struct()
        d.model_aux = None
        import time as _time
        t0 = _time.perf_counter()
        N.check_status(N.lib().tbc_batch_reload(self._h, C.byref(d)))
        self.reload_call_s = _time.perf_counter() - t0          # (the C call alone: wire encoding, a count-form batch's planning, the copy queued)
        if not hasattr(self, "_pending_n"):
            self._pending_n = []
        self._pending_n.append(nh)
        return self

    def input_info(self):
        i = N.InputInfo()
        N.check_status(N.lib().tbc_batch_input_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in N.InputInfo._fields_}

    def results(self, copy_witness=True):
        return [_result_dict(self._res[i], copy_witness) for i in range(self.n_hist)]

    def verdicts(self):
        return np.array([self._res[i].valid for i in range(self.n_hist)], np.int32)

    def timing_ns(self):
        t = (C.c_uint64 * 4)()
        N.check_status(N.lib().tbc_batch_last_timing(self._h, t))
        w = C.c_uint64(0)
        N.check_status(N.lib().tbc_batch_last_turn_wait(self._h, C.byref(w)))
        return {"init": t[0], "pack": t[1], "search": t[2], "retries": t[3], "turn_wait": w.value}

    def progress(self):
        """How far a run that is out has come (tbc_batch_progress): the one call another thread may make while run() is in flight."""
        p = N.Progress()
        N.check_status(N.lib().tbc_batch_progress(self._h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in N.Progress._fields_}

    def counters(self):
        c = N.Counters()
        N.check_status(N.lib().tbc_batch_last_counters(self._h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in N.Counters._fields_}

    # ---- one history (or a small batch) over several GPUs: jepsen-tigerbeetle_amd/shard.py drives these
    def set_shard(self, rank, world):
        N.check_status(N.lib().tbc_batch_set_shard(self._h, rank, world))

    def sweep_partial(self):
        N.check_status(N.lib().tbc_batch_sweep_partial(self._h))

    def sweep_table(self):
        """(device pointer, bytes) of the relation table of the last sweep_partial()."""
        p, n = C.c_void_p(), C.c_uint64()
        N.check_status(N.lib().tbc_batch_sweep_table(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def sweep_finish(self, merged: np.ndarray):
        m = np.ascontiguousarray(merged, np.uint8)
        st = N.lib().tbc_batch_sweep_finish(self._h, m.ctypes.data_as(C.c_void_p), C.c_uint64(m.nbytes), self._res)
        N.check_status(st)
        return self

    def sweep_table_tensor(self):
        """The relation table of the last sweep_partial() as a torch uint8 tensor over the library's own HBM (no copy)."""
        import torch
        from .shard import _DeviceBytes
        ptr, nbytes = self.sweep_table()
        return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=torch.device("cuda", torch.cuda.current_device()))

    def sweep_merge(self, gathered, world):
        """`gathered`: a torch CUDA uint8 tensor holding the `world` ranks' tables back to back (all_gather_into_tensor):
        OR-ed on the device, composed; the exchanged bytes never pass through the host."""
        # the producer of `gathered` (an all_gather_into_tensor, a torch.cat) was only ENQUEUED on torch's current stream; the library
        # reads the buffer on the batch's own non-blocking stream, which has no ordering against it: wait for the producer first
        import torch
        if gathered.is_cuda:
            torch.cuda.current_stream(gathered.device).synchronize()
        st = N.lib().tbc_batch_sweep_merge(self._h, C.c_void_p(gathered.data_ptr()), C.c_uint64(gathered.numel() * gathered.element_size()),
                                           C.c_uint32(world), self._res)
        N.check_status(st)
        return self

    def sweep_info(self):
        i = N.SweepInfo()
        N.check_status(N.lib().tbc_batch_sweep_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in N.SweepInfo._fields_}

    def device_bytes(self):
        return int(N.lib().tbc_batch_device_bytes(self._h))

    def search_width(self):
        """Configs per round of the depth-first search (what search_width=0 resolved to for this batch)."""
        return int(N.lib().tbc_batch_search_width(self._h))

    def lanes_per_history(self):
        """8 / 16 / 32 when several histories share a wavefront (one config per iteration), else 64."""
        return int(N.lib().tbc_batch_lanes_per_history(self._h))

    def list_order(self):
        """N.ORDER_SLOT / _COMPLETION / _WRITES_LAST or 16 + W: the order of the fronts' lists this batch's search runs over."""
        return int(N.lib().tbc_batch_list_order(self._h))

    def last_raced(self):
        """Histories of the last run that were searched in several list orders at once (the first pass ended at its budget)."""
        return int(N.lib().tbc_batch_last_raced(self._h))

    def close(self):
        if self._h:
            N.lib().tbc_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
