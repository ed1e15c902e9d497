"""Multi-GPU layouts.

Batch path: histories (and jepsen.independent keys) are independent units, so rank r of N simply owns every
N-th history -- no data-path collective; torch.distributed (RCCL on the GPU box, gloo in the CPU tests) only
agrees on the verdict summary and on the max-over-ranks clock.

One history over several GPUs (BASELINE.json's north_star: "the search frontier shards across the GPUs"):
the level sweep (csrc/jit_sweep.hip) cuts a history into segments whose wavefronts are independent; rank r
sweeps every N-th wavefront, ONE all-gather of the relation tables (a few hundred KB, straight out of HBM over
RCCL / xGMI) gives every rank the whole chain, and every rank composes it (`check_sharded`).  Unlike a
hash-partitioned visited set with an all-to-all per search level (>= 10^4 dependent
collectives of tens of microseconds each), the exchange happens once."""
from __future__ import annotations

import numpy as np


def shard_indices(n_items: int, rank: int, world: int) -> np.ndarray:
    """Round-robin: item i belongs to rank i % world (keeps per-rank work balanced
    when neighbouring histories have similar cost)."""
    return np.arange(rank, n_items, world, dtype=np.int64)


def merge_verdicts(local_verdicts: np.ndarray, n_items: int, rank: int, world: int, dist=None) -> np.ndarray:
    """All ranks end up with the full verdict vector (1 valid / 0 invalid / -1 unknown).
    One small all_reduce(SUM) over an int32 vector offset by +2 so that 'not mine' = 0."""
    full = np.zeros(n_items, np.int32)
    full[shard_indices(n_items, rank, world)] = np.asarray(local_verdicts, np.int32) + 2
    if world > 1:
        import torch
        t = torch.from_numpy(full)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        full = t.cpu().numpy()
    return full - 2


def max_over_ranks(seconds: float, world: int, dist=None) -> float:
    if world == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class _DeviceBytes:
    """A view of library-owned device memory that torch can wrap without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def merge_relation_tables(tables) -> np.ndarray:
    """Every (history, segment, slice) record is written by exactly one rank and all zero on the others: the merged
    table is the bitwise OR."""
    out = np.zeros_like(np.asarray(tables[0], np.uint8))
    for t in tables:
        out |= np.asarray(t, np.uint8)
    return out


def gather_relation_tables(local, world: int, dist=None):
    """all_gather of one uint8 table per rank -> list of `world` numpy arrays (on every rank).  `local` is a numpy
    array (gloo / CPU stand-in) or a torch CUDA tensor (RCCL, straight out of the library's HBM table)."""
    import torch
    if world == 1:
        return [local.cpu().numpy() if isinstance(local, torch.Tensor) else np.asarray(local, np.uint8)]
    t = local if isinstance(local, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local, np.uint8))
    if dist.get_backend() == "nccl" and not t.is_cuda:
        t = t.cuda()
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous().view(-1))
    return [x for x in out.view(world, -1).cpu().numpy()]


def check_sharded(batch, rank: int, world: int, dist=None):
    """One small batch (typically ONE history) over `world` GPUs: `batch` is this rank's core.Batch over the SAME
    histories (inputs are replicated), created with algorithm=N.ALG_LINEAR.  Returns batch.results() on every rank.

    The exchange: every rank's relation table is a tensor over the library's own HBM (`sweep_table_tensor`), ONE
    all_gather_into_tensor over RCCL puts the `world` tables back to back in device memory, and the library ORs them on
    the device and composes (`sweep_merge`) -- nothing of the exchange passes through the host.  `batch` only has to
    provide set_shard / sweep_partial / sweep_table_tensor / sweep_merge / results, which is how the CPU test drives this
    very function over gloo with the CPU restatement standing in for the kernels (tests/test_distributed_gloo.py)."""
    import torch
    batch.set_shard(rank, world)
    batch.sweep_partial()
    local = batch.sweep_table_tensor().contiguous().view(-1)
    if world == 1:
        gathered = local
    else:
        gathered = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)
        dist.all_gather_into_tensor(gathered, local)
    batch.sweep_merge(gathered, world)
    return batch.results()


class Comm:
    """tbc_comm_*: the sharded sweep's exchange BEHIND the C-ABI (csrc/tbc_comm.hip) -- what a Clojure host would call through JNA.
    `Comm.rccl(rank, world, ident, device)`: an RCCL communicator inside the library (ident = Comm.unique_id() of rank 0, carried to
    the other ranks by the caller).  `Comm.host(rank, world, allgather)`: the caller's own transport over host memory --
    allgather(send: np.uint8[n]) -> np.uint8[world * n] (the tests hand in one over gloo)."""

    def __init__(self, handle, keep=None):
        self._h, self._keep = handle, keep

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _native as N
        buf = (C.c_uint8 * N.COMM_ID_BYTES)()
        N.check_status(N.lib().tbc_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def rccl(cls, rank, world, ident: bytes, device=0):
        import ctypes as C
        from . import _native as N
        buf = (C.c_uint8 * N.COMM_ID_BYTES).from_buffer_copy(ident)
        h = C.c_void_p()
        N.check_status(N.lib().tbc_comm_init(rank, world, buf, device, C.byref(h)))
        return cls(h)

    @classmethod
    def host(cls, rank, world, allgather):
        import ctypes as C
        from . import _native as N

        def thunk(user, send, recv, nbytes):
            try:
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
                dst[:] = np.asarray(allgather(src), np.uint8).reshape(-1)
                return 0
            except Exception:          # noqa: BLE001 -- an exception must not cross the C boundary; the library reports the code
                return 1
        fn = N.ALLGATHER_FN(thunk)
        h = C.c_void_p()
        N.check_status(N.lib().tbc_comm_init_host(rank, world, fn, None, C.byref(h)))
        return cls(h, keep=fn)

    def check(self, batch):
        """tbc_batch_sweep_allgather: this rank's share of one sharded check; every rank gets the results."""
        from . import _native as N
        N.check_status(N.lib().tbc_batch_sweep_allgather(batch._h, self._h, batch._res))
        return batch.results()

    def close(self):
        if self._h:
            from . import _native as N
            N.lib().tbc_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
