"""The sharded sweep's exchange behind the C-ABI (include/tbcheck.h tbc_comm_*, tbc_batch_sweep_allgather; csrc/tbc_comm.hip): what a
host without torch.distributed -- the reference's host is a Clojure process, /root/reference/project.clj:6-8 -- calls to run its rank's
share of ONE history checked by several GPUs.  This pool has one GPU a box, so:
  * the CALLER'S transport (tbc_comm_init_host) is driven by two PROCESSES, both on the one GPU, whose all-gather runs over gloo:
    each rank sweeps its half of the wavefronts, the tables meet once, every rank composes -- verdict, failing op and the sweep's own
    counters equal the single-GPU run's, which equal oracle/sweep_ref.c's;
  * the RCCL transport (tbc_comm_unique_id / tbc_comm_init) runs at world 1: librccl is loaded by the library itself, the communicator
    is made, the one ncclAllGather of the relation table runs out of HBM into HBM and the device-side OR-merge composes it."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from jepsen_tigerbeetle_amd import _native as N, columns, core, shard, synth

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [dict(n_ops=2000, n_procs=32, seed=3, busy=0.15), dict(n_ops=2000, n_procs=32, seed=4, busy=0.15, corrupt=0.5),
         dict(n_ops=600, n_procs=8, seed=5, busy=0.3, corrupt=0.3), dict(n_ops=3000, n_procs=64, seed=6, busy=0.1)]
KEYS = ("valid", "fail_op", "prev_ok_op", "analyzer", "probes", "visited", "backtracks", "max_depth", "final_state")


def gm():
    return core.make_model(N.MODEL_CAS_REGISTER, N.NIL)


def opts():
    return core.make_opts(algorithm=N.ALG_LINEAR, want_witness=False, time_limit_ms=60000)


def hists():
    return [columns.pair_events(synth.register_events(**c)) for c in CASES]


def alone():
    out = []
    for h in hists():
        with core.Batch([h], gm(), opts()) as b:
            out.append(tuple(b.run().results()[0][k] for k in KEYS))
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    calls = []

    def allgather(send):          # the caller's transport: host memory in, host memory out
        t = torch.from_numpy(np.ascontiguousarray(send))
        out = torch.empty(world * t.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, t)
        calls.append(int(t.numel()))
        return out.numpy()

    rows = []
    with shard.Comm.host(rank, world, allgather) as comm:
        for h in hists():
            with core.Batch([h], gm(), opts()) as b:
                r = comm.check(b)[0]
                rows.append([(-7 if r[k] is None else int(r[k])) for k in KEYS])
    assert len(calls) == len(CASES) and all(c > 0 for c in calls)          # ONE exchange a history
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array(rows, np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_share_one_history_through_the_c_entry_points(tmp_path, native, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0, r1)                                           # every rank holds the same answer
    single = alone()
    for i, (row, one, c) in enumerate(zip(r0, single, CASES)):
        exp = [(-7 if v is None else int(v)) for v in one]
        assert list(row) == exp, (i, list(row), exp)                        # ... which is the single-GPU run's, counter by counter
        ops = columns.pair_events(synth.register_events(**c)).as_dict()
        seq = oracle.check(ops, {"kind": 1, "init": N.NIL}, "window", want_witness=False)
        assert row[0] == seq["valid"] and (seq["valid"] == 1 or row[1] == seq["fail_op"]), i


def test_rccl_transport_at_world_one(native):
    """librccl.so loaded by the library, an ncclUniqueId through the C-ABI, ncclCommInitRank, the all-gather out of HBM, the OR-merge."""
    ident = shard.Comm.unique_id()
    assert len(ident) == N.COMM_ID_BYTES and any(ident)
    single = alone()
    with shard.Comm.rccl(0, 1, ident, 0) as comm:
        for h, one in zip(hists(), single):
            with core.Batch([h], gm(), opts()) as b:
                r = comm.check(b)[0]
                assert tuple(r[k] for k in KEYS) == one
                again = comm.check(b)[0]                                    # the communicator and its gather buffer are re-used
                assert tuple(again[k] for k in KEYS) == one
    # a batch that does not run the level sweep has no relation table to exchange
    with shard.Comm.rccl(0, 1, shard.Comm.unique_id(), 0) as comm:
        with core.Batch(hists()[:1], gm(), core.make_opts(algorithm=N.ALG_WGL)) as b:
            with pytest.raises(N.TbcError):
                comm.check(b)
