"""N > 1 path on CPU: world_size-2 gloo.  The data path has no collective (histories
are independent); what the ranks share is the verdict summary and the max-over-ranks
clock.  The per-rank checker here is the CPU oracle standing in for the GPU batch call
(this test is about the sharding / reduction logic, not the kernel)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_hist, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jepsen_tigerbeetle_amd  # noqa: F401
    from jepsen_tigerbeetle_amd import _native as N, columns, shard, synth
    from oracle import wgl
    mine = shard.shard_indices(n_hist, rank, world)
    verdicts = []
    for i in mine:
        ops = columns.pair_events(synth.register_events(n_ops=120, n_procs=6, seed=int(i), busy=0.3,
                                                        corrupt=0.5 * (i % 3 == 0)))
        verdicts.append(wgl.check(ops.as_dict(), {"kind": 1, "init": N.NIL}, "window", want_witness=False)["valid"])
    full = shard.merge_verdicts(np.array(verdicts), n_hist, rank, world, dist)
    t = shard.max_over_ranks(1.0 + rank, world, dist)
    np.save(os.path.join(out_dir, f"v{rank}.npy"), full)
    np.save(os.path.join(out_dir, f"t{rank}.npy"), np.array([t]))
    dist.barrier()
    dist.destroy_process_group()


def test_history_sharding_world_size_2(tmp_path, native, oracle):
    from jepsen_tigerbeetle_amd import _native as N, columns, shard, synth
    n_hist, world = 11, 2
    a, b = shard.shard_indices(n_hist, 0, world), shard.shard_indices(n_hist, 1, world)
    assert sorted(np.concatenate([a, b]).tolist()) == list(range(n_hist)) and not set(a) & set(b)
    mp.spawn(_worker, args=(world, _free_port(), n_hist, str(tmp_path)), nprocs=world, join=True)
    v0, v1 = np.load(tmp_path / "v0.npy"), np.load(tmp_path / "v1.npy")
    assert np.array_equal(v0, v1)                                   # every rank holds the full summary
    expect = []
    for i in range(n_hist):
        ops = columns.pair_events(synth.register_events(n_ops=120, n_procs=6, seed=i, busy=0.3, corrupt=0.5 * (i % 3 == 0)))
        expect.append(oracle.check(ops.as_dict(), {"kind": 1, "init": N.NIL}, "window", want_witness=False)["valid"])
    assert v0.tolist() == expect and 0 in expect and 1 in expect
    assert np.load(tmp_path / "t0.npy")[0] == np.load(tmp_path / "t1.npy")[0] == 2.0   # max over ranks
