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


# ---- one history over several GPUs: the level sweep's wavefronts dealt to the ranks, one all-gather of the relation
# tables, every rank composes.  The per-rank sweep here is the CPU restatement writing the very records
# (tbc_sweep_rel) a GPU rank leaves in its table; the exchange is shard.gather_relation_tables over gloo; the
# composition is the library's own host code (tbc_sweep_compose), so everything but the kernel is the product path.
def _sweep_worker(rank, world, port, cases, seg_target, n_dom, max_segs, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    import jepsen_tigerbeetle_amd  # noqa: F401
    from jepsen_tigerbeetle_amd import _native as N, columns, shard, synth
    from oracle import wgl
    out = []
    for c in cases:
        ops = columns.pair_events(synth.register_events(**c)).as_dict()
        mine = wgl.sweep_relations(ops, {"kind": 1, "init": N.NIL}, seg_target, n_dom, max_segs, rank, world, C.sizeof(N.SweepRel))
        merged = shard.merge_relation_tables(shard.gather_relation_tables(mine, world, dist))
        rel = (N.SweepRel * (max_segs * N.SWEEP_SLICES)).from_buffer_copy(merged.tobytes())
        v = N.SweepVerdict()
        n_ret = int((ops["ret_pos"] != N.POS_CRASHED).sum())
        assert N.lib().tbc_sweep_compose(rel, max_segs, n_ret, C.byref(v)) == 0
        own = sum(1 for i in range(max_segs * N.SWEEP_SLICES) if (N.SweepRel * 1).from_buffer_copy(mine[i * C.sizeof(N.SweepRel):(i + 1) * C.sizeof(N.SweepRel)].tobytes())[0].status)
        out.append([v.valid, v.fail_level, v.n_wavefronts, own, v.probes, v.configs_total])
    np.save(os.path.join(out_dir, f"s{rank}.npy"), np.array(out, np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_one_history_sharded_over_two_ranks(tmp_path, native, oracle):
    from jepsen_tigerbeetle_amd import _native as N, columns, synth
    cases = [dict(n_ops=2000, n_procs=32, seed=3, busy=0.15), dict(n_ops=2000, n_procs=32, seed=4, busy=0.15, corrupt=0.5),
             dict(n_ops=600, n_procs=8, seed=5, busy=0.3, corrupt=0.3), dict(n_ops=3000, n_procs=64, seed=6, busy=0.1)]
    seg_target, n_dom, max_segs, world = 16, 6 + 8, 200, 2          # n_dom: nil + 0..12 (the planted impossible value is 12)
    mp.spawn(_sweep_worker, args=(world, _free_port(), cases, seg_target, n_dom, max_segs, str(tmp_path)), nprocs=world, join=True)
    s0, s1 = np.load(tmp_path / "s0.npy"), np.load(tmp_path / "s1.npy")
    assert np.array_equal(s0[:, [0, 1, 2, 4, 5]], s1[:, [0, 1, 2, 4, 5]])          # every rank reaches the same verdict
    for c, row0, row1 in zip(cases, s0, s1):
        ops = columns.pair_events(synth.register_events(**c)).as_dict()
        ref = oracle.check_sweep(ops, {"kind": 1, "init": N.NIL}, seg_target=seg_target, n_dom=n_dom)
        seq = oracle.check(ops, {"kind": 1, "init": N.NIL}, "window", want_witness=False)
        assert row0[0] == ref["valid"] == seq["valid"]
        if ref["valid"] == 0:
            rets = np.sort(ops["ret_pos"][ops["ret_pos"] != N.POS_CRASHED])
            assert int(ops["ret_pos"][ref["fail_op"]]) == int(rets[row0[1]])       # the failing completion, by rank
        assert row0[3] > 0 and row1[3] > 0                                         # the wavefronts were really split ...
        if ref["valid"] == 1:
            assert row0[2] == row0[3] + row1[3]                                    # ... and all of them composed
        assert (row0[4], row0[5]) == (ref["probes"], ref["configs_total"])         # nothing swept twice, nothing lost


# ---- shard.check_sharded itself over gloo: the function a multi-GPU caller runs (set_shard -> sweep_partial -> table tensor ->
# ONE all_gather_into_tensor -> merge -> results), with a stand-in batch whose "kernels" are the CPU restatement and whose merge
# is the bitwise OR the device kernel does + the library's own tbc_sweep_compose.  What a GPU box adds is only where the bytes live.
class _CpuSweepBatch:
    def __init__(self, ops, seg_target, n_dom, max_segs):
        self.ops, self.seg_target, self.n_dom, self.max_segs = ops, seg_target, n_dom, max_segs
        self.rank, self.world, self._table, self._res = 0, 1, None, None

    def set_shard(self, rank, world):
        self.rank, self.world = rank, world

    def sweep_partial(self):
        import ctypes as C
        from jepsen_tigerbeetle_amd import _native as N
        from oracle import wgl
        self._table = wgl.sweep_relations(self.ops, {"kind": 1, "init": N.NIL}, self.seg_target, self.n_dom, self.max_segs, self.rank, self.world, C.sizeof(N.SweepRel))

    def sweep_table_tensor(self):
        import torch
        return torch.from_numpy(np.ascontiguousarray(self._table, np.uint8))

    def sweep_merge(self, gathered, world):
        import ctypes as C
        from jepsen_tigerbeetle_amd import _native as N
        merged = np.bitwise_or.reduce(gathered.numpy().reshape(world, -1), axis=0)          # what sweep_or_kernel does in HBM
        rel = (N.SweepRel * (self.max_segs * N.SWEEP_SLICES)).from_buffer_copy(merged.tobytes())
        v = N.SweepVerdict()
        n_ret = int((self.ops["ret_pos"] != N.POS_CRASHED).sum())
        assert N.lib().tbc_sweep_compose(rel, self.max_segs, n_ret, C.byref(v)) == 0
        self._res = [{"valid": v.valid, "fail_level": v.fail_level, "probes": v.probes, "configs_total": v.configs_total}]
        return self

    def results(self):
        return self._res


def _check_sharded_worker(rank, world, port, cases, seg_target, n_dom, max_segs, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import jepsen_tigerbeetle_amd  # noqa: F401
    from jepsen_tigerbeetle_amd import columns, shard, synth
    out = []
    for c in cases:
        ops = columns.pair_events(synth.register_events(**c)).as_dict()
        r = shard.check_sharded(_CpuSweepBatch(ops, seg_target, n_dom, max_segs), rank, world, dist)[0]
        out.append([r["valid"], r["fail_level"], r["probes"], r["configs_total"]])
    np.save(os.path.join(out_dir, f"c{rank}.npy"), np.array(out, np.int64))
    dist.barrier()
    dist.destroy_process_group()


def test_check_sharded_code_path_world_size_2(tmp_path, native, oracle):
    from jepsen_tigerbeetle_amd import _native as N, columns, synth
    cases = [dict(n_ops=1500, n_procs=32, seed=13, busy=0.15), dict(n_ops=1500, n_procs=32, seed=14, busy=0.15, corrupt=0.5)]
    seg_target, n_dom, max_segs, world = 16, 14, 160, 2
    mp.spawn(_check_sharded_worker, args=(world, _free_port(), cases, seg_target, n_dom, max_segs, str(tmp_path)), nprocs=world, join=True)
    c0, c1 = np.load(tmp_path / "c0.npy"), np.load(tmp_path / "c1.npy")
    assert np.array_equal(c0, c1)
    for c, row in zip(cases, c0):
        ops = columns.pair_events(synth.register_events(**c)).as_dict()
        ref = oracle.check_sweep(ops, {"kind": 1, "init": N.NIL}, seg_target=seg_target, n_dom=n_dom)
        assert row[0] == ref["valid"] and (row[2], row[3]) == (ref["probes"], ref["configs_total"])


class StandInBatch:
    """core.Batch's surface as bench.py uses it, without a device: every history VALID except the planted one (bench.py plants an
    impossible read in the last history of every resident batch).  For the plumbing test below only."""

    def __init__(self, histories, model, opts):
        self.n_hist = len(histories)
        self.total_ops = sum(len(h) for h in histories)
        self._v = np.ones(self.n_hist, np.int32)
        self._v[-1] = 0

    def run(self):
        return self

    def verdicts(self):
        return self._v

    def timing_ns(self):
        return {"init": 1000, "pack": 2000, "search": 3000, "retries": 0, "turn_wait": 0}

    def counters(self):
        return {"steps": 10 * self.n_hist, "visited": 8 * self.n_hist, "probes": 10 * self.n_hist, "backtracks": 0, "max_depth": 1,
                "table_slots": 0, "ns_pack": 0, "ns_search": 0, "ns_total": 0}

    def search_width(self):
        return 1

    def lanes_per_history(self):
        return 8

    def device_bytes(self):
        return 0

    def close(self):
        pass


def test_bench_n_gt_1_plumbing_over_gloo(native):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), over gloo with a stand-in for
    the device batch: the sharding of the seeds, the barriers, the max-over-ranks clock, the verdict all-reduce and the ONE JSON
    line from rank 0 are the code a multi-GPU node will run."""
    import json
    import subprocess
    env = dict(os.environ, TBC_BENCH_BACKEND="gloo", TBC_BENCH_BATCH_CLASS="test_distributed_gloo:StandInBatch",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, os.environ.get("PYTHONPATH", "")]))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--batch", "6", "--ops", "300", "--procs", "8", "--in-flight", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 1 and line["scaling"] == "weak" and line["data"] == "stand-in"
    assert line["config"]["histories_per_gpu"] == 6 and line["config"]["batches_in_flight"] == 2
    # whole-job aggregate: 2 ranks x 6 histories per step; verdicts summed over the ranks' resident batches (one planted INVALID each)
    assert abs(line["value"] - 2 * 6 * 4 / (line["ms_per_step"] * 4 / 1e3)) < 0.01 * line["value"]      # (ms_per_step is rounded to a microsecond)
    assert line["extra"]["valid"] == 2 * 2 * 5 and line["extra"]["unknown"] == 0


def test_bench_extra_legs_run_in_their_own_process(monkeypatch):
    """bench.run_leg: a leg's result comes back from its process; a leg killed by a signal (what a GPU fault does to the process it
    happens in) leaves an error in its place and the measured line lives; a leg that fails an assertion fails the run."""
    import argparse
    import json
    import subprocess
    import types
    sys.path.insert(0, ROOT)
    import bench
    seen = []

    def fake_run(rc, out):
        def run(cmd, **kw):
            seen.append(cmd)
            return types.SimpleNamespace(returncode=rc, stdout=out)
        return run
    monkeypatch.delenv("TBC_BENCH_INLINE_LEGS", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "4"])
    args = argparse.Namespace()
    monkeypatch.setattr(subprocess, "run", fake_run(0, b"noise\n" + json.dumps({"leg": "set_full", "result": {"scan_ms": 1.5}}).encode() + b"\n"))
    assert bench.run_leg("set_full", args, 0) == {"scan_ms": 1.5}
    assert seen[-1][-4:] == ["--steps", "4", "--leg", "set_full"]          # the same arguments, and the leg's name
    monkeypatch.setattr(subprocess, "run", fake_run(-6, b""))
    r = bench.run_leg("workload_crashed", args, 0)
    assert set(r) == {"error"} and "signal 6" in r["error"]
    monkeypatch.setattr(subprocess, "run", fake_run(1, b""))
    with pytest.raises(SystemExit):
        bench.run_leg("tiers", args, 0)
    assert set(bench.LEGS) >= {"tiers", "set_full", "workload_2", "workload_3", "workload_crashed"}
