#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on MI355X.

  metric   : histories/sec (+ time-to-verdict, ms) on synthetic 10k-op / 64-process
             cas-register histories  (BASELINE.json `metric`, configs[1])
  step     : one pass of the hot path (pack kernel + WGL search kernel + verdict
             read-back) over one batch of B independent histories per GPU whose
             op columns are already resident in HBM (C-ABI tbc_batch_run)
  N > 1    : one process per GPU (torch.distributed / RCCL for the barrier and the
             max-over-ranks clock only); histories are independent units, so they
             are sharded across ranks with NO data-path collective -- weak scaling
             (B per GPU fixed)
  roofline : the search kernel (wgl_beam_kernel, or wgl_search_kernel at --width 1), HBM bound.  achieved = algorithmic bytes per launch
             (16 B per visited-set probe that finds a duplicate,
             32 B per probe that inserts a new config) / the kernel's average
             duration, measured with HIP events on the library's own stream
  cpu_baseline : the CPU restatement of the same search (oracle/wgl_window.c,
             "port", 1 thread) on a bounded sample of the same histories

Usage: python bench.py --gpus N --steps K --warmup W   (N>1 via torch.distributed.run)
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
WORKLOAD = "cas-register, 10k ops (invocations) / 64 processes, values 0..4, r/w/cas 1/3 each"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("TBC_BENCH_BATCH", "32768")),
                    help="histories per GPU per step")
    ap.add_argument("--ops", type=int, default=10000)
    ap.add_argument("--procs", type=int, default=64)
    ap.add_argument("--busy", type=float, default=0.1,
                    help="fraction of time a process has an op open (64 x 0.1 = 6.4 ops in flight on average)")
    ap.add_argument("--info", type=float, default=0.0, help="crashed-op (:info) rate")
    ap.add_argument("--width", type=int, default=int(os.environ.get("TBC_BENCH_WIDTH", "4")),
                    help="configs expanded per iteration: 1 = sequential knossos.wgl order, 2..16 = wide schedule")
    ap.add_argument("--visited-per-op", type=int, default=16, help="first visited-set capacity per op (0 = library default 64)")
    ap.add_argument("--round-budget", type=int, default=0,
                    help="a history that has used more rounds than this continues at width 16 (0 = off)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="histories timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libtbcheck has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import jepsen_tigerbeetle_amd  # noqa: F401
    from jepsen_tigerbeetle_amd import _native as N, columns, core, shard, synth

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic input: B distinct seeded histories per rank
    B = args.batch
    t_gen = time.time()
    seeds = shard.shard_indices(B * world, rank, world)      # history i of the job lives on rank i % world
    hists = synth.register_ops_many(seeds, n_ops=args.ops, n_procs=args.procs, busy=args.busy, info=args.info)
    t_gen = time.time() - t_gen
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    opts = core.make_opts(device=local_rank, time_limit_ms=600000, want_witness=False,
                          algorithm=N.ALG_COMPETITION, search_width=args.width, visited_per_op=args.visited_per_op,
                          round_budget=args.round_budget)
    batch = core.Batch(hists, model, opts)        # H2D happens here: inputs resident before timing

    for _ in range(args.warmup):
        batch.run()
    barrier()
    t0 = time.perf_counter()
    search_ns, pack_ns, init_ns, retry_ns = [], [], [], []
    for _ in range(args.steps):
        batch.run()
        tm = batch.timing_ns()
        search_ns.append(tm["search"]); pack_ns.append(tm["pack"]); init_ns.append(tm["init"]); retry_ns.append(tm["retries"])
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, world, dist)

    verdicts = batch.verdicts()
    counters = batch.counters()
    n_valid = int((verdicts == N.VALID).sum())
    n_unknown = int((verdicts == N.UNKNOWN).sum())
    if world > 1:
        t = torch.tensor([n_valid, n_unknown], dtype=torch.int64, device="cuda")
        dist.all_reduce(t)
        n_valid, n_unknown = int(t[0]), int(t[1])

    if rank == 0:
        total_hist = B * world * args.steps
        value = total_hist / elapsed
        # roofline of the dominant kernel (this rank's launches)
        dup = counters["probes"] - counters["visited"]
        alg_bytes = 16 * dup + 32 * counters["visited"]
        k_ms = statistics.mean(search_ns) / 1e6
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        try:   # HBM bytes per launch from the committed rocprofv3 PMC passes, when they cover this very config
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as fh:
                for e in json.load(fh)["entries"]:
                    key = (e["histories_per_gpu"], e["search_width"], e["visited_per_op"], e["ops"], e["procs"], e["busy"], e["info"])
                    if not e.get("retired") and key == (B, args.width, args.visited_per_op, args.ops, args.procs, args.busy, args.info):
                        traffic = e["traffic_bytes"]
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "histories/sec, 10k-op/64-proc cas-register histories (time-to-verdict ms in extra)",
            "value": round(value, 2), "unit": "histories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": WORKLOAD, "histories_per_gpu": B, "ops_after_pairing": int(batch.total_ops // B),
                       "processes": args.procs, "busy": args.busy, "info_rate": args.info,
                       "search_width": args.width, "round_budget": args.round_budget,
                       "parallelism": f"independent histories sharded over {world} GPU(s), no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": "wgl_search_kernel" if args.width == 1 else "wgl_beam_kernel", "kernel_ms": round(k_ms, 3),
                         "probes_per_launch": counters["probes"], "new_configs_per_launch": counters["visited"],
                         "algorithmic_bytes_per_launch": alg_bytes},
            "extra": {"valid": n_valid, "unknown": n_unknown,
                      "device_ms": {"init_memsets": round(statistics.mean(init_ns) / 1e6, 3),
                                    "pack": round(statistics.mean(pack_ns) / 1e6, 3),
                                    "search": round(k_ms, 3), "retries": round(statistics.mean(retry_ns) / 1e6, 3)},
                      "steps_per_history": counters["steps"] / B,
                      "device_GB": round(batch.device_bytes() / 1e9, 3), "gen_s": round(t_gen, 2)},
        }
        # time-to-verdict for ONE history through tbc_check (H2D + kernels + D2H), rank 0
        ttv = []
        for i in range(min(5, B)):
            r = core.check_ops(hists[i], model, core.make_opts(device=local_rank, want_witness=True,
                                                               algorithm=N.ALG_COMPETITION, search_width=args.width))
            ttv.append(r["ns_total"] / 1e6)
        bad = columns.pair_events(synth.register_events(n_ops=args.ops, n_procs=args.procs, seed=12345, busy=args.busy / 2,
                                                        info=args.info, corrupt=0.7))
        rb = core.check_ops(bad, model, core.make_opts(device=local_rank, time_limit_ms=120000,
                                                       algorithm=N.ALG_COMPETITION, search_width=args.width))
        line["extra"]["time_to_verdict_ms"] = {"valid_median": round(statistics.median(ttv), 3),
                                               "invalid_example": round(rb["ns_total"] / 1e6, 3),
                                               "invalid_example_verdict": rb["valid"], "invalid_example_steps": rb["steps"]}
        if world == 1 and not args.no_cpu:
            from oracle import wgl
            S = min(args.cpu_sample, B)
            om = {"kind": 1, "init": N.NIL}
            tc = time.perf_counter()
            ok = 0
            for i in range(S):
                ok += wgl.check(hists[i].as_dict(), om, "window", want_witness=False)["valid"] == 1
            tc = time.perf_counter() - tc
            tb = time.perf_counter()
            rbo = wgl.check(bad.as_dict(), om, "window", want_witness=False)
            tb = time.perf_counter() - tb
            # the SAME schedule the kernel runs (wide, K configs per iteration, lookahead), on one host thread:
            # how much of the speed-up is the algorithm and how much the GPU
            S2 = min(64, S)
            tw = time.perf_counter()
            okw = 0
            for i in range(S2):
                okw += wgl.check_beam(hists[i].as_dict(), om, args.width if args.width > 1 else 4, want_witness=False)["valid"] == 1
            tw = time.perf_counter() - tw
            line["cpu_baseline"] = {"value": round(S / tc, 3), "unit": "histories/s", "cores": 1, "kind": "port",
                                    "sample": f"first {S} histories of this batch, oracle/wgl_window.c (C, gcc -O2), 1 thread; "
                                              f"not stock Knossos (no JVM here)",
                                    "ms_per_history": round(tc / S * 1e3, 3),
                                    "invalid_example_ms": round(tb * 1e3, 3), "invalid_example_verdict": rbo["valid"],
                                    "same_schedule_as_kernel": {"value": round(S2 / tw, 3), "unit": "histories/s", "cores": 1,
                                                                "sample": f"first {S2} histories, oracle/wgl_beam.c with lookahead"},
                                    "host_cores_available": os.cpu_count()}
            assert okw == sum(int(v == N.VALID) for v in verdicts[:S2]), "GPU and wide oracle disagree on the sample"
            assert ok == sum(int(v == N.VALID) for v in verdicts[:S]), "GPU and oracle disagree on the sample"
        print(json.dumps(line), flush=True)
    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
