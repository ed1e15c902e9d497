#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on MI355X.

  metric   : histories/sec (+ time-to-verdict, ms) on synthetic 10k-op / 64-process
             cas-register histories  (BASELINE.json `metric`, configs[1])
  step     : one pass of the hot path (pack kernel + WGL search kernel + verdict
             read-back) over one batch of B independent histories per GPU whose
             op columns are already resident in HBM (C-ABI tbc_batch_run).  --in-flight F (default 2)
             resident batches per GPU take the steps in turn, each on its own host thread and stream:
             step i is a pass over batch i % F, and the passes of different batches overlap (one
             batch's init + pack beside another's search; the library runs the searches one at a
             time).  value = B x K / the time K steps took; extra.one_batch_at_a_time is the same
             measurement with nothing else in flight
  N > 1    : one process per GPU (torch.distributed / RCCL for the barrier and the
             max-over-ranks clock only); histories are independent units, so they
             are sharded across ranks with NO data-path collective -- weak scaling
             (B per GPU fixed)
  roofline : the search kernel (wgl_narrow_kernel when several histories share a wavefront -- the library's choice for this
             batch --, wgl_beam_kernel, or wgl_search_kernel at --width 1), HBM bound.  achieved = algorithmic bytes per launch
             (16 B per visited-set probe that finds a duplicate,
             32 B per probe that inserts a new config) / the kernel's average
             duration, measured with HIP events on the pass's own stream (from the moment the search
             has the device to its end: with two batches in flight that is the search beside the other
             batch's pack); traffic = FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 passes
             (profiles/r05_traffic.json), copied only if kernel sources and configuration match
  cpu_baseline : the CPU restatement of the same search (oracle/wgl_window.c, "port") on a bounded
             sample of the same histories: on all host cores (pthread pool, oracle/many.c) = `value`,
             and on one thread (`single_thread`)
  extra    : time-to-verdict of ONE history through tbc_check (the level sweep, jit_sweep.hip), the
             crashed-op tiers of BASELINE.md section 3 (the count form: crashed calls as counts per effect class;
             beside each the plain CPU search AND the count form's own passes on one CPU core), a second workload at
             50 % duty ("64 concurrent processes", ~32 calls in flight) and a third at 30 % with their own value /
             roofline, a batch of the headline workload with 1 % crashed calls, the H2D-inclusive rate
             (tbc_batch_create alone; the Python binding's concatenation of the columns is marshal_s), checker/set-full
             (scan roofline; end to end from compact reads, the matrix built on the device).  The tiers, the three other
             workloads and set-full each run in a process of their own (run_leg): a GPU fault in one of them costs that leg,
             not the line
  checks   : every resident batch carries ONE history with a planted impossible read (2 % in, a value inside the
             batch's domain): verdicts are compared element-wise -- that one INVALID, every other one VALID, the CPU
             sample history by history, the planted one's failing op against the oracle's

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


def workload_name(ops, procs, busy, info):
    return (f"cas-register, {ops // 1000}k ops (invocations) / {procs} processes each busy {busy:.0%} of the time "
            f"(~{procs * busy:.1f} calls in flight), values 0..4, r/w/cas 1/3 each, crashed-op rate {info:.0%}")


def kernel_sha():
    """Identity of the search kernel's sources: a committed PMC traffic figure is only quoted for the very kernel it
    was measured on (profiles/*_traffic.json carries the sha it was taken at)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("wgl_beam.hip", "wgl_narrow.hip", "wgl_narrow_impl.h", "wave_env.h", "device_common.h", "pack_open.hip", "open_walk_impl.h"):
        with open(os.path.join(ROOT, "jepsen-tigerbeetle_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def usable_cores():
    """Host threads this process can actually keep busy: the CPUs it may run on, capped by the container's CPU-time
    quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us) -- os.cpu_count() reports the machine, not the allowance (the GPU
    boxes of this pool show 256 CPUs under a quota of 16)."""
    import math
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(math.ceil(quota))))
    return n, (os.cpu_count() or 1), quota


def leg(name):
    """progress on stderr (the JSON line is the only thing on stdout): which leg a run was in if it does not end"""
    print(f"[bench {time.strftime('%H:%M:%S')}] {name}", file=sys.stderr, flush=True)


def _gpu_imports():
    import numpy as np
    import jepsen_tigerbeetle_amd  # noqa: F401
    from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
    return np, N, columns, core, synth


def leg_tiers(args, local_rank):
    """extra.tiers (its own process, see run_leg)."""
    np, N, columns, core, synth = _gpu_imports()
    from oracle import wgl
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    om = {"kind": 1, "init": N.NIL}

    # crashed-op tiers x {as generated, one bad read}; one history each, GPU limit 3 s,
    # CPU limit 2*10^7 steps.  With crashed calls the library takes the COUNT FORM (crashed calls as counts per effect
    # class, one mask word; a history the budgeted exact search leaves open is refuted relaxed, then its prefix is
    # linearized): cpu_port_ms is the plain knossos.wgl restatement, cpu_same_algorithm_ms the count form's passes on one core.
    tiers = []
    o_tier = core.make_opts(device=local_rank, want_witness=False, algorithm=N.ALG_COMPETITION, time_limit_ms=3000)
    for seed in (1, 2):      # (this process's first calls: HIP context, code objects, the persistent context's arenas -- not a tier's time)
        core.check_ops(columns.pair_events(synth.register_events(n_ops=args.ops, n_procs=args.procs, seed=seed, busy=args.busy, info=0.01 * (seed - 1))), model, o_tier)
    for info in (0.0, 0.01, 0.05):
        for corrupt in (0.0, 0.5):
            leg(f"tier info {info} corrupt {corrupt}")
            hh = columns.pair_events(synth.register_events(n_ops=args.ops, n_procs=args.procs, seed=4242, busy=args.busy, info=info, corrupt=corrupt))
            t1 = time.perf_counter()
            rg = core.check_ops(hh, model, o_tier)
            tg = (time.perf_counter() - t1) * 1e3
            t1 = time.perf_counter()
            rc = wgl.check(hh.as_dict(), om, "window", want_witness=False, max_steps=20_000_000)
            tcpu = (time.perf_counter() - t1) * 1e3
            # the count form's own passes on one CPU core (oracle/wgl_count.c): the algorithm is the CPU's too
            t1 = time.perf_counter()
            rp = wgl.check_count_pipeline(hh.as_dict(), om, width=max(rg["search_width"], 1), relaxed_sweep=True) if info else None      # (the library's passes for one history: the relaxed level sweep first)
            tpipe = (time.perf_counter() - t1) * 1e3
            # ... and the count form's passes WITHOUT the sweep in front (exact under a budget, relaxed depth-first, prefix): the better single-core
            # formulation for a valid history (one core cannot run the sweep beside the search)
            t1 = time.perf_counter()
            rq = wgl.check_count_pipeline(hh.as_dict(), om, width=max(rg["search_width"], 1)) if info else None
            tplain = (time.perf_counter() - t1) * 1e3
            if rq is not None:
                assert (rq[0], rq[1] if rq[0] == 0 else None) == (rp[0], rp[1] if rp[0] == 0 else None), (info, corrupt, "the two pipelines")
            if rp is not None:
                assert rg["valid"] == rp[0] and (rp[0] == 1 or rg["fail_op"] == rp[1]), (info, corrupt, "count form")
            if rg["valid"] != -1 and rc["valid"] != -1:
                assert rg["valid"] == rc["valid"] and (rg["valid"] == 1 or rg["fail_op"] == rc["fail_op"]), (info, corrupt)
            tiers.append({"info_rate": info, "history": "1 bad read" if corrupt else "as generated", "process_slots": int(hh.n_process),
                          "gpu_ms": round(tg, 3), "gpu_verdict": rg["valid"], "gpu_analyzer": "linear" if rg["analyzer"] == N.ALG_LINEAR else "wgl",
                          "cpu_port_ms": round(tcpu, 3), "cpu_verdict": rc["valid"],
                          "cpu_same_algorithm_ms": None if rp is None else round(tpipe, 3), "cpu_same_algorithm_passes": None if rp is None else rp[4],
                          "cpu_count_form_without_sweep_ms": None if rq is None else round(tplain, 3),
                          "cpu_best_count_form_ms": None if rp is None else round(min(tpipe, tplain), 3)})
    return tiers



def leg_workload(args, local_rank, which):
    """extra.workload_2 / workload_3 / workload_crashed (each in its own process, see run_leg)."""
    np, N, columns, core, synth = _gpu_imports()
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)

    # second workload: BASELINE.json's "64 concurrent processes" read literally is infeasible for every known
    # algorithm; busy 0.5 (~32 calls in flight) is the closest reading the dominance rules make
    # checkable, busy 0.3 (~19 in flight) a point in between.  Each with its own value and roofline and the CPU restatement
    # of the kernel's schedule on a sample beside it (thread pool, the CPUs this container may use); never mixed into `value`.
    # A history at 32 in flight can need > 10^6 configs (the tail is heavy): 2^21-entry visited sets with their stacks = 67 MB
    # each, 2,048 of them (137 GB) a batch -- a quarter of the GPU's wavefront slots; smaller first sets cost retries that take
    # longer than the whole step (measured: 4,096 histories at 2^20 entries, 134 s of retries).
    def second(busy, B2, vpo, seed0, cpu_n, cpu_cap, warm, info=0.0, in_flight=1):
        h2 = synth.register_ops_many(range(seed0, seed0 + B2), n_ops=args.ops, n_procs=args.procs, busy=busy, info=info)
        o2 = core.make_opts(device=local_rank, time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION,
                            search_width=args.width, visited_per_op=vpo, lanes_per_history=int(os.environ.get("TBC_BENCH_LEG_LANES", "0")))
        with core.Batch(h2, model, o2) as b2:
            width2, lanes2, order2 = b2.search_width(), b2.lanes_per_history(), b2.list_order()
            if warm:
                b2.run()
            t2 = time.perf_counter(); b2.run(); t2 = time.perf_counter() - t2
            c2 = b2.counters(); tm2 = b2.timing_ns(); v2 = b2.verdicts()
            raced2 = b2.last_raced() if hasattr(b2, "last_raced") else 0
            fresh2 = None
            if info:
                # the same batch object fed ANOTHER set of crashed histories (a count-form batch takes fresh inputs since round 6: the classes
                # of crashed calls are planned on the host when the input is submitted -- csrc/batch_stream.hip plan_count_input)
                h3 = synth.register_ops_many(range(seed0 + B2, seed0 + 2 * B2), n_ops=args.ops, n_procs=args.procs, busy=busy, info=info)
                tr = time.perf_counter(); b2.reload(h3); tr = time.perf_counter() - tr
                tu = time.perf_counter(); b2.run(); tu = time.perf_counter() - tu
                v3 = b2.verdicts()
                fresh2 = {"histories_per_s": round(B2 / (tr + tu), 2), "host_marshal_encode_plan_submit_s": round(tr, 3), "of_which_tbc_batch_reload_s": round(getattr(b2, "reload_call_s", 0.0), 3), "run_s": round(tu, 3),
                          "valid": int((v3 == N.VALID).sum()), "unknown": int((v3 == N.UNKNOWN).sum()),
                          "what": "tbc_batch_reload of a never-seen batch (wire encoding, the count form's planning on the host, the copy queued) + the run that consumes it, one after the other on one host thread"}
                del h3
        two = None
        if in_flight > 1:
            # the same workload with TWO batch objects in flight, each on its own host thread, as the headline's: one batch's race of list
            # orders (a few hundred wavefronts for as long as its hardest history takes) runs beside the other batch's first pass.  `value`
            # above stays one batch at a time (what rounds 3 - 5 reported); this is what a caller that keeps the device fed gets
            import threading
            sets = [h2] + [synth.register_ops_many(range(seed0 + k * B2, seed0 + (k + 1) * B2), n_ops=args.ops, n_procs=args.procs, busy=busy, info=info) for k in range(1, in_flight)]
            bb = [core.Batch(h, model, o2) for h in sets]
            try:
                for b in bb:
                    b.run()
                steps2 = 3
                def loop(b):
                    for _ in range(steps2):
                        b.run()
                tt = time.perf_counter()
                th = [threading.Thread(target=loop, args=(b,)) for b in bb]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
                tt = time.perf_counter() - tt
                vv2 = [b.verdicts() for b in bb]
                assert np.array_equal(vv2[0], v2), "the same batch, another verdict"
                two = {"value": round(in_flight * steps2 * B2 / tt, 2), "unit": "histories/s", "batches_in_flight": in_flight, "steps_each": steps2,
                       "ms_per_step": round(tt / (in_flight * steps2) * 1e3, 3), "unknown": sum(int((v == N.UNKNOWN).sum()) for v in vv2)}
            finally:
                for b in bb:
                    b.close()
        alg2 = 16 * (c2["probes"] - c2["visited"]) + 32 * c2["visited"]
        k2 = (tm2["search"] + tm2["retries"]) / 1e6
        out = {"workload": workload_name(args.ops, args.procs, busy, info), "histories_per_gpu": B2, "search_width": width2, "lanes_per_history": lanes2, "list_order": order2,
               "value": round(B2 / t2, 2), "unit": "histories/s", "ms_per_step": round(t2 * 1e3, 3),
               "valid": int((v2 == N.VALID).sum()), "unknown": int((v2 == N.UNKNOWN).sum()),
               # histories whose first pass ended at its budget (32 probes per op) and that were then searched in six list orders at once (tbcheck.h, TBC_DOM_NO_ORDER_RESTARTS)
               "raced_in_six_orders": raced2,
               "roofline": {"bound": "hbm", "achieved": round(alg2 / (k2 * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(alg2 / (k2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "traffic": None,
                            "kernel": "wgl_narrow_kernel" if lanes2 != 64 else "wgl_beam_kernel",
                            "kernel_ms": round(k2, 3), "probes_per_launch": c2["probes"], "new_configs_per_launch": c2["visited"]},
               "device_ms": {k: round(x / 1e6, 3) for k, x in tm2.items()}}
        if fresh2 is not None:
            out["fresh_input"] = fresh2
        if two is not None:
            out["two_in_flight"] = two
        if not args.no_cpu and info:
            # crashed calls: the library's count form; the CPU runs the same passes (oracle/wgl_count.c) on a thread pool
            from concurrent.futures import ThreadPoolExecutor
            from oracle import wgl
            cores, _, _ = usable_cores()
            dd = [h2[i].as_dict() for i in range(min(cpu_n, B2))]
            budget = 32 * max(len(h) for h in h2)
            tcp = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                rr = list(ex.map(lambda d: wgl.check_count_pipeline(d, {"kind": 1, "init": N.NIL}, width=width2, budget=budget), dd))
            tcp = time.perf_counter() - tcp
            assert all(r is not None and r[0] == int(v2[i]) for i, r in enumerate(rr)), "GPU and oracle disagree on the crashed workload's sample"
            out["cpu_baseline"] = {"value": round(len(dd) / tcp, 3), "unit": "histories/s", "cores": cores, "kind": "port",
                                   "sample": f"first {len(dd)} histories, oracle/wgl_count.c (the count form's own passes, {width2} configs per round) from {cores} Python threads "
                                             f"(ctypes releases the GIL)"}
        elif not args.no_cpu:
            from oracle import wgl
            cores, _, _ = usable_cores()
            dd = [h2[i].as_dict() for i in range(min(cpu_n, B2))]
            tcp = time.perf_counter()
            vv, started = wgl.check_many(dd, {"kind": 1, "init": N.NIL}, cores, max_steps=cpu_cap, beam_width=width2 if width2 > 1 else 4,
                                         list_order=order2 if order2 >= 16 else wgl.ORACLE_LIST_ORDER[order2 - 1])          # (the kernel's own list order)
            tcp = time.perf_counter() - tcp
            done = int((vv != -1).sum())
            assert all(int(vv[i]) in (-1, int(v2[i])) for i in range(len(dd))), "GPU and oracle disagree on the second workload's sample"
            out["cpu_baseline"] = {"value": round(done / tcp, 3), "unit": "histories/s", "cores": cores, "kind": "port",
                                   "sample": f"first {len(dd)} histories, oracle/wgl_beam.c (the kernel's schedule, {width2} configs per round, lookahead + eager reads + "
                                             f"twin rule) on {started} pthreads, at most {cpu_cap:.0e} probes each: {done} finished (the others are not counted)"}
        return out
    if which == "workload_2":
        return second(args.busy2, args.batch2, 256, 10_000_000, 64, 60_000_000, False)      # (a step is ~25 s: one run, no warm-up)
    if which == "workload_3":
        return second(args.busy3, args.batch3, 32, 20_000_000, 128, 60_000_000, True, in_flight=2)
    # the regime the reference produces (a nemesis makes clients time out: :info): the headline workload with 1 % of the
    # calls crashed, a batch of them -- the count form, a wavefront per history
    assert which == "workload_crashed"
    return second(args.busy, args.batch4, 8, 30_000_000, 256, 0, True, info=args.info4)


def leg_set_full(args, local_rank):
    """extra.set_full (its own process, see run_leg)."""
    np, N, columns, core, synth = _gpu_imports()

    # checker/set-full (the checker the reference runs: set_full.clj:157): the reads x elements membership scan,
    # the one streaming kernel of the path.  Synthetic: 262,144 elements x 32,768 reads (1 GB of bits), an element
    # is visible from about its acknowledgement on, 300 elements vanish in the last quarter (lost), sparse holes
    # right after the add (stale reads).  roofline = matrix bytes the scan loaded / its time, against 8 TB/s.
    from jepsen_tigerbeetle_amd.jepsen import set_full as sf
    rng = np.random.default_rng(7)
    E, R = 262144, 32768
    add_invoke = (np.sort(rng.choice(4 * (E + R), E, replace=False)) * 2).astype(np.uint32)
    read_invoke = (np.sort(rng.choice(4 * (E + R), R, replace=False)) * 2 + 1).astype(np.uint32)
    read_ok = read_invoke + (rng.integers(1, 2000, R) * 2).astype(np.uint32)
    add_ok = (add_invoke + 2001).astype(np.uint32)
    p = np.searchsorted(add_ok, read_invoke).astype(np.int64)            # row r sees the elements acknowledged before it began
    w = np.arange(E // 32, dtype=np.int64)
    M = np.where(32 * (w + 1)[None, :] <= p[:, None], 0xFFFFFFFF,
                 np.where(32 * w[None, :] >= p[:, None], 0, (1 << np.clip(p[:, None] - 32 * w[None, :], 0, 31)) - 1)).astype(np.uint32)
    lost_e = rng.choice(E // 2, 300, replace=False)
    for e in lost_e:
        M[R * 3 // 4:, e // 32] &= np.uint32(~(1 << (e % 32)) & 0xFFFFFFFF)
    holes_r = rng.integers(0, R, 200000); holes_e = np.clip(p[holes_r] - rng.integers(1, 2000, 200000), 0, E - 1)
    np.bitwise_and.at(M, (holes_r, holes_e // 32), (~(np.uint32(1) << (holes_e % 32).astype(np.uint32))).astype(np.uint32))
    # the same reads in COMPACT form (tbc_setfull_rows): a prefix of the elements and the holes in it -- what a caller that holds
    # the reads as sorted id lists hands over; the matrix is then built on the device and nothing of its size crosses PCIe
    late = np.arange(R * 3 // 4, R)
    pr = np.concatenate([np.repeat(late, len(lost_e)), holes_r])
    pe = np.concatenate([np.tile(lost_e, len(late)), holes_e])
    keep = pe < p[pr]                                                   # (an element at or above the prefix is absent anyway)
    pairs = np.unique(pr[keep].astype(np.int64) * E + pe[keep].astype(np.int64))
    exc_rows, exc = pairs // E, (pairs % E).astype(np.uint32)
    exc_off = np.zeros(R + 1, np.uint64)
    exc_off[1:] = np.cumsum(np.bincount(exc_rows, minlength=R))

    class A:
        pass
    a = A(); a.E, a.R, a.wpr = E, R, E // 32
    a.add_invoke, a.add_ok, a.read_invoke, a.read_ok, a.present = add_invoke, add_ok, read_invoke, read_ok, np.ascontiguousarray(M)
    with sf.Scan(a, device=local_rank, rows=False) as sc:
        sc.run()
        runs = [sc.run() for _ in range(5)]
    a.top, a.exc_off, a.exc = p.astype(np.uint32), exc_off, exc
    t_e2e = []
    for _ in range(3):                   # end to end: compact reads on the host -> matrix built on the device -> scan -> three indices per element back
        t1 = time.perf_counter()
        with sf.Scan(a, device=local_rank, rows=True) as sc2:
            r2 = sc2.run()
        t_e2e.append((time.perf_counter() - t1) * 1e3)
    for k in ("known", "last_present", "last_absent"):
        assert np.array_equal(r2[k], runs[0][k]), f"set-full: the matrix built on the device gives another {k}" 
    ms = statistics.mean(r["ns_scan"] for r in runs) / 1e6
    lost = int(((runs[0]["last_present"].astype(np.int64) < runs[0]["last_absent"].astype(np.int64)) & (runs[0]["last_absent"] != N.NO_OP)).sum())
    gbs = runs[0]["bytes_scanned"] / (ms * 1e-3) / 1e9
    # HBM bytes per scan from the committed rocprofv3 PMC passes (scripts/gpu_profile_setfull.sh), quoted only for the very source they were taken on
    import glob, hashlib
    sf_traffic = None
    with open(os.path.join(ROOT, "jepsen-tigerbeetle_amd", "csrc", "set_full.hip"), "rb") as fh:
        sf_sha = hashlib.sha256(fh.read()).hexdigest()[:16]
    for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_setfull_traffic.json")), reverse=True):
        try:
            with open(tf) as fh:
                tj = json.load(fh)
            if tj.get("set_full_sha") == sf_sha and (tj.get("elements"), tj.get("reads")) == (E, R):
                sf_traffic = int(tj["traffic_bytes_per_scan"])
                break
        except (OSError, ValueError, KeyError):
            pass
    return {"elements": E, "reads": R, "matrix_GB": round(runs[0]["bytes_matrix"] / 1e9, 3),
            "scan_ms": round(ms, 3), "bytes_scanned": int(runs[0]["bytes_scanned"]), "lost_elements_found": lost,
            "end_to_end_ms": round(min(t_e2e), 3), "compact_input_MB": round((exc.nbytes + exc_off.nbytes + 4 * R) / 1e6, 2),
            "end_to_end_note": "tbc_setfull_create_rows + tbc_setfull_run + destroy: allocation, H2D of the compact reads, the matrix built on the device, the scan, "
                               "three indices per element back (best of 3); the dense form moves the 1 GB matrix over PCIe instead",
            "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": sf_traffic, "kernel": "setfull_any_kernel + setfull_resolve_kernel"}}


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(line):
    """The ONE line the driver parses (last line of stdout): every field the bench contract names, `roofline`, `cpu_baseline`
    and a small `extra` -- at most 4 KB whatever the legs returned (the driver keeps a 13.7 KB tail of stdout; round 4's line
    outgrew it and the round went unmeasured).  Everything else is in the full object: gpurun_out/bench_full.json and, one
    line, on stderr.  tests/test_bench_line.py builds this from a worst-case payload and checks the size and the keys."""
    out = _pick(line, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data")
    out["config"] = _pick(line.get("config", {}), "workload", "histories_per_gpu", "ops_after_pairing", "processes", "busy", "info_rate",
                          "batches_in_flight", "search_width", "lanes_per_history", "list_order", "parallelism")
    out["roofline"] = _pick(line.get("roofline", {}), "bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                            "probes_per_launch", "new_configs_per_launch", "algorithmic_bytes_per_launch")
    cb = line.get("cpu_baseline")
    if cb is not None:
        c = _pick(cb, "value", "unit", "cores", "kind")
        c["sample"] = str(cb.get("sample", ""))[:120]
        if "single_thread" in cb:
            c["single_thread"] = _pick(cb["single_thread"], "value", "ms_per_history")
        ss = cb.get("same_schedule_as_kernel")
        if ss:
            c["same_schedule"] = {"single_thread": ss.get("value"), "all_cores": (ss.get("all_cores") or {}).get("value")}
        out["cpu_baseline"] = c
    ex, e = line.get("extra", {}), {}
    e.update(_pick(ex, "valid", "unknown", "device_GB_per_batch", "h2d_inclusive_hist_per_s"))
    fi = ex.get("fresh_input")
    if isinstance(fi, dict):          # every step consumes histories the device has never seen (PCIe copy + unpack + pack + search): first-class beside `value`
        e["fresh_input"] = _pick(fi, "histories_per_s", "steps", "ms_per_step", "pcie_GB_per_input", "pcie_GB_s_over_the_timed_region", "pcie_GB_s_while_copying", "pcie_peak_GB_s")
        ru = fi.get("re_uploaded")
        if isinstance(ru, dict):
            e["fresh_input"]["re_uploaded"] = _pick(ru, "histories_per_s", "steps", "pcie_GB_s_over_the_timed_region")
    if "device_ms" in ex:
        e["device_ms"] = _pick(ex["device_ms"], "init_memsets", "pack", "search", "retries", "search_waiting_for_its_turn")
    if "one_batch_at_a_time" in ex:
        a = ex["one_batch_at_a_time"]
        e["one_batch_at_a_time"] = {"value": a.get("value"), "search_ms": (a.get("device_ms") or {}).get("search"), "pack_ms": (a.get("device_ms") or {}).get("pack"),
                                    "roofline_frac": a.get("roofline_frac")}
    if "time_to_verdict_ms" in ex:
        e["time_to_verdict_ms"] = _pick(ex["time_to_verdict_ms"], "valid_median", "valid_min", "answered_by_sweep", "of", "invalid_example",
                                        "depth_first_with_witness_median", "vs_cpu_port_single_thread", "vs_cpu_same_schedule_single_thread", "split_us")
    if isinstance(ex.get("tiers"), list):       # six short rows: [crashed-op rate, bad read planted, GPU ms, verdict, plain CPU port ms, the same passes on one CPU core ms]
        e["tiers"] = {"cols": ["info", "bad_read", "gpu_ms", "verdict", "cpu_port_ms", "cpu_best_count_form_ms"],
                      "rows": [[t.get("info_rate"), int(t.get("history") != "as generated"), t.get("gpu_ms"), t.get("gpu_verdict"), t.get("cpu_port_ms"),
                                t.get("cpu_best_count_form_ms", t.get("cpu_same_algorithm_ms"))] for t in ex["tiers"][:6]]}
    elif "tiers" in ex:
        e["tiers"] = ex["tiers"]
    for w in ("workload_2", "workload_3", "workload_crashed"):
        d = ex.get(w)
        if isinstance(d, dict):
            e[w] = {"error": str(d["error"])[:100]} if "error" in d else {
                "value": d.get("value"), "histories": d.get("histories_per_gpu"), "unknown": d.get("unknown"), "frac": (d.get("roofline") or {}).get("frac"),
                "traffic": (d.get("roofline") or {}).get("traffic"), "kernel_ms": (d.get("roofline") or {}).get("kernel_ms"), "cpu": (d.get("cpu_baseline") or {}).get("value")}
            if isinstance(d.get("fresh_input"), dict):
                e[w]["fresh"] = d["fresh_input"].get("histories_per_s")
            if isinstance(d.get("two_in_flight"), dict):
                e[w]["two_in_flight"] = d["two_in_flight"].get("value")
    d = ex.get("set_full")
    if isinstance(d, dict):
        e["set_full"] = {"error": str(d["error"])[:100]} if "error" in d else {"scan_ms": d.get("scan_ms"), "end_to_end_ms": d.get("end_to_end_ms"),
                                                                              "frac": (d.get("roofline") or {}).get("frac")}
    bm = ex.get("bad_read_in_the_middle")
    if isinstance(bm, dict):
        e["bad_read_in_the_middle"] = {"step_ms_with_handover": (bm.get("with_handover") or {}).get("ms_per_step_alone"),
                                       "step_ms_without": (bm.get("without_handover") or {}).get("ms_per_step_alone")}
    if "one_history_over_all_gpus" in ex:
        e["one_history_over_all_gpus"] = ex["one_history_over_all_gpus"]
    e["full"] = "gpurun_out/bench_full.json"
    out["extra"] = e
    return out


def emit(line):
    """Full object -> gpurun_out/bench_full.json + stderr; the compact line -> stdout, last."""
    full = json.dumps(line)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as fh:
            fh.write(full + "\n")
    except OSError:
        pass
    print("[bench full] " + full, file=sys.stderr, flush=True)
    small = json.dumps(compact_line(line), separators=(",", ":"))
    assert len(small) < 8192, len(small)
    sys.stdout.flush()
    print(small, flush=True)


LEGS = {"tiers": leg_tiers, "set_full": leg_set_full,
        "workload_2": lambda a, d: leg_workload(a, d, "workload_2"), "workload_3": lambda a, d: leg_workload(a, d, "workload_3"),
        "workload_crashed": lambda a, d: leg_workload(a, d, "workload_crashed")}


def run_leg(name, args, local_rank):
    """One of the extra legs in a process of its own (python bench.py <the same arguments> --leg NAME prints {"leg": ..., "result": ...}):
    none of them feeds `value`, and a leg whose process is killed -- a GPU fault aborts the process it happens in; round 4 saw one
    in five runs of this file, never reproduced -- must not take the measured line with it; the line then says so
    (extra.<leg>.error).  A leg that fails an assertion fails the whole run, as before.  TBC_BENCH_INLINE_LEGS=1 runs them in this process instead."""
    leg(name.replace("_", " "))
    if os.environ.get("TBC_BENCH_INLINE_LEGS") == "1":
        return LEGS[name](args, local_rank)
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--leg", name]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, timeout=1500, env=dict(os.environ, LOCAL_RANK=str(local_rank)))   # (stderr: inherited, the leg markers)
    except subprocess.TimeoutExpired:
        return {"error": "the leg's process did not finish within 1500 s"}
    for ln in reversed(r.stdout.decode(errors="replace").splitlines()):
        if ln.startswith("{"):
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            if d.get("leg") == name and r.returncode == 0:
                return d["result"]
    if r.returncode > 0:      # a Python exception in the leg (an assertion: GPU and oracle disagree): as loud as it was in this process
        raise SystemExit(f"bench.py --leg {name} failed with return code {r.returncode} (its traceback is on stderr)")
    return {"error": f"the leg's process was killed by signal {-r.returncode} (a GPU fault aborts the process it happens in); no result"}



def main():
    t_bench0 = time.time()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed passes (each over one batch); with two batches in flight the first pack and the last search are not overlapped, so few steps under-report the steady rate (16 steps: -3 %%)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("TBC_BENCH_BATCH", "32768")),
                    help="histories per GPU per step")
    ap.add_argument("--ops", type=int, default=10000)
    ap.add_argument("--procs", type=int, default=64)
    ap.add_argument("--busy", type=float, default=0.1,
                    help="fraction of time a process has an op open (64 x 0.1 = 6.4 ops in flight on average)")
    ap.add_argument("--info", type=float, default=0.0, help="crashed-op (:info) rate")
    ap.add_argument("--width", type=int, default=int(os.environ.get("TBC_BENCH_WIDTH", "0")),
                    help="configs expanded per iteration: 0 = the library's choice (2 at low concurrency under the dominance rules, "
                         "else 4), 1 = sequential knossos.wgl order, 2..16 = wide schedule")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("TBC_BENCH_LANES", "0")),
                    help="lanes per history of the depth-first search: 0 = the library's choice (8 for this workload: eight histories per "
                         "wavefront), 8 / 16 / 32, or 64 = one history per wavefront")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("TBC_BENCH_IN_FLIGHT", "2")),
                    help="resident batches per GPU, each of --batch histories, on its own stream and host thread (how jepsen.independent's "
                         "thread pool would drive the library): a step is still ONE pass over ONE batch; the passes of different batches "
                         "overlap -- one batch's init + pack beside another's search (the library runs the searches one at a time). "
                         "1 = one batch, one pass after the other (also measured in every run: extra.one_batch_at_a_time)")
    ap.add_argument("--visited-per-op", type=int, default=4, help="first visited-set capacity per op (0 = library default 64); a history that needs more grows its set inside the kernel")
    ap.add_argument("--round-budget", type=int, default=0,
                    help="a history that has used more rounds than this continues at width 16 (0 = off)")
    ap.add_argument("--cpu-sample", type=int, default=256, help="histories timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--busy2", type=float, default=0.5, help="second workload: duty cycle of the '64 concurrent processes' reading (0 = skip)")
    ap.add_argument("--batch2", type=int, default=2048, help="second workload: histories per GPU (a 2^21-entry visited set each: no retries)")
    ap.add_argument("--busy3", type=float, default=0.3, help="a point in between: ~19 calls in flight (0 = skip)")
    ap.add_argument("--batch3", type=int, default=8192, help="third workload: histories per GPU")
    ap.add_argument("--info4", type=float, default=0.01, help="a batch of the headline workload with this share of crashed (:info) calls (0 = skip)")
    ap.add_argument("--batch4", type=int, default=8192, help="crashed workload: histories per GPU")
    ap.add_argument("--fresh-batches", type=int, default=int(os.environ.get("TBC_BENCH_FRESH", "4")),
                    help="extra.fresh_input: this many MORE batches of --batch histories each are generated on the host and written, in wire format, "
                         "into the resident batches' pinned slots; every step of that leg copies its input over PCIe, unpacks, packs and searches it "
                         "(tbc_batch_submit_input + tbc_batch_run) -- the rate a caller who checks every history once gets.  0 = skip")
    ap.add_argument("--no-tiers", action="store_true", help="skip the crashed-op tiers (extra.tiers)")
    ap.add_argument("--no-set-full", action="store_true", help="skip the checker/set-full scan (extra.set_full)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only-headline", action="store_true",
                    help="the warm-up and the timed steps only (no extra legs): what scripts/gpu_profile_r05.sh runs under rocprofv3 --kernel-trace "
                         "--stats, so that the kernel's average duration there is the one of the timed region")
    ap.add_argument("--sharded-ttv", action="store_true",
                    help="N > 1 only: also time ONE history swept by all N GPUs (shard.check_sharded: wavefronts dealt to the ranks, "
                         "one RCCL all-gather of the relation tables); off by default -- it adds a collective to the run")
    ap.add_argument("--leg", default=None, help="(internal) run this one extra leg and print its result: see run_leg")
    args = ap.parse_args()
    if args.leg is not None:
        import torch
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: libtbcheck has no CPU fallback")
        torch.cuda.set_device(lr)
        print(json.dumps({"leg": args.leg, "result": LEGS[args.leg](args, lr)}), flush=True)
        return
    if args.only_headline:
        args.no_cpu = args.no_tiers = args.no_set_full = True
        args.busy2 = args.busy3 = args.info4 = 0.0
        args.fresh_batches = 0

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist
    # TBC_BENCH_BACKEND=gloo + TBC_BENCH_BATCH_CLASS=module:Class: the N > 1 plumbing of this file (barrier, max-over-ranks clock,
    # verdict all-reduce, the one JSON line from rank 0) driven without a GPU by tests/test_distributed_gloo.py, which supplies a
    # stand-in for core.Batch; nothing is measured in that mode and the line says so ("data": "stand-in")
    backend = os.environ.get("TBC_BENCH_BACKEND", "nccl")
    on_gpu = backend == "nccl"
    if on_gpu and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libtbcheck has no CPU fallback")
    if on_gpu:
        torch.cuda.set_device(local_rank)
    else:
        args.only_headline = args.no_cpu = args.no_tiers = args.no_set_full = True
        args.busy2 = args.busy3 = args.info4 = 0.0
        args.fresh_batches = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import jepsen_tigerbeetle_amd  # noqa: F401
    from jepsen_tigerbeetle_amd import _native as N, columns, core, shard, synth

    BatchCls = core.Batch
    if not on_gpu:
        import importlib
        mod, cls = os.environ["TBC_BENCH_BATCH_CLASS"].split(":")
        BatchCls = getattr(importlib.import_module(mod), cls)

    def barrier():
        if world > 1:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    # ---- synthetic input: B distinct seeded histories per rank
    B = args.batch
    F = max(1, args.in_flight)
    t_gen = time.time()
    seeds = shard.shard_indices(F * B * world, rank, world)      # history i of the job lives on rank i % world
    hists_all = [synth.register_ops_many(seeds[k * B:(k + 1) * B], n_ops=args.ops, n_procs=args.procs, busy=args.busy, info=args.info) for k in range(F)]
    # every resident batch carries ONE history with a planted bad read (its last): a verdict mix-up cannot hide behind "all valid".
    # In the headline the read is planted 2 % into the history, as in rounds 3 and 4 (the workload is VALID histories; the guard should not
    # be what is measured): an INVALID verdict means exhausting the configs up to the failing completion, nine times a valid history's
    # search from the middle of a 10k-op history, and the slowest history of a batch is the batch's step.  What a bad read ANYWHERE costs
    # is measured by itself, in the same run: extra.bad_read_in_the_middle (asked to, the library stops a history that no longer passes
    # completions and hands it to the level sweep: tbc_opts.dominance, TBC_DOM_STALL_HANDOVER -- off by default, measured both ways there).  It reads the value 4 in a history that only ever
    # writes 0..3: impossible, and inside the batch's value domain (the generator's own planted value, 12, would widen every front
    # record of the batch from 64 to 160 bytes: another kernel instantiation)
    planted = B - 1
    for k in range(F):
        hp = columns.pair_events(synth.register_events(n_ops=args.ops, n_procs=args.procs, seed=int(seeds[k * B + planted]), busy=args.busy,
                                                       info=args.info, corrupt=0.02, n_values=4))
        assert int((hp.a == 4 + 7).sum()) == 1
        hp.a[hp.a == 4 + 7] = 4
        hists_all[k][planted] = hp
    hists = hists_all[0]
    t_gen = time.time() - t_gen
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    opts = core.make_opts(device=local_rank, time_limit_ms=600000, want_witness=False,
                          algorithm=N.ALG_COMPETITION, search_width=args.width, visited_per_op=args.visited_per_op,
                          round_budget=args.round_budget, lanes_per_history=args.lanes)
    leg("inputs generated; creating the resident batches")
    t_create = time.perf_counter()
    batches = [BatchCls(h, model, opts) for h in hists_all]        # H2D happens here: inputs resident before timing
    t_create = (time.perf_counter() - t_create) / F
    t_create_c = sum(getattr(b, "create_s", 0.0) for b in batches) / F      # tbc_batch_create alone (t_create also holds numpy's concatenation of the histories' columns)
    batch = batches[0]
    width = batch.search_width()                  # what --width 0 became for this batch
    lanes = batch.lanes_per_history()             # 8 / 16 / 32: several histories per wavefront (one config per iteration); 64: one
    narrow = lanes != 64
    kname = "wgl_narrow_kernel" if narrow else ("wgl_search_kernel" if width == 1 else "wgl_beam_kernel")
    batch_list_order = batch.list_order() if hasattr(batch, "list_order") else None      # TBC_ORDER_*: what tbc_opts.list_order = 0 became (16 + 24 for the headline batch)

    # a step = one pass (init + pack + search + verdicts back) over one resident batch.  Step i goes to batch i % F; each batch
    # has its own host thread and stream, so up to F passes are in flight (tbc_batch_run is a blocking C call that releases the GIL)
    import threading
    def passes(k, count, log):
        for _ in range(count):
            batches[k].run()
            if log is not None:
                log.append(batches[k].timing_ns())
    def in_flight(total, logs):
        th = [threading.Thread(target=passes, args=(k, len(range(k, total, F)), None if logs is None else logs[k])) for k in range(F)]
        for x in th: x.start()
        for x in th: x.join()
    leg("warm-up")
    in_flight(max(args.warmup, F), None)                      # (every resident batch is run once before the clock starts)
    leg("timed region")
    barrier()
    t0 = time.perf_counter()
    logs = [[] for _ in range(F)]
    in_flight(args.steps, logs)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, world, dist)
    tms = [tm for l in logs for tm in l]
    assert len(tms) == args.steps
    search_ns, pack_ns, init_ns, retry_ns = ([tm[k] for tm in tms] for k in ("search", "pack", "init", "retries"))
    wait_ns = [tm["turn_wait"] for tm in tms]
    leg("timed region done")
    # the same passes with nothing else in flight: one batch, one pass after the other (never `value` unless --in-flight 1)
    alone = None
    if F > 1 and not args.only_headline:
        barrier()
        ta0 = time.perf_counter()
        tma = []
        for _ in range(3):
            batch.run(); tma.append(batch.timing_ns())
        barrier()
        alone = (shard.max_over_ranks((time.perf_counter() - ta0) / 3, world, dist), tma)

    # ---- FRESH INPUTS (include/tbcheck.h "streaming"; csrc/batch_stream.hip): what a caller who checks every history ONCE gets -- the
    # reference's call pattern (checker/compose over the test's one history, core.clj:139-146; independent/checker per key,
    # set_full.clj:155-158).  The resident batches keep their arenas; G more batches of histories the device has never seen are generated
    # on the host and written in wire format (12 B an op) into the batches' pinned slots BEFORE the clock starts (the host's encoding
    # of its histories is the caller's work, as marshal_s is for the resident line); every timed step then submits one slot (PCIe copy on
    # the batch's copy stream, under the other passes) and runs it: unpack, pack, search, verdicts back.  Each batch object has its own
    # host thread, as in the headline.  `never_seen`: each of the G inputs consumed exactly once.  `re_uploaded`: the same G host batches
    # taken round-robin for 3 G more steps -- every step still copies, unpacks and packs its input from scratch (nothing of an earlier
    # pass is reused on the device: the tables are rebuilt, the visited sets are another epoch's).  Never `value`.
    device_bytes_resident = batch.device_bytes()          # (before the fresh inputs' stages and headroom)
    device_bytes_all_resident = sum(b.device_bytes() for b in batches)
    fresh = None
    G = (args.fresh_batches // F) * F
    if G > 0 and on_gpu and narrow and world == 1:        # (N > 1: the headline only -- every rank would generate and encode G more batches on the one host)
        per = G // F
        fresh_verdicts = {}
        fresh_infos = []
        def fresh_passes(k, order, keep):
            b = batches[k]
            b.map_input(order[0])                       # (waits for the slot's previous copy; the slot is not re-written)
            b.submit_input(order[0], B)
            for i, sl in enumerate(order):
                if i + 1 < len(order):
                    b.map_input(order[i + 1])
                    b.submit_input(order[i + 1], B)
                b.run()
                fresh_infos.append(b.input_info())
                if keep:
                    fresh_verdicts[(k, sl)] = b.verdicts()
        def fresh_round(orders, keep):
            th = [threading.Thread(target=fresh_passes, args=(k, orders[k], keep)) for k in range(F)]
            for x in th: x.start()
            for x in th: x.join()
        # warm-up: every batch object's OWN resident histories through the wire (slot 0) -- the arenas' one-time growth and the first
        # unpack are not in the clock, and a reloaded resident batch must answer as it did
        leg("fresh inputs: warm-up (the resident histories through the wire)")
        t_enc = time.perf_counter()
        for k in range(F):
            batches[k]._cols = None                     # (the binding's concatenated columns: the library copied them at create)
            batches[k].fill_input(0, hists_all[k])
        t_enc = (time.perf_counter() - t_enc) / F
        fresh_round([[0] for _ in range(F)], True)
        for k in range(F):
            v = fresh_verdicts[(k, 0)]
            assert int(v[planted]) == N.INVALID and bool((np.delete(v, planted) == N.VALID).all()), "a reloaded resident batch answers differently"
        leg("fresh inputs: generating and encoding %d more batches" % G)
        t_fgen = time.perf_counter()
        fseeds = shard.shard_indices(G * B * world, rank, world) + F * B * world          # seeds past the resident batches'
        fresh_sample = None
        for g in range(G):
            hs = synth.register_ops_many(fseeds[g * B:(g + 1) * B], n_ops=args.ops, n_procs=args.procs, busy=args.busy, info=args.info)
            hp = columns.pair_events(synth.register_events(n_ops=args.ops, n_procs=args.procs, seed=int(fseeds[g * B + planted]), busy=args.busy,
                                                           info=args.info, corrupt=0.02, n_values=4))
            hp.a[hp.a == 4 + 7] = 4
            hs[planted] = hp
            batches[g // per].fill_input(g % per, hs)          # (in wire format, in place in the pinned slot; the set itself is dropped)
            if g == 0:
                fresh_sample = hs[:8]
            del hs
        t_fgen = time.perf_counter() - t_fgen
        fresh_infos.clear()
        leg("fresh inputs: timed (never seen)")
        barrier()
        tf0 = time.perf_counter()
        fresh_round([list(range(per)) for _ in range(F)], True)
        barrier()
        t_never = shard.max_over_ranks(time.perf_counter() - tf0, world, dist)
        infos_never = list(fresh_infos)
        for k in range(F):
            for j in range(per):
                v = fresh_verdicts[(k, j)]
                assert int(v[planted]) == N.INVALID and bool((np.delete(v, planted) == N.VALID).all()), "a fresh input's verdict is not where it belongs"
        leg("fresh inputs: timed (re-uploaded round-robin)")
        fresh_infos.clear()
        barrier()
        tf0 = time.perf_counter()
        fresh_round([[j % per for j in range(3 * per)] for _ in range(F)], False)
        barrier()
        t_again = shard.max_over_ranks(time.perf_counter() - tf0, world, dist)
        infos_again = list(fresh_infos)
        gb = lambda infos: sum(i["bytes_copied"] for i in infos) / 1e9
        copy_rate = lambda infos: round(statistics.mean(i["bytes_copied"] / max(1, i["ns_copy"]) for i in infos), 2)
        fresh = {"histories_per_s": round(G * B * world / t_never, 2), "unit": "histories/s", "steps": G, "ms_per_step": round(t_never / G * 1e3, 3),
                 "what": f"{G} batches of {B} histories the device had never seen, each consumed once: PCIe copy of the wire columns (12 B an op) + unpack + pack + search + verdicts; "
                         f"{F} batch objects on {F} host threads, each input's copy queued before the previous input's run",
                 "pcie_GB_per_input": round(gb(infos_never) / max(1, len(infos_never)), 3),
                 "pcie_GB_s_over_the_timed_region": round(gb(infos_never) * world / t_never, 2),
                 "pcie_GB_s_while_copying": copy_rate(infos_never), "pcie_peak_GB_s": 63.0,
                 "re_uploaded": {"histories_per_s": round(3 * G * B * world / t_again, 2), "steps": 3 * G, "ms_per_step": round(t_again / (3 * G) * 1e3, 3),
                                 "pcie_GB_s_over_the_timed_region": round(gb(infos_again) * world / t_again, 2), "pcie_GB_s_while_copying": copy_rate(infos_again),
                                 "what": "the same host batches round-robin: every step copies, unpacks and packs its input again (nothing of an earlier pass is reused on the device)"},
                 "lists_regrown": max(i["lists_regrown"] for i in infos_never + infos_again),
                 "host_gen_and_encode_s": round(t_fgen, 2), "host_encode_s_per_batch": round(t_enc, 2), "device_GB_per_batch_after": round(batch.device_bytes() / 1e9, 3)}

    sharded_ms = None
    if world > 1 and args.sharded_ttv:
        o_lin = core.make_opts(device=local_rank, want_witness=False, algorithm=N.ALG_LINEAR)
        one = synth.register_ops_many([424242], n_ops=args.ops, n_procs=args.procs, busy=args.busy, info=0.0)   # the same history on every rank
        # the exchange runs INSIDE the library (csrc/tbc_comm.hip: its own RCCL communicator, ncclAllGather of the relation tables out of
        # HBM) -- what a Clojure host would call; torch.distributed only carries the 128-byte id from rank 0 to the others
        ident = [shard.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        times = []
        with shard.Comm.rccl(rank, world, ident[0], local_rank) as comm:
            for _ in range(4):
                with core.Batch(one, model, o_lin) as b1:
                    barrier()
                    t1 = time.perf_counter()
                    r1 = comm.check(b1)
                    times.append(shard.max_over_ranks(time.perf_counter() - t1, world, dist) * 1e3)
        sharded_ms = {"median_ms": round(statistics.median(times[1:]), 3), "valid": r1[0]["valid"], "gpus": world, "exchange": "tbc_batch_sweep_allgather (RCCL inside libtbcheck)"}

    verdicts = batch.verdicts()
    for b in batches:               # element-wise: the planted history is INVALID, every other one VALID, in every resident batch
        v = b.verdicts()
        assert int(v[planted]) == N.INVALID and bool((np.delete(v, planted) == N.VALID).all()), "a verdict is not where it belongs"
    all_counters = [b.counters() for b in batches]
    counters = {k: sum(c[k] for c in all_counters) // F for k in all_counters[0]}      # per launch: the mean over the resident batches
    n_valid = sum(int((b.verdicts() == N.VALID).sum()) for b in batches)            # over all resident batches (F x B histories per GPU)
    n_unknown = sum(int((b.verdicts() == N.UNKNOWN).sum()) for b in batches)
    if world > 1:
        t = torch.tensor([n_valid, n_unknown], dtype=torch.int64, device="cuda" if on_gpu else "cpu")
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
        # HBM bytes per launch from the committed rocprofv3 PMC passes (the newest round's file that has one): same config AND same kernel sources only
        import glob
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
            try:
                with open(tf) as fh:
                    for e in json.load(fh)["entries"]:
                        key = (e["histories_per_gpu"], e["search_width"], e.get("lanes_per_history", 64), e["visited_per_op"], e["ops"], e["procs"], e["busy"], e["info"])
                        if e.get("kernel_sha") == kernel_sha() and key == (B, width, lanes, args.visited_per_op, args.ops, args.procs, args.busy, args.info):
                            traffic = e["traffic_bytes"]
            except (OSError, KeyError, ValueError):
                continue
            if traffic is not None:
                break
        line = {
            "metric": "histories/sec, 10k-op/64-proc cas-register histories (time-to-verdict ms in extra)",
            "value": round(value, 2), "unit": "histories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic" if on_gpu else "stand-in",
            "config": {"workload": workload_name(args.ops, args.procs, args.busy, args.info), "histories_per_gpu": B, "ops_after_pairing": int(batch.total_ops // B),
                       "processes": args.procs, "busy": args.busy, "info_rate": args.info,
                       "batches_in_flight": F,
                       "search_width": width, "search_width_asked": args.width, "lanes_per_history": lanes, "histories_per_wavefront": 64 // lanes,
                       "list_order": batch_list_order,
                       "round_budget": args.round_budget,
                       "parallelism": f"independent histories sharded over {world} GPU(s), no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "kernel": kname, "kernel_ms": round(k_ms, 3),
                         "probes_per_launch": counters["probes"], "new_configs_per_launch": counters["visited"],
                         "algorithmic_bytes_per_launch": alg_bytes},
            "extra": {"valid": n_valid, "unknown": n_unknown,
                      "device_ms": {"init_memsets": round(statistics.mean(init_ns) / 1e6, 3),
                                    "pack": round(statistics.mean(pack_ns) / 1e6, 3),
                                    "search": round(k_ms, 3), "retries": round(statistics.mean(retry_ns) / 1e6, 3),
                                    "search_waiting_for_its_turn": round(statistics.mean(wait_ns) / 1e6, 3),
                                    "note": "per pass, HIP events on the pass's own stream; with several batches in flight a pass's init and pack "
                                            "share the GPU with another pass's search, so the four do not add up to ms_per_step"},
                      "steps_per_history": counters["steps"] / B,
                      "device_GB": round(device_bytes_all_resident / 1e9, 3), "device_GB_per_batch": round(device_bytes_resident / 1e9, 3), "gen_s": round(t_gen, 2),
                      # the same batch with its inputs NOT resident: tbc_batch_create (allocation + H2D of the op
                      # columns over PCIe) + one run; never `value`
                      # (create_h2d_s = tbc_batch_create: host SoA columns in, resident batch out; marshal_s = what the Python binding spends
                      # before it, concatenating 32,768 histories' columns into one set -- the reference side's op maps -> SoA step)
                      "h2d_inclusive_hist_per_s": round(B / (t_create_c + elapsed / args.steps), 2),
                      "create_h2d_s": round(t_create_c, 3), "marshal_s": round(t_create - t_create_c, 3), "kernel_sha": kernel_sha()},
        }
        if alone is not None:
            a_s, tma = alone
            line["extra"]["one_batch_at_a_time"] = {
                "value": round(B * world / a_s, 2), "unit": "histories/s", "ms_per_step": round(a_s * 1e3, 3),
                "device_ms": {k2: round(statistics.mean(tm[k1] for tm in tma) / 1e6, 3) for k1, k2 in (("init", "init_memsets"), ("pack", "pack"), ("search", "search"), ("retries", "retries"))},
                "roofline_frac": round(alg_bytes / (statistics.mean(tm["search"] for tm in tma) * 1e-9) / 1e9 / HBM_PEAK_GBS, 6),
                "note": "the resident batch 0 alone, 3 passes back to back right after the timed region"}
        if fresh is not None:
            line["extra"]["fresh_input"] = fresh
        if sharded_ms is not None:
            line["extra"]["one_history_over_all_gpus"] = sharded_ms
        # time-to-verdict for ONE history through tbc_check (host columns in -> verdict out: H2D + kernels + D2H), rank 0.
        # knossos.competition without a witness = the level sweep (jit_sweep.hip); with a witness = the depth-first search
        for b in batches:
            b.close()
        leg("same batch, one history per wavefront")
        if world == 1 and narrow and args.lanes == 0 and not args.only_headline:
            # the same batch with ONE history per wavefront (round 2's kernel, wgl_beam_kernel at the width it chose then): what
            # several histories per wavefront buy, measured in the same run.  Never `value`.
            with core.Batch(hists, model, core.make_opts(device=local_rank, time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION,
                                                         lanes_per_history=64, visited_per_op=args.visited_per_op)) as b4:
                b4.run()
                t4 = time.perf_counter(); b4.run(); t4 = time.perf_counter() - t4
                c4, tm4, w4 = b4.counters(), b4.timing_ns(), b4.search_width()
            alg4 = 16 * (c4["probes"] - c4["visited"]) + 32 * c4["visited"]
            line["extra"]["same_batch_one_history_per_wavefront"] = {
                "kernel": "wgl_beam_kernel", "search_width": w4, "value": round(B / t4, 2), "unit": "histories/s", "ms_per_step": round(t4 * 1e3, 3),
                "kernel_ms": round(tm4["search"] / 1e6, 3), "pack_ms": round(tm4["pack"] / 1e6, 3),
                "probes_per_launch": c4["probes"], "new_configs_per_launch": c4["visited"], "algorithmic_bytes_per_launch": alg4,
                "roofline_frac": round(alg4 / (tm4["search"] * 1e-9) / 1e9 / HBM_PEAK_GBS, 6)}
        leg("a bad read in the middle of one history of the batch")
        if world == 1 and narrow and args.lanes == 0 and not args.only_headline and on_gpu:
            # the SAME batch with its planted history's bad read half way in: one pass alone, the stall handover on and off (the library's default) -- how long one not-linearizable history holds a pass of 32,768.  Never `value`.
            mid = list(hists)
            mid[planted] = columns.pair_events(synth.register_events(n_ops=args.ops, n_procs=args.procs, seed=int(seeds[planted]), busy=args.busy,
                                                                    info=args.info, corrupt=0.5, n_values=4))
            mid[planted].a[mid[planted].a == 4 + 7] = 4
            out_mid = {"bad_read_at": 0.5}
            for name, ho in (("with_handover", True), ("without_handover", False)):
                with core.Batch(mid, model, core.make_opts(device=local_rank, time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION,
                                                           visited_per_op=args.visited_per_op, stall_handover=ho)) as bm:
                    bm.run()
                    tmid = time.perf_counter(); bm.run(); tmid = time.perf_counter() - tmid
                    vm, tmm = bm.verdicts(), bm.timing_ns()
                    rm = bm.results()[planted]
                assert int(vm[planted]) == N.INVALID and int((vm == N.VALID).sum()) == B - 1
                out_mid[name] = {"ms_per_step_alone": round(tmid * 1e3, 3), "search_ms": round(tmm["search"] / 1e6, 3), "after_the_search_ms": round(tmm["retries"] / 1e6, 3),
                                 "fail_op": rm["fail_op"], "answered_by": "linear" if rm["analyzer"] == N.ALG_LINEAR else "wgl"}
            assert out_mid["with_handover"]["fail_op"] == out_mid["without_handover"]["fail_op"]
            line["extra"]["bad_read_in_the_middle"] = out_mid
        leg("time to verdict (one history through tbc_check)")
        if on_gpu:      # (one history through tbc_check: needs the device)
            ttv, ttv_dfs, analyzers = [], [], []
            o_sweep = core.make_opts(device=local_rank, want_witness=False, algorithm=N.ALG_COMPETITION)
            o_dfs = core.make_opts(device=local_rank, want_witness=True, algorithm=N.ALG_COMPETITION, search_width=args.width)
            core.check_ops(hists[0], model, o_sweep)                     # first call sizes the persistent context
            parts = []
            for i in range(min(10, B)):
                best, bp = 1e9, None
                for _ in range(3):          # (best of 3 per history, the median over the histories: what round 4's form legs reported)
                    t1 = time.perf_counter(); r = core.check_ops(hists[i], model, o_sweep); dt = (time.perf_counter() - t1) * 1e3
                    if dt < best:
                        best, bp = dt, (r["ns_pack"] / 1e3, r["ns_search"] / 1e3, r["ns_total"] / 1e3)
                ttv.append(best); parts.append(bp)
                analyzers.append(r["analyzer"])
                assert r["valid"] == verdicts[i]
            med = lambda k: round(statistics.median(p[k] for p in parts), 1)
            core.check_ops(hists[0], model, o_dfs)
            for i in range(min(5, B)):
                t1 = time.perf_counter(); r = core.check_ops(hists[i], model, o_dfs); ttv_dfs.append((time.perf_counter() - t1) * 1e3)
            bad = columns.pair_events(synth.register_events(n_ops=args.ops, n_procs=args.procs, seed=12345, busy=args.busy,
                                                            info=args.info, corrupt=0.5))
            t1 = time.perf_counter(); rb = core.check_ops(bad, model, o_sweep); tb_gpu = (time.perf_counter() - t1) * 1e3
            line["extra"]["time_to_verdict_ms"] = {"valid_median": round(statistics.median(ttv), 3), "valid_min": round(min(ttv), 3),
                                                   "answered_by_sweep": sum(a == N.ALG_LINEAR for a in analyzers), "of": len(analyzers),
                                                   "depth_first_with_witness_median": round(statistics.median(ttv_dfs), 3),
                                                   "invalid_example": round(tb_gpu, 3),
                                                   "invalid_example_verdict": rb["valid"], "invalid_example_steps": rb["steps"],
                                                   # where a call's time goes (medians over the histories' best runs): HIP events around the pack and around the
                                                   # sweep (+ what follows it), the rest of the wall time inside tbc_check (H2D, memsets, launches, D2H, the host's
                                                   # composition of the segments' relations), and what the ctypes binding adds around the call
                                                   "split_us": {"device_pack": med(0), "device_sweep": med(1), "rest_inside_tbc_check": round(med(2) - med(0) - med(1), 1),
                                                                "binding": round(statistics.median(ttv) * 1e3 - med(2), 1)}}
        leg("CPU baselines")
        if world == 1 and not args.no_cpu:
            from concurrent.futures import ThreadPoolExecutor
            from oracle import wgl
            S = min(args.cpu_sample, B - 1)          # (the planted history is checked by itself below, not timed in the sample)
            om = {"kind": 1, "init": N.NIL}
            dicts = [hists[i].as_dict() for i in range(S)]
            wgl.check(dicts[0], om, "window", want_witness=False)          # loads / builds the oracle
            S1 = min(64, S)
            tc = time.perf_counter()
            ok1 = sum(wgl.check(d, om, "window", want_witness=False)["valid"] == 1 for d in dicts[:S1])
            tc = time.perf_counter() - tc
            # all host cores: the same sample on a pthread pool inside the C oracle (oracle/many.c), one history per
            # thread at a time -- how stock Knossos would spread independent keys over a thread pool
            cores, cores_visible, cpu_quota = usable_cores()
            reps = max(1, (64 * cores + S - 1) // S)
            work = dicts * reps
            wgl.check_many(work[:cores], om, cores)                                     # spin up / page in
            ta = time.perf_counter()
            oka, started = wgl.check_many(work, om, cores)
            ta = time.perf_counter() - ta
            tb = time.perf_counter()
            rbo = wgl.check(bad.as_dict(), om, "window", want_witness=False, max_steps=50_000_000)
            tb = time.perf_counter() - tb
            # the SAME schedule the kernel runs (wide, K configs per iteration, lookahead, eager reads, twin rule) on one
            # host thread: how much of the speed-up is the algorithm and how much the GPU
            lo = batch_list_order if batch_list_order is not None and batch_list_order >= 16 else wgl.ORACLE_LIST_ORDER[(batch_list_order or 1) - 1]      # TBC_ORDER_* -> the oracle's
            sk = dict(width=1, round_pairs=lanes, rules_at_any_round_size=True, branch_lists=True, list_order=lo) if narrow else dict(width=width if width > 1 else 4, list_order=lo)
            tw = time.perf_counter()
            okw = sum(wgl.check_beam(d, om, want_witness=False, **sk)["valid"] == 1 for d in dicts[:S1])
            tw = time.perf_counter() - tw
            # ... and on every CPU this container may use (oracle/many.c runs wgl_beam.c on the same thread pool)
            twa = time.perf_counter()
            okwa, _ = wgl.check_many(work, om, cores, beam_width=sk["width"], round_pairs=lanes if narrow else 64, list_order=lo, branch_lists=narrow)
            twa = time.perf_counter() - twa
            ts = time.perf_counter()
            oks = sum(wgl.check_sweep(d, om)["valid"] == 1 for d in dicts[:S1])
            ts = time.perf_counter() - ts
            line["cpu_baseline"] = {"value": round(len(work) / ta, 3), "unit": "histories/s", "cores": cores, "kind": "port",
                                    "sample": f"first {S} histories of this batch x {reps}, oracle/wgl_window.c (C restatement of "
                                              f"knossos.wgl, gcc -O2) on {started} pthreads (oracle/many.c) = the CPUs this container may use "
                                              f"({cores_visible} visible, CPU-time quota {cpu_quota}); not stock Knossos (no JVM here)",
                                    "single_thread": {"value": round(S1 / tc, 3), "unit": "histories/s", "cores": 1,
                                                      "ms_per_history": round(tc / S1 * 1e3, 3), "sample": f"first {S1} histories"},
                                    "ms_per_history": round(tc / S1 * 1e3, 3),
                                    "invalid_example_ms": round(tb * 1e3, 3), "invalid_example_verdict": rbo["valid"],
                                    "same_schedule_as_kernel": {"value": round(S1 / tw, 3), "unit": "histories/s", "cores": 1,
                                                                "sample": f"first {S1} histories, oracle/wgl_beam.c with lookahead + eager reads + twin rule, {sk['width']} config(s) per iteration, {lanes if narrow else 64} pairs per round",
                                                                "all_cores": {"value": round(len(work) / twa, 3), "unit": "histories/s", "cores": cores,
                                                                              "sample": f"the all-cores sample ({len(work)} histories) on {cores} pthreads"}},
                                    "level_sweep_on_cpu": {"value": round(S1 / ts, 3), "unit": "histories/s", "cores": 1,
                                                           "sample": f"first {S1} histories, oracle/sweep_ref.c"},
                                    "host_cores_visible": cores_visible, "host_cpu_quota": cpu_quota}
            # element-wise, not sums: history i's verdict on the GPU is history i's verdict on the CPU, in every formulation
            assert okw == oks == ok1 == S1 and all(int(v) == N.VALID for v in verdicts[:S1]), "GPU and oracles disagree on the sample"
            assert [int(v) for v in oka[:S]] == [int(v) for v in verdicts[:S]], "GPU and oracle disagree on the sample"
            assert [int(v) for v in okwa[:S]] == [int(v) for v in verdicts[:S]], "GPU and oracle (wide schedule, thread pool) disagree on the sample"
            if fresh is not None:          # the first eight histories of the first never-seen input, history by history
                for i, h in enumerate(fresh_sample):
                    assert wgl.check(h.as_dict(), om, "window", want_witness=False)["valid"] == int(fresh_verdicts[(0, 0)][i]), "a fresh input: GPU and oracle disagree"
            rp = wgl.check(hists[planted].as_dict(), om, "window", want_witness=False, max_steps=50_000_000)
            assert rp["valid"] == 0 == int(verdicts[planted]), "GPU and oracle disagree on the planted history"
            rgp = core.check_ops(hists[planted], model, o_sweep)
            assert rgp["valid"] == 0 and rgp["fail_op"] == rp["fail_op"], "the planted history's failing op"
            line["extra"]["time_to_verdict_ms"]["vs_cpu_port_single_thread"] = round((tc / S1 * 1e3) / statistics.median(ttv), 2)
            # ... and against the best single-thread CPU formulation of this repo (the kernel's own schedule with the dominance rules)
            line["extra"]["time_to_verdict_ms"]["vs_cpu_same_schedule_single_thread"] = round((tw / S1 * 1e3) / statistics.median(ttv), 2)

            if not args.no_tiers:
                line["extra"]["tiers"] = run_leg("tiers", args, local_rank)

        # the legs below run in processes of their own (run_leg): their inputs are their own, none feeds `value`
        if world == 1 and args.busy2 > 0:
            line["extra"]["workload_2"] = run_leg("workload_2", args, local_rank)
            if args.busy3 > 0:
                line["extra"]["workload_3"] = run_leg("workload_3", args, local_rank)
            if args.info4 > 0:
                line["extra"]["workload_crashed"] = run_leg("workload_crashed", args, local_rank)
        if world == 1 and not args.no_set_full:
            line["extra"]["set_full"] = run_leg("set_full", args, local_rank)
        emit(line)
    for b in batches:
        b.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
