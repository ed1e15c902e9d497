"""GPU parity sweep (run on the MI355X box via gpurun): HIP search vs the CPU
restatement (oracle/wgl_window.c) on seeded synthetic cas-register histories.
Compares verdict, failing op, witness, final state and traversal counters.
Prints one line per mismatch and a summary; exits non-zero on any mismatch."""
import argparse
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

import jepsen_tigerbeetle_amd  # noqa: E402,F401
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth  # noqa: E402
from oracle import wgl  # noqa: E402


def compare(tag, ops, gpu, ora):
    bad = []
    if gpu["valid"] != ora["valid"]:
        bad.append(f"valid gpu={gpu['valid']} oracle={ora['valid']}")
    else:
        if ora["valid"] == 0:
            if gpu["fail_op"] != ora["fail_op"]:
                bad.append(f"fail_op gpu={gpu['fail_op']} oracle={ora['fail_op']}")
            po = None if ora["prev_ok_op"] == N.NO_OP else ora["prev_ok_op"]
            if gpu["prev_ok_op"] != po:
                bad.append(f"prev_ok gpu={gpu['prev_ok_op']} oracle={po}")
        if ora["valid"] == 1:
            if gpu["final_state"] != ora["final_state"]:
                bad.append(f"final_state gpu={gpu['final_state']} oracle={ora['final_state']}")
            if gpu["witness"] is None or not np.array_equal(gpu["witness"], ora["witness"]):
                bad.append("witness differs")
        for k in ("steps", "visited", "probes", "backtracks", "max_depth"):
            if gpu[k] != ora[k]:
                bad.append(f"{k} gpu={gpu[k]} oracle={ora[k]}")
    if bad:
        print(f"MISMATCH {tag} n={len(ops)} W={ops.n_process}: " + "; ".join(bad), flush=True)
    return not bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    print("devices:", N.lib().tbc_device_count(), flush=True)
    model = {"kind": N.MODEL_CAS_REGISTER, "init": N.NIL}
    gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    cases = []
    for n_ops, procs in ((8, 3), (40, 4), (200, 8), (1000, 16), (3000, 64)):
        for seed in range(6 if args.quick else 12):
            cases.append((n_ops, procs, seed, 0.0, 0.0, 0.5))            # valid, no crashes
            cases.append((n_ops, procs, seed, 0.02, 0.0, 0.5))           # valid, 2 % crashed ops
            cases.append((n_ops, procs, seed, 0.0, 0.6, min(0.5, 3.0 / procs)))   # violation late
            cases.append((n_ops, procs, seed, 0.02, 0.1, min(0.5, 3.0 / procs)))  # violation early, crashes
    hists, tags = [], []
    for (n_ops, procs, seed, info, corrupt, busy) in cases:
        ev = synth.register_events(n_ops=n_ops, n_procs=procs, seed=seed, busy=busy, info=info, corrupt=corrupt)
        hists.append(columns.pair_events(ev))
        tags.append(f"n{n_ops}p{procs}s{seed}i{info}c{corrupt}")
    t = time.time()
    oras = [wgl.check(h.as_dict(), model, "window", max_steps=3_000_000) for h in hists]
    print(f"oracle: {len(hists)} histories in {time.time() - t:.2f}s; "
          f"valid={sum(o['valid'] == 1 for o in oras)} invalid={sum(o['valid'] == 0 for o in oras)} "
          f"unknown={sum(o['valid'] == -1 for o in oras)}", flush=True)
    # single-history entry point on a few
    ok = True
    for i in list(range(0, len(hists), max(1, len(hists) // 12))):
        if oras[i]["valid"] == -1:
            continue
        g = core.check_ops(hists[i], gm, core.make_opts(time_limit_ms=20000))
        ok &= compare("single:" + tags[i], hists[i], g, oras[i])
    print("single-history pass:", ok, flush=True)
    # batch entry point on all (grouped by mask width so wide windows do not slow narrow ones)
    t = time.time()
    keep = [i for i in range(len(hists)) if oras[i]["valid"] != -1]
    with core.Batch([hists[i] for i in keep], gm, core.make_opts(time_limit_ms=60000)) as b:
        b.run()
        res = b.results()
        print("batch timing ns:", b.timing_ns(), "counters:", b.counters(), flush=True)
    for j, i in enumerate(keep):
        ok &= compare("batch:" + tags[i], hists[i], res[j], oras[i])
    print(f"batch pass: {ok} ({len(keep)} histories, {time.time() - t:.2f}s)", flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
