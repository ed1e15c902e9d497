"""The fused pack (csrc/pack_one_impl.h: batch form and one-history form with open counts) under the workgroup emulator on random
small batches -- sizes around the chunk and wavefront boundaries, crashed calls in both forms, a broken row in one history of five
(invocations out of order, a live call of an unknown process, an op the model does not know, two completions on one row), list
arenas too small -- every word against the restatements (tests/emu/emu_pack.cpp, host_tables.h).  usage: fuzz_pack_emu.py [rounds] [seed]
Round 4: 400 rounds x 3 forms, no mismatch (the one it found was the harness's own: host_tables.h indexed by an unknown process).
Round 5: + the 64-slot batch geometry (Batch64Geo) in the mask and the count form."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jepsen_tigerbeetle_amd  # noqa: F401
from jepsen_tigerbeetle_amd import columns, synth
import emu

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for it in range(rounds):
    hs = []
    for k in range(6):
        n = rng.choice([1, 2, 5, 30, 64, 65, 127, 200, 513, 900])
        p = rng.choice([1, 2, 3, 8, 17, 40, 64])
        h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=rng.randrange(10 ** 6), busy=rng.choice([0.1, 0.5, 1.0]), info=rng.choice([0.0, 0.0, 0.05, 0.3])))
        d = dict(h.as_dict())
        for key in ("f", "a", "b", "process", "inv_pos", "ret_pos"):
            d[key] = np.array(d[key], copy=True)
        m = len(d["f"])
        live = np.flatnonzero(d["ret_pos"] != 0xFFFFFFFF)
        r = rng.random()
        if m >= 4 and r < 0.08:
            i = rng.randrange(m - 1); d["inv_pos"][i], d["inv_pos"][i + 1] = d["inv_pos"][i + 1], d["inv_pos"][i]
        elif len(live) and r < 0.14:
            d["process"][int(rng.choice(list(live)))] = d["n_process"] + 3
        elif m >= 2 and r < 0.2:
            d["f"][rng.randrange(m)] = 9
        elif len(live) >= 2 and r < 0.26:
            d["ret_pos"][live[-1]] = d["ret_pos"][live[0]]
        if d["n_process"] <= 256:
            hs.append(d)
    if not hs:
        continue
    cap = rng.choice([0, 0, 0, 40, 400])
    hs64 = [d for d in hs if d["n_process"] <= 64]              # Batch64Geo: at most 64 process slots (a crashed call of the mask form takes one)
    for kw in ({"branch": bool(it & 1), "lst_cap": cap}, {"count": True, "branch": bool(it & 2)}, {"one": True, "branch": True, "lst_cap": cap},
               {"slots64": True, "branch": bool(it & 1), "lst_cap": cap}, {"slots64": True, "count": True, "branch": bool(it & 2)}):
        if kw.get("slots64") and not hs64:
            continue
        r = emu.pack_wg_check(hs64 if kw.get("slots64") else hs, seed=it, per_launch=rng.choice([0, 1, 4]), **kw)
        if r is not None:
            bad += 1
            print("MISMATCH", it, kw, r, flush=True)
print("rounds", rounds, "mismatches", bad)
sys.exit(1 if bad else 0)
