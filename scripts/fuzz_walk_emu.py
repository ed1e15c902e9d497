"""The front walk with lane = front (csrc/open_walk_impl.h) under the wavefront emulator on random small batches -- 1 .. 64 process slots,
quiet to always-busy processes, crashed calls (they keep their slots: many candidates whose completion rank is "never"), every table
format, full and branch lists, every list order (slot, completion, writes last, a :write W ranks later with W in 1 .. 300) -- every word
against tables built on the host from the definitions (tests/emu/host_tables.h).  usage: fuzz_walk_emu.py [rounds] [seed]
Round 5 (after the list-order places went to unique keys in registers): 400 rounds over two seeds, no mismatch."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jepsen_tigerbeetle_amd  # noqa: F401
from jepsen_tigerbeetle_amd import columns, synth
import emu

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for it in range(rounds):
    hs = []
    for k in range(rng.choice([1, 3, 5])):
        n = rng.choice([1, 3, 40, 64, 65, 130, 300, 450, 700])
        p = rng.choice([1, 2, 5, 16, 33, 48, 58, 64])
        info = rng.choice([0.0, 0.0, 0.01, 0.05]) if p < 60 else 0.0
        h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=rng.randrange(10 ** 6), busy=rng.choice([0.1, 0.5, 0.9, 1.0]), info=info,
                                                      n_values=rng.choice([2, 5])))
        if h.n_process <= 64 and len(h.f):
            hs.append(h)
    if not hs:
        continue
    by_ret = rng.choice([0, 1, 2, 16 + rng.choice([1, 2, 3, 24, 24, 60, 300])])
    front, branch = rng.choice([("plain", False), ("wide", False), ("wide", True), ("compact", True), ("compact", False)])
    twin, look = rng.random() < 0.85, rng.random() < 0.85
    r = emu.walk_check(hs, 8, twin=twin, look=look, branch=branch, front=front, by_ret=by_ret)
    if r is not None:
        bad += 1
        print("MISMATCH", it, dict(by_ret=by_ret, front=front, branch=branch, twin=twin, look=look), r, flush=True)
print("rounds", rounds, "mismatches", bad)
sys.exit(1 if bad else 0)
