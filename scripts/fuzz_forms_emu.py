"""The forms of the level sweep (plain walk, compact walk, narrow passes by one wavefront) and the lists in every order
of the narrow search under the emulators on random small histories -- sizes, concurrency, planted bad reads (inside the value domain),
crashed calls, wavefronts per workgroup, set sizes, segment lengths, interleaving seeds -- every record / counter against the oracle.
usage: fuzz_forms_emu.py [rounds] [seed]      Round 4: 650 rounds over five seeds; one find -- the ring form's missing barrier (wavefronts at
different workgroup barriers: a hang on the device), fixed -- and no mismatch since."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jepsen_tigerbeetle_amd  # noqa: F401
from jepsen_tigerbeetle_amd import _native as N, columns, synth
import test_sweep_wg_emu as TS
import test_narrow_emu as TN

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad = 0
for it in range(rounds):
    n = rng.choice([20, 60, 150, 300, 500]); p = rng.choice([2, 4, 8, 16, 24]); busy = rng.choice([0.2, 0.5, 0.8, 1.0])
    corrupt = rng.choice([0.0, 0.0, 0.3, 0.7]); info = rng.choice([0.0, 0.0, 0.0, 0.02])
    seed = rng.randrange(10 ** 6)
    h = TN._in_domain(n, p, seed, busy, info, corrupt)
    # ---- the sweep's forms
    form = rng.choice([{"compact": True}, {"compact": 2}, {"compact": 2}, {}])
    # (the geometries tests/emu/emu_sweep.cpp instantiates for each form)
    if form.get("compact"):
        waves, cap = rng.choice([(2, 512), (2, 1024), (4, 1024), (8, 1024), (8, 2048), (16, 2048)])
    else:
        waves, cap = rng.choice([(2, 1024), (4, 1024), (8, 1024), (4, 512), (8, 2048), (16, 2048)])
    seg = rng.choice([0, 16, 32, 32])
    if os.environ.get("TBC_FUZZ_TRACE"):        # (an emulator abort -- a divergent barrier -- ends the process: name the case first)
        print("case", it, (n, p, busy, corrupt, info, seed), form, waves, cap, seg, flush=True)
    try:
        try:
            TS._compare(h, seg, 6, waves, cap=cap, seed=rng.randrange(1000), **form)
        except AssertionError as e:
            if e.args:
                raise
            TS._compare(h, seg, 6, waves, cap=cap, seed=rng.randrange(1000), expect_overflow=True, **form)      # (the bare assert: a level outgrew the sets)
    except Exception as e:
        bad += 1; print("SWEEP MISMATCH", it, (n, p, busy, corrupt, info, seed), form, waves, cap, seg, repr(e)[:300], flush=True)
    # ---- the narrow kernel over the fronts' lists in every order
    if h.n_process <= 64:
        try:
            by_ret = ((16 + rng.choice([1, 4, 16, 24, 200]) if it & 32 else 2) if it & 16 else 1) if it & 2 else 0        # (2: ... with the :write calls last, TBC_NARROW_ORDER=2) the fronts' lists in order of completion (a witness's absorbed reads in that order too)
            TN.compare([h], TN.CAS, rng.choice([8, 16, 32]), tag="fuzz", pool_words=8_000_000, entries_per_op=rng.choice([1, 4, 8]),
                       want_witness=bool(it & 1), epochs=rng.choice([0, 0, 2]), by_ret=by_ret)
        except Exception as e:
            a = e.args[0] if e.args else None
            if isinstance(a, tuple) and len(a) == 4 and a[1] == -1 and a[3] == 3:
                continue                                  # (the emulator's growth pool ran out under an exhaustive search: a limit, not a mismatch)
            bad += 1; print("ORDER MISMATCH", it, (n, p, busy, corrupt, info, seed), repr(e)[:300], flush=True)
print("rounds", rounds, "mismatches", bad)
sys.exit(1 if bad else 0)
