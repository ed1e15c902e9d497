"""The lazy rule of the commutative models (set, bank; tbcheck.h TBC_DOM_NO_LAZY_COMMUTING) against the plain wide search: verdict and failing
op on random crash-heavy histories, rule on vs off (oracle/wgl_beam.c).  usage: stress_lazy_commuting.py <first seed> <last seed>
(round 4: seeds 1000 .. 24999 on six processes: 24,000 histories compared, 9,475 of them invalid, 4.9 million configurations saved, 0 mismatches)."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jepsen_tigerbeetle_amd  # noqa
from helpers import bank_history, set_history
from jepsen_tigerbeetle_amd.knossos import _analysis, model as M
from oracle import wgl

lo, hi = int(sys.argv[1]), int(sys.argv[2])
rng = random.Random(lo)
tot = bad = inv = skipped = 0
saved = 0
t0 = time.time()
for seed in range(lo, hi):
    n, procs = rng.choice([10, 20, 40, 80]), rng.choice([2, 3, 4, 6])
    busy, info = rng.choice([0.3, 0.6, 0.9]), rng.choice([0.05, 0.15, 0.3])
    if seed % 2:
        h = set_history(n, procs, seed, busy=busy, info=info, corrupt=rng.choice([None, None, "lost", "phantom"]))
        e = _analysis.Encoded(M.set(), h)
        om = {"kind": 5, "init": 0, "pool": e.ops.pool, "n_adds": e.n_adds}
    else:
        h = bank_history(n, procs, seed, accounts=[1, 2, 3, 4], busy=busy, info=info, corrupt=rng.random() < 0.4)
        e = _analysis.Encoded(M.bank([1, 2, 3, 4]), h)
        om = {"kind": 6, "init": 0, "pool": e.ops.pool, "n_accounts": 4}
    if e.native_model[0].kind not in (5, 6):
        skipped += 1
        continue
    d = e.ops.as_dict()
    off = wgl.check_beam(d, om, rng.choice([1, 4, 16]), lazy_commuting=False, max_probes=3_000_000, want_witness=False)
    if off["valid"] == -1:
        skipped += 1
        continue
    on = wgl.check_beam(d, om, rng.choice([1, 4, 16]), lazy_commuting=True, want_witness=False)
    tot += 1; inv += off["valid"] == 0; saved += off["visited"] - on["visited"]
    if (on["valid"], on["fail_op"]) != (off["valid"], off["fail_op"]):
        bad += 1
        print("MISMATCH", seed, om["kind"], off["valid"], on["valid"], off["fail_op"], on["fail_op"], flush=True)
print("range", lo, hi, "compared", tot, "mismatches", bad, "invalid", inv, "skipped (plain search over its limit, or another model)", skipped,
      "configs saved", saved, "%.0fs" % (time.time() - t0), flush=True)
