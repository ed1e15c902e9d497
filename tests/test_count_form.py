"""The COUNT FORM of the search (oracle/wgl_count.c): crashed calls as a count per effect class, process slots re-used,
the lazy ("hot") rule, the Pareto rule, the relaxed refutation and the prefix target -- pinned, like every other
restatement here, against the brute-force definition and the sequential restatement (PARITY UNPINNED: oracle_model.h)."""
import random

import numpy as np
import pytest

from helpers import op_tuples
from jepsen_tigerbeetle_amd import _native as N, columns, synth
from oracle import brute

CAS = {"kind": 1, "init": N.NIL}


def verdict_pipeline(oracle, ops, width, budget):
    v, fo, _, _, how = oracle.check_count_pipeline(ops, CAS, width=width, budget=budget)
    return v, fo, how


def crashy(seed, rng, small):
    n_ops = rng.choice([6, 7, 8] if small else [12, 20, 30, 40, 60, 100])
    procs = rng.choice([2, 3] if small else [2, 3, 4, 6, 8])
    ev = synth.register_events(n_ops=n_ops, n_procs=procs, seed=seed, busy=rng.choice([0.3, 0.6, 0.9]), info=rng.choice([0.15, 0.3, 0.5]),
                               n_values=rng.choice([2, 3, 5]), corrupt=rng.choice([0.0, 0.0, 0.4, 0.8]))
    ops = columns.pair_events(ev)
    if seed % 2 == 0:      # a stale-but-plausible value
        rd = [i for i in range(len(ops)) if ops.f[i] == N.F_READ and ops.a[i] != N.NIL and ops.ret_pos[i] != N.POS_CRASHED]
        if rd:
            ops.a[rng.choice(rd)] = rng.randrange(3)
    return ops


@pytest.mark.parametrize("block", range(3))
def test_count_form_matches_brute_force(native, oracle, block):
    rng = random.Random(block)
    n_invalid = n_class_steps = 0
    for seed in range(block * 200, block * 200 + 200):
        ops = crashy(seed, rng, small=True)
        tup = op_tuples(ops)
        bad = brute.first_bad_completion(CAS, tup)
        n_invalid += bad is not None
        for width in (1, 4):
            for look in (True, False):
                r = oracle.check_count(ops.as_dict(), CAS, width=width, lookahead=look)
                assert r["valid"] == (1 if bad is None else 0), (seed, width, look)
                if bad is not None:
                    assert r["fail_op"] == bad, (seed, width, look)
                else:
                    assert brute.check_witness(CAS, tup, [int(x) for x in r["witness"]]) == r["final_state"], (seed, width, look)
                n_class_steps += r["class_steps"]
    assert n_invalid > 20 and n_class_steps > 100


@pytest.mark.parametrize("block", range(3))
def test_count_form_and_its_pipeline_match_the_sequential_search(native, oracle, block):
    rng = random.Random(100 + block)
    how = {}
    for seed in range(block * 300, block * 300 + 300):
        ops = crashy(seed, rng, small=False)
        e = oracle.check(ops.as_dict(), CAS, "window", want_witness=False, max_steps=3_000_000)
        if e["valid"] == -1:
            continue
        r = oracle.check_count(ops.as_dict(), CAS, width=rng.choice([1, 2, 4, 16]))
        assert r["valid"] == e["valid"], seed
        if e["valid"] == 0:
            assert r["fail_op"] == e["fail_op"], seed
        else:
            assert brute.check_witness(CAS, op_tuples(ops), [int(x) for x in r["witness"]]) == r["final_state"], seed
        v, fo, h = verdict_pipeline(oracle, ops.as_dict(), rng.choice([1, 4]), rng.choice([1, 5, 20, 100]))
        how[h] = how.get(h, 0) + 1
        assert v == e["valid"] and (v == 1 or fo == e["fail_op"]), (seed, h)
    assert how.get("prefix", 0) > 50 and how.get("exact", 0) > 50, how


def test_slots_are_reused_and_masks_stay_narrow(native, oracle):
    ops = columns.pair_events(synth.register_events(n_ops=3000, n_procs=16, seed=5, busy=0.3, info=0.05))
    assert ops.n_process > 64                      # the mask form needs a slot per crashed call
    r = oracle.check_count(ops.as_dict(), CAS, width=4, want_slots=True)
    assert r["valid"] == 1 and r["n_slots"] <= 16 and r["n_classes"] >= 10 and r["count_bits"] <= 128
    live = ops.ret_pos != N.POS_CRASHED
    assert (r["slots"][~live] == 0xFFFFFFFF).all() and (r["slots"][live] < 16).all()
    # two live calls that overlap never share a slot
    order = np.argsort(ops.inv_pos)
    open_until = {}
    for i in order:
        if not live[i]:
            continue
        s = int(r["slots"][i])
        assert open_until.get(s, -1) < ops.inv_pos[i], i
        open_until[s] = int(ops.ret_pos[i])
    assert brute.check_witness(CAS, op_tuples(ops), [int(x) for x in r["witness"]]) == r["final_state"]


def test_the_bench_tiers_get_verdicts(native, oracle):
    """BASELINE.md section 3's crashed tiers (10k ops, 64 processes, 1 % / 5 % crashed; as generated / one bad read)."""
    for info in (0.01, 0.05):
        for corrupt in (0.0, 0.5):
            hh = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=4242, busy=0.1, info=info, corrupt=corrupt)).as_dict()
            v, fo, how = verdict_pipeline(oracle, hh, 4, 32 * len(hh["f"]))
            assert v == (0 if corrupt else 1), (info, corrupt)
            assert how == ("prefix" if corrupt else "exact")
            if corrupt:
                assert hh["a"][fo] == 12          # the planted value (n_values + 7)


def test_the_pipeline_with_the_relaxed_sweep_in_front(native, oracle):
    """check_count_pipeline(relaxed_sweep=True) -- the library's passes for a handful of histories -- gives the verdict and the failing op of
    the pipeline without it and of the sequential restatement; the relaxed level sweep and the relaxed depth-first search are the same
    relaxation (same verdict, same completion)."""
    import random
    rng = random.Random(7)
    n = refuted = 0
    for it in range(250):
        d = columns.pair_events(synth.register_events(n_ops=rng.choice([10, 30, 60, 120]), n_procs=rng.choice([2, 4, 8]), seed=rng.randrange(10 ** 6),
                                                      busy=rng.choice([0.3, 0.7, 1.0]), info=rng.choice([0.05, 0.15, 0.3, 0.5]), corrupt=rng.choice([0, 0.3, 0.7]),
                                                      n_values=rng.choice([2, 5]))).as_dict()
        a = oracle.check_count_pipeline(d, CAS, width=4, budget=rng.choice([1, 10, 100, None]), relaxed_sweep=True)
        b = oracle.check_count_pipeline(d, CAS, width=4)
        if a is None or b is None:
            assert a is None and b is None
            continue
        ref = oracle.check(d, CAS, "window", max_steps=3_000_000, want_witness=False)
        if ref["valid"] == -1:
            continue
        n += 1
        assert (a[0], a[1] if a[0] == 0 else None) == (b[0], b[1] if b[0] == 0 else None) == (ref["valid"], ref["fail_op"] if ref["valid"] == 0 else None), (it, a[4], b[4])
        refuted += a[4].startswith("relaxed sweep")
        sw = oracle.check_sweep(d, CAS, relaxed=True, seg_target=rng.choice([0, 8, 32]))
        rl = oracle.check_count(d, CAS, width=4, relaxed=True, want_witness=False)
        assert (sw["valid"], sw["fail_op"] if sw["valid"] == 0 else None) == (rl["valid"], rl["fail_op"] if rl["valid"] == 0 else None), it
    assert n > 150 and refuted > 50
