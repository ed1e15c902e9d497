"""The oracle pinned as far as it can be: the reference holds no golden vectors
(PARITY UNPINNED, oracle/oracle_model.h), so the CPU restatements are checked
against (1) hand-derived known-answer histories, (2) an independent brute-force
definition of linearizability on small random histories, (3) each other (the
published DLL/bit-set form vs the windowed-key form the HIP kernel uses)."""
import numpy as np
import pytest

from helpers import load_kats, op_tuples, oracle_model
from jepsen_tigerbeetle_amd import _native as N, columns, synth
from jepsen_tigerbeetle_amd.knossos import _analysis
from helpers import MODELS
from oracle import brute

KATS = load_kats()


@pytest.mark.parametrize("name,model,hist,valid,fail_index", KATS, ids=[k[0] for k in KATS])
def test_kat_oracles(native, oracle, name, model, hist, valid, fail_index):
    enc = _analysis.Encoded(MODELS[model](), hist)
    om = oracle_model(model)
    for alg in ("ref", "window"):
        r = oracle.check(enc.ops.as_dict(), om, alg)
        assert r["valid"] == (1 if valid else 0), (alg, r)
        if not valid:
            assert enc.op_completion(r["fail_op"])["index"] == fail_index
    # brute force agrees with the hand derivation too
    tup = op_tuples(enc.ops)
    bad = brute.first_bad_completion(om, tup)
    assert (bad is None) == valid
    if not valid:
        assert enc.op_completion(bad)["index"] == fail_index


@pytest.mark.parametrize("block", range(4))
def test_oracles_match_brute_force_on_small_histories(native, oracle, block):
    m = {"kind": 1, "init": N.NIL}
    n_invalid = 0
    for seed in range(block * 150, block * 150 + 150):
        ev = synth.register_events(n_ops=7, n_procs=3, seed=seed, busy=0.8, info=0.15, n_values=3,
                                   corrupt=(0.5 if seed % 2 else 0.0))
        ops = columns.pair_events(ev)
        if seed % 3 == 0:   # a second kind of corruption: a stale-but-plausible value
            rd = [i for i in range(len(ops)) if ops.f[i] == N.F_READ and ops.ret_pos[i] != N.POS_CRASHED]
            if rd:
                ops.a[rd[len(rd) // 2]] = (seed // 3) % 3
        tup = op_tuples(ops)
        bad = brute.first_bad_completion(m, tup)
        n_invalid += bad is not None
        for alg in ("ref", "window"):
            r = oracle.check(ops.as_dict(), m, alg)
            assert r["valid"] == (1 if bad is None else 0), (seed, alg)
            if bad is not None:
                assert r["fail_op"] == bad, (seed, alg)
            else:
                assert brute.check_witness(m, tup, list(r["witness"])) == r["final_state"]
    assert n_invalid > 20


@pytest.mark.parametrize("n_ops,procs,info,corrupt,busy", [
    (300, 8, 0.0, 0.0, 0.5), (300, 8, 0.03, 0.0, 0.5), (300, 8, 0.0, 0.7, 0.3), (300, 8, 0.03, 0.2, 0.3),
    (2000, 64, 0.0, 0.0, 0.5), (2000, 64, 0.01, 0.0, 0.5), (1000, 16, 0.0, 0.5, 0.2),
])
def test_windowed_form_equals_published_form(native, oracle, n_ops, procs, info, corrupt, busy):
    """Same traversal: verdict, failing op, witness and every counter."""
    m = {"kind": 1, "init": N.NIL}
    for seed in range(5):
        ops = columns.pair_events(synth.register_events(n_ops=n_ops, n_procs=procs, seed=seed, busy=busy,
                                                        info=info, corrupt=corrupt))
        a = oracle.check(ops.as_dict(), m, "ref", max_steps=2_000_000)
        b = oracle.check(ops.as_dict(), m, "window", max_steps=2_000_000)
        for k in ("valid", "fail_op", "prev_ok_op", "final_state", "n_witness", "steps", "visited", "backtracks", "max_depth"):
            assert a[k] == b[k], (seed, k)
        if a["valid"] == 1:
            assert np.array_equal(a["witness"], b["witness"])
            # a witness is a legal run that respects real-time order
            brute.check_witness(m, op_tuples(ops), list(a["witness"]))


def test_committed_golden_fixtures_still_hold(native, oracle):
    """tests/golden/synth_golden.json (made by tests/golden/make_synth_golden.py): the oracle and the
    seeded generator have not drifted."""
    import hashlib
    import json
    import os
    from helpers import GOLDEN
    cases = json.load(open(os.path.join(GOLDEN, "synth_golden.json")))["cases"]
    m = {"kind": 1, "init": N.NIL}
    for g in cases:
        ops = columns.pair_events(synth.register_events(**g["case"]))
        assert len(ops) == g["n_ops"] and ops.n_process == g["n_process"]
        r = oracle.check(ops.as_dict(), m, "window")
        assert r["valid"] == g["valid"] and r["steps"] == g["sequential"]["steps"]
        if r["valid"] == 1:
            assert hashlib.sha256(r["witness"].astype("<u4").tobytes()).hexdigest()[:16] == g["sequential"]["witness_sha"]
        else:
            assert r["fail_op"] == g["fail_op"]
        w = oracle.check_beam(ops.as_dict(), m, 8, eager_reads=False, twin_rule=False)
        assert (w["probes"], w["visited"]) == (g["wide8"]["probes"], g["wide8"]["visited"])
        w = oracle.check_beam(ops.as_dict(), m, 8)          # library default: + eager reads + twin rule
        assert (w["probes"], w["visited"]) == (g["wide8_rules"]["probes"], g["wide8_rules"]["visited"])
        if w["valid"] == 1:
            assert hashlib.sha256(w["witness"].astype("<u4").tobytes()).hexdigest()[:16] == g["wide8_rules"]["witness_sha"]


def test_lookahead_never_changes_a_verdict(oracle):
    """tbc_opts.lookahead sets aside configs from which one of the next 8 completions can never be
    linearized and expands them only if the search would otherwise end INVALID: the verdict must be
    the one of the plain search (and of the sequential oracle) on valid and invalid histories alike,
    an invalid history must end with the plain search's failing op and counters, and valid ones
    must get cheaper."""
    from jepsen_tigerbeetle_amd import columns, synth
    m = {"kind": 1, "init": N.NIL}
    saved = 0
    for (n, p, busy, info, corrupt) in [(40, 4, 0.5, 0.05, 0.0), (60, 8, 0.8, 0.2, 0.0), (60, 6, 0.6, 0.1, 0.3),
                                        (300, 16, 0.3, 0.05, 0.0), (300, 8, 0.4, 0.02, 0.2), (1000, 32, 0.15, 0.02, 0.0)]:
        for s in range(12 if n <= 300 else 4):
            d = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=500 + s, busy=busy, info=info, corrupt=corrupt)).as_dict()
            seq = oracle.check(d, m, "window", max_steps=3_000_000, want_witness=False)
            for K in (2, 16):
                a = oracle.check_beam(d, m, K, max_probes=3_000_000, want_witness=False, lookahead=True, eager_reads=False, twin_rule=False)
                b = oracle.check_beam(d, m, K, max_probes=3_000_000, want_witness=False, lookahead=False, eager_reads=False, twin_rule=False)
                if -1 in (a["valid"], b["valid"]):
                    continue
                assert a["valid"] == b["valid"], (n, p, s, K)
                if seq["valid"] != -1:
                    assert a["valid"] == seq["valid"], (n, p, s, K)
                if a["valid"] == 0:
                    # set-aside configs are expanded before the verdict: the exact search's numbers
                    assert a["fail_op"] == b["fail_op"] == seq["fail_op"]
                    assert (a["visited"], a["probes"], a["expanded"]) == (b["visited"], b["probes"], b["expanded"])
                else:
                    saved += b["probes"] - a["probes"]
    assert saved > 0
    # and the rule itself never called a live config dead: no linearization was found only after the
    # set-aside configs had to be taken up (randomized sweeps of 17,000 histories while developing: 0)
    import ctypes as C
    late = oracle.lib().wgl_beam_late_valid
    late.restype = C.c_uint64
    assert late() == 0


def test_dominance_rules_never_change_a_verdict(oracle):
    """tbc_opts.dominance (eager reads, twin rule; kernel: wgl_beam.hip): a read that is viable now
    can be linearized now without loss of generality, so every child absorbs all of them.  Same verdict
    and failing op as the plain search, witnesses that replay legally under the independent checker,
    and far less work with no heavy tail."""
    from jepsen_tigerbeetle_amd import columns, synth
    m = {"kind": 1, "init": N.NIL}
    for (n, p, busy, info, corrupt) in [(40, 4, 0.5, 0.05, 0.0), (60, 8, 0.8, 0.2, 0.3), (300, 16, 0.3, 0.05, 0.0),
                                        (300, 8, 0.4, 0.02, 0.3), (1000, 32, 0.15, 0.02, 0.0)]:
        for s in range(10 if n <= 300 else 4):
            ops = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=300 + s, busy=busy, info=info, corrupt=corrupt))
            d = ops.as_dict()
            plain = oracle.check_beam(d, m, 4, max_probes=3_000_000, lookahead=False, eager_reads=False, twin_rule=False)
            for la, twin in ((False, False), (True, False), (True, True)):
                # twin rule: of several open calls with the same effect the one completing first goes first
                e = oracle.check_beam(d, m, 4, max_probes=3_000_000, lookahead=la, eager_reads=True, twin_rule=twin)
                if -1 in (plain["valid"], e["valid"]):
                    continue
                assert e["valid"] == plain["valid"], (n, p, s, la, twin)
                if e["valid"] == 0:
                    assert e["fail_op"] == plain["fail_op"], (n, p, s, la, twin)
                else:
                    w = [int(x) for x in e["witness"]]
                    assert brute.check_witness(m, op_tuples(ops), w) == e["final_state"], (n, p, s, la)
    big = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=3659, busy=0.1)).as_dict()   # hardest of 4,096 seeds
    a = oracle.check_beam(big, m, 4, want_witness=False, eager_reads=False, twin_rule=False)
    b = oracle.check_beam(big, m, 4, want_witness=False, eager_reads=True, twin_rule=False)
    assert a["valid"] == b["valid"] == 1 and b["rounds"] * 4 < a["rounds"]


def test_crashed_twin_shortcut_agrees_with_the_list_walk(oracle):
    """Groundwork for the next kernel change: under the twin rule crashed calls of one effect are
    linearized in invocation order, so a crashed candidate only has to look at the live entries of its front and at
    the PREVIOUS crashed call of its effect.  wgl_beam.c evaluates both forms on every such test when asked to."""
    from jepsen_tigerbeetle_amd import columns, synth
    m = {"kind": 1, "init": -2147483648}
    checked = 0
    for seed in range(4):
        for (n, p, info, busy, corrupt) in ((1500, 24, 0.03, 0.3, 0.0), (1000, 16, 0.05, 0.5, 0.3)):
            ops = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=seed, busy=busy, info=info, corrupt=corrupt))
            for w in (2, 16):
                r = oracle.check_beam(ops.as_dict(), m, w, max_probes=1_000_000, want_witness=False, twin_selfcheck=True)
                checked += r["twin_checked"]
                assert r["twin_mismatch"] == 0
    assert checked > 100_000


def test_thread_pool_runs_both_restatements(oracle):
    """oracle/many.c: the sequential restatement and the wide schedule on a pthread pool give the verdicts of the
    one-at-a-time calls (bench.py's cpu_baseline uses both)."""
    from jepsen_tigerbeetle_amd import columns, synth
    m = {"kind": 1, "init": -2147483648}
    hs = [columns.pair_events(synth.register_events(n_ops=600, n_procs=12, seed=s, busy=0.3, info=0.01 if s % 3 == 0 else 0.0,
                                                    corrupt=0.5 if s % 4 == 1 else 0.0)).as_dict() for s in range(24)]
    one = [oracle.check(h, m, "window", want_witness=False)["valid"] for h in hs]
    assert 0 in one and 1 in one
    for bw in (0, 2, 4):
        v, started = oracle.check_many(hs, m, 4, beam_width=bw)
        assert started >= 1 and list(v) == one, bw


def test_narrow_round_schedules_keep_verdict_and_failing_op(oracle):
    """The schedules a wavefront shared by several histories would run: 8 / 16 / 32 pairs per round,
    1 or 2 configs per round, lookahead + eager reads + twin rule.  Against the sequential restatement on histories
    it finishes (and brute force through it, test_oracles_match_brute_force): same verdict, same failing op, a legal
    witness; and the round size changes rounds only -- probes, new configs and the witness are the 64-pair schedule's."""
    from helpers import op_tuples
    from jepsen_tigerbeetle_amd import columns, synth
    m = {"kind": 1, "init": N.NIL}
    n_checked = 0
    for (n, p, busy, info, corrupt) in [(60, 6, 0.6, 0.1, 0.3), (300, 16, 0.3, 0.02, 0.0), (300, 8, 0.4, 0.0, 0.3), (1500, 32, 0.15, 0.0, 0.0)]:
        for s in range(6 if n <= 300 else 3):
            ops = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=900 + s, busy=busy, info=info, corrupt=corrupt))
            d = ops.as_dict()
            seq = oracle.check(d, m, "window", max_steps=3_000_000, want_witness=False)
            if seq["valid"] == -1:
                continue
            for K in (1, 2):
                ref = oracle.check_beam(d, m, K, max_probes=3_000_000, rules_at_any_round_size=True)
                for rp in (8, 16, 32):
                    r = oracle.check_beam(d, m, K, max_probes=3_000_000, round_pairs=rp, rules_at_any_round_size=True)
                    assert r["valid"] == seq["valid"] == ref["valid"], (n, p, s, K, rp)
                    if r["valid"] == 0:
                        assert r["fail_op"] == seq["fail_op"]
                    else:
                        brute.check_witness(m, op_tuples(ops), [int(x) for x in r["witness"]])
                    if info == 0.0 and r["valid"] == 1:      # (with crashed calls a round's pairs are cut differently: counters may differ)
                        assert (r["probes"], r["visited"]) == (ref["probes"], ref["visited"]) and r["rounds"] >= ref["rounds"]
                    n_checked += 1
    assert n_checked > 60


def test_list_order_never_changes_a_verdict(native, oracle):
    """The schedules behind tbc_opts.list_order (a front's candidates in order of completion, with the :write calls last, with a :write
    24 ranks later -- the library's default; wgl_beam_set_list_order) are schedules of the SAME search: verdict and failing op are
    the sequential restatement's on random histories, valid and invalid, with crashed calls -- at one config a round over 8 pairs
    (the narrow kernel's), at 2 and 4 configs a round (the wide kernel's)."""
    import random
    rng = random.Random(1)
    cas = {"kind": 1, "init": N.NIL}
    n_invalid = n = 0
    for it in range(140):
        shape = dict(n_ops=rng.choice([10, 30, 80, 200]), n_procs=rng.choice([2, 4, 8, 16]), busy=rng.choice([0.3, 0.7, 1.0]),
                     info=rng.choice([0, 0, 0.05]), corrupt=rng.choice([0, 0.3, 0.7]), n_values=rng.choice([2, 5]))
        d = columns.pair_events(synth.register_events(seed=rng.randrange(10 ** 6), **shape)).as_dict()
        ref = oracle.check(d, cas, "window", max_steps=5_000_000, want_witness=False)
        if ref["valid"] == -1:
            continue
        n += 1
        n_invalid += ref["valid"] == 0
        for width, kw in ((1, dict(round_pairs=8, rules_at_any_round_size=True, branch_lists=True)), (4, {}), (2, {})):
            for order in (1, oracle.ORACLE_LIST_ORDER[2], oracle.ORACLE_LIST_ORDER[16 + 24]):          # (PackOpenArgs.list_order 1, 2, 16 + 24)
                r = oracle.check_beam(d, cas, width, want_witness=False, list_order=order, max_probes=20_000_000, **kw)
                if r["valid"] == -1:
                    continue
                assert r["valid"] == ref["valid"], (it, width, order, shape)
                if ref["valid"] == 0:
                    assert r["fail_op"] == ref["fail_op"], (it, width, order, shape)
    assert n > 100 and n_invalid > 40
