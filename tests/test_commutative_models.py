"""set (knossos.model/set; BASELINE.json config 3) and bank (new model; the reference's
ledger->bank mapping, tests/ledger.clj:89-114; BASELINE.json config 5) as COMMUTATIVE device
models: configs carry no state, a read is checked against the calls completed before the front
plus the open calls already linearized.  CPU: both oracles against brute force, which uses the
ordinary stateful semantics.  GPU: the wide kernel against its oracle, bit for bit."""
import numpy as np
import pytest

from helpers import bank_history, op_tuples, set_history
from jepsen_tigerbeetle_amd import _native as N, core
from jepsen_tigerbeetle_amd.jepsen import checker as jc, independent
from jepsen_tigerbeetle_amd.knossos import _analysis, model as M
from oracle import brute


def enc_set(hist):
    e = _analysis.Encoded(M.set(), hist)
    assert e.native_model[0].kind == N.MODEL_SET
    return e, {"kind": 5, "init": 0, "pool": e.ops.pool, "n_adds": e.n_adds}


def enc_bank(hist, accounts=range(1, 9)):
    e = _analysis.Encoded(M.bank(accounts), hist)
    assert e.native_model[0].kind == N.MODEL_BANK
    return e, {"kind": 6, "init": 0, "pool": e.ops.pool, "n_accounts": len(list(accounts))}


@pytest.mark.parametrize("which", ["set", "bank"])
def test_oracles_against_brute_force(native, oracle, which):
    n_bad = 0
    for seed in range(150):
        if which == "set":
            hist = set_history(7, 3, seed, busy=0.7, info=0.12, corrupt=[None, "lost", None, "phantom"][seed % 4])
            e, om = enc_set(hist)
        else:
            hist = bank_history(7, 3, seed, accounts=[1, 2, 3], busy=0.7, info=0.12, corrupt=seed % 2 == 1)
            e, om = enc_bank(hist, [1, 2, 3])
        bad = brute.first_bad_completion(om, op_tuples(e.ops))
        n_bad += bad is not None
        w = oracle.check(e.ops.as_dict(), om, "window")
        b = oracle.check_beam(e.ops.as_dict(), om, 4)
        for r in (w, b):
            assert r["valid"] == (1 if bad is None else 0), seed
            if bad is not None:
                assert r["fail_op"] == bad, seed
    assert n_bad > 10


@pytest.mark.parametrize("which", ["set", "bank"])
def test_lazy_rule_changes_no_answer(native, oracle, which):
    """The lazy rule of the commutative models (a mutating call is linearized only when it completes at the front or an open read
    could take it): against brute force on small crash-heavy histories, and against the plain search (rule off) where that still
    finishes -- same verdict, same failing op, never more configs."""
    n_bad = saved = 0
    for seed in range(200, 420):
        if which == "set":
            hist = set_history(8, 3, seed, busy=0.8, info=0.3, corrupt=[None, "lost", None, "phantom"][seed % 4])
            e, om = enc_set(hist)
        else:
            hist = bank_history(8, 3, seed, accounts=[1, 2, 3], busy=0.8, info=0.3, corrupt=seed % 2 == 1)
            e, om = enc_bank(hist, [1, 2, 3])
        bad = brute.first_bad_completion(om, op_tuples(e.ops))
        n_bad += bad is not None
        for width in (1, 4):
            r = oracle.check_beam(e.ops.as_dict(), om, width, lazy_commuting=True)
            assert r["valid"] == (1 if bad is None else 0), (seed, width)
            if bad is not None:
                assert r["fail_op"] == bad, (seed, width)
            else:
                brute.check_witness(om, op_tuples(e.ops), [int(x) for x in r["witness"]])
    assert n_bad > 20
    for seed in range(12):
        if which == "set":
            hist = set_history(400, 6, seed, busy=0.3, info=0.02, corrupt=[None, "lost", "phantom"][seed % 3])
            e, om = enc_set(hist)
        else:
            hist = bank_history(400, 6, seed, busy=0.3, info=0.01, corrupt=seed % 2 == 1)
            e, om = enc_bank(hist)
        a = oracle.check_beam(e.ops.as_dict(), om, 8, lazy_commuting=True)
        b = oracle.check_beam(e.ops.as_dict(), om, 8, lazy_commuting=False, max_probes=20_000_000)
        if b["valid"] == -1:
            continue
        assert (a["valid"], a["fail_op"]) == (b["valid"], b["fail_op"]), seed
        assert a["visited"] <= b["visited"]
        saved += b["visited"] - a["visited"]
    assert saved > 0


def test_fallbacks(native):
    # duplicate elements / non-empty initial set / bank that forbids overdrafts: memo table instead
    dup = [{"type": "invoke", "f": "add", "value": 1, "process": 0}, {"type": "ok", "f": "add", "value": 1, "process": 0},
           {"type": "invoke", "f": "add", "value": 1, "process": 1}, {"type": "ok", "f": "add", "value": 1, "process": 1}]
    assert _analysis.Encoded(M.set(), dup).native_model[0].kind == N.MODEL_TABLE
    t = [{"type": "invoke", "f": "transfer", "value": {"debit-acct": 1, "credit-acct": 2, "amount": 1}, "process": 0}]
    assert _analysis.Encoded(M.bank([1, 2], negative_balances=False), t).native_model[0].kind == N.MODEL_TABLE


@pytest.mark.gpu
@pytest.mark.parametrize("which,n_ops,procs,info,corrupt", [
    ("set", 60, 4, 0.05, None), ("set", 60, 4, 0.0, "lost"), ("set", 800, 8, 0.02, None), ("set", 800, 8, 0.0, "phantom"),
    ("set", 3000, 16, 0.01, None), ("set", 6000, 5, 0.02, None), ("set", 6000, 5, 0.02, "lost"),      # (~60 crashed adds: feasible under the lazy rule only)
    ("bank", 60, 4, 0.05, False), ("bank", 60, 4, 0.0, True), ("bank", 800, 8, 0.02, False), ("bank", 800, 8, 0.0, True),
    ("bank", 3000, 16, 0.0, False)])
def test_gpu_matches_oracle(native, oracle, which, n_ops, procs, info, corrupt):
    for seed in range(3):
        if which == "set":
            hist = set_history(n_ops, procs, seed, busy=0.2, info=info, corrupt=corrupt)
            e, om = enc_set(hist)
        else:
            hist = bank_history(n_ops, procs, seed, busy=0.2, info=info, corrupt=corrupt)
            e, om = enc_bank(hist)
        exp = oracle.check_beam(e.ops.as_dict(), om, 8)
        got = core.check_ops(e.ops, e.native_model, core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=8))
        assert got["valid"] == exp["valid"] == (0 if corrupt else 1), seed
        if n_ops <= 800:        # the rule switched off on both sides: the plain search, bit for bit too
            exp0 = oracle.check_beam(e.ops.as_dict(), om, 8, lazy_commuting=False)
            got0 = core.check_ops(e.ops, e.native_model, core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=8, lazy_commuting=False))
            assert (got0["valid"], got0["probes"], got0["visited"]) == (exp0["valid"], exp0["probes"], exp0["visited"]), seed
        assert (got["probes"], got["visited"], got["backtracks"]) == (exp["probes"], exp["visited"], exp["expanded"])
        if exp["valid"] == 1:
            assert np.array_equal(got["witness"], exp["witness"])
        else:
            assert got["fail_op"] == exp["fail_op"]
            w = oracle.check(e.ops.as_dict(), om, "window", max_steps=5_000_000)
            assert w["valid"] == -1 or (w["valid"] == 0 and w["fail_op"] == got["fail_op"])      # (the plain sequential search may not finish a crash-heavy history)
        # ALG_WGL asks for the sequential order; these models exist in the wide kernel only, same verdict
        again = core.check_ops(e.ops, e.native_model, core.make_opts(time_limit_ms=60000))
        assert again["valid"] == exp["valid"]


@pytest.mark.gpu
def test_set_full_shape_through_independent_checker(native):
    """The composition at set_full.clj:155-158 with a :linear entry: 5 ledger keys, one batch launch."""
    t = independent.tuple_
    hist = []
    subs = {k: set_history(400, 6, 100 + k, busy=0.2, info=0.02, corrupt="lost" if k == 3 else None) for k in range(1, 6)}
    for k, h in subs.items():
        for o in h:
            hist.append(dict(o, process=o["process"] * 8 + k, value=t(k, o["value"])))
    c = independent.checker(jc.compose({"linear": jc.linearizable({"model": M.set(), "algorithm": "linear"})}))
    r = c.check({}, hist, {})
    assert r["valid?"] is False and r["failures"] == [3]
    assert r["results"][1]["linear"]["valid?"] is True
