"""multi-register (knossos.model/multi-register; BASELINE.json config 4's model): the device model
(state = 4 bits per key, txn micro-ops in the value pool) against brute force and the oracles on
CPU, and the HIP kernels against the oracle on the GPU."""
import numpy as np
import pytest

from helpers import multi_register_history, op_tuples
from jepsen_tigerbeetle_amd import _native as N, core
from jepsen_tigerbeetle_amd.knossos import _analysis, linear, model as M, wgl as kwgl
from oracle import brute


def encode(hist):
    enc = _analysis.Encoded(M.multi_register({}), hist)
    assert enc.native_model[0].kind == N.MODEL_MULTI_REGISTER       # direct device model, not the memo table
    return enc, {"kind": 4, "init": 0, "pool": enc.ops.pool}


def test_small_histories_against_brute_force(native, oracle):
    n_bad = 0
    for seed in range(160):
        hist = multi_register_history(7, 3, seed, n_keys=2, n_values=2, busy=0.7, info=0.1, corrupt=seed % 2 == 1)
        if seed % 4 == 2:   # a plausible-but-stale value instead of an impossible one
            oks = [o for o in hist if o["type"] == "ok"]
            for o in oks[len(oks) // 2:]:
                rd = [m for m in o["value"] if m[0] == "r"]
                if rd:
                    rd[0][2] = seed % 2
                    break
        enc, om = encode(hist)
        bad = brute.first_bad_completion(om, op_tuples(enc.ops))
        n_bad += bad is not None
        for alg in ("ref", "window"):
            r = oracle.check(enc.ops.as_dict(), om, alg)
            assert r["valid"] == (1 if bad is None else 0), (seed, alg)
            if bad is not None:
                assert r["fail_op"] == bad
        rb = oracle.check_beam(enc.ops.as_dict(), om, 4)
        assert rb["valid"] == (1 if bad is None else 0) and (bad is None or rb["fail_op"] == bad)
    assert n_bad > 15


def test_the_multi_register_rules_change_no_answer(native, oracle):
    """The two rules the wide kernel applies to multi-register (csrc kRuleTxnEager / kRuleTxnIndep; specified in oracle/wgl_beam.c):
    eager txns -- an open txn of micro-reads only that the state allows is linearized at once --, and txn independence -- the candidates
    of a config are the closure of the call completing at its front under "one writes a key the other reads or writes".  Each alone and
    both together: same verdict and failing op as brute force on small crash-heavy histories (2 - 4 keys: closures that do and do not
    cover the open calls), the witness replayed by the independent checker, fewer probes on larger ones."""
    n_bad = 0
    for seed in range(300):
        hist = multi_register_history(8, 3, 1000 + seed, n_keys=2 + seed % 3, n_values=2, busy=0.8, info=0.15, corrupt=seed % 2 == 1)
        enc, om = encode(hist)
        bad = brute.first_bad_completion(om, op_tuples(enc.ops))
        n_bad += bad is not None
        for width in (1, 4):
            for eager, indep in ((True, False), (False, True), (True, True)):
                r = oracle.check_beam(enc.ops.as_dict(), om, width, eager_txns=eager, txn_independence=indep)
                assert r["valid"] == (1 if bad is None else 0) and (bad is None or r["fail_op"] == bad), (seed, width, eager, indep)
                if bad is None:          # the witness (absorbed txns included) replayed by the independent checker
                    assert brute.check_witness(om, op_tuples(enc.ops), [int(x) for x in r["witness"]]) == r["final_state"]
    assert n_bad > 30
    fewer = [0, 0, 0]
    for seed in range(6):
        hist = multi_register_history(3000, 32, seed, n_keys=8, n_values=5, busy=0.12, info=0.0, corrupt=seed % 3 == 2)
        enc, om = encode(hist)
        plain = oracle.check_beam(enc.ops.as_dict(), om, 8, want_witness=False, eager_txns=False, txn_independence=False)
        answer = (plain["valid"], plain["fail_op"] if plain["valid"] == 0 else None)
        for i, (eager, indep) in enumerate(((True, False), (False, True), (True, True))):
            r = oracle.check_beam(enc.ops.as_dict(), om, 8, want_witness=False, eager_txns=eager, txn_independence=indep)
            assert (r["valid"], r["fail_op"] if r["valid"] == 0 else None) == answer
            fewer[i] += r["probes"] < plain["probes"]
    assert min(fewer) >= 5


def test_memo_fallback_when_too_many_keys(native):
    hist = [{"type": "invoke", "f": "txn", "value": [["w", k, 1] for k in range(10)], "process": 0},
            {"type": "ok", "f": "txn", "value": [["w", k, 1] for k in range(10)], "process": 0}]
    enc = _analysis.Encoded(M.multi_register({}), hist)
    assert enc.native_model[0].kind == N.MODEL_TABLE


@pytest.mark.gpu
@pytest.mark.parametrize("n_ops,procs,info,corrupt", [(60, 4, 0.05, False), (60, 4, 0.0, True), (600, 8, 0.02, False),
                                                       (600, 8, 0.0, True), (2500, 24, 0.0, False)])
def test_gpu_matches_oracle(native, oracle, n_ops, procs, info, corrupt):
    for seed in range(3):
        hist = multi_register_history(n_ops, procs, seed, n_keys=8, n_values=5, busy=0.25, info=info, corrupt=corrupt)
        enc, om = encode(hist)
        exp = oracle.check(enc.ops.as_dict(), om, "window", max_steps=5_000_000)
        assert exp["valid"] != -1
        got = core.check_ops(enc.ops, enc.native_model, core.make_opts(time_limit_ms=60000))
        assert got["valid"] == exp["valid"]
        for k in ("steps", "visited", "backtracks", "max_depth"):
            assert got[k] == exp[k], k
        if exp["valid"] == 1:
            assert np.array_equal(got["witness"], exp["witness"]) and got["final_state"] == exp["final_state"]
        else:
            assert got["fail_op"] == exp["fail_op"]
        # the wide kernel: under both rules (the default), under each alone, under none -- every counter and the witness the oracle's
        for eager, indep in ((True, True), (True, False), (False, True), (False, False)):
            expb = oracle.check_beam(enc.ops.as_dict(), om, 8, eager_txns=eager, txn_independence=indep)
            gotb = core.check_ops(enc.ops, enc.native_model, core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=8,
                                                                            eager_txns=eager, txn_independence=indep))
            assert gotb["valid"] == expb["valid"] == exp["valid"], (eager, indep)
            assert (gotb["probes"], gotb["visited"], gotb["backtracks"]) == (expb["probes"], expb["visited"], expb["backtracks"]), (eager, indep)
            if exp["valid"] == 1:
                assert np.array_equal(gotb["witness"], expb["witness"]), (eager, indep)
                assert brute.check_witness(om, op_tuples(enc.ops), [int(x) for x in gotb["witness"]]) == gotb["final_state"]
            else:
                assert gotb["fail_op"] == exp["fail_op"]
        # and through the knossos surface (result map carries the decoded model)
        a = kwgl.analysis(M.multi_register({}), hist)
        assert a["valid?"] is (exp["valid"] == 1)
        if exp["valid"] == 1:
            assert isinstance(a["configs"][0]["model"], M.MultiRegister)
        assert linear.analysis(M.multi_register({}), hist)["valid?"] is (exp["valid"] == 1)


@pytest.mark.gpu
def test_the_rules_at_256_processes_and_with_crashed_txns(native, oracle):
    """Four mask words (up to 256 process slots, crashed processes included: wider windows go to the sequential kernel), crashed txns among the candidates (they join a closure like any open call, the eager rule
    never takes them), widths 2 .. 16, a planted bad read: verdict, failing op, witness and counters equal the oracle's under the rules."""
    cases = [(4000, 256, 0.02, 0.0, False, 8), (4000, 200, 0.025, 0.01, False, 4), (3000, 200, 0.03, 0.0, True, 16), (1500, 70, 0.06, 0.02, False, 2),
             (1500, 130, 0.03, 0.015, True, 8)]
    for i, (n_ops, procs, busy, info, corrupt, width) in enumerate(cases):
        hist = multi_register_history(n_ops, procs, 44 if i == 4 else 40 + i, n_keys=8, n_values=5, busy=busy, info=info, corrupt=corrupt)
        enc, om = encode(hist)
        exp = oracle.check_beam(enc.ops.as_dict(), om, width, max_probes=20_000_000)
        assert exp["valid"] == (0 if corrupt else 1), i
        got = core.check_ops(enc.ops, enc.native_model, core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, search_width=width))
        assert got["valid"] == exp["valid"], i
        assert (got["probes"], got["visited"], got["backtracks"], got["max_depth"]) == (exp["probes"], exp["visited"], exp["backtracks"], exp["max_depth"]), i
        if corrupt:
            assert got["fail_op"] == exp["fail_op"]
        else:
            assert np.array_equal(got["witness"], exp["witness"]) and got["final_state"] == exp["final_state"]
            assert brute.check_witness(om, op_tuples(enc.ops), [int(x) for x in got["witness"]]) == got["final_state"]
