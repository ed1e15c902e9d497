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


def test_eager_pure_read_txns_change_no_answer(native, oracle):
    """The eager rule for multi-register as a design study of the oracle (oracle/wgl_beam.c, g_eager_txns; no kernel speaks it):
    an open txn of micro-reads only that the state allows is linearized at once.  Same verdict and failing op as brute force and
    as the plain search on small crash-heavy histories, fewer probes on larger ones."""
    n_bad = fewer = 0
    for seed in range(240):
        hist = multi_register_history(8, 3, 1000 + seed, n_keys=2, n_values=2, busy=0.8, info=0.15, corrupt=seed % 2 == 1)
        enc, om = encode(hist)
        bad = brute.first_bad_completion(om, op_tuples(enc.ops))
        n_bad += bad is not None
        for width in (1, 4):
            r = oracle.check_beam(enc.ops.as_dict(), om, width, eager_txns=True)
            assert r["valid"] == (1 if bad is None else 0) and (bad is None or r["fail_op"] == bad), (seed, width)
            if bad is None:          # the witness (absorbed txns included) replayed by the independent checker
                assert brute.check_witness(om, op_tuples(enc.ops), [int(x) for x in r["witness"]]) == r["final_state"]
    assert n_bad > 25
    for seed in range(6):
        hist = multi_register_history(3000, 32, seed, n_keys=8, n_values=5, busy=0.12, info=0.0, corrupt=seed % 3 == 2)
        enc, om = encode(hist)
        plain = oracle.check_beam(enc.ops.as_dict(), om, 8, want_witness=False)
        eager = oracle.check_beam(enc.ops.as_dict(), om, 8, want_witness=False, eager_txns=True)
        assert (plain["valid"], plain["fail_op"] if plain["valid"] == 0 else None) == (eager["valid"], eager["fail_op"] if eager["valid"] == 0 else None)
        fewer += eager["probes"] < plain["probes"]
    assert fewer >= 5


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
        expb = oracle.check_beam(enc.ops.as_dict(), om, 8)
        gotb = core.check_ops(enc.ops, enc.native_model, core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=8))
        assert gotb["valid"] == expb["valid"] == exp["valid"]
        assert (gotb["probes"], gotb["visited"]) == (expb["probes"], expb["visited"])
        if exp["valid"] == 1:
            assert np.array_equal(gotb["witness"], expb["witness"])
        else:
            assert gotb["fail_op"] == exp["fail_op"]
        # and through the knossos surface (result map carries the decoded model)
        a = kwgl.analysis(M.multi_register({}), hist)
        assert a["valid?"] is (exp["valid"] == 1)
        if exp["valid"] == 1:
            assert isinstance(a["configs"][0]["model"], M.MultiRegister)
        assert linear.analysis(M.multi_register({}), hist)["valid?"] is (exp["valid"] == 1)
