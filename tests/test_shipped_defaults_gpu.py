"""What SHIPS, end to end: tests/conftest.py pins the mask form and process-slot lists for the tests that compare counters with the slot-order
oracles -- this file pins nothing.  Every history goes through the library's own defaults (tbc_opts all zero but the algorithm: the count
form for crashed calls, the relaxed sweep beside the search of a handful of them, the fronts' lists in order of completion with a :write 24
ranks later, the level sweep for few histories without a witness, several histories per wavefront for big quiet batches) and the verdict,
the failing op and -- where a witness comes back -- the witness's legality are checked against the sequential restatement and brute force's
replay, which know nothing of any of that."""
import numpy as np
import pytest

from helpers import op_tuples
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import brute

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(240)]          # (pytest-timeout: nothing here may hold the GPU tier)

CAS = {"kind": 1, "init": N.NIL}
SHAPES = [(8, 3, 0.8, 0.1, 0.0), (8, 3, 0.8, 0.1, 0.5), (40, 4, 0.5, 0.05, 0.5), (200, 8, 0.5, 0.02, 0.0), (200, 8, 0.3, 0.0, 0.6), (1000, 16, 0.5, 0.0, 0.0),
          (1000, 16, 0.3, 0.01, 0.0), (1000, 16, 0.2, 0.03, 0.6), (3000, 64, 0.1, 0.0, 0.0), (3000, 64, 0.05, 0.0, 0.6), (2000, 64, 0.1, 0.02, 0.5),
          (600, 4, 0.8, 0.2, 0.7), (400, 70, 0.5, 0.05, 0.0), (1500, 24, 0.25, 0.01, 0.3)]


@pytest.fixture(autouse=True)
def _the_shipped_defaults():
    old = core.DEFAULT_COUNT_FORM, core.DEFAULT_LIST_ORDER
    core.DEFAULT_COUNT_FORM, core.DEFAULT_LIST_ORDER = True, 0
    yield
    core.DEFAULT_COUNT_FORM, core.DEFAULT_LIST_ORDER = old


def _hists():
    return [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=8800 + 7 * i + s, busy=busy, info=info, corrupt=corrupt))
            for i, (n, p, busy, info, corrupt) in enumerate(SHAPES) for s in range(2)]


def _same(got, ref, h, tag):
    if got["valid"] == N.UNKNOWN and got["cause"] == N.CAUSE_TIME_LIMIT:
        return          # (a limit is an honest answer; nothing here should need 20 s, and nothing may hold the GPU tier longer)
    assert got["valid"] == ref["valid"], (tag, got["valid"], ref["valid"], got["cause"])
    if ref["valid"] == 0:
        assert got["fail_op"] == ref["fail_op"], tag
    elif got["witness"] is not None:
        assert brute.check_witness(CAS, op_tuples(h), [int(x) for x in got["witness"]]) == got["final_state"], tag


@pytest.mark.parametrize("algorithm,witness", [(N.ALG_COMPETITION, False), (N.ALG_COMPETITION, True), (N.ALG_WGL, True), (N.ALG_LINEAR, False)])
def test_one_history_at_a_time_and_as_batches(native, oracle, algorithm, witness):
    hists = _hists()
    if algorithm in (N.ALG_WGL, N.ALG_LINEAR):
        # (:algorithm :wgl / :linear keep a mask bit per crashed call -- the published forms: an invalid history with crashed calls is
        # exponential for them -- the first version of this test spent ten GPU minutes there.  Crash-free or small.)
        hists = [h for h in hists if len(h) <= 250 or not (np.asarray(h.ret_pos) == 0xFFFFFFFF).any()]
    refs = [oracle.check(h.as_dict(), CAS, "window", max_steps=20_000_000, want_witness=False) for h in hists]
    assert sum(r["valid"] == 0 for r in refs) >= 5 and sum(r["valid"] == 1 for r in refs) >= 5
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    opts = core.make_opts(time_limit_ms=20000, algorithm=algorithm, want_witness=witness)
    for i, (h, ref) in enumerate(zip(hists, refs)):
        if ref["valid"] != -1:
            _same(core.check_ops(h, model, opts), ref, h, ("single", i))
    n = len(hists)
    for idx in (list(range(min(8, n))), list(range(min(8, n), n)), [k % n for k in range((4 if algorithm != N.ALG_COMPETITION else 20) * n)]):
        if not idx:
            continue          # a handful (the sweeps), a few dozen, some hundreds (a wavefront each)
        with core.Batch([hists[k] for k in idx], model, opts) as b:
            res = b.run().results()
            again = b.run().results()
        for i, (k, g) in enumerate(zip(idx, res)):
            if refs[k]["valid"] != -1:
                _same(g, refs[k], hists[k], ("batch", len(idx), i))
            assert (again[i]["valid"], again[i]["fail_op"]) == (g["valid"], g["fail_op"])


def test_a_big_quiet_batch_under_the_defaults(native, oracle):
    """>= 24,576 histories at low concurrency: eight to a wavefront by the library's own choice, the lists in the default order; valid and
    not, crash-free (the count form keeps a wavefront per history)"""
    base = [columns.pair_events(synth.register_events(n_ops=150, n_procs=16, seed=9900 + s, busy=0.2, corrupt=0.5 if s % 5 == 0 else 0.0)) for s in range(64)]
    refs = [oracle.check(h.as_dict(), CAS, "window", want_witness=False) for h in base]
    NB = 24576 + 64
    with core.Batch([base[i % 64] for i in range(NB)], core.make_model(N.MODEL_CAS_REGISTER, N.NIL),
                    core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, want_witness=False)) as b:
        assert (b.lanes_per_history(), b.list_order()) == (8, 16 + 24)
        res = b.run().results()
    for k in range(NB):
        ref = refs[k % 64]
        assert res[k]["valid"] == ref["valid"] and (ref["valid"] == 1 or res[k]["fail_op"] == ref["fail_op"]), k


def test_the_other_models_and_the_narrow_shapes_under_the_defaults(native, oracle):
    """The parity files pinned by tests/conftest.py (mask form, slot order), a second time with NOTHING pinned, at verdict / failing-op level:
    set and bank (the lazy rule), multi-register, the narrow kernel's shapes as one batch of >= 24,576 histories (the library then takes
    eight to a wavefront, the default list order and -- for what the wide kernel gets -- the race of list orders on its own)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import bank_history, multi_register_history, set_history
    from test_commutative_models import enc_bank, enc_set
    from test_multi_register import encode as enc_mr
    o = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False)
    n_bad = 0
    for which, n_ops, procs, info, corrupt in [("set", 800, 8, 0.02, None), ("set", 800, 8, 0.0, "phantom"), ("set", 3000, 5, 0.02, "lost"),
                                                ("bank", 800, 8, 0.02, False), ("bank", 800, 8, 0.0, True), ("mr", 600, 8, 0.02, False), ("mr", 600, 8, 0.0, True)]:
        for seed in range(2):
            if which == "set":
                e, om = enc_set(set_history(n_ops, procs, seed, busy=0.2, info=info, corrupt=corrupt))
            elif which == "bank":
                e, om = enc_bank(bank_history(n_ops, procs, seed, busy=0.2, info=info, corrupt=corrupt))
            else:
                e, om = enc_mr(multi_register_history(n_ops, procs, seed, n_keys=8, n_values=5, busy=0.25, info=info, corrupt=corrupt))
            exp = oracle.check_beam(e.ops.as_dict(), om, 8)
            got = core.check_ops(e.ops, e.native_model, o)
            assert got["valid"] == exp["valid"] != N.UNKNOWN, (which, seed)
            n_bad += exp["valid"] == 0           # (a planted loss may leave a history linearizable: the oracle says, not the generator)
            if exp["valid"] == 0:
                assert got["fail_op"] == exp["fail_op"], (which, seed)
    assert n_bad >= 6
    shapes = [(40, 4, 0.0, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3), (1000, 16, 0.0, 0.0, 0.5), (1000, 16, 0.0, 0.6, 0.2), (3000, 64, 0.0, 0.0, 0.1), (3000, 64, 0.0, 0.6, 0.05)]
    base = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=7700 + s, busy=busy, info=info, corrupt=corrupt))
            for (n, p, info, corrupt, busy) in shapes for s in range(4)]
    refs = [oracle.check(h.as_dict(), CAS, "window", max_steps=20_000_000, want_witness=False) for h in base]
    NB = 24576 + len(base)
    with core.Batch([base[i % len(base)] for i in range(NB)], core.make_model(N.MODEL_CAS_REGISTER, N.NIL), o) as b:
        res = b.run().results()
    for k in range(NB):
        ref = refs[k % len(base)]
        if ref["valid"] != -1:
            assert res[k]["valid"] == ref["valid"] and (ref["valid"] == 1 or res[k]["fail_op"] == ref["fail_op"]), k
