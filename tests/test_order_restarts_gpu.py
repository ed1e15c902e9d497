"""ORDER RESTARTS (include/tbcheck.h TBC_DOM_NO_ORDER_RESTARTS; csrc/batch_run.hip order_restarts / race_orders): a history of the wide
depth-first search that has not ended within the budget is searched again in other list orders.
  * with a witness wanted: one order after the other -- pass by pass against oracle/wgl.py check_restart_pipeline (wgl_beam.c in each pass's
    order under each pass's budget): verdict, failing op and the counters summed over the passes; the witness replays legally (the
    answering pass's own order decides how its absorbed reads are replayed);
  * without: all orders at once, the first to decide a history stops the others (a race): verdict and failing op equal the oracle's,
    whichever order answers; the counters depend on who wins and are NOT compared (they are at least the first pass's)."""
import numpy as np
import pytest

from helpers import op_tuples
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import brute

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
CAS = {"kind": 1, "init": N.NIL}


def gm():
    return core.make_model(N.MODEL_CAS_REGISTER, N.NIL)


def hard(seed, corrupt=0.0, n_ops=2500, n_procs=48, busy=0.6):
    return columns.pair_events(synth.register_events(n_ops=n_ops, n_procs=n_procs, seed=seed, busy=busy, corrupt=corrupt))


@pytest.fixture(autouse=True)
def _the_librarys_own_order():
    old = core.DEFAULT_LIST_ORDER
    core.DEFAULT_LIST_ORDER = 0          # (tests/conftest.py pins slot order for the single-pass parity tests; the restarts ARE the default order's)
    yield
    core.DEFAULT_LIST_ORDER = old


def test_passes_equal_the_oracles_pipeline(native, oracle):
    hists = [hard(900 + s) for s in range(20)] + [hard(950 + s, corrupt=0.5, n_ops=1200, n_procs=24, busy=0.5) for s in range(4)]
    budget = 32 * max(len(h) for h in hists)
    with core.Batch(hists, gm(), core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, search_width=4, want_witness=True)) as b:
        assert b.lanes_per_history() == 64 and b.list_order() == 16 + 24
        res = b.run().results()
        again = b.run().results()
        assert b.last_raced() == 0
    later = 0
    for i, (h, g) in enumerate(zip(hists, res)):
        v, fo, last, tot, how = oracle.check_restart_pipeline(h.as_dict(), CAS, width=4, budget=budget)
        assert g["valid"] == v, (i, how, g["cause"])
        if v == 0:
            assert g["fail_op"] == fo, (i, how)
        assert (g["probes"], g["visited"], g["backtracks"], g["max_depth"]) == (tot["probes"], tot["visited"], tot["expanded"], tot["max_stack"]), (i, how)
        later += not how.startswith("pass 0")
        assert (again[i]["valid"], again[i]["probes"]) == (g["valid"], g["probes"]), i          # the same run after run
        ref = oracle.check(h.as_dict(), CAS, "window", max_steps=30_000_000, want_witness=False)      # ... and the sequential restatement, whatever the passes
        if ref["valid"] != -1:
            assert g["valid"] == ref["valid"] and (ref["valid"] == 1 or g["fail_op"] == ref["fail_op"]), i
    assert later >= 3, later          # (some histories did need another order: else this test tests nothing)


def test_a_witness_of_a_later_pass_replays_legally_and_the_switch_switches_them_off(native, oracle):
    hists = [hard(900 + s) for s in range(12)]
    budget = 32 * max(len(h) for h in hists)
    with core.Batch(hists, gm(), core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, search_width=4, want_witness=True)) as b:
        res = b.run().results()
    for i, (h, g) in enumerate(zip(hists, res)):
        assert g["valid"] == 1, i
        assert brute.check_witness(CAS, op_tuples(h), [int(x) for x in g["witness"]]) == g["final_state"], i
    with core.Batch(hists, gm(), core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, search_width=4, want_witness=False, order_restarts=False)) as b:
        one = b.run().results()
    for i, (h, g) in enumerate(zip(hists, one)):
        exp = oracle.check_beam(h.as_dict(), CAS, 4, want_witness=False, list_order=16 + 24, max_probes=40_000_000)
        assert (g["valid"], g["probes"], g["visited"]) == (exp["valid"], exp["probes"], exp["visited"]), i
    assert any(g["probes"] > budget for g in one)


def test_the_race_of_orders_answers_as_the_oracle_does(native, oracle):
    """No witness wanted: what the budgeted first pass leaves undecided is searched in six orders at once.  Valid and invalid histories,
    ~29 calls in flight; every verdict and failing op against the sequential restatement, run after run."""
    hists = [hard(900 + s) for s in range(20)] + [hard(950 + s, corrupt=0.5, n_ops=1200, n_procs=24, busy=0.5) for s in range(4)]
    budget = 32 * max(len(h) for h in hists)
    # (the reference: the oracle's own restart pipeline -- the plain sequential restatement does not end on every one of these within 3 * 10^7
    # steps; where it does it agrees: test_passes_equal_the_oracles_pipeline)
    refs = []
    for h in hists:
        v, fo, _, _, _ = oracle.check_restart_pipeline(h.as_dict(), CAS, width=4, budget=budget)
        refs.append({"valid": v, "fail_op": fo})
    assert all(r["valid"] != -1 for r in refs)
    with core.Batch(hists, gm(), core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, search_width=4, want_witness=False)) as b:
        for rnd in range(3):
            res = b.run().results()
            assert b.last_raced() >= 3, b.last_raced()          # (some histories did pass the budget: else nothing raced)
            for i, (g, ref) in enumerate(zip(res, refs)):
                assert g["valid"] == ref["valid"], (rnd, i, g["cause"])
                if ref["valid"] == 0:
                    assert g["fail_op"] == ref["fail_op"], (rnd, i)
                    assert g["configs"], (rnd, i)                # the answering search's stuck configs come along
            assert sum(g["probes"] > budget for g in res) == b.last_raced()
