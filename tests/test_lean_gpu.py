"""The lean tables and / or the lists in order of completion ON THE DEVICE (csrc/tbc_internal.h kLeanCands | kLeanLook; TBC_NARROW_LEAN=1, read once per process: this file
runs only in a process started with it -- tests/test_zy_forms_gpu.py does that): the narrow kernel over list entries {call, twin mask}
and 8 B lookahead records against the oracle's schedule with the lean record's reading of three or more open producers (look_two):
verdict, failing op, every counter.  That the lean formats were really in effect is part of the comparison: on the two-valued busy
histories look_two and the plain lookahead give different counters, and the device must give look_two's."""
import os

import numpy as np
import pytest

from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

LEAN = os.environ.get("TBC_NARROW_LEAN") in ("1", "2")
LAZY = os.environ.get("TBC_NARROW_LEAN") == "2"        # + the lookahead at once only for the config popped next (the oracle's lazy_look)
ORDER = int(os.environ.get("TBC_NARROW_ORDER", "0")) if os.environ.get("TBC_NARROW_ORDER", "").isdigit() and (int(os.environ["TBC_NARROW_ORDER"]) in (1, 2) or 16 <= int(os.environ["TBC_NARROW_ORDER"]) <= 16 + 4096) else 0      # (2: ... with the :write calls last) the fronts' lists in order of completion (PackOpenArgs.list_order); a witness replays its absorbed reads in that order
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (LEAN or ORDER), reason="a process started with TBC_NARROW_LEAN=1|2 and / or TBC_NARROW_ORDER=1|2 only")]

CAS = {"kind": 1, "init": N.NIL}
SHAPES = [(8, 3, 0.0, 0.0, 0.8), (40, 4, 0.0, 0.5, 0.5), (200, 8, 0.0, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3), (1000, 16, 0.0, 0.0, 0.5),
          (1000, 16, 0.01, 0.0, 0.3), (1000, 16, 0.0, 0.6, 0.2), (3000, 64, 0.0, 0.0, 0.1), (3000, 64, 0.0, 0.6, 0.05)]


def _in_domain(n, p, s, busy, info, corrupt, n_values=5):
    h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt, n_values=n_values - 1 if corrupt else n_values))
    h.a[h.a == n_values - 1 + 7] = n_values - 1
    return h


def _expect(oracle, h, L, look_two=None, list_order=None, **kw):
    return oracle.check_beam(h.as_dict(), CAS, 1, round_pairs=L, rules_at_any_round_size=True, branch_lists=True,
                             look_two=LEAN if look_two is None else look_two, list_order=oracle.ORACLE_LIST_ORDER[ORDER] if list_order is None else list_order,
                             lazy_look=LAZY and look_two is None, **kw)


@pytest.mark.parametrize("L", [8, 16])
def test_lean_tables_match_the_oracles_look_two_schedule(native, oracle, L):
    hists = [_in_domain(n, p, s, busy, info, corrupt) for (n, p, info, corrupt, busy) in SHAPES for s in range(3)]
    hists += [columns.pair_events(synth.register_events(n_ops=300, n_procs=24, seed=s, busy=1.0, n_values=2)) for s in range(6)]
    hists = [h for h in hists if h.n_process <= 64]
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    # (enough histories that the library does not add the level sweep beside the search; a witness every time: under ORDER the library
    # replays the absorbed reads in completion order, as the oracle's list_order does)
    n1 = len(hists)
    hists = hists * 10
    with core.Batch(hists, model, core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, lanes_per_history=L, want_witness=True)) as b:
        assert b.lanes_per_history() == L
        res = b.run().results()
        again = b.run().results()
    differs = 0
    for i, (h, got) in enumerate(zip(hists, res)):
        if i >= n1 and i % 7:
            continue
        exp = _expect(oracle, h, L, want_witness=True)
        assert got["valid"] == exp["valid"], (i, got["valid"], exp["valid"], got["cause"])
        assert (got["probes"], got["visited"], got["backtracks"], got["max_depth"]) == (exp["probes"], exp["visited"], exp["expanded"], exp["max_stack"]), i
        if exp["valid"] == 0:
            assert got["fail_op"] == exp["fail_op"], i
        elif got["witness"] is not None:
            assert np.array_equal(got["witness"], exp["witness"]), i
        assert (again[i]["valid"], again[i]["probes"]) == (got["valid"], got["probes"])
        plain = _expect(oracle, h, L, look_two=False, list_order=0, want_witness=False)
        differs += (plain["probes"], plain["rounds"]) != (exp["probes"], exp["rounds"])
    assert differs >= 1          # (else this run could not tell these tables from the default ones)


def test_lean_tables_in_a_big_batch_with_the_queue(native, oracle):
    """4,096 bench-shaped histories (1,000 ops each) by 8 lanes per history: wavefronts refill from the queue, sets grow inside the kernel"""
    base = [_in_domain(1000, 64, 9000 + s, 0.1, 0.0, 0.5 * (s % 8 == 0)) for s in range(64)]
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    with core.Batch([base[i % 64] for i in range(4096)], model, core.make_opts(time_limit_ms=60000, want_witness=False, algorithm=N.ALG_COMPETITION, lanes_per_history=8, visited_per_op=1)) as b:
        res = b.run().results()
    for i in range(64):
        exp = _expect(oracle, base[i], 8, want_witness=False)
        for k in (i, i + 64 * 17, i + 64 * 63):
            got = res[k]
            assert (got["valid"], got["probes"], got["visited"]) == (exp["valid"], exp["probes"], exp["visited"]), (i, k)


@pytest.mark.skipif(not ORDER, reason="TBC_NARROW_ORDER=1|2 only")
@pytest.mark.parametrize("width", [2, 4])
def test_wide_schedule_over_lists_in_order_of_completion(native, oracle, width):
    """a wavefront per history (the kernel of workloads 2 / 3) takes its pairs from the same lists: against the oracle's wide schedule
    with list_order = 1 -- verdict, failing op, every counter -- at 6, 19 and 32 calls in flight; and it needs fewer probes there"""
    cases = [(200, 8, 0.0, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3), (1000, 16, 0.0, 0.0, 0.5), (1000, 16, 0.0, 0.6, 0.2), (2000, 64, 0.0, 0.0, 0.1),
             (2000, 64, 0.0, 0.0, 0.3), (2000, 64, 0.0, 0.0, 0.5), (1500, 64, 0.0, 0.5, 0.3)]
    hists = [_in_domain(n, p, s, busy, info, corrupt) for (n, p, info, corrupt, busy) in cases for s in range(3)]
    n1 = len(hists)
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    with core.Batch(hists * 11, model, core.make_opts(time_limit_ms=60000, search_width=width, algorithm=N.ALG_COMPETITION, want_witness=False)) as b:
        assert b.lanes_per_history() == 64
        res = b.run().results()
    fewer = 0
    for i, h in enumerate(hists):
        exp = oracle.check_beam(h.as_dict(), CAS, width, max_probes=20_000_000, want_witness=False, list_order=oracle.ORACLE_LIST_ORDER[ORDER])
        plain = oracle.check_beam(h.as_dict(), CAS, width, max_probes=20_000_000, want_witness=False)
        for k in (i, i + 5 * n1):
            got = res[k]
            assert got["valid"] == exp["valid"], (i, k)
            if exp["valid"] == 0:
                assert got["fail_op"] == exp["fail_op"], (i, k)
            assert (got["probes"], got["visited"], got["backtracks"], got["max_depth"]) == (exp["probes"], exp["visited"], exp["expanded"], exp["max_stack"]), (i, k)
        fewer += exp["probes"] < plain["probes"]
    assert fewer >= n1 // 2
