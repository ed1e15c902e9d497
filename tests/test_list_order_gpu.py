"""tbc_opts.list_order ON THE DEVICE: the fronts' lists in order of completion (plain, with the :write calls last, with a :write as if it
completed W ranks later -- 16 + 24 is what the library takes by itself wherever the order applies) under the narrow kernel (several
histories per wavefront) and the wide one (a wavefront per history), against the oracle's schedule with the same list order: verdict,
failing op, witness, every counter.  Round 4 built the orders under the emulator; round 5's first device call ran them
(profiles/r05_lean_gpu_order_*.txt: every parity assertion green) and made the write delay the default."""
import numpy as np
import pytest

from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

pytestmark = pytest.mark.gpu

CAS = {"kind": 1, "init": N.NIL}
SHAPES = [(8, 3, 0.0, 0.0, 0.8), (40, 4, 0.0, 0.5, 0.5), (200, 8, 0.0, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3), (1000, 16, 0.0, 0.0, 0.5),
          (1000, 16, 0.01, 0.0, 0.3), (1000, 16, 0.0, 0.6, 0.2), (3000, 64, 0.0, 0.0, 0.1), (3000, 64, 0.0, 0.6, 0.05)]
# (what is asked, what tbc_batch_list_order must answer, the oracle's list order)
ORDERS = [(N.ORDER_DEFAULT, 16 + 24), (N.ORDER_SLOT, N.ORDER_SLOT), (N.ORDER_COMPLETION, N.ORDER_COMPLETION), (N.ORDER_WRITES_LAST, N.ORDER_WRITES_LAST), (16 + 5, 16 + 5)]


def oracle_order(oracle, reported):
    """TBC_ORDER_* as tbc_batch_list_order reports it -> wgl_beam_set_list_order's numbering"""
    return reported if reported >= 16 else oracle.ORACLE_LIST_ORDER[reported - 1]


def _in_domain(n, p, s, busy, info, corrupt, n_values=5):
    h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt, n_values=n_values - 1 if corrupt else n_values))
    h.a[h.a == n_values - 1 + 7] = n_values - 1
    return h


@pytest.mark.parametrize("asked,reported", ORDERS)
@pytest.mark.parametrize("L", [8, 16])
def test_narrow_kernel_follows_the_oracles_schedule_in_every_list_order(native, oracle, L, asked, reported):
    hists = [_in_domain(n, p, s, busy, info, corrupt) for (n, p, info, corrupt, busy) in SHAPES for s in range(3)]
    hists += [columns.pair_events(synth.register_events(n_ops=300, n_procs=24, seed=s, busy=1.0, n_values=2)) for s in range(6)]
    hists = [h for h in hists if h.n_process <= 64]
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    # (enough histories that the library does not add the level sweep beside the search; a witness every time: the library replays the
    # absorbed reads in the lists' order, as the oracle does)
    n1 = len(hists)
    hists = hists * 10
    with core.Batch(hists, model, core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, lanes_per_history=L, want_witness=True, list_order=asked, count_form=False)) as b:
        assert b.lanes_per_history() == L and b.list_order() == reported
        res = b.run().results()
        again = b.run().results()
    differs = 0
    for i, (h, got) in enumerate(zip(hists, res)):
        if i >= n1 and i % 7:
            continue
        exp = oracle.check_beam(h.as_dict(), CAS, 1, round_pairs=L, rules_at_any_round_size=True, branch_lists=True, list_order=oracle_order(oracle, reported), want_witness=True)
        assert got["valid"] == exp["valid"], (i, got["valid"], exp["valid"], got["cause"])
        assert (got["probes"], got["visited"], got["backtracks"], got["max_depth"]) == (exp["probes"], exp["visited"], exp["expanded"], exp["max_stack"]), i
        if exp["valid"] == 0:
            assert got["fail_op"] == exp["fail_op"], i
        elif got["witness"] is not None:
            assert np.array_equal(got["witness"], exp["witness"]), i
        assert (again[i]["valid"], again[i]["probes"]) == (got["valid"], got["probes"])
        if reported != N.ORDER_SLOT and i < n1:
            plain = oracle.check_beam(h.as_dict(), CAS, 1, round_pairs=L, rules_at_any_round_size=True, branch_lists=True, want_witness=False)
            differs += (plain["probes"], plain["rounds"]) != (exp["probes"], exp["rounds"])
    assert differs >= 1 or reported == N.ORDER_SLOT         # (else this run could not tell the order from slot order)


def test_default_list_order_in_a_big_batch_with_the_queue(native, oracle):
    """4,096 bench-shaped histories (1,000 ops each) by 8 lanes per history: wavefronts refill from the queue, sets grow inside the kernel"""
    base = [_in_domain(1000, 64, 9000 + s, 0.1, 0.0, 0.5 * (s % 8 == 0)) for s in range(64)]
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    with core.Batch([base[i % 64] for i in range(4096)], model, core.make_opts(time_limit_ms=60000, want_witness=False, algorithm=N.ALG_COMPETITION, lanes_per_history=8, visited_per_op=1, list_order=N.ORDER_DEFAULT)) as b:
        assert b.list_order() == 16 + 24
        res = b.run().results()
    for i in range(64):
        exp = oracle.check_beam(base[i].as_dict(), CAS, 1, round_pairs=8, rules_at_any_round_size=True, branch_lists=True, list_order=16 + 24, want_witness=False)
        for k in (i, i + 64 * 17, i + 64 * 63):
            got = res[k]
            assert (got["valid"], got["probes"], got["visited"]) == (exp["valid"], exp["probes"], exp["visited"]), (i, k)


@pytest.mark.parametrize("asked,reported", [o for o in ORDERS if o[1] != N.ORDER_SLOT])
@pytest.mark.parametrize("width", [2, 4])
def test_wide_schedule_over_lists_in_order_of_completion(native, oracle, width, asked, reported):
    """a wavefront per history (the kernel of workloads 2 / 3) takes its pairs from the same lists: against the oracle's wide schedule
    with the same list order -- verdict, failing op, every counter -- at 6, 19 and 32 calls in flight"""
    cases = [(200, 8, 0.0, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3), (1000, 16, 0.0, 0.0, 0.5), (1000, 16, 0.0, 0.6, 0.2), (2000, 64, 0.0, 0.0, 0.1),
             (2000, 64, 0.0, 0.0, 0.3), (2000, 64, 0.0, 0.0, 0.5), (1500, 64, 0.0, 0.5, 0.3)]
    hists = [_in_domain(n, p, s, busy, info, corrupt) for (n, p, info, corrupt, busy) in cases for s in range(3)]
    n1 = len(hists)
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    with core.Batch(hists * 11, model, core.make_opts(time_limit_ms=60000, search_width=width, algorithm=N.ALG_COMPETITION, want_witness=False, list_order=asked, order_restarts=False)) as b:      # (ONE pass in the asked order: the restarts are tests/test_order_restarts_gpu.py)
        assert b.lanes_per_history() == 64 and b.list_order() == reported
        res = b.run().results()
    total = total_plain = 0
    for i, h in enumerate(hists):
        exp = oracle.check_beam(h.as_dict(), CAS, width, max_probes=20_000_000, want_witness=False, list_order=oracle_order(oracle, reported))
        for k in (i, i + 5 * n1):
            got = res[k]
            assert got["valid"] == exp["valid"], (i, k)
            if exp["valid"] == 0:
                assert got["fail_op"] == exp["fail_op"], (i, k)
            assert (got["probes"], got["visited"], got["backtracks"], got["max_depth"]) == (exp["probes"], exp["visited"], exp["expanded"], exp["max_stack"]), (i, k)
        total += exp["probes"]
        total_plain += oracle.check_beam(h.as_dict(), CAS, width, max_probes=20_000_000, want_witness=False)["probes"]
    assert total != total_plain          # (round 4 asserted "fewer probes on most histories": false on these small ones -- 6 of 24 -- and no parity matter)


def test_list_order_where_it_does_not_apply_is_slot_order(native):
    """the count form, the level sweep beside the search, two mask words: the lists stay in slot order whatever is asked, and the batch says so"""
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    crashed = [columns.pair_events(synth.register_events(n_ops=400, n_procs=8, seed=s, busy=0.5, info=0.05)) for s in range(4)]
    with core.Batch(crashed, model, core.make_opts(algorithm=N.ALG_COMPETITION, search_width=4, want_witness=False, list_order=16 + 24, count_form=True)) as b:      # count form
        assert b.list_order() == N.ORDER_SLOT
    few = [columns.pair_events(synth.register_events(n_ops=400, n_procs=8, seed=s, busy=0.5)) for s in range(4)]
    with core.Batch(few, model, core.make_opts(algorithm=N.ALG_COMPETITION, want_witness=False, list_order=N.ORDER_DEFAULT)) as b:      # the level sweep takes a handful of histories
        assert b.sweep_info()["enabled"] == 1 and b.list_order() == N.ORDER_SLOT
    wide = [columns.pair_events(synth.register_events(n_ops=600, n_procs=100, seed=s, busy=0.05)) for s in range(4)]
    with core.Batch(wide, model, core.make_opts(algorithm=N.ALG_COMPETITION, search_width=4, want_witness=False, list_order=N.ORDER_DEFAULT)) as b:
        assert b.list_order() == N.ORDER_SLOT
    with pytest.raises(N.TbcError):
        core.Batch(few, model, core.make_opts(algorithm=N.ALG_COMPETITION, list_order=7))
