"""Fresh inputs into a batch's arenas (include/tbcheck.h "streaming"; csrc/batch_stream.hip): tbc_batch_map_input /
tbc_batch_submit_input / tbc_batch_reload.  The reference checks every history ONCE (checker/compose over the test's one history,
/root/reference/src/tigerbeetle/core.clj:139-146; independent/checker per key, workloads/set_full.clj:155-158) -- so what counts is a
batch that takes NEW histories, and that a consumed input is answered exactly as a batch created from the same histories would answer it:
verdict, failing op, every counter -- which in turn is the oracle's schedule bit for bit (oracle/wgl_beam.c)."""
import numpy as np
import pytest

from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

pytestmark = pytest.mark.gpu

CAS = {"kind": 1, "init": N.NIL}
KEYS = ("valid", "fail_op", "prev_ok_op", "probes", "visited", "backtracks", "max_depth", "steps", "final_state")


def gm():
    return core.make_model(N.MODEL_CAS_REGISTER, N.NIL)


def hists_of(seed0, count, n_ops=500, n_procs=8, busy=0.3, info=0.0, bad_every=5, **kw):
    return [columns.pair_events(synth.register_events(n_ops=n_ops + 17 * (s % 7), n_procs=n_procs, seed=seed0 + s, busy=busy, info=info,
                                                      corrupt=0.4 if bad_every and s % bad_every == 0 else 0.0, **kw)) for s in range(count)]


def key(r):
    return tuple(r[k] for k in KEYS)


def fresh(hists, opts):
    with core.Batch(hists, gm(), opts) as b:
        return [key(r) for r in b.run().results()]


def narrow_expect(oracle, h, L):
    return oracle.check_beam(h.as_dict(), CAS, 1, round_pairs=L, rules_at_any_round_size=True, branch_lists=True)


@pytest.mark.parametrize("lanes", [8, 64])
def test_reload_three_times_matches_a_batch_created_from_the_same_histories_and_the_oracle(native, oracle, lanes):
    """The verdict's bar (round 5, next item 1): a batch is reloaded three times with different histories and answers each input as its own
    batch would -- same verdicts, failing ops and counters --, which for several histories per wavefront is the oracle's schedule."""
    opts = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, lanes_per_history=lanes, search_width=0 if lanes == 8 else 4)
    first = hists_of(1000, 96)
    with core.Batch(first, gm(), opts) as b:
        assert [key(r) for r in b.run().results()] == fresh(first, opts)
        for rnd in range(3):
            nxt = hists_of(2000 + 500 * rnd, 96 - 7 * rnd)          # (fewer histories each time, other lengths)
            b.reload(nxt)
            res = b.run().results()
            assert len(res) == len(nxt)
            assert [key(r) for r in res] == fresh(nxt, opts), rnd
            info = b.input_info()
            assert info["n_hist"] == len(nxt) and info["pending"] == 0 and info["inputs_consumed"] == rnd + 1
            assert info["bytes_copied"] == 12 * sum(len(h) for h in nxt)
            if lanes == 8:
                for i in (0, 5, 10, len(nxt) - 1):
                    exp = narrow_expect(oracle, nxt[i], 8)
                    got = res[i]
                    assert (got["valid"], got["probes"], got["visited"]) == (exp["valid"], exp["probes"], exp["visited"]), (rnd, i)
                    if exp["valid"] == 0:
                        assert got["fail_op"] == exp["fail_op"], (rnd, i)
            else:
                for i in (0, 5, 10):
                    assert res[i]["valid"] == oracle.check(nxt[i].as_dict(), CAS, "window", want_witness=False)["valid"], (rnd, i)
            # the resident input once more: the same answers (its lists' places are dealt again on the device)
            assert [key(r) for r in b.run().results()] == [key(r) for r in res]


def test_mapped_slots_are_consumed_in_order_and_two_may_wait(native, oracle):
    opts = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, lanes_per_history=8)
    sets = [hists_of(100 * k, 40, n_ops=300 + 20 * k) for k in range(5)]
    expect = [fresh(s, opts) for s in sets]
    with core.Batch(sets[0], gm(), opts) as b:
        b.run()
        # three slots, filled in place; two inputs wait at most
        for k in (1, 2):
            b.submit_input(k, b.fill_input(k, sets[k]))
        with pytest.raises(N.TbcError):
            b.submit_input(3, b.fill_input(3, sets[3]))          # a third may not wait
        assert [key(r) for r in b.run().results()] == expect[1]
        b.submit_input(3, 40)                                    # (slot 3 is filled already)
        assert [key(r) for r in b.run().results()] == expect[2]
        b.submit_input(1, b.fill_input(1, sets[4]))              # slot 1 again: map_input waits for its previous copy
        assert [key(r) for r in b.run().results()] == expect[3]
        assert [key(r) for r in b.run().results()] == expect[4]
        assert b.input_info()["inputs_consumed"] == 4
        # an empty history and a single op travel too
        tiny = [sets[1][0], columns.pair_events(synth.register_events(n_ops=1, n_procs=1, seed=3, busy=0.5)), sets[1][2]]
        b.reload(tiny)
        got = [key(r) for r in b.run().results()]
        assert got == fresh(tiny, opts)


def test_what_a_batch_cannot_take_is_refused_by_name(native):
    opts = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, lanes_per_history=8)
    first = hists_of(10, 32, bad_every=0, n_values=3)          # values 0..2: the batch's tables hold nil + 0..2
    with core.Batch(first, gm(), opts) as b:
        b.run()
        ok = [key(r) for r in b.results()]
        wider = hists_of(50, 32, bad_every=0, n_values=5)      # a value the batch has no row for
        b.reload(wider)
        with pytest.raises(N.TbcError) as e:
            b.run()
        assert e.value.status == N.ERR_UNSUPPORTED and "value domain" in e.value.detail
        with pytest.raises(N.TbcError):
            b.run()                                           # nothing resident to run
        crashed = hists_of(60, 32, bad_every=0, n_values=3, info=0.05)
        b.reload(crashed)
        with pytest.raises(N.TbcError) as e:
            b.run()
        assert e.value.status == N.ERR_UNSUPPORTED and "crashed" in e.value.detail
        b.reload(first)                                       # and the batch is as good as new
        assert [key(r) for r in b.run().results()] == ok
        many = hists_of(70, 33, bad_every=0, n_values=3)
        with pytest.raises(N.TbcError):
            b.reload(many)                                    # more histories than the batch holds
        wide = [columns.pair_events(synth.register_events(n_ops=800, n_procs=40, seed=s, busy=0.3, info=0.0, n_values=3)) for s in range(4)]
        assert max(h.n_process for h in wide) <= 64
        b.reload(wide)
        assert [key(r) for r in b.run().results()] == fresh(wide, opts)
    # batches that take no fresh inputs say so
    for o in (core.make_opts(algorithm=N.ALG_LINEAR, want_witness=False), core.make_opts(algorithm=N.ALG_WGL)):
        with core.Batch(first[:4], gm(), o) as b:
            with pytest.raises(N.TbcError) as e:
                b.map_input(0)
            assert e.value.status == N.ERR_UNSUPPORTED


def test_lists_that_outgrow_their_arena_fall_back_once_and_the_arena_grows(native, oracle):
    """The list arenas are sized by the first input; an input with far more calls in flight does not fit: the histories beyond the arena
    are answered by the sequential kernel THAT run (same verdicts), the arenas grow, and the same input then runs as its own batch would."""
    opts = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, lanes_per_history=8)
    quiet = hists_of(300, 48, n_ops=600, busy=0.05)
    dense = hists_of(400, 48, n_ops=600, busy=0.5)
    exp = fresh(dense, opts)
    with core.Batch(quiet, gm(), opts) as b:
        b.run()
        b.reload(dense)
        res = b.run().results()
        assert [(r["valid"], r["fail_op"]) for r in res] == [(e[0], e[1]) for e in exp]
        assert any(r["search_width"] != 1 for r in res) or b.input_info()["lists_regrown"] == 0      # (someone fell back, unless everything fitted)
        b.reload(dense)
        assert [key(r) for r in b.run().results()] == exp
        assert b.input_info()["lists_regrown"] <= 1


def test_crashed_calls_travel_when_the_batch_was_created_with_some(native, oracle):
    """The mask form (a bit per crashed call): a batch created from histories with crashed calls takes others."""
    opts = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, lanes_per_history=8, count_form=False, max_steps=200000)
    first = hists_of(500, 24, n_ops=400, info=0.02)
    nxt = hists_of(600, 24, n_ops=400, info=0.03)
    with core.Batch(first, gm(), opts) as b:
        b.run()
        b.reload(nxt)
        assert [key(r) for r in b.run().results()] == fresh(nxt, opts)


@pytest.mark.parametrize("lanes", [64, 8])
def test_a_count_form_batch_takes_fresh_inputs(native, oracle, lanes):
    """Crashed calls with an effect under the default rules -- what the reference's nemesis makes (core.clj:106-125) -- are searched in the
    COUNT FORM (classes of crashed calls, process slots re-used: planned on the host).  A batch created that way takes fresh inputs since
    round 6: the planning runs over the wire columns when the input is submitted.  Three inputs of other sizes, crash rates and list
    lengths (the third with twice the calls in flight: the list arena grows BEFORE the run -- a count-form batch has no sequential fallback) are answered
    exactly as batches created from them: verdict, failing op, every counter; and the verdicts are the sequential oracle's."""
    opts = core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, want_witness=False, count_form=True, lanes_per_history=lanes)
    first = hists_of(7000, 80, n_ops=900, info=0.03)
    with core.Batch(first, gm(), opts) as b:
        assert [key(r) for r in b.run().results()] == fresh(first, opts)
        for rnd, (cnt, n_ops, info, busy) in enumerate([(80, 600, 0.05, 0.3), (61, 450, 0.02, 0.3), (80, 800, 0.04, 0.7)]):
            nxt = hists_of(7100 + 300 * rnd, cnt, n_ops=n_ops, info=info, busy=busy, bad_every=4)
            b.reload(nxt)
            res = b.run().results()
            assert [key(r) for r in res] == fresh(nxt, opts), rnd
            assert sum(1 for r in res if r["valid"] == N.INVALID) >= cnt // 8 and all(r["valid"] != N.UNKNOWN for r in res)
            assert b.input_info()["inputs_consumed"] == rnd + 1
            for i in (0, 1, 4, cnt - 1):
                ref = oracle.check(nxt[i].as_dict(), CAS, "window", max_steps=3_000_000, want_witness=False)
                if ref["valid"] != -1:
                    assert res[i]["valid"] == ref["valid"] and (ref["valid"] == 1 or res[i]["fail_op"] == ref["fail_op"]), (rnd, i)
        assert b.input_info()["lists_regrown"] >= 1
        # a slot submitted again as it is (mapped, not re-filled): the caller's words were not touched by the planning -- same answers
        again = hists_of(7800, 30, n_ops=500, info=0.03)
        b.fill_input(0, again)
        b.submit_input(0, len(again))
        exp = fresh(again, opts)
        assert [key(r) for r in b.run().results()] == exp
        b.map_input(0)
        b.submit_input(0, len(again))
        assert [key(r) for r in b.run().results()] == exp
        # a crash-free input into a count-form batch: no classes; its own batch would not be a count-form one (another schedule, other
        # counters), so verdict, failing op and the op before it are what must agree
        calm = hists_of(7900, 40, n_ops=500, info=0.0)
        b.reload(calm)
        assert [key(r)[:3] for r in b.run().results()] == [k[:3] for k in fresh(calm, opts)]
