"""Parity proper: the HIP search, called through the C-ABI, against the CPU
oracle on the same seeded inputs -- bit-exact verdict, failing op, witness,
final state and traversal counters -- plus the hand-derived known-answer
histories and, at BASELINE.json's full size, size-independent properties
(the witness is a legal real-time-respecting run; batch == single; re-runs are
idempotent)."""
import numpy as np
import pytest

from helpers import MODELS, load_kats, op_tuples, oracle_model
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from jepsen_tigerbeetle_amd.jepsen import checker as jc, independent
from jepsen_tigerbeetle_amd.knossos import _analysis, competition, linear, model as M, op as kop, wgl
from oracle import brute

pytestmark = pytest.mark.gpu

CAS = {"kind": 1, "init": N.NIL}


def gm():
    return core.make_model(N.MODEL_CAS_REGISTER, N.NIL)


SEQ = dict(algorithm=N.ALG_WGL)   # the sequential knossos.wgl order (search_width 1)


def assert_same(got, exp, tag=""):
    assert got["valid"] == exp["valid"], (tag, got["valid"], exp["valid"])
    if exp["valid"] == 0:
        assert got["fail_op"] == exp["fail_op"], tag
        assert got["prev_ok_op"] == (None if exp["prev_ok_op"] == N.NO_OP else exp["prev_ok_op"]), tag
    if exp["valid"] == 1:
        assert got["final_state"] == exp["final_state"], tag
        assert np.array_equal(got["witness"], exp["witness"]), tag
    for k in ("steps", "visited", "probes", "backtracks", "max_depth"):
        assert got[k] == exp[k], (tag, k, got[k], exp[k])


KATS = load_kats()


@pytest.mark.parametrize("name,model,hist,valid,fail_index", KATS, ids=[k[0] for k in KATS])
def test_kat_through_knossos_surface(native, name, model, hist, valid, fail_index):
    for analysis in (wgl.analysis, linear.analysis, competition.analysis):
        a = analysis(MODELS[model](), hist)
        assert a["valid?"] is valid
        if not valid:
            assert a["op"]["index"] == fail_index and a["op"]["type"] == "ok"


@pytest.mark.parametrize("n_ops,procs,info,corrupt,busy", [
    (8, 3, 0.1, 0.0, 0.8), (8, 3, 0.1, 0.5, 0.8), (40, 4, 0.0, 0.0, 0.5), (40, 4, 0.05, 0.5, 0.5),
    (200, 8, 0.02, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3), (1000, 16, 0.0, 0.0, 0.5), (1000, 16, 0.01, 0.0, 0.5),
    (1000, 16, 0.0, 0.6, 0.2), (1000, 16, 0.02, 0.1, 0.2), (3000, 64, 0.0, 0.0, 0.5), (3000, 64, 0.0, 0.6, 0.05),
])
def test_single_history_matches_oracle(native, oracle, n_ops, procs, info, corrupt, busy):
    for seed in range(4):
        ops = columns.pair_events(synth.register_events(n_ops=n_ops, n_procs=procs, seed=seed, busy=busy,
                                                        info=info, corrupt=corrupt))
        exp = oracle.check(ops.as_dict(), CAS, "window", max_steps=3_000_000)
        if exp["valid"] == -1:
            continue
        got = core.check_ops(ops, gm(), core.make_opts(time_limit_ms=30000))
        assert_same(got, exp, f"seed{seed}")


@pytest.mark.parametrize("rules", [True, False])
@pytest.mark.parametrize("lookahead", [True, False])
@pytest.mark.parametrize("width", [2, 8, 16])
def test_wide_schedule_matches_its_oracle(native, oracle, width, lookahead, rules):
    """search_width > 1: the (config, open call)-pair-per-lane kernel against oracle/wgl_beam.c --
    verdict, failing op, witness, final state and counters, bit for bit; and the verdict /
    failing op also against the sequential oracle (they are properties of the history).  With the
    lookahead (tbc_opts.lookahead, default on) the kernel decides deadness from the per-rank records
    pack_open builds, the oracle from the definition; an invalid verdict is re-searched without it on
    both sides, so failing op and counters of invalid histories are the exact search's.
    rules = tbc_opts.dominance (eager reads + twin rule, the library default): the kernel absorbs reads
    through the per-front read masks and tests twins through the per-entry twin masks of pack_open, the
    oracle walks the open-call lists; the witness is the kernel's chain of branching calls expanded by
    the host (tbc_api.hip) against the oracle's own replay."""
    cases = [(8, 3, 0.1, 0.5, 0.8), (40, 4, 0.05, 0.0, 0.5), (200, 8, 0.02, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3),
             (1000, 16, 0.02, 0.0, 0.5), (1000, 16, 0.0, 0.6, 0.2), (3000, 64, 0.02, 0.0, 0.1), (3000, 64, 0.0, 0.6, 0.05)]
    if rules:   # the concurrency the rules make feasible: 6, 19 and 32 calls in flight of 64 processes
        cases += [(2000, 64, 0.0, 0.0, 0.1), (2000, 64, 0.0, 0.0, 0.3), (2000, 64, 0.0, 0.0, 0.5), (2000, 64, 0.01, 0.0, 0.3),
                  (1500, 64, 0.0, 0.5, 0.3)]
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in cases for s in range(3)]
    opts = core.make_opts(time_limit_ms=60000, search_width=width, algorithm=N.ALG_COMPETITION, lookahead=lookahead,
                          eager_reads=rules, twin_rule=rules)
    with core.Batch(hists, gm(), opts) as b:
        res = b.run().results()
    single = core.check_ops(hists[5], gm(), opts)
    for i, (h, got) in enumerate(zip(hists, res)):
        exp = oracle.check_beam(h.as_dict(), CAS, width, lookahead=lookahead, eager_reads=rules, twin_rule=rules, max_probes=20_000_000)
        seq = oracle.check(h.as_dict(), CAS, "window", max_steps=5_000_000, want_witness=False)
        if exp["valid"] == -1:
            continue
        assert got["valid"] == exp["valid"], i
        if seq["valid"] != -1:
            assert got["valid"] == seq["valid"], i
        if exp["valid"] == 0:
            assert got["fail_op"] == exp["fail_op"], i
            if seq["valid"] != -1:
                assert got["fail_op"] == seq["fail_op"], i
            assert got["prev_ok_op"] == (None if exp["prev_ok_op"] == N.NO_OP else exp["prev_ok_op"]), i
        if exp["valid"] == 1:
            assert got["final_state"] == exp["final_state"], i
            assert np.array_equal(got["witness"], exp["witness"]), i
            assert brute.check_witness(CAS, op_tuples(h), [int(x) for x in got["witness"]]) == got["final_state"]
        assert (got["probes"], got["visited"], got["backtracks"], got["max_depth"]) == \
               (exp["probes"], exp["visited"], exp["expanded"], exp["max_stack"]), i
    assert single["valid"] == res[5]["valid"] and single["probes"] == res[5]["probes"]


def test_wide_schedule_overflow_retry_and_default_algorithm(native, oracle):
    ops = columns.pair_events(synth.register_events(n_ops=1000, n_procs=16, seed=1, busy=0.5, info=0.01, corrupt=0.5))
    got = core.check_ops(ops, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, visited_per_op=4))
    assert got["search_width"] in (2, 4)     # nobody named a width: the library's choice for this concurrency (8 live + the crashed calls)
    exp = oracle.check_beam(ops.as_dict(), CAS, got["search_width"])
    assert got["valid"] == exp["valid"] == 0 and got["fail_op"] == exp["fail_op"]
    assert got["visited"] == exp["visited"] and got["table_slots"] > 16 * len(ops)
    small = core.check_ops(ops, gm(), core.make_opts(algorithm=N.ALG_COMPETITION, max_visited_bytes=64 * 1024))
    assert small["valid"] == N.UNKNOWN and small["cause"] == N.CAUSE_VISITED_FULL
    # knossos.linear (the level sweep) keeps no visited set: the same cap does not concern it
    lin = core.check_ops(ops, gm(), core.make_opts(algorithm=N.ALG_LINEAR, max_visited_bytes=64 * 1024, want_witness=False))
    assert lin["valid"] == 0 and lin["fail_op"] == exp["fail_op"] and lin["analyzer"] == N.ALG_LINEAR


def test_batch_matches_oracle_and_single(native, oracle):
    hists = []
    for seed in range(48):
        n_ops, procs = [(30, 3), (300, 8), (1500, 32)][seed % 3]
        hists.append(columns.pair_events(synth.register_events(
            n_ops=n_ops, n_procs=procs, seed=seed, busy=0.3, info=0.02 * (seed % 2), corrupt=0.5 * (seed % 4 == 3))))
    exps = [oracle.check(h.as_dict(), CAS, "window", max_steps=3_000_000) for h in hists]
    keep = [i for i, e in enumerate(exps) if e["valid"] != -1]
    with core.Batch([hists[i] for i in keep], gm(), core.make_opts(time_limit_ms=60000)) as b:
        first = b.run().results()
        second = b.run().results()          # inputs stay resident; a re-run is idempotent
    for j, i in enumerate(keep):
        assert_same(first[j], exps[i], f"hist{i}")
        assert_same(second[j], exps[i], f"hist{i} rerun")
    assert any(e["valid"] == 0 for e in exps) and any(e["valid"] == 1 for e in exps)


def test_wide_window_many_crashed_processes(native, oracle):
    """> 64 processes (crashed ops retire their process id): 2- and 4-word masks."""
    for n_ops, info in ((1500, 0.05), (3000, 0.05)):
        ops = columns.pair_events(synth.register_events(n_ops=n_ops, n_procs=48, seed=3, busy=0.12, info=info))
        assert ops.n_process > 64
        exp = oracle.check(ops.as_dict(), CAS, "window", max_steps=20_000_000)
        assert exp["valid"] != -1
        got = core.check_ops(ops, gm(), core.make_opts(time_limit_ms=120000))
        assert_same(got, exp)


def test_visited_set_overflow_is_retried(native, oracle):
    """An invalid history whose search outgrows its first table (forced small: 4 entries per op)."""
    ops = columns.pair_events(synth.register_events(n_ops=1000, n_procs=16, seed=1, busy=0.5, info=0.01, corrupt=0.5))
    exp = oracle.check(ops.as_dict(), CAS, "window")
    assert exp["valid"] == 0 and exp["visited"] > 16 * len(ops)
    got = core.check_ops(ops, gm(), core.make_opts(time_limit_ms=60000, visited_per_op=4))
    assert_same(got, exp)
    assert got["table_slots"] > 16 * len(ops)
    # and with the cap too small to ever fit: :unknown, cause memory -- never a wrong verdict
    small = core.check_ops(ops, gm(), core.make_opts(max_visited_bytes=64 * 1024))
    assert small["valid"] == N.UNKNOWN and small["cause"] == N.CAUSE_VISITED_FULL


def test_step_limit_gives_unknown(native):
    ops = columns.pair_events(synth.register_events(n_ops=1000, n_procs=16, seed=1, busy=0.5, info=0.01, corrupt=0.5))
    r = core.check_ops(ops, gm(), core.make_opts(max_steps=1000))
    assert r["valid"] == N.UNKNOWN and r["cause"] == N.CAUSE_STEP_LIMIT


def test_full_size_properties_10k_ops_64_procs(native, oracle):
    """BASELINE.json config 2 shape.  Size-independent properties + the oracle."""
    for seed, info in ((0, 0.0), (1, 0.01)):
        ops = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=seed, busy=0.1, info=info))
        got = core.check_ops(ops, gm(), core.make_opts(time_limit_ms=60000))
        assert got["valid"] == N.VALID       # linearizable by construction
        wide = core.check_ops(ops, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION))
        assert wide["valid"] == N.VALID
        assert brute.check_witness(CAS, op_tuples(ops), [int(x) for x in wide["witness"]]) == wide["final_state"]
        tup = op_tuples(ops)
        assert brute.check_witness(CAS, tup, [int(x) for x in got["witness"]]) == got["final_state"]
        assert_same(got, oracle.check(ops.as_dict(), CAS, "window"))
    # one corrupted read => not linearizable, and the op reported is that read
    ev = synth.register_events(n_ops=10000, n_procs=64, seed=5, busy=0.04, info=0.0, corrupt=0.7)
    ops = columns.pair_events(ev)
    for alg in (N.ALG_WGL, N.ALG_COMPETITION):
        got = core.check_ops(ops, gm(), core.make_opts(time_limit_ms=60000, algorithm=alg))
        assert got["valid"] == N.INVALID
        assert ops.a[got["fail_op"]] == 5 + 7    # the impossible value the generator planted


def test_rejects_malformed_ops(native):
    ops = columns.pair_events(synth.register_events(n_ops=50, n_procs=4, seed=2))
    bad = columns.OpColumns(ops.f.copy(), ops.a.copy(), ops.b.copy(), ops.process.copy(), ops.inv_pos.copy(),
                            ops.ret_pos.copy(), ops.n_events, ops.n_process)
    bad.inv_pos[3], bad.inv_pos[4] = bad.inv_pos[4], bad.inv_pos[3]          # not ascending
    with pytest.raises(N.TbcError) as e:
        core.check_ops(bad, gm())
    assert e.value.status == N.ERR_BAD_HISTORY
    reg = core.make_model(N.MODEL_REGISTER, N.NIL)                           # :cas on a plain register
    if (ops.f == N.F_CAS).any():
        with pytest.raises(N.TbcError) as e:
            core.check_ops(ops, reg)
        assert e.value.status == N.ERR_MODEL


def test_mutex_and_table_models(native, oracle):
    h = []
    for i in range(40):   # two processes handing a lock back and forth, one bad double-acquire at the end
        p = i % 2
        h += [kop.invoke(p, "acquire", None), kop.ok(p, "acquire", None), kop.invoke(p, "release", None), kop.ok(p, "release", None)]
    assert wgl.analysis(M.mutex(), h)["valid?"] is True
    h += [kop.invoke(0, "acquire", None), kop.ok(0, "acquire", None), kop.invoke(1, "acquire", None), kop.ok(1, "acquire", None)]
    a = wgl.analysis(M.mutex(), h)
    assert a["valid?"] is False and a["op"]["index"] == len(h) - 1
    # memo-table path: the set model and multi-register through knossos.model.memo
    s = [kop.invoke(0, "add", 1), kop.invoke(1, "add", 2), kop.ok(1, "add", 2), kop.invoke(2, "read", None),
         kop.ok(2, "read", [2]), kop.ok(0, "add", 1), kop.invoke(2, "read", None), kop.ok(2, "read", [1, 2])]
    assert wgl.analysis(M.set(), s)["valid?"] is True
    s[-1] = kop.ok(2, "read", [2])      # add 1 completed before this read was invoked
    a = wgl.analysis(M.set(), s)
    assert a["valid?"] is False and a["op"]["index"] == 7
    t = [kop.invoke(0, "txn", [["w", "x", 1], ["w", "y", 1]]), kop.ok(0, "txn", [["w", "x", 1], ["w", "y", 1]]),
         kop.invoke(1, "txn", [["r", "x", None], ["r", "y", None]]), kop.ok(1, "txn", [["r", "x", 1], ["r", "y", 1]])]
    assert wgl.analysis(M.multi_register({}), t)["valid?"] is True
    t[-1] = kop.ok(1, "txn", [["r", "x", 1], ["r", "y", None]])
    assert wgl.analysis(M.multi_register({}), t)["valid?"] is True      # nil read matches anything
    t[-1] = kop.ok(1, "txn", [["r", "x", 1], ["r", "y", 2]])
    assert wgl.analysis(M.multi_register({}), t)["valid?"] is False


def test_jepsen_checker_surface(native):
    """checker/linearizable inside checker/compose inside independent/checker -- the
    composition the reference uses at set_full.clj:155-158 -- batched on the device."""
    t = independent.tuple_
    h = []
    for k in (1, 2, 3):
        h += [kop.invoke(k, "write", t(k, k)), kop.ok(k, "write", t(k, k)),
              kop.invoke(k, "read", t(k, None)), kop.ok(k, "read", t(k, k if k != 2 else 99))]
    h.insert(3, {"type": "info", "f": "start-partition", "process": "nemesis", "value": None})
    c = independent.checker(jc.linearizable({"model": M.cas_register(), "algorithm": "wgl"}))
    r = c.check({}, h, {})
    assert r["valid?"] is False and r["failures"] == [2]
    assert r["results"][1]["valid?"] is True and r["results"][2]["op"]["value"] == 99
    comp = jc.compose({"linear": jc.linearizable({"model": M.cas_register(), "algorithm": "linear"}),
                       "wgl": jc.linearizable({"model": M.cas_register()})})
    r2 = comp.check({}, independent.subhistory(1, h), {})
    assert r2["valid?"] is True and r2["linear"]["analyzer"] == "linear"


@pytest.mark.parametrize("alg", [N.ALG_WGL, N.ALG_COMPETITION])
def test_invalid_verdict_carries_the_stuck_configs(native, oracle, alg):
    """knossos :configs on failure: the (model state, linearized pending calls) pairs stuck at the
    failing completion, sorted, first 10.  The sequential order (:wgl) reports the plain search's set; the wide
    schedule (competition + witness) the set of the search under the dominance rules -- configs in normal form
    (reads absorbed, twins ordered), a subset -- each against the oracle of its own schedule."""
    for seed in range(4):
        ops = columns.pair_events(synth.register_events(n_ops=1000, n_procs=16, seed=seed, busy=0.2, corrupt=0.6))
        got = core.check_ops(ops, gm(), core.make_opts(time_limit_ms=60000, algorithm=alg))
        if alg == N.ALG_WGL:
            exp = oracle.check(ops.as_dict(), CAS, "window")
            total, rows = oracle.last_configs("window")
        else:
            assert got["search_width"] == 2                  # 3 calls in flight, none crashed: the narrow default
            exp = oracle.check_beam(ops.as_dict(), CAS, got["search_width"])
            total, rows = oracle.last_configs("beam")
        assert exp["valid"] == 0
        assert got["valid"] == N.INVALID and got["fail_op"] == exp["fail_op"]
        assert len(got["configs"]) == min(total, 10) and total <= 256
        for c, row in zip(got["configs"], rows):
            assert c["state"] == row[0]
            assert c["n_pending"] <= 16
            mask = 0
            for k, op in enumerate(c["pending"]):
                assert ops.inv_pos[op] < ops.ret_pos[exp["fail_op"]] and (ops.ret_pos[op] >= ops.ret_pos[exp["fail_op"]])
                if c["linearized_mask"] >> k & 1:
                    mask |= 1 << int(ops.process[op])
            assert mask == row[1]
        assert exp["fail_op"] in got["configs"][0]["pending"]          # the failing call is open, never linearized
    a = wgl.analysis(M.cas_register(), [kop.invoke(0, "write", 1), kop.ok(0, "write", 1), kop.invoke(1, "read", None), kop.ok(1, "read", 2)])
    assert a["valid?"] is False and a["configs"] and a["configs"][0]["model"] == M.CASRegister(1)
    assert [o["f"] for o in a["configs"][0]["pending"]] == ["read"]


def test_default_width_follows_the_calls_in_flight(native, oracle):
    """tbc_opts.search_width = 0: 2 configs per round for a register-family batch under both dominance rules at no
    more than 10 calls in flight on average, else 4; a named width is taken as named.  Whatever it becomes is reported
    (tbc_result.search_width, tbc_batch_search_width) and is the oracle's schedule at that width, bit for bit."""
    quiet = [columns.pair_events(synth.register_events(n_ops=2000, n_procs=64, seed=s, busy=0.1)) for s in range(3)]     # 6.4 in flight
    busy = [columns.pair_events(synth.register_events(n_ops=2000, n_procs=64, seed=s, busy=0.3)) for s in range(3)]     # 19 in flight
    for hists, kw, want in ((quiet, {}, 2), (busy, {}, 4), (quiet, {"search_width": 8}, 8), (quiet, {"twin_rule": False}, 4),
                            (quiet, {"algorithm": N.ALG_WGL}, 1)):
        opts = core.make_opts(**{"time_limit_ms": 60000, "algorithm": N.ALG_COMPETITION, **kw})
        with core.Batch(hists, gm(), opts) as b:
            assert b.search_width() == want, (kw, b.search_width())
            res = b.run().results()
        if want == 1:
            continue
        for h, got in zip(hists, res):
            exp = oracle.check_beam(h.as_dict(), CAS, want, twin_rule=kw.get("twin_rule", True))
            assert got["search_width"] == want and got["valid"] == exp["valid"] == 1
            assert np.array_equal(got["witness"], exp["witness"])
            assert (got["probes"], got["visited"], got["backtracks"]) == (exp["probes"], exp["visited"], exp["expanded"])


def test_round_budget_widens_stragglers_in_place(native, oracle):
    """tbc_opts.round_budget: a history that has used more rounds than the budget continues at width 16
    (same table, same stack).  Still the deterministic schedule of oracle/wgl_beam.c (widen_after)."""
    hists = [columns.pair_events(synth.register_events(n_ops=1500, n_procs=24, seed=s, busy=0.25, info=0.01)) for s in range(24)]
    exp4 = [oracle.check_beam(h.as_dict(), CAS, 4) for h in hists]
    rounds = sorted(e["rounds"] for e in exp4)
    budget = rounds[len(rounds) // 3]               # most histories cross it
    opts = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=4, round_budget=budget)
    with core.Batch(hists, gm(), opts) as b:
        res = b.run().results()
    n_wide = 0
    for h, got, e4 in zip(hists, res, exp4):
        exp = oracle.check_beam(h.as_dict(), CAS, 4, widen_after=budget)
        n_wide += e4["rounds"] > budget
        assert got["valid"] == exp["valid"] == 1 and got["cause"] == 0
        assert np.array_equal(got["witness"], exp["witness"])
        assert (got["probes"], got["visited"], got["backtracks"]) == (exp["probes"], exp["visited"], exp["expanded"])
        if e4["rounds"] > budget:
            assert exp["probes"] != e4["probes"]      # the schedule really changed
    assert n_wide >= 8


def test_against_committed_golden_fixtures(native):
    """The HIP kernels against tests/golden/synth_golden.json alone (no live oracle in the loop)."""
    import hashlib
    import json
    import os
    from helpers import GOLDEN
    cases = json.load(open(os.path.join(GOLDEN, "synth_golden.json")))["cases"]
    hists = [columns.pair_events(synth.register_events(**g["case"])) for g in cases]
    sha = lambda w: hashlib.sha256(np.asarray(w).astype("<u4").tobytes()).hexdigest()[:16]
    for opts, key in ((core.make_opts(time_limit_ms=60000), "sequential"),
                      (core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=8, eager_reads=False, twin_rule=False), "wide8"),
                      (core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=8), "wide8_rules")):
        with core.Batch(hists, gm(), opts) as b:
            res = b.run().results()
        for g, got in zip(cases, res):
            assert got["valid"] == g["valid"]
            if g["valid"] == 1:
                assert got["final_state"] == g[key]["final_state"] and sha(got["witness"]) == g[key]["witness_sha"]
            else:
                assert got["fail_op"] == g["fail_op"]
            assert got["visited"] == g[key]["visited"]
            assert got["probes"] == (g[key]["steps"] if key == "sequential" else g[key]["probes"])


@pytest.mark.gpu
def test_lookahead_value_range_crashed_writers_and_plain_register(native, oracle):
    """Lookahead corner cases, kernel records (pack_open.hip) against the oracle's definition: register
    values outside 0..31 (those ranks constrain nothing), many crashed calls (crashed writers stay
    producers for ever; the mask grows past one word), a plain register, and heavy concurrency."""
    cases = [dict(n_ops=1500, n_procs=16, n_values=40, busy=0.3, info=0.0),
             dict(n_ops=1500, n_procs=16, n_values=40, busy=0.3, info=0.02, corrupt=0.3),
             dict(n_ops=600, n_procs=24, n_values=3, busy=0.3, info=0.08),
             dict(n_ops=400, n_procs=16, n_values=3, busy=0.3, info=0.04, corrupt=0.4),
             dict(n_ops=2000, n_procs=64, n_values=2, busy=0.15, info=0.01),
             dict(n_ops=800, n_procs=8, n_values=5, busy=0.5, read=0.5, write=0.5)]          # plain register
    for ci, c in enumerate(cases):
        model_kind = N.MODEL_REGISTER if c.get("read", 0) + c.get("write", 0) == 1 else N.MODEL_CAS_REGISTER
        om = {"kind": 0 if model_kind == N.MODEL_REGISTER else 1, "init": N.NIL}
        hists = [columns.pair_events(synth.register_events(seed=40 + s, **c)) for s in range(4)]
        for width in (4, 16):
            with core.Batch(hists, core.make_model(model_kind, N.NIL),
                            core.make_opts(time_limit_ms=60000, search_width=width, algorithm=N.ALG_COMPETITION)) as b:
                res = b.run().results()
            for i, (h, got) in enumerate(zip(hists, res)):
                exp = oracle.check_beam(h.as_dict(), om, width, max_probes=20_000_000)
                if exp["valid"] == -1:
                    continue
                assert got["valid"] == exp["valid"], (ci, width, i)
                assert (got["probes"], got["visited"], got["backtracks"]) == (exp["probes"], exp["visited"], exp["expanded"]), (ci, width, i)
                if exp["valid"] == 1:
                    assert np.array_equal(got["witness"], exp["witness"]), (ci, width, i)
                else:
                    assert got["fail_op"] == exp["fail_op"], (ci, width, i)


def test_each_dominance_rule_alone(native, oracle):
    """tbc_opts.dominance bit by bit: eager reads without the twin rule and the twin rule without eager
    reads are schedules of their own (oracle/wgl_beam.c flags) and must match bit for bit as well."""
    cases = [(300, 8, 0.03, 0.0, 0.5), (300, 8, 0.0, 0.5, 0.4), (1500, 32, 0.01, 0.0, 0.3), (1500, 64, 0.0, 0.0, 0.4)]
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=70 + s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in cases for s in range(3)]
    for eager, twin in ((True, False), (False, True)):
        opts = core.make_opts(time_limit_ms=60000, search_width=4, algorithm=N.ALG_COMPETITION, eager_reads=eager, twin_rule=twin)
        with core.Batch(hists, gm(), opts) as b:
            res = b.run().results()
        for i, (h, got) in enumerate(zip(hists, res)):
            exp = oracle.check_beam(h.as_dict(), CAS, 4, eager_reads=eager, twin_rule=twin, max_probes=20_000_000)
            if exp["valid"] == -1:
                continue
            assert got["valid"] == exp["valid"], (eager, twin, i)
            assert (got["probes"], got["visited"], got["backtracks"]) == (exp["probes"], exp["visited"], exp["expanded"]), (eager, twin, i)
            if exp["valid"] == 1:
                assert np.array_equal(got["witness"], exp["witness"]), (eager, twin, i)
                assert brute.check_witness(CAS, op_tuples(h), [int(x) for x in got["witness"]]) == got["final_state"]
            else:
                assert got["fail_op"] == exp["fail_op"], (eager, twin, i)


def test_independent_keys_share_one_value_encoding(native):
    """jepsen.independent checks all keys in ONE batch with ONE model struct: the register values of all keys are
    interned together, so a key that sees a string does not shift the meaning of the initial value for the others
    (round-1 advisor finding), and the remaining options (max-steps) reach the batch."""
    T = independent.tuple_
    h = []
    def op(p, f, k, v_inv, v_ok):
        h.append(kop.invoke(p, f, T(k, v_inv))); h.append(kop.ok(p, f, T(k, v_ok)))
    op(0, "read", "a", None, 5)          # key a: ints only; the initial value 5 is readable
    op(0, "write", "a", 1, 1); op(1, "read", "a", None, 1)
    op(2, "read", "b", None, 5)          # key b: a string among its values; the initial value is still 5
    op(2, "write", "b", "x", "x"); op(3, "read", "b", None, "x")
    op(4, "read", "c", None, 7)          # key c: 7 was never written and is not the initial value
    for alg in ("wgl", "linear", None):
        r = independent.checker(jc.linearizable({"model": M.register(5), "algorithm": alg})).check(None, h, None)
        assert r["results"]["a"]["valid?"] is True, alg
        assert r["results"]["b"]["valid?"] is True, alg
        assert r["results"]["c"]["valid?"] is False and r["failures"] == ["c"], alg
    # a budget of a single step: every key must come back :unknown (the option used to be dropped on this path)
    r = independent.checker(jc.linearizable({"model": M.register(5), "algorithm": "wgl", "max-steps": 1})).check(None, h, None)
    assert r["results"]["a"]["valid?"] == "unknown" and r["results"]["a"]["cause"] == "step-limit"


def test_pack_rejects_out_of_range_pool_references(native):
    """Malformed C-ABI input for the pool-backed models must end as TBC_ERR_MODEL, not as an out-of-bounds read
    (round-1 advisor finding): a set read whose record runs past the pool, an add index past the add count, a bank
    transfer between accounts that do not exist."""
    from jepsen_tigerbeetle_amd.knossos import _analysis as A
    hs = [kop.invoke(0, "add", 1), kop.ok(0, "add", 1), kop.invoke(1, "read", None), kop.ok(1, "read", [1])]
    e = A.Encoded(M.SetModel(), hs)
    good = core.check_ops(e.ops, e.native_model, core.make_opts(algorithm=N.ALG_COMPETITION))
    assert good["valid"] == N.VALID
    for mutate in ("read_offset", "add_index"):
        ops = columns.OpColumns(e.ops.f.copy(), e.ops.a.copy(), e.ops.b.copy(), e.ops.process.copy(), e.ops.inv_pos.copy(),
                                e.ops.ret_pos.copy(), e.ops.n_events, e.ops.n_process)
        ops.pool = e.ops.pool.copy()
        if mutate == "read_offset":
            ops.a[ops.f == N.F_READ] = len(ops.pool) - 1
        else:
            ops.a[ops.f == N.F_ADD] = 1000
        with pytest.raises(N.TbcError) as err:
            core.check_ops(ops, e.native_model, core.make_opts(algorithm=N.ALG_COMPETITION))
        assert err.value.status == N.ERR_MODEL, mutate


# ---- K5n: several histories per wavefront (wgl_narrow.hip; the same body runs emulated in tests/test_narrow_emu.py)
NARROW_SHAPES = [(8, 3, 0.1, 0.0, 0.8), (8, 3, 0.1, 0.5, 0.8), (40, 4, 0.0, 0.0, 0.5), (40, 4, 0.05, 0.5, 0.5), (200, 8, 0.02, 0.0, 0.5),
                 (200, 8, 0.0, 0.6, 0.3), (1000, 16, 0.0, 0.0, 0.5), (1000, 16, 0.01, 0.0, 0.3), (1000, 16, 0.0, 0.6, 0.2),
                 (3000, 64, 0.0, 0.0, 0.1), (3000, 64, 0.0, 0.6, 0.05)]


def _narrow_expect(oracle, h, L, model=None, **kw):
    # (under the eager rule the narrow kernel's lists hold :write / :cas only and its root starts in normal form: branch_lists)
    return oracle.check_beam(h.as_dict(), model or CAS, 1, round_pairs=L, rules_at_any_round_size=True,
                             branch_lists=kw.get("eager_reads", True), **kw)


def _assert_narrow(got, exp, tag):
    assert got["valid"] == exp["valid"], (tag, got["valid"], exp["valid"], got["cause"])
    assert got["search_width"] == 1, tag
    assert (got["probes"], got["visited"], got["backtracks"], got["max_depth"]) == (exp["probes"], exp["visited"], exp["expanded"], exp["max_stack"]), tag
    if exp["valid"] == 1:
        assert got["final_state"] == exp["final_state"], tag
        if got["witness"] is not None:
            assert np.array_equal(got["witness"], exp["witness"]), tag
    elif exp["valid"] == 0:
        assert got["fail_op"] == exp["fail_op"], tag


@pytest.mark.parametrize("L", [8, 16, 32])
def test_narrow_kernel_matches_its_oracle(native, oracle, L):
    """lanes_per_history = L: 64 / L histories per wavefront, each on the oracle's schedule for one config per iteration and
    L pairs per round -- every shape x 3 seeds in ONE launch, so wavefronts mix lengths, verdicts and ends."""
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in NARROW_SHAPES for s in range(3)]
    with core.Batch(hists, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, lanes_per_history=L)) as b:
        assert b.lanes_per_history() == L and b.search_width() == 1
        res = b.run().results()
        again = b.run().results()
    for i, (h, got) in enumerate(zip(hists, res)):
        _assert_narrow(got, _narrow_expect(oracle, h, L), (L, i))
        assert (again[i]["valid"], again[i]["probes"], again[i]["visited"]) == (got["valid"], got["probes"], got["visited"])      # idempotent


def test_narrow_kernel_rules_lookahead_growth_and_limits(native, oracle):
    hists = [columns.pair_events(synth.register_events(n_ops=600, n_procs=8, seed=s, busy=0.3, info=0.01, corrupt=c)) for s in range(6) for c in (0.0, 0.5)]
    for kw, okw in (({"eager_reads": False}, {"eager_reads": False}), ({"twin_rule": False}, {"twin_rule": False}), ({"lookahead": False}, {"lookahead": False}),
                    ({"eager_reads": False, "twin_rule": False, "lookahead": False}, {"eager_reads": False, "twin_rule": False, "lookahead": False})):
        with core.Batch(hists, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, lanes_per_history=8, **kw)) as b:
            res = b.run().results()
        for i, (h, got) in enumerate(zip(hists, res)):
            _assert_narrow(got, _narrow_expect(oracle, h, 8, **okw), (str(kw), i))
    # first visited sets of one entry per op: histories of one wavefront outgrow theirs inside the kernel (growth pool)
    big = [columns.pair_events(synth.register_events(n_ops=2500, n_procs=16, seed=s, busy=0.25, corrupt=c)) for s in range(8) for c in (0.0, 0.4)]
    with core.Batch(big, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, lanes_per_history=8, visited_per_op=1)) as b:
        res = b.run().results()
    for i, (h, got) in enumerate(zip(big, res)):
        _assert_narrow(got, _narrow_expect(oracle, h, 8), ("grow", i))
    assert max(r["table_slots"] for r in res) >= 4 * 4096
    # the step limit, counted as the oracle counts it (after the round that exceeded it)
    with core.Batch(hists, gm(), core.make_opts(algorithm=N.ALG_COMPETITION, lanes_per_history=16, max_steps=57, want_witness=False)) as b:
        res = b.run().results()
    for i, (h, got) in enumerate(zip(hists, res)):
        exp = _narrow_expect(oracle, h, 16, max_probes=57)
        assert got["valid"] == exp["valid"] and (exp["valid"] != -1 or got["cause"] == N.CAUSE_STEP_LIMIT), i
        assert (got["probes"], got["visited"]) == (exp["probes"], exp["visited"]), i


def test_narrow_kernel_wide_masks_and_other_models(native, oracle):
    h2 = [columns.pair_events(synth.register_events(n_ops=1200, n_procs=24, seed=s, busy=0.15, info=0.05)) for s in range(4)]
    assert all(64 < h.n_process <= 128 for h in h2)
    h4 = [columns.pair_events(synth.register_events(n_ops=2000, n_procs=32, seed=s, busy=0.1, info=0.08)) for s in range(3)]
    assert all(128 < h.n_process <= 256 for h in h4)
    for hists, L in ((h2, 8), (h2, 16), (h4, 8)):
        with core.Batch(hists, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, lanes_per_history=L, max_steps=60000)) as b:
            res = b.run().results()
        for i, (h, got) in enumerate(zip(hists, res)):
            _assert_narrow(got, _narrow_expect(oracle, h, L, max_probes=60000), (L, h.n_process, i))
    # plain register (no :cas): the same kernel
    reg = [columns.pair_events(synth.register_events(n_ops=500, n_procs=8, seed=s, busy=0.3, info=0.01, corrupt=c, read=0.5, write=0.5)) for s in range(4) for c in (0.0, 0.5)]
    with core.Batch(reg, core.make_model(N.MODEL_REGISTER, N.NIL), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, lanes_per_history=8)) as b:
        res = b.run().results()
    for i, (h, got) in enumerate(zip(reg, res)):
        _assert_narrow(got, _narrow_expect(oracle, h, 8, model={"kind": 0, "init": N.NIL}), ("register", i))
    # what the narrow kernel does not do is refused by name, not answered some other way
    with pytest.raises(N.TbcError):
        core.Batch(reg, core.make_model(N.MODEL_REGISTER, N.NIL), core.make_opts(algorithm=N.ALG_WGL, lanes_per_history=8))
    with pytest.raises(N.TbcError):
        core.Batch(reg, core.make_model(N.MODEL_REGISTER, N.NIL), core.make_opts(algorithm=N.ALG_COMPETITION, lanes_per_history=8, search_width=4))


def test_batches_in_flight_from_several_threads(native, oracle):
    """tbcheck.h, "several batches in flight": three resident batches, each run four times from its own host thread at the same
    time (their searches take the device in turn, everything else floats) -- every pass of every batch gives what the batch
    gives alone, which is what its oracle gives."""
    import threading
    sets = [[columns.pair_events(synth.register_events(n_ops=200 + 40 * k, n_procs=16, seed=100 * k + s, busy=0.2, corrupt=0.3 if s % 5 == 0 else 0.0))
             for s in range(96)] for k in range(3)]
    opts = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, lanes_per_history=8)
    batches = [core.Batch(h, gm(), opts) for h in sets]
    try:
        alone = []
        for b in batches:
            res = b.run().results()
            alone.append([(r["valid"], r["probes"], r["visited"], r["backtracks"], r["fail_op"]) for r in res])
        for k in range(3):
            for i in (0, 5, 17, 95):
                _assert_narrow(batches[k].results()[i], _narrow_expect(oracle, sets[k][i], 8), (k, i))
        seen = [[] for _ in batches]
        errors = []
        go = threading.Barrier(3)
        def work(k):
            try:
                go.wait()
                for _ in range(4):
                    res = batches[k].run().results()
                    seen[k].append([(r["valid"], r["probes"], r["visited"], r["backtracks"], r["fail_op"]) for r in res])
                    assert batches[k].timing_ns()["search"] > 0
            except Exception as e:          # noqa: BLE001 -- reported by the asserting thread below
                errors.append((k, repr(e)))
        th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
        for t in th: t.start()
        for t in th: t.join()
        assert not errors, errors
        for k in range(3):
            assert len(seen[k]) == 4 and all(x == alone[k] for x in seen[k]), k
    finally:
        for b in batches:
            b.close()


def test_big_quiet_batches_take_the_narrow_kernel_by_default(native, oracle):
    """tbc_opts.lanes_per_history = 0 and search_width = 0: a batch of >= 24,576 register-family histories at low concurrency
    under both rules runs 8 to a wavefront (more wavefronts than the GPU holds at once: groups take their next history off the
    queue as they finish one); smaller or busier batches keep a wavefront each."""
    base = [columns.pair_events(synth.register_events(n_ops=120, n_procs=16, seed=s, busy=0.2)) for s in range(64)]
    opts = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False)
    NB = 24576 + 4096 + 8
    with core.Batch([base[i % 64] for i in range(NB)], gm(), opts) as b:
        assert (b.lanes_per_history(), b.search_width()) == (8, 1)
        res = b.run().results()
    for i in range(64):
        exp = _narrow_expect(oracle, base[i], 8)
        for k in list(range(0, NB - 64, 64 * 37)) + [NB - 64 - (NB % 64)]:
            _assert_narrow(res[i + k], exp, (i, k))
    assert all(r["valid"] == 1 and r["probes"] == res[j % 64]["probes"] for j, r in enumerate(res))      # every one of them, refilled groups included
    with core.Batch([base[i % 64] for i in range(4096)], gm(), opts) as b:
        assert (b.lanes_per_history(), b.search_width()) == (64, 2)


def test_narrow_kernel_at_the_bench_configuration(native, oracle):
    """bench.py's own configuration, bit for bit: 24,576 histories of 10k invocations / 64 processes at 10 % duty in ONE batch
    (so the library takes the narrow kernel by itself: 8 histories per wavefront, compact 64 B front records, branch lists, first
    visited sets of 4 entries per op that grow inside the kernel, more histories than group slots: the work queue refills) --
    a sample of 64 valid histories and three with a planted bad read against the oracle's schedule for that kernel: verdict,
    failing op and every counter (probes, new configs, configs expanded, deepest stack)."""
    B = 24576
    hists = synth.register_ops_many(range(7_000_000, 7_000_000 + B), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
    planted = [11, 4097, B - 5]
    for i in planted:
        hists[i] = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=7_000_000 + i, busy=0.1, corrupt=0.5))
    opts = core.make_opts(time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=4, list_order=N.ORDER_DEFAULT)      # bench.py's options
    with core.Batch(hists, gm(), opts) as b:
        assert (b.lanes_per_history(), b.search_width(), b.list_order()) == (8, 1, 16 + 24)      # the shipped default: in order of completion, a :write 24 ranks later
        b.run()
        verdicts = b.verdicts()
        res = b.results()
    sample = sorted(set(list(range(0, B, B // 61))[:61] + planted + [B - 1, B - 2, 1]))
    for i in sample:
        _assert_narrow(res[i], _narrow_expect(oracle, hists[i], 8, want_witness=False, list_order=16 + 24), i)
    assert [int(verdicts[i]) for i in planted] == [0, 0, 0]
    assert int((verdicts == 1).sum()) == B - len(planted)                      # element-wise: only the planted ones are invalid
