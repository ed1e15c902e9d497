"""K5n on the CPU: the narrow search kernel's own body (jepsen-tigerbeetle_amd/csrc/wgl_narrow_impl.h, several histories
per wavefront) compiled for the wavefront emulator of tests/emu/ and compared with the oracle's schedule for one config per
iteration and L pairs per round (oracle/wgl_beam.c, wgl_beam_check_rp(K = 1, round_pairs = L)) -- verdict, failing op,
witness chain, final state and every counter, bit for bit.  The tables the kernel reads are built on the host from their
definitions (tests/emu/emu_narrow.cpp), a second formulation of what the pack kernels write.  The same comparisons run on
the device in tests/test_gpu_parity.py; here a schedule bug costs seconds to find and needs no GPU.

The emulator is test infrastructure: only this file and tests/emu/ use it; libtbcheck.so has no CPU path."""
import numpy as np
import pytest

import emu
from jepsen_tigerbeetle_amd import _native as N, columns, synth
from oracle import wgl

CAS = {"kind": 1, "init": N.NIL}


def expand_chain(d, chain, model, eager, branch=False, completion_order=False):
    """The chain of branching calls replayed from the initial state, absorbing reads as the search does (a third
    formulation, besides oracle/wgl_beam.c's and tbc_api.hip's expand_eager_witness).  branch: the root starts in normal
    form (its own reads come first).  completion_order: the lists are in order of completion (tbc_opts.list_order), not by slot."""
    if not eager:
        return [int(x) for x in chain]
    f, a, b, proc = (np.asarray(d[k]) for k in ("f", "a", "b", "process"))
    inv, ret = np.asarray(d["inv_pos"], np.int64), np.asarray(d["ret_pos"], np.int64)
    n = len(f)
    live = [i for i in range(n) if ret[i] != 0xFFFFFFFF]
    by_ret = sorted(live, key=lambda i: ret[i])
    R = len(by_ret)
    rank_of_pos = {int(ret[i]): r for r, i in enumerate(by_ret)}
    inv_rank = [sum(1 for i2 in by_ret if ret[i2] < inv[i]) for i in range(n)] if n < 400 else None
    if inv_rank is None:
        rets = np.sort(ret[live])
        inv_rank = np.searchsorted(rets, inv, side="left").tolist()
    ret_rank = {i: rank_of_pos[int(ret[i])] for i in live}
    done, out = set(), []
    front, state = 0, model["init"]

    def advance():
        nonlocal front
        moved = False
        while front < R and by_ret[front] in done:
            front += 1
            moved = True
        return moved

    for op in ([None] if branch else []) + list(chain):
        if op is not None:
            op = int(op)
            state = int(a[op]) if f[op] == 1 else (int(b[op]) if f[op] == 2 else state)
            done.add(op); out.append(op)
            advance()
        again = True
        while again and front < R:
            opens = sorted((i for i in live if i not in done and inv_rank[i] <= front <= ret_rank[i]), key=(lambda i: ret_rank[i]) if completion_order else (lambda i: proc[i]))
            for x in opens:
                if f[x] == 0 and (a[x] == N.NIL or a[x] == state):
                    done.add(x); out.append(x)
            again = advance()
    return out


def compare(hists, model, L, kind=1, tag="", **kw):
    ds = [h.as_dict() for h in hists]
    got = emu.run(ds, kind, model["init"], L, **kw)
    for i, (d, g) in enumerate(zip(ds, got)):
        e = wgl.check_beam(d, model, 1, round_pairs=L, rules_at_any_round_size=True, lookahead=kw.get("lookahead", True),
                           eager_reads=bool(g["rules"] & 1), twin_rule=bool(g["rules"] & 2), max_probes=kw.get("max_steps", 0),
                           branch_lists=bool(g["rules"] & 4), list_order=wgl.ORACLE_LIST_ORDER[int(kw.get("by_ret", 0))])
        t = (tag, i, L)
        assert g["valid"] == e["valid"], (t, g["valid"], e["valid"], g["cause"])
        for a_, b_ in (("probes", "probes"), ("visited", "visited"), ("backtracks", "expanded"), ("max_depth", "max_stack"), ("bucket_reads", "rounds")):
            assert g[a_] == e[b_], (t, a_, g[a_], e[b_])
        if e["valid"] == 0:
            assert (g["fail_op"], g["prev_ok_op"]) == (e["fail_op"], e["prev_ok_op"]), t
        if e["valid"] == 1 and len(d["f"]):
            assert g["final_state"] == e["final_state"], t
            if g["chain"] is not None:
                    assert expand_chain(d, g["chain"], model, bool(g["rules"] & 1), bool(g["rules"] & 4), bool(kw.get("by_ret"))) == [int(x) for x in e["witness"]], t
                    if g["rules"] & 1:        # ... and the product's own replay (csrc/witness_expand.h: what tbc_api.hip runs on the device's chain)
                        assert emu.expand_witness(d, g["chain"], model["init"], bool(g["rules"] & 4), bool(kw.get("by_ret"))) == [int(x) for x in e["witness"]], t
    return got


SHAPES = [(8, 3, 0.1, 0.0, 0.8), (8, 3, 0.1, 0.5, 0.8), (40, 4, 0.0, 0.0, 0.5), (40, 4, 0.05, 0.5, 0.5), (200, 8, 0.02, 0.0, 0.5),
          (200, 8, 0.0, 0.6, 0.3), (1000, 16, 0.0, 0.0, 0.5), (1000, 16, 0.01, 0.0, 0.3), (1000, 16, 0.0, 0.6, 0.2), (2000, 64, 0.0, 0.0, 0.1)]


@pytest.mark.parametrize("L", [8, 16])
def test_every_shape_one_batch_per_width(L):
    """all shapes x 3 seeds in ONE launch: wavefronts hold histories of different lengths, verdicts and ends"""
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in SHAPES for s in range(3)]
    compare(hists, CAS, L, tag="shapes", pool_words=4_000_000)


@pytest.mark.parametrize("L", [4, 32])
def test_other_group_sizes(L):
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in SHAPES[:7] for s in range(2)]
    compare(hists, CAS, L, tag="sizes", pool_words=4_000_000)


def test_without_a_witness_no_parent_links_are_kept():
    """want_witness = 0: verdict, failing op and counters as ever; the kernel writes no parent links (and reads none)"""
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in SHAPES[:9] for s in range(2)]
    got = compare(hists, CAS, 8, tag="no-links", pool_words=4_000_000, want_witness=False)
    assert all(g["depth"] == 0 for g in got)


def test_wide_front_records_too():
    """compact = False: the 128 B front records (what batches with more than six register states, or more than 64 slots, get)"""
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in SHAPES[:9] for s in range(2)]
    compare(hists, CAS, 8, tag="wide-records", pool_words=4_000_000, compact=False)
    many = [columns.pair_events(synth.register_events(n_ops=500, n_procs=8, seed=s, busy=0.3, n_values=9)) for s in range(4)]      # values 0..8: no compact form
    compare(many, CAS, 8, tag="nine-values", pool_words=4_000_000)


def test_full_lists_too():
    """branch lists off: the per-front lists hold every live open call (what the wide kernel and the sweep read), the root as given"""
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in SHAPES[:8] for s in range(2)]
    compare(hists, CAS, 8, tag="full-lists", pool_words=4_000_000, branch_lists=False)


def test_rules_and_lookahead_switched_off_one_by_one():
    hists = [columns.pair_events(synth.register_events(n_ops=300, n_procs=6, seed=s, busy=0.3, info=0.01, corrupt=c)) for s in range(4) for c in (0.0, 0.5)]
    for rules in (0, 1, 2, 3):
        for look in (True, False):
            compare(hists, CAS, 8, tag=f"rules{rules}look{look}", rules=rules, lookahead=look, pool_words=4_000_000)


def test_visited_sets_grow_inside_the_kernel():
    """first visited sets of 1 entry per op (1,024 at least): several histories of a wavefront outgrow theirs, some twice"""
    hists = [columns.pair_events(synth.register_events(n_ops=2500, n_procs=16, seed=s, busy=0.25, info=0.0, corrupt=c)) for s in range(5) for c in (0.0, 0.4)]
    got = compare(hists, CAS, 8, tag="grow", entries_per_op=1, pool_words=8_000_000)
    first = [max(10, int(np.ceil(np.log2(max(1, len(h.as_dict()["f"])))))) for h in hists]
    assert sum(g["tab_log2"] > f0 for g, f0 in zip(got, first)) >= 5 and max(g["tab_log2"] - f0 for g, f0 in zip(got, first)) >= 2
    # without a pool a history that outgrows its set ends UNKNOWN / VISITED_FULL (the host then retries with a larger one)
    small = emu.run([h.as_dict() for h in hists[:3]], 1, N.NIL, 8, entries_per_op=1, pool_words=0)
    assert all(g["valid"] == N.UNKNOWN and g["cause"] == N.CAUSE_VISITED_FULL for g in small)


def test_step_limit_as_the_oracle_counts_it():
    hists = [columns.pair_events(synth.register_events(n_ops=600, n_procs=8, seed=s, busy=0.4, info=0.0, corrupt=0.0)) for s in range(6)]
    for limit in (1, 57, 300):
        got = compare(hists, CAS, 8, tag=f"limit{limit}", max_steps=limit, pool_words=1_000_000)
        assert all(g["valid"] == N.UNKNOWN and g["cause"] == N.CAUSE_STEP_LIMIT for g in got)


def test_plain_register():
    reg = {"kind": 0, "init": N.NIL}
    hists = [columns.pair_events(synth.register_events(n_ops=400, n_procs=8, seed=s, busy=0.3, info=0.01, corrupt=c, read=0.5, write=0.5)) for s in range(3) for c in (0.0, 0.5)]
    compare(hists, reg, 8, kind=0, tag="register", pool_words=1_000_000)


def test_wide_masks_many_crashed_processes():
    """crashed calls retire their process slot: 2 and 4 mask words; the crashed candidates' twin walk"""
    h2 = [columns.pair_events(synth.register_events(n_ops=1200, n_procs=24, seed=s, busy=0.15, info=0.05, corrupt=0.0)) for s in range(3)]
    assert all(64 < h.n_process <= 128 for h in h2), [h.n_process for h in h2]
    compare(h2, CAS, 8, tag="mw2", pool_words=4_000_000, max_steps=60_000)
    compare(h2, CAS, 16, tag="mw2", pool_words=4_000_000, max_steps=60_000)
    h4 = [columns.pair_events(synth.register_events(n_ops=2000, n_procs=32, seed=s, busy=0.1, info=0.08, corrupt=0.0)) for s in range(2)]
    assert all(128 < h.n_process <= 256 for h in h4), [h.n_process for h in h4]
    compare(h4, CAS, 8, tag="mw4", pool_words=4_000_000, max_steps=40_000)


@pytest.mark.parametrize("L,waves", [(8, 1), (8, 2), (16, 3)])
def test_groups_take_more_work_as_they_finish(L, waves):
    """fewer wavefronts than the batch needs (a launch sized to the GPU, not to the batch): a group that has finished its
    history reports it and takes the next one off the queue -- every history still gets its own schedule's answer"""
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in SHAPES for s in range(3)]
    compare(hists, CAS, L, tag="refill", pool_words=4_000_000, max_waves=waves)
    compare(hists[:20], CAS, L, tag="refill-small-tables", pool_words=4_000_000, max_waves=1, entries_per_op=1)


def test_the_bench_configuration_at_full_size():
    """BASELINE.json configs[1] as bench.py runs it: 10k-invocation / 64-process cas-register histories at 10 % duty, 8 lanes per
    history, compact front records, branch lists, first visited sets of 4 entries per op, no witness -- 24 histories (three
    wavefronts' worth, two of them waiting on the queue) against the oracle's schedule, every counter."""
    hists = synth.register_ops_many(range(7000, 7024), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
    got = compare(hists, CAS, 8, tag="bench", entries_per_op=4, pool_words=8_000_000, want_witness=False, max_waves=1)
    assert all(g["valid"] == 1 for g in got) and sum(g["probes"] for g in got) > 200_000
    bad = [columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=s, busy=0.1, corrupt=0.5)) for s in (12345, 12346)]
    got = compare(bad, CAS, 8, tag="bench-invalid", entries_per_op=4, pool_words=8_000_000, want_witness=False)
    assert all(g["valid"] == 0 for g in got)


def test_epoch_tags_instead_of_zeroed_visited_sets():
    """libtbcheck does not zero a batch's visited sets before every pass any more: keys carry the pass number (BeamArgs.epoch) and
    another pass's entries read as empty.  Three passes over ONE arena that is never cleared, each under its own tag -- with the
    invalid histories' crowded tables, growth into the (zeroed) pool, and both record forms -- give what one pass over a zeroed
    arena gives, which is what the oracle gives."""
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in SHAPES[:9] for s in range(2)]
    for kw in ({}, {"compact": False}, {"want_witness": False}):
        compare(hists, CAS, 8, tag=f"epochs{kw}", pool_words=4_000_000, epochs=3, **kw)
    grow = [columns.pair_events(synth.register_events(n_ops=2500, n_procs=16, seed=s, busy=0.25, info=0.0, corrupt=c)) for s in range(3) for c in (0.0, 0.4)]
    compare(grow, CAS, 8, tag="epochs-grow", entries_per_op=1, pool_words=8_000_000, epochs=3)
    wide = [columns.pair_events(synth.register_events(n_ops=600, n_procs=12, seed=s, busy=0.15, info=0.08)) for s in range(2)]
    compare(wide, CAS, 8, tag="epochs-two-words", pool_words=4_000_000, epochs=2, max_steps=30000)


def compare_count(hists, L, tag="", relaxed=False, targets=None, **kw):
    """K5n in the COUNT FORM (crashed calls as counts per effect class) against oracle/wgl_count.c at one config per iteration and L pairs per round."""
    ds = [h.as_dict() for h in hists]
    got = emu.run(ds, 1, N.NIL, L, count=True, relaxed=relaxed, targets=targets, mw=kw.pop("mw", 1), **kw)
    n_cls = 0
    for i, (d, g) in enumerate(zip(ds, got)):
        e = wgl.check_count(d, CAS, width=1, round_pairs=L, lookahead=kw.get("lookahead", True), relaxed=relaxed,
                            target=int(targets[i]) if targets is not None else 0, max_probes=kw.get("max_steps", 0))
        assert e is not None
        t = (tag, i, L)
        assert g["valid"] == e["valid"], (t, g["valid"], e["valid"], g["cause"])
        for a_, b_ in (("probes", "probes"), ("visited", "visited"), ("backtracks", "expanded"), ("max_depth", "max_stack"), ("bucket_reads", "rounds")):
            assert g[a_] == e[b_], (t, a_, g[a_], e[b_])
        if e["valid"] == 0:
            assert (g["fail_op"], g["prev_ok_op"]) == (e["fail_op"], e["prev_ok_op"]), t
        if e["valid"] == 1 and len(d["f"]) and g["chain"] is not None and not relaxed and targets is None:
            assert g["final_state"] == e["final_state"], t
            # the chain of branching calls (crashed calls among them) replayed with the eager rule = the oracle's witness
            assert expand_chain_count(d, g["chain"]) == [int(x) for x in e["witness"]], t
        n_cls += e["class_steps"]
    return got, n_cls


def expand_chain_count(d, chain):
    """as expand_chain, for a chain that may hold crashed calls (they set the state and hold no slot); lists in SLOT order (the count form's
    re-used slots, from the oracle)"""
    slots = wgl.check_count(d, CAS, width=1, want_slots=True)["slots"]
    f, a, b = (np.asarray(d[k]) for k in ("f", "a", "b"))
    inv, ret = np.asarray(d["inv_pos"], np.int64), np.asarray(d["ret_pos"], np.int64)
    n = len(f)
    live = [i for i in range(n) if ret[i] != 0xFFFFFFFF]
    by_ret = sorted(live, key=lambda i: ret[i])
    R = len(by_ret)
    rets = np.sort(ret[live])
    inv_rank = np.searchsorted(rets, inv, side="left").tolist()
    ret_rank = {i: r for r, i in enumerate(by_ret)}
    done, out = set(), []
    front, state = 0, N.NIL

    def advance():
        nonlocal front
        moved = False
        while front < R and by_ret[front] in done:
            front += 1
            moved = True
        return moved

    for op in chain:
        op = int(op)
        state = int(a[op]) if f[op] == 1 else (int(b[op]) if f[op] == 2 else state)
        done.add(op); out.append(op)
        advance()
        again = True
        while again and front < R:
            opens = sorted((i for i in live if i not in done and inv_rank[i] <= front <= ret_rank[i]), key=lambda i: slots[i])
            for x in opens:
                if f[x] == 0 and (a[x] == N.NIL or a[x] == state):
                    done.add(x); out.append(x)
            again = advance()
    return out


COUNT_SHAPES = [(40, 4, 0.3, 0.0, 0.5), (40, 4, 0.3, 0.5, 0.5), (300, 8, 0.05, 0.0, 0.3), (300, 8, 0.05, 0.5, 0.3), (1000, 16, 0.03, 0.0, 0.3),
                (1000, 16, 0.03, 0.6, 0.3), (600, 4, 0.2, 0.0, 0.8), (2000, 64, 0.02, 0.0, 0.1)]


@pytest.mark.parametrize("L", [8, 16, 32])
def test_count_form_several_histories_per_wavefront(L):
    """crashed calls as counts per effect class in the narrow kernel: classes as candidates, hot configs, the Pareto chain walk in the
    probe, re-used slots -- every shape x 2 seeds in one launch; a step limit that stops the invalid histories' exhaustion where the
    oracle's stops"""
    hists = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=40 + s, busy=busy, info=info, corrupt=corrupt))
             for (n, p, info, corrupt, busy) in COUNT_SHAPES for s in range(2)]
    hists = [h for h in hists if wgl.check_count(h.as_dict(), CAS, width=1, max_probes=1) is not None]
    got, n_cls = compare_count(hists, L, tag="count", pool_words=8_000_000, max_steps=40000)
    assert n_cls > 500 and sum(g["valid"] == 1 for g in got) >= 6


def test_count_form_relaxed_prefix_growth_and_epochs():
    hists = [columns.pair_events(synth.register_events(n_ops=400, n_procs=8, seed=60 + s, busy=0.3, info=0.06, corrupt=0.6)) for s in range(6)]
    rel, _ = compare_count(hists, 8, tag="relaxed", relaxed=True, pool_words=4_000_000)
    assert all(g["valid"] == 0 for g in rel)
    # the prefix of each history before the completion the relaxed search could not pass: a linearization of it ends the search VALID
    targets = np.array([g["max_front"] for g in rel], np.uint32)
    assert (targets > 0).all()
    pre, _ = compare_count(hists, 8, tag="prefix", targets=targets, pool_words=4_000_000, want_witness=False)
    assert all(g["valid"] == 1 for g in pre)
    # small first tables: growth with the count words carried along; several passes over one arena under epoch tags; two mask words
    valid = [columns.pair_events(synth.register_events(n_ops=1500, n_procs=16, seed=70 + s, busy=0.25, info=0.03)) for s in range(4)]
    compare_count(valid, 8, tag="grow", entries_per_op=1, pool_words=8_000_000)
    compare_count(valid[:2], 16, tag="epochs", pool_words=4_000_000, epochs=3, want_witness=False)
    wide = [columns.pair_events(synth.register_events(n_ops=500, n_procs=70, seed=80 + s, busy=0.5, info=0.05)) for s in range(2)]
    compare_count(wide, 8, tag="two-words", pool_words=4_000_000, mw=2, max_steps=30000)


# ---- the fronts' lists in order of completion (tbc_opts.list_order; csrc PackOpenArgs.list_order): the kernel reads what it is given, the
# schedule is the oracle's with that list order -- the call that completes soonest is tried first.  16 + 24 is the library's default
# wherever the order applies (round 5); 1 / 2 are the plain forms of the same walk.
def _in_domain(n, p, s, busy, info, corrupt):
    """a history whose planted bad read (if any) stays inside the value domain 0..4 (compact front records)"""
    h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt, n_values=4 if corrupt else 5))
    h.a[h.a == 4 + 7] = 4
    return h


ORDER_SHAPES = [(8, 3, 0.0, 0.0, 0.8), (40, 4, 0.0, 0.5, 0.5), (200, 8, 0.0, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3), (1000, 16, 0.0, 0.0, 0.5),
                (1000, 16, 0.01, 0.0, 0.3), (1000, 16, 0.0, 0.6, 0.2), (2000, 64, 0.0, 0.0, 0.1), (600, 24, 0.03, 0.0, 0.6)]


@pytest.mark.parametrize("by_ret", [1, 2, 16 + 24, 16 + 5])
def test_lists_in_order_of_completion_every_counter(by_ret):
    hists = [_in_domain(n, p, s, busy, info, corrupt) for (n, p, info, corrupt, busy) in ORDER_SHAPES for s in range(2)]
    hists = [h for h in hists if h.n_process <= 64]
    # (the chain's absorbed reads come out in list order: the third formulation above, expand_chain, replays them in that order too)
    compare(hists, CAS, 8, tag="by ret", pool_words=4_000_000, by_ret=by_ret)
    compare(hists[:10], CAS, 16, tag="by ret 16", pool_words=4_000_000, by_ret=by_ret, want_witness=False)
    compare(hists, CAS, 4, tag="by ret 4", pool_words=4_000_000, by_ret=by_ret, want_witness=False)      # (16 histories a wavefront)
    h = [columns.pair_events(synth.register_events(n_ops=300, n_procs=24, seed=s, busy=1.0, n_values=2)) for s in range(4)]      # many backtracks
    compare(h, CAS, 8, tag="by ret busy", pool_words=8_000_000, by_ret=by_ret, want_witness=False)


def test_default_list_order_growth_epochs_queue():
    hists = [_in_domain(1500, 16, s, 0.4, 0.0, 0.5 * (s % 2)) for s in range(12)]
    compare(hists, CAS, 8, tag="order growth", entries_per_op=1, pool_words=6_000_000, by_ret=16 + 24, want_witness=False)
    compare(hists, CAS, 8, tag="order epochs", epochs=3, pool_words=6_000_000, by_ret=16 + 24)
    compare(hists, CAS, 8, tag="order queue", max_waves=1, pool_words=6_000_000, by_ret=16 + 24, want_witness=False)


def test_the_default_list_order_needs_fewer_rounds_at_the_bench_configuration():
    """what the order buys (oracle counts, the kernel following them): on the bench workload about a fifth fewer rounds than slot order,
    and the write delay a few per cent over plain completion order; at 19 calls in flight much more"""
    hists = synth.register_ops_many(range(7000, 7004), n_ops=10000, n_procs=64, busy=0.1, info=0.0) + \
            synth.register_ops_many(range(7200, 7202), n_ops=10000, n_procs=64, busy=0.3, info=0.0)
    soft = compare(hists, CAS, 8, tag="a write 24 ranks later, bench", entries_per_op=4, pool_words=1 << 25, want_witness=False, by_ret=16 + 24)
    slot = [wgl.check_beam(h.as_dict(), CAS, 1, round_pairs=8, rules_at_any_round_size=True, branch_lists=True, want_witness=False) for h in hists]
    plain = [wgl.check_beam(h.as_dict(), CAS, 1, round_pairs=8, rules_at_any_round_size=True, branch_lists=True, want_witness=False, list_order=1) for h in hists]
    assert sum(g["bucket_reads"] for g in soft) < 0.97 * sum(p["rounds"] for p in plain)
    assert sum(g["bucket_reads"] for g in soft[:4]) < 0.9 * sum(p["rounds"] for p in slot[:4])


def test_a_history_that_stops_passing_completions_is_stopped():
    """BeamArgs.stall_checks (the library: 64 looks at the clock = 4,096 rounds; tbc_api.hip hands such a history to the level sweep): valid bench
    histories -- the one of them whose burst of concurrency costs 1,500 rounds included -- run to their end with the oracle's counters, a history
    with a bad read in its middle is stopped (UNKNOWN, step limit) long before it has exhausted what lies in front of that read"""
    hists = synth.register_ops_many(range(7000, 7004), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
    compare(hists, CAS, 8, tag="stall, valid", entries_per_op=4, pool_words=1 << 25, want_witness=False, by_ret=16 + 24, stall=64)
    bad = _in_domain(10000, 64, 7100, 0.1, 0.0, 0.5)
    full = wgl.check_beam(bad.as_dict(), CAS, 1, round_pairs=8, rules_at_any_round_size=True, branch_lists=True, list_order=16 + 24, want_witness=False)
    assert full["valid"] == 0
    got = emu.run([bad.as_dict()], 1, N.NIL, 8, entries_per_op=4, pool_words=1 << 25, want_witness=False, by_ret=16 + 24, stall=64)[0]
    assert (got["valid"], got["cause"]) == (-1, 2)
    assert got["bucket_reads"] < 0.5 * full["rounds"] and got["bucket_reads"] > 4096
