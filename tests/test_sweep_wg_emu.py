"""K6w -- the level sweep with a WORKGROUP per segment (jepsen-tigerbeetle_amd/csrc/jit_sweep_wg_impl.h) -- run on the CPU under the
workgroup emulator (tests/emu/wave_env_wg_emu.h: NW x 64 fibers, wavefronts interleaved in seeded orders) against oracle/sweep_ref.c:
every record a workgroup leaves (status, levels, level sizes, sub-rounds, probes, the relation, the origins' last levels) is the
record the oracle writes for that (segment, slice), and the library's own composition of them gives the oracle's verdict."""
import ctypes as C

import numpy as np
import pytest

import jepsen_tigerbeetle_amd  # noqa: F401
from jepsen_tigerbeetle_amd import _native as N, columns, synth
from oracle import wgl

import emu

CAS = {"kind": 1, "init": N.NIL}


def _records(buf):
    return np.frombuffer(buf, dtype=np.uint8).reshape(-1, C.sizeof(N.SweepRel))


def _compare(ops, seg_target, n_dom, waves, cap=1024, rules=None, seed=1, expect_overflow=False, second_pass=False, compact=False, relaxed=False):
    d = ops.as_dict()
    R = int((np.asarray(d["ret_pos"]) != 0xFFFFFFFF).sum())
    max_segs = max(1, min(512, (R + seg_target - 1) // seg_target)) if seg_target else 1
    eager = twin = True if rules is None else bool(rules)
    L = wgl.lib()
    buf = np.zeros(max_segs * 4 * C.sizeof(N.SweepRel), np.uint8)
    L.sweep_set_export(buf.ctypes.data_as(C.c_void_p), C.c_uint32(max_segs), C.c_uint32(0), C.c_uint32(1))
    try:
        wgl.check_sweep(d, CAS, eager_reads=eager, twin_rule=twin, seg_target=seg_target, n_dom=n_dom, relaxed=relaxed)
    finally:
        L.sweep_set_export(None, C.c_uint32(0), C.c_uint32(0), C.c_uint32(1))
    got = emu.sweep_wg(d, 1, N.NIL, seg_target, n_dom, max_segs, waves=waves, cap=cap, rules=None if rules is None else (3 if rules else 0), seed=seed, compact=compact, relaxed=relaxed)
    want = [N.SweepRel.from_buffer_copy(r.tobytes()) for r in _records(buf)]
    have = [N.SweepRel.from_buffer_copy(r.tobytes()) for r in _records(got)]
    if second_pass:      # the segments that overflowed, once more with sets of 2,048 configs and eight wavefronts (what launch_sweep does)
        again = [(i // 4, i % 4) for i, g in enumerate(have) if g.status == 2]
        assert (len(again) > 0) == expect_overflow
        expect_overflow = False
        got = emu.sweep_wg(d, 1, N.NIL, seg_target, n_dom, max_segs, waves=8, cap=2048, rules=None if rules is None else (3 if rules else 0), seed=seed + 5, again=again, first=got)
        have = [N.SweepRel.from_buffer_copy(r.tobytes()) for r in _records(got)]
    n_swept = overflowed = 0
    for i, (w, g) in enumerate(zip(want, have)):
        if g.status == 2:
            overflowed += 1
            continue
        assert g.status == w.status, (i, g.status, w.status)
        if w.status == 0:
            continue
        n_swept += 1
        for k in ("F0", "F1", "max_level", "subrounds", "n_end", "configs_total", "probes"):
            assert getattr(g, k) == getattr(w, k), (i, k, getattr(g, k), getattr(w, k))
        assert [list(r) for r in g.M] == [list(r) for r in w.M], (i, "relation")
        assert list(g.last_level) == list(w.last_level), (i, "last levels")
    assert (overflowed > 0) == expect_overflow
    if not overflowed:      # the library's own composition (host code, no device) of K6w's records = the oracle's verdict
        ref = wgl.check_sweep(d, CAS, eager_reads=eager, twin_rule=twin, seg_target=seg_target, n_dom=n_dom, relaxed=relaxed)
        v = N.SweepVerdict()
        recs = (N.SweepRel * len(have))(*have)
        assert N.lib().tbc_sweep_compose(recs, C.c_uint32(max_segs), C.c_uint32(R), C.byref(v)) == 0
        assert v.valid == ref["valid"]
        if ref["valid"] == 0:
            order = np.argsort(np.asarray(d["ret_pos"], np.int64) + (np.asarray(d["ret_pos"]) == 0xFFFFFFFF) * (1 << 40), kind="stable")
            assert int(order[v.fail_level]) == ref["fail_op"]
        else:
            assert v.probes == ref["probes"] and v.configs_total == ref["configs_total"]
    return n_swept


@pytest.mark.parametrize("waves", [2, 4, 8])
def test_small_histories_every_record(waves):
    n = 0
    for seed in range(6):
        for corrupt in (0.0, 0.4):
            h = columns.pair_events(synth.register_events(n_ops=300, n_procs=8, seed=100 + seed, busy=0.5, info=0.0, corrupt=corrupt))
            n += _compare(h, 32, 6, waves, seed=seed)
    assert n > 20


def _compare_sample(ops, seg_target, n_dom, waves, cap=1024, seed=1, heavy=24, every=12, **form):
    """As _compare, for the experimental forms on a full-size history, on a SAMPLE of its workgroups (the emulator runs one
    workgroup after the other): the `heavy` (segment, slice) pairs with the most probes -- the bursts, where the forms differ from
    the plain walk -- and every `every`-th of the rest.  The plain form's test sweeps every workgroup of such a history."""
    d = ops.as_dict()
    R = int((np.asarray(d["ret_pos"]) != 0xFFFFFFFF).sum())
    max_segs = max(1, min(512, (R + seg_target - 1) // seg_target))
    L = wgl.lib()
    buf = np.zeros(max_segs * 4 * C.sizeof(N.SweepRel), np.uint8)
    L.sweep_set_export(buf.ctypes.data_as(C.c_void_p), C.c_uint32(max_segs), C.c_uint32(0), C.c_uint32(1))
    try:
        wgl.check_sweep(d, CAS, seg_target=seg_target, n_dom=n_dom, relaxed=bool(form.get("relaxed")))
    finally:
        L.sweep_set_export(None, C.c_uint32(0), C.c_uint32(0), C.c_uint32(1))
    want = [N.SweepRel.from_buffer_copy(r.tobytes()) for r in _records(buf)]
    swept = [i for i, w in enumerate(want) if w.status == 1]
    by_probes = sorted(swept, key=lambda i: -want[i].probes)
    pick = sorted(set(by_probes[:heavy]) | set(swept[::every]))
    got = emu.sweep_wg(d, 1, N.NIL, seg_target, n_dom, max_segs, waves=waves, cap=cap, seed=seed, again=[(i // 4, i % 4) for i in pick],
                       first=np.zeros_like(buf), **form)
    have = [N.SweepRel.from_buffer_copy(r.tobytes()) for r in _records(got)]
    compared = 0
    for i in pick:
        w, g = want[i], have[i]
        if g.status == 2 and form.get("relaxed"):          # (a relaxed burst that outgrows the sets: reported, the library's depth-first passes take the history)
            continue
        assert g.status == w.status, (i, g.status, w.status)
        for k in ("F0", "F1", "max_level", "subrounds", "n_end", "configs_total", "probes"):
            assert getattr(g, k) == getattr(w, k), (i, k, getattr(g, k), getattr(w, k))
        assert [list(r) for r in g.M] == [list(r) for r in w.M], (i, "relation")
        assert list(g.last_level) == list(w.last_level), (i, "last levels")
        compared += 1
    return compared


def test_rules_off_and_one_segment():
    for seed in range(3):
        h = columns.pair_events(synth.register_events(n_ops=120, n_procs=5, seed=200 + seed, busy=0.6, info=0.0, corrupt=0.0))
        _compare(h, 32, 6, 4, rules=False)
        _compare(h, 0, 6, 4)


def test_wavefront_interleavings_do_not_matter():
    """the same history under eight schedules of the wavefronts (a read of LDS another wavefront writes with no barrier between
    them would differ somewhere)"""
    h = columns.pair_events(synth.register_events(n_ops=600, n_procs=16, seed=7, busy=0.5, info=0.0, corrupt=0.0))
    for seed in range(8):
        _compare(h, 32, 6, 8, seed=1000 + 17 * seed)


def test_a_bench_history_with_its_bursts():
    """one 10k-invocation / 64-process history of the bench: ~340 workgroups, among them the burst segment (a level of hundreds of
    configs, sub-rounds of thousands of pairs) that the workgroup form exists for"""
    h = synth.register_ops_many([3], n_ops=10000, n_procs=64, busy=0.1, info=0.0)[0]
    assert _compare(h, 32, 6, 8) > 250


def test_histories_with_crashed_calls():
    """crashed calls stay candidates for ever (K6's list behind the live calls, their twins computed from the level's records) and
    end the cutting at the first of them; a level or a sub-round set past the capacity ends the segment with status 2"""
    for seed in (49, 54, 58, 59):
        h = columns.pair_events(synth.register_events(n_ops=300, n_procs=8, seed=seed, busy=0.4, info=0.015, corrupt=0.0))
        assert int((np.asarray(h.as_dict()["ret_pos"]) == 0xFFFFFFFF).sum()) >= 5
        _compare(h, 32, 6, 4)
    h = columns.pair_events(synth.register_events(n_ops=400, n_procs=8, seed=31, busy=0.5, info=0.03, corrupt=0.0))     # levels of 11,520 configs
    _compare(h, 32, 6, 4, expect_overflow=True)


def test_overflow_is_reported_not_mis_swept():
    """sets of 512 configs against a burst that needs more: the segment ends with status 2 (the host then sweeps it again with
    K6's big sets), every other record is still the oracle's"""
    h = synth.register_ops_many([4], n_ops=10000, n_procs=64, busy=0.1, info=0.0)[0]
    ref = wgl.check_sweep(h.as_dict(), CAS, seg_target=32, n_dom=6)
    assert max(ref["max_level"], ref["max_pending"]) > 512
    _compare(h, 32, 6, 4, cap=512, expect_overflow=True)
    _compare(h, 32, 6, 4, cap=512, expect_overflow=True, second_pass=True)        # ... and the second pass makes every record the oracle's


# ---- the compact walk (COMPACT): a sub-round's passes take 64 x NW CHILDREN, not 64 x NW (config, call) slots
def test_the_compact_walk_every_record():
    """the same records as the plain form: small histories valid and invalid, rules off (reads are candidates: more children per
    config), crashed calls (one long segment, classes of candidates past the live calls), fewer and more wavefronts"""
    n = 0
    for seed in range(4):
        for corrupt in (0.0, 0.4):
            h = columns.pair_events(synth.register_events(n_ops=300, n_procs=8, seed=100 + seed, busy=0.5, info=0.0, corrupt=corrupt))
            n += _compare(h, 32, 6, (2, 4, 8, 8)[seed], seed=seed, compact=True)
    assert n > 12
    h = columns.pair_events(synth.register_events(n_ops=400, n_procs=8, seed=5, busy=0.6, info=0.0, corrupt=0.0))
    _compare(h, 32, 6, 8, rules=False, seed=3, compact=True)
    h = columns.pair_events(synth.register_events(n_ops=400, n_procs=16, seed=5, busy=0.8, info=0.0, corrupt=0.0))      # levels past 1,024 configs without the rules
    _compare(h, 32, 6, 8, rules=False, seed=3, compact=True, expect_overflow=True)
    h = columns.pair_events(synth.register_events(n_ops=300, n_procs=8, seed=58, busy=0.4, info=0.015, corrupt=0.0))
    _compare(h, 32, 6, 8, seed=4, compact=True)
    _compare(h, 0, 6, 4, seed=5, compact=True)                    # one segment


def test_the_compact_walk_on_a_bench_history_its_bursts_and_overflow():
    """a 10k-op bench history (349 workgroups; its bursts are the sub-rounds of more than two plain passes the compact walk takes:
    blocks of 512 configs, several insertion passes a block); the history whose burst overflows the small sets: reported"""
    h = synth.register_ops_many([3], n_ops=10000, n_procs=64, busy=0.1, info=0.0)[0]
    assert _compare_sample(h, 32, 6, 8, compact=True) > 40
    h = columns.pair_events(synth.register_events(n_ops=600, n_procs=16, seed=7, busy=0.5, info=0.0, corrupt=0.0))
    for seed in range(2):
        _compare(h, 32, 6, 2, cap=512, seed=3000 + 17 * seed, compact=True)       # (two wavefronts: 128 children a pass -- many blocks, many passes)
    h4 = synth.register_ops_many([4], n_ops=10000, n_procs=64, busy=0.1, info=0.0)[0]
    _compare(h4, 32, 6, 4, cap=512, expect_overflow=True, compact=True)


# ---- narrow passes by wavefront 0 alone (SOLO, with the compact walk): the library's first pass since round 5
def test_solo_passes_every_record():
    """a level of at most 64 configs / a sub-round of at most 64 slots is wavefront 0's alone (one workgroup barrier instead of three):
    the same records -- small histories (nearly every pass is narrow), rules off, crashed calls, one segment, several interleavings"""
    n = 0
    for seed in range(4):
        for corrupt in (0.0, 0.4):
            h = columns.pair_events(synth.register_events(n_ops=300, n_procs=8, seed=100 + seed, busy=0.5, info=0.0, corrupt=corrupt))
            n += _compare(h, 32, 6, (2, 4, 8, 8)[seed], seed=seed, compact=2)
    assert n > 12
    h = columns.pair_events(synth.register_events(n_ops=400, n_procs=8, seed=5, busy=0.6, info=0.0, corrupt=0.0))
    _compare(h, 32, 6, 8, rules=False, seed=3, compact=2)
    h = columns.pair_events(synth.register_events(n_ops=300, n_procs=8, seed=58, busy=0.4, info=0.015, corrupt=0.0))
    _compare(h, 32, 6, 8, seed=4, compact=2)
    _compare(h, 0, 6, 4, seed=5, compact=2)
    h = columns.pair_events(synth.register_events(n_ops=600, n_procs=16, seed=7, busy=0.5, info=0.0, corrupt=0.0))
    for seed in range(3):
        _compare(h, 32, 6, 8, seed=4000 + 19 * seed, compact=2)


def test_solo_passes_on_a_bench_history_and_overflow():
    h = synth.register_ops_many([3], n_ops=10000, n_procs=64, busy=0.1, info=0.0)[0]
    assert _compare_sample(h, 32, 6, 8, compact=2) > 40
    h4 = synth.register_ops_many([4], n_ops=10000, n_procs=64, busy=0.1, info=0.0)[0]
    _compare(h4, 32, 6, 4, cap=512, expect_overflow=True, compact=2)


# ---- the RELAXED sweep of a history with crashed calls (RLX; csrc/reach_table.h; oracle/sweep_ref.c sweep_set_relaxed): the crashed calls
# as classes in unlimited supply, a sub-round's pairs compound steps -- every record of every workgroup against the oracle's
def test_the_relaxed_sweep_every_record():
    n = 0
    for seed in range(6):
        for corrupt in (0.0, 0.4):
            for info in (0.05, 0.2):
                h = columns.pair_events(synth.register_events(n_ops=300, n_procs=8, seed=300 + seed, busy=0.5, info=info, corrupt=corrupt))
                n += _compare(h, 32, 6, (2, 8, 8)[seed % 3], seed=seed, relaxed=True)
    assert n > 40
    # two values, busy processes, many crashed writes: hops that absorb reads, chains of class steps
    for seed in range(4):
        h = columns.pair_events(synth.register_events(n_ops=200, n_procs=12, seed=400 + seed, busy=0.9, info=0.3, n_values=2, corrupt=0.3 * (seed % 2)))
        _compare(h, 16, 4, 8, seed=seed, relaxed=True)
    h = columns.pair_events(synth.register_events(n_ops=400, n_procs=8, seed=5, busy=0.6, info=0.1))
    _compare(h, 0, 6, 8, seed=5, relaxed=True)                    # one segment
    _compare(h, 32, 6, 4, cap=512, seed=6, relaxed=True)


def test_the_relaxed_sweep_on_bench_tiers_and_the_second_pass():
    """the bench's crashed-op tiers (10k ops, 1 % and 5 % crashed, one bad read planted): a sample of the workgroups, the bursts among them"""
    for info, seed in ((0.01, 4242), (0.05, 4242)):
        h = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=seed, busy=0.1, info=info, corrupt=0.5))
        assert _compare_sample(h, 32, 6, 8, cap=2048, relaxed=True) > 30
