"""K1 for one history -- pack by a workgroup's sixteen wavefronts (csrc/pack_one_impl.h, the file hipcc compiles into libtbcheck.so;
what tbc_check packs one history with) -- on the CPU under the workgroup emulator of tests/emu, against a restatement of what
pack.hip's header defines (tests/emu/emu_pack.cpp): every record and sentinel, list start, completion table entry, rank and place in
the scratch arena, n_ret and status, word for word, under several seeded interleavings of the wavefronts.  Test infrastructure
only: the product has no CPU path."""
import numpy as np
import pytest

import emu
from jepsen_tigerbeetle_amd import _native as N, columns, synth

SHAPES = [  # n_ops, n_procs, busy, info
    (300, 16, 0.3, 0.0), (500, 64, 0.1, 0.0), (400, 64, 0.9, 0.0), (260, 8, 1.0, 0.02), (350, 24, 0.5, 0.05), (64, 3, 0.5, 0.0),
    (65, 64, 1.0, 0.0), (1, 1, 0.5, 0.0), (2, 2, 1.0, 0.0), (700, 33, 0.2, 0.01), (1023, 7, 0.7, 0.0), (1024, 64, 0.4, 0.0), (1025, 5, 1.0, 0.1),
    (63, 1, 1.0, 0.0), (128, 2, 1.0, 0.0)]


def _hists(seeds=(1, 2), shapes=SHAPES):
    return [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info))
            for (n, p, busy, info) in shapes for s in seeds]


def _d(h):
    d = dict(h.as_dict())
    for k in ("f", "a", "b", "process", "inv_pos", "ret_pos"):
        d[k] = np.array(d[k], copy=True)
    return d


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_every_word_of_the_pack(seed):
    """all shapes (full and ragged chunks, fewer chunks than wavefronts, one op, crashed calls, 1 .. 64+ slots), one workgroup each,
    under three interleavings of the sixteen wavefronts"""
    assert emu.pack_one_check(_hists(), seed=seed) is None


def test_launched_a_few_at_a_time():
    """the histories of a batch packed by several launches (PackArgs.h0 moves), as tbc_api.hip's range launches do"""
    h = _hists(seeds=(5,))
    assert emu.pack_one_check(h, per_launch=4, seed=7) is None
    assert emu.pack_one_check(h, per_launch=1, seed=8) is None


def test_count_form_inputs():
    """count form: live calls on re-used slots, a crashed call holds none -- ranks and scratch only, no record"""
    h = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info))
         for (n, p, busy, info) in [(600, 16, 0.5, 0.1), (900, 64, 0.2, 0.3), (130, 4, 1.0, 0.5), (64, 8, 0.5, 1.0)] for s in (1, 2)]
    assert any((np.asarray(x.as_dict()["ret_pos"]) == 0xFFFFFFFF).sum() > 20 for x in h)
    assert emu.pack_one_check(h, count=True, seed=3) is None
    assert emu.pack_one_check(h, count=False, seed=4) is None            # the same histories in the mask form: a slot per crashed call


def test_full_size_histories():
    """BASELINE.json configs[1]: 10k invocations / 64 processes (157 chunks: ten per wavefront, the last wavefront short), at 10 % and
    50 % duty, with and without crashed calls; and a 60k-op history (the LDS bitmap nearly full)"""
    h = synth.register_ops_many(range(7000, 7003), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
    busy = synth.register_ops_many(range(7100, 7101), n_ops=10000, n_procs=64, busy=0.5, info=0.0)
    crashed = synth.register_ops_many(range(7200, 7201), n_ops=10000, n_procs=64, busy=0.1, info=0.01)
    assert emu.pack_one_check(h + busy + crashed, seed=11) is None
    assert emu.pack_one_check(crashed, count=True, seed=12) is None
    big = synth.register_ops_many(range(7300, 7301), n_ops=60000, n_procs=64, busy=0.3, info=0.0)
    assert len(big[0]) > 40000
    assert emu.pack_one_check(big, seed=13) is None


def test_many_slots():
    """a crash-heavy history in the mask form: hundreds of process slots (the scan over the slots runs past one wavefront)"""
    h = [columns.pair_events(synth.register_events(n_ops=3000, n_procs=64, seed=s, busy=0.5, info=0.2)) for s in (1, 2)]
    assert max(x.n_process for x in h) > 300
    assert emu.pack_one_check(h, seed=5) is None


def test_other_models_and_what_they_refuse():
    h = _hists(seeds=(9,), shapes=[(300, 16, 0.5, 0.0)])[0]
    assert (np.asarray(h.as_dict()["f"]) == N.F_CAS).any()
    assert emu.pack_one_check([h], model_kind=N.MODEL_CAS_REGISTER) is None
    assert emu.pack_one_check([h], model_kind=N.MODEL_REGISTER) is None          # a :cas in a plain register's history: TBC_ERR_MODEL on both sides
    d = _d(h)
    d["f"][:] = np.where(np.arange(len(d["f"])) % 2 == 0, N.F_ACQUIRE, N.F_RELEASE)
    assert emu.pack_one_check([d], model_kind=N.MODEL_MUTEX) is None
    d["f"][:] = N.F_CLASS
    d["a"][:] = np.arange(len(d["a"])) % 7
    assert emu.pack_one_check([d], model_kind=N.MODEL_TABLE, n_classes=7) is None
    assert emu.pack_one_check([d], model_kind=N.MODEL_TABLE, n_classes=6) is None  # class 6 is out of range: refused


def test_histories_pack_refuses():
    """every check of pack.hip's header, one broken row each (status and n_ret as pack_kernel leaves them); a good history beside it"""
    good = _hists(seeds=(4,), shapes=[(400, 12, 0.6, 0.02)])[0]
    cases = []
    d = _d(good); d["inv_pos"][100], d["inv_pos"][101] = d["inv_pos"][101], d["inv_pos"][100]; cases.append(d)       # invocations not ascending
    d = _d(good); i = int(np.argmax(d["ret_pos"] != 0xFFFFFFFF)); d["ret_pos"][i] = d["inv_pos"][i]; cases.append(d)  # completion not after invocation
    d = _d(good); d["process"][7] = d["n_process"]; cases.append(d)                                                  # process out of range
    d = _d(good); d["process"][9] = -1; cases.append(d)
    d = _d(good)
    live = np.flatnonzero(d["ret_pos"] != 0xFFFFFFFF)
    later = [j for j in live if d["inv_pos"][j] < d["ret_pos"][live[0]]]
    d["ret_pos"][later[-1]] = d["ret_pos"][live[0]]; cases.append(d)                                                # two completions on one row
    d = _d(good)
    p0 = np.flatnonzero(d["process"] == d["process"][0])
    d["ret_pos"][p0[0]] = d["inv_pos"][p0[1]] + 1
    cases.append(d)                                                                                                 # two calls of one process open at once ...
    d = _d(good); d["f"][33] = 77; cases.append(d)                                                                  # an op the model does not know
    d = _d(good); d["f"][33] = 77; d["process"][200] = 5000; cases.append(d)                                        # ... and a bad row: the model error wins
    for k, d in enumerate(cases):
        assert emu.pack_one_check([good, d, good], seed=20 + k) is None, k
    # a position past the history's rows
    d = _d(good)
    ne = int(max(d["inv_pos"].max(), d["ret_pos"][d["ret_pos"] != 0xFFFFFFFF].max())) + 1
    assert emu.pack_one_check([d], n_events=[ne - 1]) is None
    assert emu.pack_one_check([d], n_events=[ne + 1000]) is None          # rows after the last op (nemesis lines): fine


# ---- the batch form: four wavefronts per history, pack + open counts in one pass (BatchGeo / Batch64Geo)
@pytest.mark.parametrize("seed", [1, 2])
@pytest.mark.parametrize("branch", [False, True])
def test_batch_form_every_word_of_pack_and_open_counts(seed, branch):
    """all shapes, with the full lists and with branch lists (live reads left out, the reads' completions subtracted), crashed calls in
    the mask form (ncr[], the crashed-call list) -- every word pack_kernel and open_counts_kernel would leave"""
    assert emu.pack_wg_check(_hists(), branch=branch, seed=seed) is None
    assert emu.pack_wg_check(_hists(seeds=(3,)), branch=branch, look=False, rk8=False, vpad=32, seed=seed + 10) is None


def test_batch_form_count_form_and_crash_heavy():
    h = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info))
         for (n, p, busy, info) in [(600, 16, 0.5, 0.1), (900, 64, 0.2, 0.3), (130, 4, 1.0, 0.5), (64, 8, 0.5, 1.0)] for s in (1, 2)]
    assert emu.pack_wg_check(h, count=True, seed=3) is None                 # count form: no ncr[], no crashed list from this kernel
    assert emu.pack_wg_check(h, count=True, branch=True, seed=4) is None
    small = [x for x in h if x.n_process <= 256]
    assert len(small) >= 4
    assert emu.pack_wg_check(small, count=False, seed=5) is None            # the same in the mask form: a slot per crashed call
    assert emu.pack_wg_check(small, count=False, branch=True, per_launch=3, seed=6) is None


def test_batch_form_full_size_histories():
    """BASELINE.json configs[1] (10k invocations / 64 processes: 7,3xx completions of the 8,192 the LDS histogram holds) at 10 % and
    50 % duty, with crashed calls in both forms"""
    h = synth.register_ops_many(range(7000, 7002), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
    busy = synth.register_ops_many(range(7100, 7101), n_ops=10000, n_procs=64, busy=0.5, info=0.0)
    crashed = synth.register_ops_many(range(7200, 7201), n_ops=10000, n_procs=64, busy=0.1, info=0.01)
    assert emu.pack_wg_check(h + busy, branch=True, seed=11) is None
    assert emu.pack_wg_check(h[:1] + crashed, branch=False, seed=12) is None
    assert emu.pack_wg_check(crashed, count=True, branch=True, seed=13) is None


def test_batch_form_for_at_most_64_slots():
    """Batch64Geo (what a batch of the narrow kernel's histories is packed by: a byte per histogram entry, 19 KB of LDS): all shapes with at
    most 64 slots in both list forms, the bench histories, all 64 slots invoked between two completions (an entry of 64), crashed calls
    in both forms, refused histories and a list arena too small"""
    hs = [h for h in _hists() if h.n_process <= 64]
    assert len(hs) >= 8
    for branch in (False, True):
        assert emu.pack_wg_check(hs, branch=branch, slots64=True, seed=1 + branch) is None
    h = synth.register_ops_many(range(7000, 7002), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
    busy = synth.register_ops_many(range(7100, 7101), n_ops=10000, n_procs=64, busy=1.0, info=0.0)      # every process always in a call
    crashed = synth.register_ops_many(range(7200, 7201), n_ops=10000, n_procs=56, busy=0.1, info=0.001)
    assert all(x.n_process <= 64 for x in h + busy + crashed)
    assert emu.pack_wg_check(h + busy, branch=True, slots64=True, seed=3) is None
    assert emu.pack_wg_check(crashed + h[:1], branch=False, slots64=True, seed=4) is None
    assert emu.pack_wg_check(crashed, count=True, branch=True, slots64=True, seed=5) is None
    small = _hists(seeds=(4,), shapes=[(400, 12, 0.6, 0.02), (300, 16, 0.3, 0.0)])
    assert emu.pack_wg_check(small, lst_cap=50, slots64=True, seed=6) is None
    good = small[0]
    e = _d(good); e["f"][33] = 77
    assert emu.pack_wg_check([good, e, good], branch=True, slots64=True, seed=7) is None
    wide = [x for x in _hists() if x.n_process > 64]
    if wide:
        assert emu.pack_wg_check(wide[:1], slots64=True) == ("does not fit", 0, 0, 0, 0) or emu.pack_wg_check(wide[:1], slots64=True)[0] == "does not fit"


def test_batch_form_lists_that_do_not_fit_and_histories_pack_refuses():
    """a list arena too small: BeamHist.status 1, lst_need says how much; a refused history: no lists, as open_counts_kernel leaves it"""
    hs = _hists(seeds=(4,), shapes=[(400, 12, 0.6, 0.02), (300, 16, 0.3, 0.0)])
    assert emu.pack_wg_check(hs, lst_cap=50, seed=1) is None
    good = hs[0]
    d = _d(good); d["inv_pos"][100], d["inv_pos"][101] = d["inv_pos"][101], d["inv_pos"][100]
    e = _d(good); e["f"][33] = 77
    g = _d(good)
    p0 = np.flatnonzero(g["process"] == g["process"][0])
    g["ret_pos"][p0[0]] = g["inv_pos"][p0[1]] + 1              # two calls of one process open at once: found by the last phase
    assert emu.pack_wg_check([good, d, e, g, good], branch=True, seed=2) is None
    one = columns.pair_events(synth.register_events(n_ops=1, n_procs=1, seed=1, busy=0.5, info=1.0))      # no completion at all
    assert emu.pack_wg_check([one, good], seed=3) is None


def test_batch_form_crashed_calls_invoked_after_the_last_completion():
    """crashed calls invoked when nothing completes any more: in the crashed-call list, in no front's count (their rank is R)"""
    good = _hists(seeds=(6,), shapes=[(300, 8, 0.5, 0.05)])[0]
    d = _d(good)
    last = int(max(d["inv_pos"].max(), d["ret_pos"][d["ret_pos"] != 0xFFFFFFFF].max()))
    k = 4
    d["f"] = np.concatenate([d["f"], np.full(k, N.F_WRITE, np.uint8)])
    d["a"] = np.concatenate([d["a"], np.arange(k, dtype=np.int32)])
    d["b"] = np.concatenate([d["b"], np.zeros(k, np.int32)])
    d["process"] = np.concatenate([d["process"], np.arange(k, dtype=np.int32) + int(d["n_process"])])
    d["inv_pos"] = np.concatenate([d["inv_pos"], (last + 1 + np.arange(k)).astype(np.uint32)])
    d["ret_pos"] = np.concatenate([d["ret_pos"], np.full(k, 0xFFFFFFFF, np.uint32)])
    d["n_process"] = int(d["n_process"]) + k
    for branch in (False, True):
        assert emu.pack_wg_check([d, good], branch=branch, seed=1) is None


# ---- the one-history form with open counts (OneCountsGeo: sixteen wavefronts)
def test_one_history_form_with_open_counts():
    """every word of pack and open counts by sixteen wavefronts: all shapes, both list forms, both crashed-call forms, many slots,
    a full-size history"""
    assert emu.pack_wg_check(_hists(), branch=True, one=True, seed=1) is None
    assert emu.pack_wg_check(_hists(seeds=(2,)), branch=False, one=True, seed=2) is None
    crash = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=1, busy=busy, info=info))
             for (n, p, busy, info) in [(600, 16, 0.5, 0.1), (900, 64, 0.2, 0.3), (3000, 64, 0.5, 0.2)]]
    assert max(x.n_process for x in crash) > 300                                # (mask form: hundreds of slots)
    assert emu.pack_wg_check(crash, one=True, seed=3) is None
    assert emu.pack_wg_check(crash, count=True, branch=True, one=True, seed=4) is None
    full = synth.register_ops_many(range(7000, 7001), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
    assert emu.pack_wg_check(full, branch=True, one=True, seed=5) is None
