"""The front walk with lane = front (csrc/open_walk_impl.h -- the file hipcc compiles into libtbcheck.so) on the CPU, under the
lane-accurate wavefront emulator of tests/emu, against tables built on the host from the definitions (tests/emu/host_tables.h):
every list entry, twin mask, open-read row / front-record mask word and lookahead record, word for word.  Test infrastructure
only: the product has no CPU path."""
import numpy as np
import pytest

import emu
from jepsen_tigerbeetle_amd import columns, synth

SHAPES = [  # n_ops, n_procs, busy, info, corrupt
    (300, 16, 0.3, 0.0, 0.0), (500, 64, 0.1, 0.0, 0.0), (400, 64, 0.9, 0.0, 0.0), (260, 8, 1.0, 0.02, 0.0),
    (350, 24, 0.5, 0.05, 0.2), (64, 3, 0.5, 0.0, 0.0), (65, 64, 1.0, 0.0, 0.0), (1, 1, 0.5, 0.0, 0.0), (700, 33, 0.2, 0.01, 0.0)]


def _hists(seeds=(1, 2), shapes=SHAPES, **kw):
    out = [columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=info, corrupt=corrupt, **kw))
           for (n, p, busy, info, corrupt) in shapes for s in seeds]
    assert all(h.n_process <= 64 for h in out), [h.n_process for h in out]      # (a crashed call keeps its slot: new processes take new ones)
    return out


@pytest.mark.parametrize("front,branch", [("plain", False), ("wide", False), ("wide", True), ("compact", True), ("compact", False)])
def test_every_word_of_the_walk(front, branch):
    """all shapes in one batch (chunks of different histories, full and ragged last chunks, crashed calls, 1 .. 64 slots)"""
    assert emu.walk_check(_hists(), 8, twin=True, look=True, branch=branch, front=front) is None


def test_without_the_optional_tables():
    h = _hists(seeds=(3,))
    assert emu.walk_check(h, 8, twin=False, look=True, front="plain") is None
    assert emu.walk_check(h, 8, twin=True, look=False, front="plain") is None
    assert emu.walk_check(h, 0, twin=False, look=False, front="plain") is None          # no rows at all (rules off)
    assert emu.walk_check(h, 0, twin=False, look=True, branch=False, front="wide") is None   # front records without masks


def test_wider_rows():
    """values up to 30: rows of 16 / 32 entries (the 32-entry instantiation)"""
    for vmax, vpad in ((9, 16), (30, 32)):
        h = [columns.pair_events(synth.register_events(n_ops=300, n_procs=24, seed=s, busy=0.6, n_values=vmax + 1)) for s in range(3)]
        assert max(int(np.asarray(x.as_dict()["a"]).max()) for x in h) == vmax
        assert emu.walk_check(h, vpad, front="plain") is None
        assert emu.walk_check(h, vpad, branch=True, front="wide") is None


def test_many_twins():
    """two values only and every process busy: most open writes have a twin, often several"""
    h = [columns.pair_events(synth.register_events(n_ops=400, n_procs=40, seed=s, busy=1.0, n_values=2)) for s in range(3)]
    assert emu.walk_check(h, 8, front="plain") is None
    assert emu.walk_check(h, 8, branch=True, front="compact") is None


def test_full_size_histories():
    """BASELINE.json configs[1]: 10k invocations / 64 processes (the bench's histories, 115 chunks each), every table format"""
    h = synth.register_ops_many(range(7000, 7006), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
    busy = synth.register_ops_many(range(7100, 7102), n_ops=10000, n_procs=64, busy=0.5, info=0.0)      # ~32 calls in flight
    assert emu.walk_check(h, 8, branch=True, front="compact") is None
    assert emu.walk_check(h + busy, 8, branch=False, front="plain") is None
    assert emu.walk_check(busy, 8, branch=True, front="wide") is None


# ---- the fronts' lists in order of completion (tbc_opts.list_order; PackOpenArgs.list_order 1, 2, 16 + W -- 16 + 24 is the library's default)
@pytest.mark.parametrize("by_ret", [1, 2, 16 + 24, 16 + 3])
def test_lists_in_order_of_completion_every_word(by_ret):
    """the same tables with every front's list sorted by completion rank; 2: the :write calls after everything else (full lists -- reads,
    then cas, then writes apart -- and branch lists); 16 + W: a :write as if it completed W ranks later"""
    assert emu.walk_check(_hists(), 8, branch=True, front="compact", by_ret=by_ret) is None
    assert emu.walk_check(_hists(seeds=(3,)), 8, branch=False, front="compact", by_ret=by_ret) is None
    h = [columns.pair_events(synth.register_events(n_ops=400, n_procs=40, seed=s, busy=1.0, n_values=2)) for s in range(2)]
    full = synth.register_ops_many(range(7000, 7002), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
    busy = synth.register_ops_many(range(7100, 7101), n_ops=10000, n_procs=64, busy=0.5, info=0.0)
    assert emu.walk_check(h + full + busy, 8, branch=True, front="compact", by_ret=by_ret) is None
    assert emu.walk_check(full[:1] + busy, 8, branch=False, front="plain", by_ret=by_ret) is None          # (the wide kernel's tables: plain rows, full lists)
    # more than 128 candidates in a chunk (58 - 62 slots all busy + the 71 completions of the window), crashed calls among them: the
    # three register sets of the keys, and crashed calls' keys above every live one
    crowd = [columns.pair_events(synth.register_events(n_ops=400, n_procs=58, seed=s, busy=1.0, info=0.01)) for s in range(3)]
    assert all(58 <= h.n_process <= 64 for h in crowd), [h.n_process for h in crowd]
    assert emu.walk_check(crowd, 8, branch=True, front="compact", by_ret=by_ret) is None
    assert emu.walk_check(crowd, 8, branch=False, front="plain", by_ret=by_ret) is None
