"""The count form on the MI355X (wgl_beam.hip, CNT instantiation) against oracle/wgl_count.c: verdict, failing op, witness and
every counter of every pass, through the C-ABI (tbc_check / tbc_batch_*)."""
import numpy as np
import pytest

from helpers import op_tuples
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import brute
from test_count_form import CAS

pytestmark = pytest.mark.gpu


def gm():
    return core.make_model(N.MODEL_CAS_REGISTER, N.NIL)


def oracle_pipeline(oracle, ops, width, want_witness=False, budget=None):
    return oracle.check_count_pipeline(ops, CAS, width=width, budget=budget, want_witness=want_witness)


SHAPES = [(300, 8, 0.3, 0.05, 0.0), (300, 8, 0.3, 0.05, 0.5), (1000, 16, 0.3, 0.03, 0.0), (1000, 16, 0.3, 0.03, 0.6),
          (2000, 64, 0.1, 0.02, 0.0), (2000, 64, 0.1, 0.02, 0.5), (600, 4, 0.8, 0.2, 0.0), (600, 4, 0.8, 0.2, 0.7),
          (1500, 24, 0.25, 0.01, 0.3), (400, 70, 0.5, 0.05, 0.0)]


@pytest.mark.parametrize("width", [2, 4, 16])
def test_count_form_kernel_equals_its_oracle(native, oracle, width):
    hists = []
    for n, p, busy, info, corrupt in SHAPES:
        for s in range(3):
            h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=1000 + s, busy=busy, info=info, corrupt=corrupt))
            if oracle.check_count(h.as_dict(), CAS, width=width, max_probes=1) is not None:      # (None: more than 128 bits of counts)
                hists.append(h)
    assert len(hists) > 20
    budget = 32 * max(len(h) for h in hists)
    exp = [oracle_pipeline(oracle, h.as_dict(), width, want_witness=True, budget=budget) for h in hists]
    # all of them as one batch ...
    with core.Batch(hists, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=width, count_form=True)) as b:
        res = b.run().results()
    how = {}
    for i, (h, (v, fo, last, tot, hw), g) in enumerate(zip(hists, exp, res)):
        how[hw] = how.get(hw, 0) + 1
        assert g["valid"] == v, (i, hw)
        assert (g["probes"], g["visited"], g["backtracks"]) == (tot["probes"], tot["visited"], tot["expanded"]), (i, hw)
        if v == 1:
            assert np.array_equal(g["witness"], last["witness"]), i
            assert g["final_state"] == last["final_state"]
            assert brute.check_witness(CAS, op_tuples(h), [int(x) for x in g["witness"]]) == g["final_state"]
        else:
            assert g["fail_op"] == fo, (i, hw)
    assert how.get("exact", 0) >= 10 and how.get("prefix", 0) >= 3, how
    # ... and one at a time through tbc_check (its budget is its own length's)
    for i in (0, 4, 10, len(hists) - 1):
        one = core.check_ops(hists[i], gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=width, count_form=True, want_witness=False))
        v, fo, last, tot, hw = oracle_pipeline(oracle, hists[i].as_dict(), width)
        assert (one["valid"], one["probes"], one["visited"]) == (v, tot["probes"], tot["visited"]), (i, hw)


def test_count_form_agrees_with_the_mask_form(native, oracle):
    """Two formulations of one search on the device: the same verdicts and failing ops (the counters are each form's own)."""
    hists = [columns.pair_events(synth.register_events(n_ops=300, n_procs=12, seed=s, busy=0.3, info=0.03, corrupt=0.5 * (s % 2))) for s in range(12)]
    res = {}
    for cf in (True, False):
        with core.Batch(hists, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=4, count_form=cf, want_witness=False)) as b:
            res[cf] = [(r["valid"], r["fail_op"]) for r in b.run().results()]
    assert res[True] == res[False]
    assert any(v == 0 for v, _ in res[True]) and all(v != -1 for v, _ in res[True])


@pytest.mark.parametrize("info,corrupt", [(0.01, 0.0), (0.01, 0.5), (0.05, 0.0), (0.05, 0.5)])
def test_the_crashed_tiers_at_full_size(native, oracle, info, corrupt):
    """BASELINE.md section 3's crashed tiers, 10k ops / 64 processes: the library's default path (knossos.competition, no
    witness) through tbc_check -- the relaxed level sweep, then the budgeted exact search or, for the planted bad read the sweep
    refutes, the prefix search -- pass by pass against the oracle."""
    h = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=4242, busy=0.1, info=info, corrupt=corrupt))
    g = core.check_ops(h, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, count_form=True))
    width = g["search_width"]
    v, fo, last, tot, how = oracle.check_count_pipeline(h.as_dict(), CAS, width=width, relaxed_sweep=True)
    assert how == ("relaxed sweep, prefix" if corrupt else "exact")
    assert g["valid"] == v == (0 if corrupt else 1)
    if corrupt:
        assert g["fail_op"] == fo and h.a[fo] == 12
    assert (g["probes"], g["visited"], g["backtracks"]) == (tot["probes"], tot["visited"], tot["expanded"])
    # with a witness: the linearization replays legally
    if not corrupt:
        w = core.check_ops(h, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=True, count_form=True))
        assert brute.check_witness(CAS, op_tuples(h), [int(x) for x in w["witness"]]) == w["final_state"]


@pytest.mark.parametrize("L", [8, 16])
def test_count_form_several_histories_per_wavefront(native, oracle, L):
    """The count form in the narrow kernel (lanes_per_history = L): one config per iteration, L pairs per round -- every pass of the
    library's pipeline against oracle/wgl_count.c at that schedule, the crashed tiers at full size among the histories."""
    hists = []
    for n, p, busy, info, corrupt in SHAPES[:6] + [(10000, 64, 0.1, 0.01, 0.0), (10000, 64, 0.1, 0.05, 0.0), (10000, 64, 0.1, 0.01, 0.5)]:
        for s in range(2):
            h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=2000 + s, busy=busy, info=info, corrupt=corrupt))
            if oracle.check_count(h.as_dict(), CAS, width=1, max_probes=1) is not None:
                hists.append(h)
    budget = 32 * max(len(h) for h in hists)
    with core.Batch(hists, gm(), core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, lanes_per_history=L, count_form=True, want_witness=False)) as b:
        assert (b.lanes_per_history(), b.search_width()) == (L, 1)
        res = b.run().results()
    how = {}
    for i, (h, g) in enumerate(zip(hists, res)):
        v, fo, last, tot, hw = oracle.check_count_pipeline(h.as_dict(), CAS, width=1, budget=budget, round_pairs=L)
        how[hw] = how.get(hw, 0) + 1
        assert g["valid"] == v, (i, hw)
        assert (g["probes"], g["visited"], g["backtracks"]) == (tot["probes"], tot["visited"], tot["expanded"]), (i, hw)
        if v == 0:
            assert g["fail_op"] == fo, (i, hw)
    assert how.get("exact", 0) >= 8 and how.get("prefix", 0) >= 3, how
    # with a witness: the linearization replays legally (crashed calls included)
    with core.Batch(hists[:6], gm(), core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, lanes_per_history=L, count_form=True)) as b:
        for h, g in zip(hists[:6], b.run().results()):
            if g["valid"] == 1:
                assert brute.check_witness(CAS, op_tuples(h), [int(x) for x in g["witness"]]) == g["final_state"]


def test_crash_heavy_edn_goldens_in_the_count_form(native, monkeypatch):
    """tests/golden/edn/synth_*_i0.0[368]*.edn (14 - 29 crashed calls each; what scripts/knossos_crosscheck.clj feeds to stock Knossos) through
    the EDN reader and jepsen.checker/linearizable with the library's defaults -- the count form -- against the committed expectations."""
    import json
    import os
    from helpers import GOLDEN, MODELS
    from jepsen_tigerbeetle_amd.jepsen import checker as jc, edn
    monkeypatch.setattr(core, "DEFAULT_COUNT_FORM", True)
    cases = [c for c in json.load(open(os.path.join(GOLDEN, "edn", "expected.json")))["cases"] if "_i0.0" in c["file"] and "_i0.0_" not in c["file"]]
    assert len(cases) >= 8
    n_invalid = 0
    for c in cases:
        h = edn.read_history(os.path.join(GOLDEN, "edn", c["file"]))
        a = jc.linearizable({"model": MODELS[c["model"]](), "algorithm": None}).check(None, h, None)
        assert a["valid?"] is c["valid?"], c["file"]
        if c["valid?"] is False:
            assert a["op"]["index"] == c["op-index"], c["file"]
            n_invalid += 1
    assert n_invalid >= 3


def test_relaxed_sweep_in_front_of_the_count_form_search(native, oracle):
    """tbc_check / a handful of histories with crashed calls, knossos.competition, no witness, no schedule named: the RELAXED LEVEL SWEEP
    runs first (tbc_batch::rsweep).  A history it refutes goes to the prefix search alone, one it finds valid under the relaxation
    never takes the relaxed depth-first search: verdict, failing op and the depth-first passes' counters against the oracle's statement
    of the same pipeline -- crash-heavy small histories one by one and eight to a batch, valid and with a planted bad read."""
    hists = []
    for n, p, busy, info, corrupt in [(300, 8, 0.3, 0.05, 0.0), (300, 8, 0.3, 0.05, 0.5), (1000, 16, 0.3, 0.03, 0.0), (1000, 16, 0.3, 0.03, 0.6), (2000, 64, 0.1, 0.02, 0.5),
                                      (600, 4, 0.8, 0.2, 0.0), (600, 4, 0.8, 0.2, 0.7), (1500, 24, 0.25, 0.01, 0.3), (200, 12, 0.9, 0.3, 0.4)]:
        for s in range(3):
            h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=5000 + s, busy=busy, info=info, corrupt=corrupt, n_values=2 if p == 12 else 5))
            if h.n_process and oracle.check_count(h.as_dict(), CAS, width=4, max_probes=1) is not None:
                hists.append(h)
    assert len(hists) >= 24
    opts = core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, count_form=True)
    n_refuted = 0
    for i, h in enumerate(hists):
        g = core.check_ops(h, gm(), opts)
        exp = oracle.check_count_pipeline(h.as_dict(), CAS, width=g["search_width"], relaxed_sweep=True)
        if exp is None:
            continue
        v, fo, last, tot, how = exp
        assert g["valid"] == v, (i, how)
        if v == 0:
            assert g["fail_op"] == fo, (i, how)
        assert (g["probes"], g["visited"], g["backtracks"]) == (tot["probes"], tot["visited"], tot["expanded"]), (i, how)
        n_refuted += how.startswith("relaxed sweep")
        ref = oracle.check(h.as_dict(), CAS, "window", max_steps=5_000_000, want_witness=False)      # ... and the sequential restatement, whatever the passes
        if ref["valid"] != -1:
            assert (g["valid"], g["fail_op"] if v == 0 else None) == (ref["valid"], ref["fail_op"] if ref["valid"] == 0 else None), i
    assert n_refuted >= 6
    for lo in range(0, len(hists) - 7, 8):
        group = hists[lo:lo + 8]
        with core.Batch(group, gm(), opts) as b:
            res = b.run().results()
            again = b.run().results()
        for h, g, g2 in zip(group, res, again):
            exp = oracle.check_count_pipeline(h.as_dict(), CAS, width=g["search_width"], budget=32 * max(len(x) for x in group), relaxed_sweep=True)
            if exp is None:
                continue
            assert (g["valid"], g["fail_op"] if exp[0] == 0 else None) == (exp[0], exp[1] if exp[0] == 0 else None), lo
            assert (g["probes"], g["visited"]) == (exp[3]["probes"], exp[3]["visited"]), (lo, exp[4])
            assert (g2["valid"], g2["probes"]) == (g["valid"], g["probes"])
