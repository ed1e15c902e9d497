"""The count form on the MI355X (wgl_beam.hip, CNT instantiation) against oracle/wgl_count.c: verdict, failing op, witness and
every counter of every pass, through the C-ABI (tbc_check / tbc_batch_*)."""
import numpy as np
import pytest

from helpers import op_tuples
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import brute
from test_count_form import CAS, completion_rank

pytestmark = pytest.mark.gpu


def gm():
    return core.make_model(N.MODEL_CAS_REGISTER, N.NIL)


def oracle_pipeline(oracle, ops, width, want_witness=False):
    """The library's passes over the oracle, counters summed as the library sums them (tbc_api.hip, batch_run_impl)."""
    n = len(ops["f"])
    tot = {"probes": 0, "visited": 0, "expanded": 0}

    def add(r):
        for k in tot:
            tot[k] += r[k]
        return r
    g = add(oracle.check_count(ops, CAS, width=width, want_witness=want_witness, max_probes=32 * n))
    if g["valid"] != -1:
        return g["valid"], g["fail_op"], g, tot, "exact"
    r = add(oracle.check_count(ops, CAS, width=width, want_witness=False, relaxed=True))
    if r["valid"] == 1:
        g = add(oracle.check_count(ops, CAS, width=width, want_witness=want_witness))
        return g["valid"], g["fail_op"], g, tot, "exact, no budget"
    t = completion_rank(ops, r["fail_op"])
    if t == 0:
        return 0, r["fail_op"], r, tot, "relaxed"
    g = add(oracle.check_count(ops, CAS, width=width, want_witness=False, target=t))
    if g["valid"] == 1:
        return 0, r["fail_op"], g, tot, "prefix"
    return g["valid"], g["fail_op"], g, tot, "prefix exhausted"


SHAPES = [(300, 8, 0.3, 0.05, 0.0), (300, 8, 0.3, 0.05, 0.5), (1000, 16, 0.3, 0.03, 0.0), (1000, 16, 0.3, 0.03, 0.6),
          (2000, 64, 0.1, 0.02, 0.0), (2000, 64, 0.1, 0.02, 0.5), (600, 4, 0.8, 0.2, 0.0), (600, 4, 0.8, 0.2, 0.7),
          (1500, 24, 0.25, 0.01, 0.3), (400, 70, 0.5, 0.05, 0.0)]


@pytest.mark.parametrize("width", [2, 4, 16])
def test_count_form_kernel_equals_its_oracle(native, oracle, width):
    hists, exp = [], []
    for n, p, busy, info, corrupt in SHAPES:
        for s in range(3):
            h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=1000 + s, busy=busy, info=info, corrupt=corrupt))
            e = oracle.check_count(h.as_dict(), CAS, width=width)
            if e is None:
                continue
            hists.append(h); exp.append(e)
    assert len(hists) > 20
    # one history at a time and all of them as one batch: the same answers
    with core.Batch(hists, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=width, count_form=True, max_steps=10 ** 9)) as b:
        res = b.run().results()
    n_inv = n_cls = 0
    for i, (h, e, g) in enumerate(zip(hists, exp, res)):
        assert g["valid"] == e["valid"], i
        assert (g["probes"], g["visited"], g["backtracks"], g["max_depth"]) == (e["probes"], e["visited"], e["expanded"], e["max_stack"]), (i, SHAPES[i // 3])
        if e["valid"] == 1:
            assert np.array_equal(g["witness"], e["witness"]), i
            assert g["final_state"] == e["final_state"]
            assert brute.check_witness(CAS, op_tuples(h), [int(x) for x in g["witness"]]) == g["final_state"]
        else:
            assert g["fail_op"] == e["fail_op"], i
            n_inv += 1
        n_cls += e["class_steps"]
    assert n_inv >= 5 and n_cls > 1000
    one = core.check_ops(hists[5], gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=width, count_form=True, max_steps=10 ** 9))
    assert (one["valid"], one["probes"], one["visited"]) == (exp[5]["valid"], exp[5]["probes"], exp[5]["visited"])


def test_count_form_agrees_with_the_mask_form(native, oracle):
    """Two formulations of one search on the device: the same verdicts and failing ops (the counters are each form's own)."""
    hists = [columns.pair_events(synth.register_events(n_ops=800, n_procs=12, seed=s, busy=0.3, info=0.04, corrupt=0.5 * (s % 2))) for s in range(12)]
    res = {}
    for cf in (True, False):
        with core.Batch(hists, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, search_width=4, count_form=cf, want_witness=False)) as b:
            res[cf] = [(r["valid"], r["fail_op"]) for r in b.run().results()]
    assert res[True] == res[False]
    assert any(v == 0 for v, _ in res[True])


@pytest.mark.parametrize("info,corrupt", [(0.01, 0.0), (0.01, 0.5), (0.05, 0.0), (0.05, 0.5)])
def test_the_crashed_tiers_at_full_size(native, oracle, info, corrupt):
    """BASELINE.md section 3's crashed tiers, 10k ops / 64 processes: the library's default path (knossos.competition, no
    witness) through tbc_check -- the budgeted exact search, then for the planted bad read the relaxed refutation and the
    prefix search -- pass by pass against the oracle."""
    h = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=4242, busy=0.1, info=info, corrupt=corrupt))
    g = core.check_ops(h, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, count_form=True))
    width = g["search_width"]
    v, fo, last, tot, how = oracle_pipeline(oracle, h.as_dict(), width)
    assert how == ("prefix" if corrupt else "exact")
    assert g["valid"] == v == (0 if corrupt else 1)
    if corrupt:
        assert g["fail_op"] == fo and h.a[fo] == 12
    assert (g["probes"], g["visited"], g["backtracks"]) == (tot["probes"], tot["visited"], tot["expanded"])
    # with a witness: the linearization replays legally
    if not corrupt:
        w = core.check_ops(h, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=True, count_form=True))
        assert brute.check_witness(CAS, op_tuples(h), [int(x) for x in w["witness"]]) == w["final_state"]
