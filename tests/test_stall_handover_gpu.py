"""A big quiet batch (several histories per wavefront) with NOT LINEARIZABLE histories in it: a pass is as long as its slowest history, and
exhausting the configs in front of a bad read in the middle of a history costs many times a valid history's search.  When asked (tbc_opts.dominance,
TBC_DOM_STALL_HANDOVER -- off by default: valid histories stall too and an all-valid batch pays for them, tbcheck.h) the narrow
kernel stops a history that has not passed a completion for 4,096 rounds (BeamArgs.stall_checks) and the library checks it again with the
level sweep in a small batch of its own (tbc_api.hip, hand_over_stalled).
Verdict and failing op are the search's either way; who answered is in tbc_result.analyzer."""
import numpy as np
import pytest

from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

pytestmark = pytest.mark.gpu

CAS = {"kind": 1, "init": N.NIL}


def _in_domain(n, p, s, busy, corrupt):
    h = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=s, busy=busy, info=0.0, corrupt=corrupt, n_values=4 if corrupt else 5))
    h.a[h.a == 4 + 7] = 4
    return h


def test_stalled_histories_are_handed_to_the_level_sweep(native, oracle):
    base = [_in_domain(10000, 64, 31000 + s, 0.1, (0.3 + 0.05 * (s % 7)) if s % 6 == 0 else 0.0) for s in range(48)]
    bad = [i for i in range(48) if i % 6 == 0]
    hists = [base[i % 48] for i in range(2048)]
    model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    ref = [oracle.check_sweep(h.as_dict(), CAS) for h in base]
    assert all(ref[i]["valid"] == (0 if i in bad else 1) for i in range(48))
    runs = {}
    for handover in (True, False):
        opts = core.make_opts(time_limit_ms=120000, want_witness=False, algorithm=N.ALG_COMPETITION, lanes_per_history=8, visited_per_op=4,
                              list_order=N.ORDER_DEFAULT, stall_handover=handover)
        with core.Batch(hists, model, opts) as b:
            assert b.lanes_per_history() == 8
            res = b.run().results()
            tm = b.timing_ns()
            again = b.run().results()
        runs[handover] = (res, tm)
        for k, g in enumerate(res):
            e = ref[k % 48]
            assert g["valid"] == e["valid"], (handover, k)
            if e["valid"] == 0:
                assert g["fail_op"] == e["fail_op"], (handover, k)
            assert (again[k]["valid"], again[k]["fail_op"]) == (g["valid"], g["fail_op"]), (handover, k)
    on, off = runs[True][0], runs[False][0]
    # without the handover: every history by the narrow schedule, counter by counter its oracle's (the invalid ones' exhaustion included)
    for k in list(range(48)) + [48 * 17 + 4, 2047]:
        e = oracle.check_beam(base[k % 48].as_dict(), CAS, 1, round_pairs=8, rules_at_any_round_size=True, branch_lists=True, list_order=16 + 24, want_witness=False)
        assert (off[k]["probes"], off[k]["visited"], off[k]["backtracks"]) == (e["probes"], e["visited"], e["expanded"]), k
        assert off[k]["analyzer"] == N.ALG_WGL
    # with it: the valid ones untouched, invalid ones that stalled answered by the sweep -- and the stalled ones did stall: fewer probes spent on them
    handed = [k for k in range(2048) if on[k]["analyzer"] == N.ALG_LINEAR]
    assert handed and all((k % 48) in bad for k in handed)
    for k in range(2048):
        if k not in handed and (k % 48) not in bad:
            assert (on[k]["probes"], on[k]["visited"]) == (off[k]["probes"], off[k]["visited"]), k
    print("handed over", len(handed), "of", sum(1 for k in range(2048) if (k % 48) in bad), "invalid; search ms with / without:",
          runs[True][1]["search"] / 1e6, runs[False][1]["search"] / 1e6)
