"""BASELINE.json configs 3, 4 and 5 as parity cases, at their stated sizes and shapes.  Difficulty
(how many calls are open at once, how many crash) is kept where the Wing-Gong/Lowe search
terminates at all -- the config space is exponential in both, on any machine.

  config 3  set-full add/read history, 50k ops over 5 keys (set_full.clj:151: keys 1..#nodes),
            checked as knossos.model/set through independent/checker -- one batch launch
  config 4  100k-op multi-register (8 keys) history, 256 processes, ONE non-decomposable history
  config 5  batch of independent 5k-op bank-transfer histories (1024 in BASELINE.json; 64 here so the
            CPU oracle finishes in seconds; the batch path is the same)
"""
import numpy as np
import pytest

from helpers import bank_history, multi_register_history, op_tuples, set_history
from jepsen_tigerbeetle_amd import _native as N, core
from jepsen_tigerbeetle_amd.jepsen import checker as jc, independent
from jepsen_tigerbeetle_amd.knossos import _analysis, model as M
from oracle import brute

pytestmark = pytest.mark.gpu


def test_config3_set_full_50k_ops_5_keys(native, oracle):
    t = independent.tuple_
    subs = {k: set_history(10000, 5, 300 + k, busy=0.3, info=0.0005, corrupt="lost" if k == 4 else None)
            for k in range(1, 6)}
    hist = [dict(o, process=o["process"] * 8 + k, value=t(k, o["value"])) for k, h in subs.items() for o in h]
    hist.insert(100, {"type": "info", "f": "start-partition", "process": "nemesis", "value": ["isolated", {}]})
    assert sum(1 for o in hist if o["type"] == "invoke") == 50000
    c = independent.checker(jc.linearizable({"model": M.set(), "algorithm": "linear"}))
    r = c.check({}, hist, {})
    assert r["failures"] == [4] and r["valid?"] is False
    for k in (1, 2, 3, 5):
        assert r["results"][k]["valid?"] is True
    # the failing read is the corrupted one: its value misses the last element of a consistent read
    e = _analysis.Encoded(M.set(), subs[4])
    exp = oracle.check_beam(e.ops.as_dict(), {"kind": 5, "init": 0, "pool": e.ops.pool, "n_adds": e.n_adds}, 16, want_witness=False)
    assert exp["valid"] == 0
    assert r["results"][4]["op"]["value"] == e.op_completion(exp["fail_op"])["value"]


def test_config4_multi_register_100k_ops_256_procs(native, oracle):
    hist = multi_register_history(100000, 256, 7, n_keys=8, n_values=5, busy=0.012, info=0.0)
    e = _analysis.Encoded(M.multi_register({}), hist)
    assert e.native_model[0].kind == N.MODEL_MULTI_REGISTER and e.ops.n_process == 256 and len(e.ops) == 100000
    om = {"kind": 4, "init": 0, "pool": e.ops.pool}
    exp = oracle.check_beam(e.ops.as_dict(), om, 8)
    got = core.check_ops(e.ops, e.native_model, core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, search_width=8))
    assert got["valid"] == exp["valid"] == N.VALID
    assert np.array_equal(got["witness"], exp["witness"]) and got["probes"] == exp["probes"]
    assert brute.check_witness(om, op_tuples(e.ops), [int(x) for x in got["witness"]]) == got["final_state"]
    seq = core.check_ops(e.ops, e.native_model, core.make_opts(time_limit_ms=120000))      # sequential order, 4 mask words
    exps = oracle.check(e.ops.as_dict(), om, "window")
    assert seq["valid"] == N.VALID and np.array_equal(seq["witness"], exps["witness"]) and seq["steps"] == exps["steps"]


def test_config5_batch_of_5k_op_bank_histories(native, oracle):
    hists = [bank_history(5000, 8, 500 + i, busy=0.25, info=0.0, corrupt=(i % 16 == 5)) for i in range(64)]
    encs = [_analysis.Encoded(M.bank(), h) for h in hists]
    assert all(e.native_model[0].kind == N.MODEL_BANK for e in encs)
    with core.Batch([e.ops for e in encs], encs[0].native_model,
                    core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, search_width=8)) as b:
        res = b.run().results()
    for i, (e, got) in enumerate(zip(encs, res)):
        exp = oracle.check_beam(e.ops.as_dict(), {"kind": 6, "init": 0, "pool": e.ops.pool, "n_accounts": 8}, 8)
        assert got["valid"] == exp["valid"] == (0 if i % 16 == 5 else 1), i
        assert (got["probes"], got["visited"]) == (exp["probes"], exp["visited"]), i
        if exp["valid"] == 1:
            assert np.array_equal(got["witness"], exp["witness"]), i
        else:
            assert got["fail_op"] == exp["fail_op"], i
