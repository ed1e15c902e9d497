"""BASELINE.json configs 3, 4 and 5 as parity cases, at their stated sizes and shapes.

  config 3  set-full add/read history, 50k ops over 5 keys (set_full.clj:151: keys 1..#nodes), :info timeouts
            clustered in two partition windows per key (set_full.clj:107-110), checked as knossos.model/set
            through independent/checker -- one batch launch
  config 4  100k-op multi-register (8 keys) history, 256 processes, ONE non-decomposable history
  config 5  batch of 1,024 independent 5k-op bank-transfer histories: all 1,024 on the GPU; the CPU oracle
            compares a 64-history sample bit for bit and every 4th witness is replayed by the independent checker

What is NOT at BASELINE.json's face value is concurrency, and the reason is the problem, not the machine: the
config space of the Wing-Gong/Lowe search is exponential in the calls open at once and in crashed mutating
calls.  The register family has dominance rules (eager reads, twin rule, lookahead) that carry it to ~32 calls
in flight (tests/test_gpu_parity.py, bench.py workload_2).  The commutative models (set, bank) have the lazy rule (a
mutating call is linearized only when it completes at the front or an open read could take it): config 3's partition
windows now crash ~60 adds per key (the plain search exceeds 5*10^7 probes at 50 crashed adds in a 10k-op history; under
the rule such a history costs ~2*10^4).  multi-register has two rules of its own since round 6 (eager pure-read txns, txn
independence: tbcheck.h TBC_DOM_NO_EAGER_TXNS / TBC_DOM_NO_TXN_INDEPENDENCE), and the CPU oracle measures where they carry: at 256
processes the plain search costs 2.5*10^6 probes per 20k ops at 5.1 calls in flight and > 3*10^7 at 7.7; under the rules 4 - 7*10^6 at 7.7
and > 3*10^7 at 11.5 -- so config 4 runs at 7.7.
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
    from helpers import partition_windows
    part = partition_windows([(2000, 2400), (6000, 6400)], 0.18)       # two partitions: the clients' timeouts come in bursts
    subs = {k: set_history(10000, 5, 300 + k, busy=0.3, info=part, corrupt="lost" if k == 4 else None)
            for k in range(1, 6)}
    crashed = {k: [i for i, o in enumerate(h) if o["type"] == "info"] for k, h in subs.items()}
    crashed_adds = {k: sum(1 for i in v if subs[k][i]["f"] == "add") for k, v in crashed.items()}
    assert min(crashed_adds.values()) >= 50 and all(2 * 2000 - 400 < i < 2 * 6400 + 800 for v in crashed.values() for i in v)
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
    # ~7.7 calls in flight: under the two multi-register rules of the wide kernel (eager txns, txn independence: round 6) 2.2 * 10^7 probes;
    # the plain search passes 3 * 10^7 within the first 20k ops (round 5 ran this test at 4.1 in flight for that reason)
    hist = multi_register_history(100000, 256, 7, n_keys=8, n_values=5, busy=0.03, info=0.0)
    e = _analysis.Encoded(M.multi_register({}), hist)
    assert e.native_model[0].kind == N.MODEL_MULTI_REGISTER and e.ops.n_process == 256 and len(e.ops) == 100000
    om = {"kind": 4, "init": 0, "pool": e.ops.pool}
    exp = oracle.check_beam(e.ops.as_dict(), om, 8)
    got = core.check_ops(e.ops, e.native_model, core.make_opts(time_limit_ms=300000, algorithm=N.ALG_COMPETITION, search_width=8))
    assert got["valid"] == exp["valid"] == N.VALID
    assert np.array_equal(got["witness"], exp["witness"]) and (got["probes"], got["visited"]) == (exp["probes"], exp["visited"])
    assert brute.check_witness(om, op_tuples(e.ops), [int(x) for x in got["witness"]]) == got["final_state"]
    # the sequential order (4 mask words, no rules) at the concurrency it can still do
    hist = multi_register_history(100000, 256, 7, n_keys=8, n_values=5, busy=0.016, info=0.0)     # ~4.1 calls in flight
    e = _analysis.Encoded(M.multi_register({}), hist)
    om = {"kind": 4, "init": 0, "pool": e.ops.pool}
    seq = core.check_ops(e.ops, e.native_model, core.make_opts(time_limit_ms=120000))
    exps = oracle.check(e.ops.as_dict(), om, "window")
    assert seq["valid"] == N.VALID and np.array_equal(seq["witness"], exps["witness"]) and seq["steps"] == exps["steps"]


def _bank_case(i):
    h = bank_history(5000, 8, 500 + i, busy=0.5, info=0.0, corrupt=(i % 16 == 5))     # (8 client processes, each busy half of the time)
    e = _analysis.Encoded(M.bank(), h)
    return e.ops, e.native_model[0].kind


def test_config5_batch_of_1024_5k_op_bank_histories(native, oracle):
    from concurrent.futures import ProcessPoolExecutor
    import os
    n_hist = 1024
    with ProcessPoolExecutor(min(32, os.cpu_count() or 1)) as ex:          # generating + encoding is host-side Python
        cases = list(ex.map(_bank_case, range(n_hist), chunksize=8))
    ops = [c[0] for c in cases]
    assert all(c[1] == N.MODEL_BANK for c in cases) and sum(len(o) for o in ops) > 4_000_000
    with core.Batch(ops, core.make_model(N.MODEL_BANK, 0, n_keys=8),
                    core.make_opts(time_limit_ms=120000, algorithm=N.ALG_COMPETITION, search_width=8)) as b:
        res = b.run().results()
    for i, got in enumerate(res):
        assert got["valid"] == (0 if i % 16 == 5 else 1), i
    for i in range(0, n_hist, 16):                  # the CPU oracle on a 64-history sample: bit for bit
        j = i + 5 if (i // 16) % 2 else i           # half of the sample are the corrupted ones
        o, got = ops[j], res[j]
        exp = oracle.check_beam(o.as_dict(), {"kind": 6, "init": 0, "pool": o.pool, "n_accounts": 8}, 8)
        assert got["valid"] == exp["valid"], j
        assert (got["probes"], got["visited"]) == (exp["probes"], exp["visited"]), j
        if exp["valid"] == 1:
            assert np.array_equal(got["witness"], exp["witness"]), j
        else:
            assert got["fail_op"] == exp["fail_op"], j
    om = lambda o: {"kind": 6, "init": 0, "pool": o.pool, "n_accounts": 8}
    for i in range(0, n_hist, 4):                   # every 4th witness replayed by the independent checker
        if res[i]["valid"] == 1:
            brute.check_witness(om(ops[i]), op_tuples(ops[i]), [int(x) for x in res[i]["witness"]])
