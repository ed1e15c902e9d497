"""knossos.search's limits on the device (`knossos.search`: a time limit turns into {:valid? :unknown :cause ...}):
`tbc_opts.time_limit_ms` must FIRE -- TBC_UNKNOWN / TBC_CAUSE_TIME_LIMIT, in about the time named -- in every search kernel (K3 sequential,
K5 a wavefront per history, K5n several histories per wavefront, K5 in the count form) and behind the level sweep (K6 / K6w have no clock of
their own: a segment is a few hundred levels and always ends; what the sweep cannot finish goes to the depth-first search, whose clock it is).
Round 5's tests only tolerated the cause (tests/test_shipped_defaults_gpu.py); none made it happen.

And :configs of a count-form INVALID verdict (result map): the prefix search's end config, no longer empty."""
import time

import numpy as np
import pytest

from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def gm():
    return core.make_model(N.MODEL_CAS_REGISTER, N.NIL)


def hard(seed, info=0.0):
    """~27 calls in flight and a bad read half way in: without the dominance rules the exhaustion in front of it is hopeless"""
    return columns.pair_events(synth.register_events(n_ops=4000, n_procs=30, seed=seed, busy=0.9, info=info, corrupt=0.5))


PLAIN = dict(eager_reads=False, twin_rule=False, lookahead=False)


@pytest.mark.parametrize("name,kw", [
    ("K3 sequential", dict(algorithm=N.ALG_WGL)),
    ("K5 a wavefront per history", dict(algorithm=N.ALG_COMPETITION, search_width=4, lanes_per_history=64, **PLAIN)),
    ("K5n eight histories per wavefront", dict(algorithm=N.ALG_COMPETITION, lanes_per_history=8, **PLAIN)),
    ("K5 count form", dict(algorithm=N.ALG_COMPETITION, search_width=4, count_form=True, max_steps=1 << 40)),
    ("behind the level sweep", dict(algorithm=N.ALG_LINEAR, **PLAIN)),
])
def test_the_time_limit_fires(native, name, kw):
    info = 0.05 if "count form" in name else 0.0
    hists = [hard(31 + s, info) for s in range(4)]
    opts = core.make_opts(time_limit_ms=60, want_witness=False, **kw)
    t0 = time.perf_counter()
    with core.Batch(hists, gm(), opts) as b:
        res = b.run().results()
    dt = time.perf_counter() - t0
    hit = [r for r in res if r["valid"] == N.UNKNOWN]
    assert hit, (name, [(r["valid"], r["cause"]) for r in res])
    for r in hit:
        assert r["cause"] == N.CAUSE_TIME_LIMIT, (name, r["cause"])
        assert r["fail_op"] is None and r["configs"] == []
    for r in res:          # whoever did finish within 60 ms finished with a verdict, not with a limit of another kind
        assert r["valid"] in (N.VALID, N.INVALID) or r["cause"] == N.CAUSE_TIME_LIMIT, (name, r["valid"], r["cause"])
    # the limit is per pass of a kernel (retries with larger visited sets start their own clocks): seconds at most, never the minutes
    # the exhaustion would take
    assert dt < 20.0, (name, dt)
    # one history through tbc_check: the same answer
    r1 = core.check_ops(hists[0], gm(), opts)
    assert (r1["valid"], r1["cause"]) == (N.UNKNOWN, N.CAUSE_TIME_LIMIT) or r1["valid"] in (N.VALID, N.INVALID), name


def test_no_limit_named_means_no_limit(native, oracle):
    """time_limit_ms = 0: the same kernels finish a history that takes longer than any limit above would allow"""
    h = columns.pair_events(synth.register_events(n_ops=3000, n_procs=16, seed=5, busy=0.6, corrupt=0.5))
    ref = oracle.check(h.as_dict(), {"kind": 1, "init": N.NIL}, "window", want_witness=False, max_steps=50_000_000)
    assert ref["valid"] == 0
    for kw in (dict(algorithm=N.ALG_COMPETITION, search_width=4, lanes_per_history=64), dict(algorithm=N.ALG_COMPETITION, lanes_per_history=8)):
        g = core.check_ops(h, gm(), core.make_opts(time_limit_ms=0, want_witness=False, **kw))
        assert (g["valid"], g["fail_op"]) == (0, ref["fail_op"])


def test_configs_of_a_count_form_invalid_verdict(native, oracle):
    """The default path for histories with crashed calls (what the reference's nemesis makes: core.clj:106-125): relaxed refutation, then
    the exact search of the prefix.  Round 5 returned an empty :configs there; now the config the prefix's linearization ended in: the
    failing call is among its pending calls and not linearized, its state is a register value."""
    old = core.DEFAULT_COUNT_FORM
    core.DEFAULT_COUNT_FORM = True
    try:
        seen = 0
        for s in range(6):
            h = columns.pair_events(synth.register_events(n_ops=2000, n_procs=16, seed=700 + s, busy=0.3, info=0.03, corrupt=0.5))
            g = core.check_ops(h, gm(), core.make_opts(time_limit_ms=60000, algorithm=N.ALG_COMPETITION, want_witness=False, search_width=4))
            exp = oracle.check_count_pipeline(h.as_dict(), {"kind": 1, "init": N.NIL}, width=4)
            if exp is None or exp[0] != 0 or not exp[4].endswith("prefix"):
                continue
            assert g["valid"] == 0 and g["fail_op"] == exp[1]
            assert len(g["configs"]) == 1, (s, g["configs"])
            c = g["configs"][0]
            assert g["fail_op"] in c["pending"] or c["n_pending"] > 16
            if g["fail_op"] in c["pending"]:
                assert not (c["linearized_mask"] >> c["pending"].index(g["fail_op"])) & 1
            assert c["state"] == N.NIL or 0 <= c["state"] <= 12
            # every pending call was invoked before the failing completion and is not complete by then
            P = int(h.ret_pos[g["fail_op"]])
            for o in c["pending"]:
                assert int(h.inv_pos[o]) < P and (int(h.ret_pos[o]) == 0xFFFFFFFF or int(h.ret_pos[o]) >= P)
            seen += 1
        assert seen >= 2
    finally:
        core.DEFAULT_COUNT_FORM = old


def test_progress_is_readable_from_another_thread_while_the_run_is_out(native):
    """tbc_batch_progress (reference: knossos.search's reporter, which logs how far a running search has come): a second thread polls the
    batch while tbc_batch_run is in flight -- the count of decided histories only grows, the run is seen running, and afterwards the
    count is what the run handed back."""
    import threading
    import time
    hists = synth.register_ops_many(range(8192), n_ops=4000, n_procs=32, busy=0.2, info=0.0)
    gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
    with core.Batch(hists, gm, core.make_opts(time_limit_ms=120000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=8)) as b:
        p0 = b.progress()
        assert p0["n_histories"] == 8192 and p0["running"] == 0 and p0["n_decided"] == 0 and p0["phase"] == N.PHASE_IDLE
        seen, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                seen.append(b.progress())
                time.sleep(0.0005)

        t = threading.Thread(target=poll)
        t.start()
        try:
            for _ in range(3):
                b.run()
        finally:
            stop.set()
            t.join()
        v = b.verdicts()
        after = b.progress()
        assert after["running"] == 0 and after["phase"] == N.PHASE_IDLE and after["n_decided"] == int((v != N.UNKNOWN).sum()) == 8192
        running = [p for p in seen if p["running"]]
        assert len(running) >= 3 and all(p["elapsed_ns"] > 0 and p["phase"] in (N.PHASE_PACK, N.PHASE_RETRIES) for p in running)
        assert all(0 <= p["n_decided"] <= 8192 for p in seen)
        # inside one run the count only grows: split the samples at the points where a new run zeroed it
        partial = [p["n_decided"] for p in running if 0 < p["n_decided"] < 8192]
        assert partial, "no sample fell inside a search (the kernels count as they store their results)"
    # one history through the wide kernel and through the sequential one: counted by those kernels too
    one = synth.register_ops_many(range(1), n_ops=3000, n_procs=16, busy=0.3, info=0.0)
    for alg, width in ((N.ALG_COMPETITION, 8), (N.ALG_WGL, 0)):
        with core.Batch(one, gm, core.make_opts(time_limit_ms=60000, algorithm=alg, search_width=width)) as b:
            b.run()
            assert b.progress()["n_decided"] == 1
