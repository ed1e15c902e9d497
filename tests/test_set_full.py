"""jepsen.checker/set-full (the checker the reference runs:
/root/reference/src/tigerbeetle/workloads/set_full.clj:157).  CPU: the restatement on hand-derived
cases and the host encoder.  GPU: the scan kernel (csrc/set_full.hip) against the restatement's
known / last-present / last-absent per element on generated grow-only-set histories with lost, stale
and never-read elements, through the Checker surface and through independent/checker as the
reference composes it; and a large synthetic matrix against a numpy reduction."""
import numpy as np
import pytest

from helpers import set_history
from jepsen_tigerbeetle_amd import _native as N
from jepsen_tigerbeetle_amd.jepsen import checker as jc, independent, set_full as sf
from oracle import set_full as osf


def _h(rows):
    return [{"type": t, "f": f, "value": v, "process": p, "index": i} for i, (t, f, v, p) in enumerate(rows)]


def test_restatement_on_hand_derived_cases():
    # stable at once, stale (absent after the ack, present later), never read, lost
    h = _h([("invoke", "add", 1, 0), ("ok", "add", 1, 0), ("invoke", "read", None, 1), ("ok", "read", [1], 1),
            ("invoke", "add", 2, 0), ("ok", "add", 2, 0), ("invoke", "read", None, 1), ("ok", "read", [1], 1),
            ("invoke", "read", None, 1), ("ok", "read", [1, 2], 1), ("invoke", "add", 3, 0), ("info", "add", 3, 0)])
    r = osf.check(h, linearizable=True)
    assert r["valid?"] is False and r["stale"] == [2] and r["never-read"] == [3] and r["stable-count"] == 2 and r["lost-count"] == 0
    assert osf.check(h, linearizable=False)["valid?"] is True
    lost = _h([("invoke", "add", 7, 0), ("ok", "add", 7, 0), ("invoke", "read", None, 1), ("ok", "read", [7], 1),
               ("invoke", "read", None, 1), ("ok", "read", [], 1)])
    r = osf.check(lost)
    assert r["valid?"] is False and r["lost"] == [7]
    assert osf.check(_h([("invoke", "add", 1, 0), ("ok", "add", 1, 0)]))["valid?"] == "unknown"     # nothing ever read
    # a read that completed before the add was even invoked says nothing about the element
    early = _h([("invoke", "read", None, 1), ("ok", "read", [], 1), ("invoke", "add", 5, 0), ("ok", "add", 5, 0),
                ("invoke", "read", None, 1), ("ok", "read", [5], 1)])
    st = osf.element_states(early)[0]
    assert st["last_absent"] == osf.NONE and st["last_present"] == 4 and st["known"] == 3
    # known = the read's completion when it sees the element before the add is acknowledged
    seen = _h([("invoke", "add", 9, 0), ("invoke", "read", None, 1), ("ok", "read", [9], 1), ("ok", "add", 9, 0)])
    assert osf.element_states(seen)[0]["known"] == 2


def test_encoder_builds_the_membership_matrix():
    h = _h([("invoke", "add", 10, 0), ("ok", "add", 10, 0), ("invoke", "add", 11, 1), ("invoke", "read", None, 2),
            ("ok", "read", [10, 99], 2), ("info", "add", 11, 1), ("invoke", "read", None, 2),
            ("ok", "read", [11, 10], 2)])
    e = sf.Encoded(h)
    assert e.elements == [10, 11] and e.R == 2 and e.wpr == 1
    assert e.add_invoke.tolist() == [0, 2] and e.add_ok.tolist() == [1, N.NO_OP]
    assert e.read_invoke.tolist() == [3, 6] and e.read_ok.tolist() == [4, 7]
    assert e.present[:, 0].tolist() == [0b01, 0b11]          # 99 was never added: not a column


def test_compact_rows_say_what_the_matrix_says():
    """tbc_setfull_rows (top / exc_off / exc: what the checker hands to the device, which builds the matrix there) against the
    dense matrix, read by read: ones below top, the listed exceptions flipped -- on histories with lost, stale and never-read
    elements, re-adds, and reads that hold values nobody added."""
    for seed in range(6):
        hist = _lossy_set_history(seed, n_ops=600)[0]
        first = next(o["value"] for o in hist if o["f"] == "add")
        if seed % 2:
            hist = hist + _h([("invoke", "add", first, 0), ("ok", "add", first, 0), ("invoke", "read", None, 1),
                              ("ok", "read", [first, 10 ** 7], 1)])
            for i, o in enumerate(hist):
                o["index"] = i
        enc = sf.Encoded(hist)
        dense = enc.present
        assert enc.exc_off[0] == 0 and len(enc.exc) == enc.exc_off[-1] and (np.diff(enc.exc_off.astype(np.int64)) >= 0).all()
        n_exc = 0
        for r in range(enc.R):
            row = np.zeros(enc.wpr * 32, bool)
            row[:enc.top[r]] = True
            ex = enc.exc[int(enc.exc_off[r]):int(enc.exc_off[r + 1])]
            assert len(np.unique(ex)) == len(ex) and (ex < enc.E).all()
            row[ex] ^= True
            assert np.array_equal(np.packbits(row, bitorder="little").view(np.uint32), dense[r]), (seed, r)
            n_exc += len(ex)
        assert n_exc > 0


def _lossy_set_history(seed, n_ops=3000, lose=3, stale=4):
    """A grow-only-set history from the simulated store, then damaged: `lose` elements vanish from every read after some
    point (lost), `stale` elements are hidden from a few reads right after their add (stale), late adds stay unread."""
    import random
    rng = random.Random(seed)
    h = set_history(n_ops, 6, seed, busy=0.4, info=0.002)
    added = [o["value"] for o in h if o["type"] == "ok" and o["f"] == "add"]
    victims = rng.sample(added[: len(added) // 2], lose)
    shy = rng.sample(added[len(added) // 4: len(added) * 3 // 4], stale)
    ok_at = {o["value"]: i for i, o in enumerate(h) if o["type"] == "ok" and o["f"] == "add"}
    cut = len(h) * 2 // 3
    for i, o in enumerate(h):
        if o["type"] == "ok" and o["f"] == "read" and o["value"] is not None:
            v = [x for x in o["value"] if not (x in victims and i > cut)]
            v = [x for x in v if not (x in shy and ok_at[x] < i < ok_at[x] + 40)]
            o["value"] = v
    return h, set(victims), set(shy)


@pytest.mark.gpu
def test_scan_matches_the_restatement(native):
    for seed in range(4):
        h, victims, shy = _lossy_set_history(seed)
        enc = sf.Encoded(h)
        with sf.Scan(enc) as s:
            st = s.run()
        exp = osf.element_states(h)
        assert enc.elements == [e["element"] for e in exp]
        assert st["known"].tolist() == [e["known"] for e in exp]
        assert st["last_present"].tolist() == [e["last_present"] for e in exp]
        assert st["last_absent"].tolist() == [e["last_absent"] for e in exp]
        for lin in (True, False):
            got, want = sf.result_map(enc, st, lin), osf.check(h, lin)
            for k in ("valid?", "attempt-count", "stable-count", "lost-count", "lost", "never-read-count", "never-read", "stale-count", "stale"):
                assert got[k] == want[k], (seed, lin, k)
            assert [w["element"] for w in got["worst-stale"]][:3] == [w["element"] for w in want["worst-stale"]][:3]
        assert set(want["lost"]) == victims and got["valid?"] is False
        assert st["bytes_scanned"] <= 2 * st["bytes_matrix"] + 4096


@pytest.mark.gpu
def test_matrix_built_on_the_device_equals_the_matrix_from_the_host(native):
    """tbc_setfull_create_rows (compact reads in, the matrix built by setfull_rows_kernel) against tbc_setfull_create (the dense
    matrix over PCIe): the same three indices for every element; malformed compact inputs are refused before any device work."""
    for seed in range(4):
        enc = sf.Encoded(_lossy_set_history(10 + seed, n_ops=2500)[0])
        with sf.Scan(enc, rows=True) as a, sf.Scan(enc, rows=False) as b:
            ra, rb = a.run(), b.run()
        for k in ("known", "last_present", "last_absent"):
            assert np.array_equal(ra[k], rb[k]), (seed, k)
        assert ra["bytes_scanned"] == rb["bytes_scanned"]
    bad = sf.Encoded(_lossy_set_history(3, n_ops=300)[0])
    bad.exc = bad.exc.copy(); bad.exc[0] = bad.E + 5
    with pytest.raises(N.TbcError):
        sf.Scan(bad, rows=True)
    # an element listed twice in one read's exceptions: the kernel flips the listed bits, the second flip would undo the first -- refused
    dup = sf.Encoded(_lossy_set_history(3, n_ops=300)[0])
    r = int(np.argmax(np.diff(dup.exc_off) >= 2))
    assert dup.exc_off[r + 1] - dup.exc_off[r] >= 2
    dup.exc = dup.exc.copy(); dup.exc[int(dup.exc_off[r]) + 1] = dup.exc[int(dup.exc_off[r])]
    with pytest.raises(N.TbcError, match="twice"):
        sf.Scan(dup, rows=True)
    # ... in whatever order it is listed; and an UNSORTED list without duplicates is what it always was: the same scan
    enc = sf.Encoded(_lossy_set_history(12, n_ops=2500)[0])
    with sf.Scan(enc, rows=True) as a:
        ra = a.run()
    rev = sf.Encoded(_lossy_set_history(12, n_ops=2500)[0])
    rev.exc = rev.exc.copy()
    for r in range(len(rev.exc_off) - 1):
        lo, hi = int(rev.exc_off[r]), int(rev.exc_off[r + 1])
        rev.exc[lo:hi] = rev.exc[lo:hi][::-1]
    assert not np.array_equal(rev.exc, enc.exc)
    with sf.Scan(rev, rows=True) as b:
        rb = b.run()
    for k in ("known", "last_present", "last_absent"):
        assert np.array_equal(ra[k], rb[k]), k
    dup2 = sf.Encoded(_lossy_set_history(3, n_ops=300)[0])
    lo = int(dup2.exc_off[r_dup := int(np.argmax(np.diff(dup2.exc_off) >= 3))])
    if dup2.exc_off[r_dup + 1] - lo >= 3:
        dup2.exc = dup2.exc.copy(); dup2.exc[lo + 2] = dup2.exc[lo]; dup2.exc[lo], dup2.exc[lo + 1] = dup2.exc[lo + 1], dup2.exc[lo]
        with pytest.raises(N.TbcError, match="twice"):
            sf.Scan(dup2, rows=True)


@pytest.mark.gpu
def test_checker_surface_as_the_reference_composes_it(native):
    """(independent/checker (checker/compose {:set-full (checker/set-full {:linearizable? true}) ...})), set_full.clj:155-158."""
    t = independent.tuple_
    good = set_history(1500, 5, 21, busy=0.3)
    bad, victims, _ = _lossy_set_history(22, n_ops=1500, lose=2, stale=0)
    hist = [dict(o, value=t(k, o["value"]), process=o["process"] * 4 + k) for k, hh in ((1, good), (2, bad)) for o in hh]
    c = independent.checker(jc.compose({"set-full": jc.set_full({"linearizable?": True})}))
    r = c.check({}, hist, {})
    assert r["results"][1]["set-full"]["valid?"] in (True, False)          # (a generated history may hold a stale read)
    assert r["results"][2]["set-full"]["valid?"] is False and set(r["results"][2]["set-full"]["lost"]) == victims
    for k in (1, 2):
        want = osf.check(good if k == 1 else bad, True)
        got = r["results"][k]["set-full"]
        assert all(got[f] == want[f] for f in ("valid?", "stable-count", "lost", "never-read", "stale")), k


@pytest.mark.gpu
def test_large_matrix_against_numpy(native):
    """65,536 elements x 8,192 reads (64 MB): element e appears from read first[e] on, a few elements vanish again, a few
    reads miss recent elements.  numpy computes the three reductions directly."""
    rng = np.random.default_rng(5)
    E, R = 65536, 8192
    add_invoke = np.sort(rng.choice(4 * (E + R), E, replace=False)).astype(np.uint32) * 2
    read_invoke = (np.sort(rng.choice(4 * (E + R), R, replace=False)).astype(np.uint32) * 2 + 1)
    read_ok = read_invoke + rng.integers(1, 2000, R).astype(np.uint32) * 2
    add_ok = np.where(rng.random(E) < 0.02, N.NO_OP, add_invoke + rng.integers(1, 3000, E).astype(np.uint32) * 2 + 1).astype(np.uint32)
    vis = np.where(add_ok == N.NO_OP, add_invoke + 500, add_ok)                      # visible from about its ack on
    present = read_invoke[:, None] > vis[None, :]
    gone = rng.choice(E, 200, replace=False)
    present[R * 3 // 4:, gone] = False                                              # lost
    present &= ~((rng.random((R, E)) < 0.0005) & (read_invoke[:, None] < vis[None, :] + 4000))     # stale holes near the add
    class A:                                                                        # the arrays a Scan needs
        pass
    a = A(); a.E, a.R, a.wpr = E, R, E // 32
    a.add_invoke, a.add_ok, a.read_invoke, a.read_ok = add_invoke, add_ok, read_invoke, read_ok
    a.present = np.ascontiguousarray(np.packbits(present, axis=1, bitorder="little").view(np.uint32))
    with sf.Scan(a) as s:
        st = s.run()
        st2 = s.run()
    valid = read_ok[:, None] > add_invoke[None, :]
    inv = read_invoke.astype(np.int64)[:, None]
    lp = np.where(present & valid, inv, -1).max(axis=0)
    la = np.where(~present & valid, inv, -1).max(axis=0)
    kn = np.where(present & valid, read_ok.astype(np.int64)[:, None], 2 ** 40).min(axis=0)
    kn = np.minimum(kn, np.where(add_ok == N.NO_OP, 2 ** 40, add_ok.astype(np.int64)))
    assert np.array_equal(np.where(st["last_present"] == N.NO_OP, -1, st["last_present"].astype(np.int64)), lp)
    assert np.array_equal(np.where(st["last_absent"] == N.NO_OP, -1, st["last_absent"].astype(np.int64)), la)
    assert np.array_equal(np.where(st["known"] == N.NO_OP, 2 ** 40, st["known"].astype(np.int64)), kn)
    assert np.array_equal(st["known"], st2["known"]) and st["bytes_scanned"] == st2["bytes_scanned"]     # idempotent
    assert 0.2 * st["bytes_matrix"] < st["bytes_scanned"] < 1.6 * st["bytes_matrix"]


@pytest.mark.gpu
@pytest.mark.parametrize("E,R", [(1000, 700), (4096 + 77, 3000), (130, 5000)])
def test_rows_of_a_multiple_of_16_bytes_and_rows_that_are_not(native, E, R):
    """the streaming pass takes four word columns a lane (16 B loads) when a row is a multiple of four words and one otherwise: the
    same matrix at words_per_row = ceil(E / 32) rounded up to four, + 1, + 2 and + 3 -- the same three indices as numpy's, the
    same bytes counted"""
    rng = np.random.default_rng(E + R)
    add_invoke = np.sort(rng.choice(4 * (E + R), E, replace=False)).astype(np.uint32) * 2
    read_invoke = (np.sort(rng.choice(4 * (E + R), R, replace=False)).astype(np.uint32) * 2 + 1)
    read_ok = read_invoke + rng.integers(1, 300, R).astype(np.uint32) * 2
    add_ok = np.where(rng.random(E) < 0.05, N.NO_OP, add_invoke + rng.integers(1, 400, E).astype(np.uint32) * 2 + 1).astype(np.uint32)
    vis = np.where(add_ok == N.NO_OP, add_invoke + 100, add_ok)
    present = read_invoke[:, None] > vis[None, :]
    present[R * 2 // 3:, rng.choice(E, max(1, E // 50), replace=False)] = False          # lost
    present &= ~(rng.random((R, E)) < 0.002)                                               # holes anywhere
    valid = read_ok[:, None] > add_invoke[None, :]
    inv = read_invoke.astype(np.int64)[:, None]
    lp = np.where(present & valid, inv, -1).max(axis=0)
    la = np.where(~present & valid, inv, -1).max(axis=0)
    kn = np.where(present & valid, read_ok.astype(np.int64)[:, None], 2 ** 40).min(axis=0)
    kn = np.minimum(kn, np.where(add_ok == N.NO_OP, 2 ** 40, add_ok.astype(np.int64)))
    w4 = (((E + 31) // 32) + 3) // 4 * 4
    scanned = set()
    for wpr in (w4, w4 + 1, w4 + 2, w4 + 3):
        bits = np.zeros((R, wpr * 32), bool); bits[:, :E] = present
        class A:
            pass
        a = A(); a.E, a.R, a.wpr = E, R, wpr
        a.add_invoke, a.add_ok, a.read_invoke, a.read_ok = add_invoke, add_ok, read_invoke, read_ok
        a.present = np.ascontiguousarray(np.packbits(bits, axis=1, bitorder="little").view(np.uint32))
        with sf.Scan(a) as s:
            st = s.run()
        assert np.array_equal(np.where(st["last_present"] == N.NO_OP, -1, st["last_present"].astype(np.int64)), lp), wpr
        assert np.array_equal(np.where(st["last_absent"] == N.NO_OP, -1, st["last_absent"].astype(np.int64)), la), wpr
        assert np.array_equal(np.where(st["known"] == N.NO_OP, 2 ** 40, st["known"].astype(np.int64)), kn), wpr
        scanned.add(int(st["bytes_scanned"]))
    assert len(scanned) == 1, scanned


@pytest.mark.gpu
def test_more_than_256_chunks_of_reads(native):
    """n_reads > 256 * 2048: the resolve pass must consult every chunk's summary, not the first 256 (round 2 held four
    summaries per lane).  600,000 reads of 64 elements; the deciding rows lie in the LAST chunks: every element is
    present in all late reads, one is absent in the very last read, one appears only in the last 1,000 reads."""
    rng = np.random.default_rng(11)
    E, R = 64, 600_000
    add_invoke = (np.arange(E, dtype=np.uint32) * 2)
    add_ok = add_invoke + 1
    read_invoke = (2 * E + np.arange(R, dtype=np.uint32) * 4).astype(np.uint32)
    read_ok = read_invoke + rng.integers(1, 40, R).astype(np.uint32) * 4 + 1      # overlapping reads
    present = np.ones((R, E), bool)
    present[: R - 1000, 7] = False            # element 7 is seen late only
    present[R - 1, 9] = False                 # element 9 is missing from the very last read (lost)
    present[R // 2, 11] = False               # a hole half way: last_absent in a middle chunk
    class A:
        pass
    a = A(); a.E, a.R, a.wpr = E, R, E // 32
    a.add_invoke, a.add_ok, a.read_invoke, a.read_ok = add_invoke, add_ok, read_invoke, read_ok
    a.present = np.ascontiguousarray(np.packbits(present, axis=1, bitorder="little").view(np.uint32))
    with sf.Scan(a) as s:
        st = s.run()
    inv = read_invoke.astype(np.int64)[:, None]
    lp = np.where(present, inv, -1).max(axis=0)
    la = np.where(~present, inv, -1).max(axis=0)
    kn = np.minimum(np.where(present, read_ok.astype(np.int64)[:, None], 2 ** 40).min(axis=0), add_ok.astype(np.int64))
    assert np.array_equal(np.where(st["last_present"] == N.NO_OP, -1, st["last_present"].astype(np.int64)), lp)
    assert np.array_equal(np.where(st["last_absent"] == N.NO_OP, -1, st["last_absent"].astype(np.int64)), la)
    assert np.array_equal(np.where(st["known"] == N.NO_OP, 2 ** 40, st["known"].astype(np.int64)), kn)
    assert lp[9] == read_invoke[R - 2] and la[9] == read_invoke[R - 1] and la[7] == read_invoke[R - 1001]


def test_unsorted_inputs_are_rejected_before_any_device_work(native_lib_loaded=None):
    """tbc_setfull_create checks the documented orders (add_invoke / read_invoke strictly ascending) -- on a machine without
    a GPU it answers TBC_ERR_NO_DEVICE first, which is fine: nothing is computed either way."""
    import ctypes as C
    lib = N.lib()
    ai = np.array([4, 2], np.uint32); ao = np.array([5, 3], np.uint32)
    ri = np.array([6], np.uint32); ro = np.array([7], np.uint32); pr = np.array([3], np.uint32)
    inp = N.SetFullIn(2, 1, 1, 0, ai.ctypes.data_as(C.POINTER(C.c_uint32)), ao.ctypes.data_as(C.POINTER(C.c_uint32)),
                      ri.ctypes.data_as(C.POINTER(C.c_uint32)), ro.ctypes.data_as(C.POINTER(C.c_uint32)), pr.ctypes.data_as(C.POINTER(C.c_uint32)))
    h = C.c_void_p()
    st = lib.tbc_setfull_create(C.byref(inp), C.byref(h))
    assert st in (N.ERR_INVALID_ARG, N.ERR_NO_DEVICE)
    if lib.tbc_device_count() > 0:
        assert st == N.ERR_INVALID_ARG and b"ascending" in lib.tbc_last_error()


def test_recalled_jepsen_details_latency_duplicates_readd():
    """(1) latencies are whole milliseconds of :time (a sub-millisecond gap is not a stale read);
    (2) an element twice in one read -> :duplicated, :valid? false; (3) adding an element again starts it afresh."""
    ms = 1_000_000
    rows = [("invoke", "add", 1, 0, 0), ("ok", "add", 1, 0, 10), ("invoke", "read", None, 1, 20), ("ok", "read", [], 1, 30),
            ("invoke", "read", None, 1, 40), ("ok", "read", [1], 1, 50)]
    h = [{"type": t, "f": f, "value": v, "process": p, "index": i, "time": tm} for i, (t, f, v, p, tm) in enumerate(rows)]
    r = osf.check(h, linearizable=True)                      # absent 10 ns after the ack: 0 ms -> not stale
    assert r["valid?"] is True and r["stale"] == [] and r["stable-latencies"] == {0: 0, 0.5: 0, 0.95: 0, 0.99: 0, 1: 0}
    slow = [dict(o, time=o["time"] * ms) for o in h]          # the same gaps in milliseconds: stale by 10 ms (20 ms + 1 ns - 10 ms, truncated)
    r = osf.check(slow, linearizable=True)
    assert r["valid?"] is False and r["stale"] == [1] and r["worst-stale"][0]["stable-latency"] == 10
    enc = sf.Encoded(slow)
    st = {"known": np.array([1], np.uint32), "last_present": np.array([4], np.uint32), "last_absent": np.array([2], np.uint32)}
    g = sf.result_map(enc, st, True)
    assert g["valid?"] is False and g["stale"] == [1] and g["worst-stale"][0]["stable-latency"] == 10
    assert sf.result_map(sf.Encoded(h), st, True)["valid?"] is True
    # duplicates
    d = _h([("invoke", "add", 1, 0), ("ok", "add", 1, 0), ("invoke", "read", None, 1), ("ok", "read", [1, 1, 1], 1)])
    r = osf.check(d)
    assert r["valid?"] is False and r["duplicated"] == {1: 3} and r["duplicated-count"] == 1 and r["stable-count"] == 1
    e = sf.Encoded(d)
    assert e.duplicated == {1: 3}
    g = sf.result_map(e, {"known": np.array([1], np.uint32), "last_present": np.array([2], np.uint32), "last_absent": np.array([N.NO_OP], np.uint32)}, False)
    assert g["valid?"] is False and g["duplicated"] == {1: 3}
    # re-adding: the read before the second :add invocation no longer counts for the element
    ra = _h([("invoke", "add", 5, 0), ("ok", "add", 5, 0), ("invoke", "read", None, 1), ("ok", "read", [], 1),
             ("invoke", "add", 5, 0), ("ok", "add", 5, 0), ("invoke", "read", None, 1), ("ok", "read", [5], 1)])
    s5 = osf.element_states(ra)[0]
    assert (s5["add_invoke"], s5["known"], s5["last_absent"], s5["last_present"]) == (4, 5, osf.NONE, 6)
    e = sf.Encoded(ra)
    assert e.add_invoke.tolist() == [4] and e.add_ok.tolist() == [5]
