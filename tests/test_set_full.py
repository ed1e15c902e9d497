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
