"""Host-side logic of the boundary (no GPU): tbc_pair_events, tbc_memo_build,
the knossos.history / jepsen mirrors, value interning."""
import numpy as np
import pytest

from jepsen_tigerbeetle_amd import _native as N, columns
from jepsen_tigerbeetle_amd.columns import EventColumns
from jepsen_tigerbeetle_amd.knossos import _analysis, history as H, memo, model as M, op as kop
from jepsen_tigerbeetle_amd.jepsen import checker as jc, independent


def ev(rows):
    t = {"invoke": 0, "ok": 1, "fail": 2, "info": 3}
    return EventColumns(np.array([t[r[0]] for r in rows], np.uint8), np.array([r[1] for r in rows], np.int32),
                        np.array([r[2] for r in rows], np.uint8), np.array([r[3] for r in rows], np.int32),
                        np.array([r[4] for r in rows], np.int32))


def test_pair_events_complete_without_failures(native):
    rows = [("invoke", 7, N.F_READ, N.NIL, 0), ("invoke", 9, N.F_CAS, 1, 2), ("ok", 7, N.F_READ, 4, 0),
            ("fail", 9, N.F_CAS, 1, 2), ("invoke", 9, N.F_WRITE, 3, 0), ("info", 9, N.F_WRITE, 3, 0),
            ("invoke", 7, N.F_WRITE, 5, 0)]
    ops = columns.pair_events(ev(rows))
    assert len(ops) == 3 and ops.n_events == 7
    assert list(ops.f) == [N.F_READ, N.F_WRITE, N.F_WRITE]
    assert list(ops.a) == [4, 3, 5]                      # the read learned its value
    assert list(ops.inv_pos) == [0, 4, 6]
    assert list(ops.ret_pos) == [2, N.POS_CRASHED, N.POS_CRASHED]
    assert list(ops.process) == [0, 1, 0] and ops.n_process == 2   # dense, first appearance


@pytest.mark.parametrize("rows", [
    [("invoke", 0, 0, 0, 0), ("invoke", 0, 0, 0, 0)],                       # double invoke
    [("ok", 0, 0, 0, 0)],                                                   # completion without invoke
    [("invoke", 0, 1, 1, 0), ("info", 0, 1, 1, 0), ("invoke", 0, 1, 2, 0)],  # invoke after crash
])
def test_pair_events_rejects_malformed(native, rows):
    with pytest.raises(N.TbcError) as e:
        columns.pair_events(ev(rows))
    assert e.value.status == N.ERR_BAD_HISTORY


def test_memo_table_matches_model(native):
    ops = [{"f": "write", "value": v} for v in range(3)] + [{"f": "read", "value": v} for v in (None, 0, 1, 2)] + \
          [{"f": "cas", "value": [a, b]} for a in range(3) for b in range(3)]
    info = memo.memo(M.cas_register(), ops)
    assert len(info["states"]) == 4                      # nil, 0, 1, 2
    for s, mod in enumerate(info["states"]):
        for c, o in enumerate(info["class_ops"]):
            nxt = mod.step(o)
            t = info["table"][s][c]
            if M.inconsistent_p(nxt):
                assert t == 0xFFFF
            else:
                assert info["states"][t] == nxt


def test_memo_gives_up_past_cap(native):
    ops = [{"f": "add", "value": v} for v in range(20)]
    with pytest.raises(memo.MemoTooLarge):
        memo.memo(M.set(), ops, max_states=256)


def test_history_helpers():
    h = [kop.invoke(0, "read", None), {"type": "info", "f": "start", "process": "nemesis", "value": None},
         kop.ok(0, "read", 3), kop.invoke(1, "cas", [1, 2]), kop.fail(1, "cas", [1, 2]), kop.invoke(2, "write", 1)]
    h = H.index(h)
    assert [o["index"] for o in h] == list(range(6))
    c = H.complete(h)
    assert c[0]["value"] == 3 and c[3].get("fails?")
    assert [o["f"] for o in H.without_failures(c)] == ["read", "start", "read", "write"]
    assert [o["f"] for o in H.unmatched_invokes(h)] == ["write"]
    p = H.pair_index(h)
    assert p[0] == 2 and p[2] == 0 and H.completion(h, p, 3)["type"] == "fail"


def test_encoding_interns_non_integer_values(native):
    h = [kop.invoke(0, "write", "a"), kop.ok(0, "write", "a"), kop.invoke(1, "read", None), kop.ok(1, "read", "a"),
         {"type": "info", "f": "start", "process": "nemesis", "value": None}]
    enc = _analysis.Encoded(M.register(), h)
    assert len(enc.ops) == 2 and enc.ops.a[0] == enc.ops.a[1] and enc.ops.n_events == 4
    assert enc.model_of_state(int(enc.ops.a[0])) == M.Register("a")
    with pytest.raises(ValueError):
        _analysis.Encoded(M.register(), [kop.invoke(0, "cas", [1, 2])])


def test_merge_valid_and_independent_split():
    assert jc.merge_valid([True, True]) is True
    assert jc.merge_valid([True, "unknown"]) == "unknown"
    assert jc.merge_valid([True, "unknown", False]) is False
    t = independent.tuple_
    h = [kop.invoke(0, "add", t(1, 9)), kop.invoke(1, "add", t(2, 10)), kop.ok(0, "add", t(1, 9)),
         {"type": "info", "f": "start", "process": "nemesis", "value": None}, kop.ok(1, "add", t(2, 10))]
    assert independent.history_keys(h) == [1, 2]
    s1 = independent.subhistory(1, h)
    assert [o["value"] for o in s1 if o["process"] != "nemesis"] == [9, 9] and len(s1) == 3


def test_the_committed_traffic_figure_is_a_measurement():
    """bench.py quotes roofline.traffic only for the kernel sources it was measured on (profiles/r*_traffic.json carries their hash; any
    other tree reports traffic: null until scripts/gpu_profile_r05.sh has run its PMC passes on it).  The file itself must be a
    measurement: written by scripts/update_traffic.py from FETCH_SIZE / WRITE_SIZE passes, no hand-edited `rekeyed` essay standing
    in for one (round 4's did)."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "r*_traffic.json")), reverse=True)
    assert files
    if os.path.basename(files[0]) < "r05":
        return          # (round 4's file: superseded as soon as round 5's passes have run)
    entries = json.load(open(files[0]))["entries"]
    assert entries
    for e in entries:
        assert "rekeyed" not in e and e["traffic_bytes"] == e["fetch_bytes"] + e["write_bytes"] and len(e["kernel_sha"]) == 16
