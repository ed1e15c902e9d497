"""tests/golden/edn/*.edn (Jepsen history.edn format) -> EDN reader -> knossos surface -> tbc_check, compared
with the committed expected.json.  Closes SURVEY.md section 8f row 1 end to end on the device, and is the same set of
files scripts/knossos_crosscheck.clj feeds to stock Knossos on an outside JVM (expected.json's provenance
field says who produced each expectation)."""
import json
import os

import pytest

from helpers import GOLDEN, MODELS
from jepsen_tigerbeetle_amd.jepsen import checker as jc, edn

EDN_DIR = os.path.join(GOLDEN, "edn")
CASES = json.load(open(os.path.join(EDN_DIR, "expected.json")))["cases"]


def _adopt_stock_knossos(cases, path):
    """`make -C clj crosscheck` (any machine with a JVM) leaves stock Knossos's answers in expected_knossos.json, one JSON object per
    line (scripts/knossos_crosscheck.clj).  When the file is there its verdicts REPLACE this repository's own expectations: that is
    what pins parity ("parity unpinned" until then)."""
    if not os.path.exists(path):
        return 0
    stock = {}
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            o = json.loads(line)
            if "linear" in o:
                stock[o["file"]] = o
    n = 0
    for c in cases:
        s = stock.get(c["file"])
        if s is not None and c.get("model") != "bank":
            c["valid?"], c["op-index"], c["provenance"] = s["linear"]["valid?"], s["linear"]["op-index"], "stock-knossos"
            n += 1
    return n


ADOPTED = _adopt_stock_knossos(CASES, os.path.join(EDN_DIR, "expected_knossos.json"))


def test_edn_goldens_are_committed_and_well_formed():
    assert len(CASES) >= 34
    for c in CASES:
        h = edn.read_history(os.path.join(EDN_DIR, c["file"]))
        assert all(op["index"] == i for i, op in enumerate(h)), c["file"]
        assert c["provenance"] in ("hand-derived", "oracle", "stock-knossos")
        assert (c["op-index"] is None) == (c["valid?"] is True), c["file"]
        if c["op-index"] is not None:
            assert h[c["op-index"]]["type"] == "ok", c["file"]


@pytest.mark.gpu
@pytest.mark.parametrize("algorithm", ["wgl", "linear", None])
def test_edn_files_through_the_device(native, algorithm):
    for c in CASES:
        h = edn.read_history(os.path.join(EDN_DIR, c["file"]))
        a = jc.linearizable({"model": MODELS[c["model"]](), "algorithm": algorithm}).check(None, h, None)
        assert a["valid?"] is c["valid?"], (c["file"], algorithm)
        if c["valid?"] is False:
            assert a["op"]["index"] == c["op-index"], (c["file"], algorithm)


def test_stock_knossos_answers_replace_the_expectations_when_present(tmp_path):
    cases = [{"file": "a.edn", "model": "cas-register", "valid?": True, "op-index": None, "provenance": "oracle"},
             {"file": "b.edn", "model": "bank", "valid?": True, "op-index": None, "provenance": "oracle"}]
    p = tmp_path / "expected_knossos.json"
    assert _adopt_stock_knossos(cases, str(p)) == 0
    p.write_text(json.dumps({"file": "a.edn", "model": "cas-register", "provenance": "stock-knossos", "valid?": False, "op-index": 7,
                             "wgl": {"valid?": False}, "linear": {"valid?": False, "op-index": 7}}) + "\n" +
                 json.dumps({"file": "b.edn", "model": "bank", "skipped": "no such model in Knossos"}) + "\n")
    assert _adopt_stock_knossos(cases, str(p)) == 1
    assert (cases[0]["valid?"], cases[0]["op-index"], cases[0]["provenance"]) == (False, 7, "stock-knossos")
    assert cases[1]["provenance"] == "oracle"
