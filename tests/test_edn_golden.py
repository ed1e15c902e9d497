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
