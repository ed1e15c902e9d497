"""tests/golden/edn_checkers/*.edn: externally checkable goldens for the checkers the reference actually runs --
`checker/set-full` (set_full.clj:157), `checker/linearizable` with knossos.model/set, and this repository's bank model
(specified from tests/ledger.clj:89-114).  CPU: the files are well formed and the restatements that produced the
expectations still produce them.  GPU: the product path (EDN reader -> checker surface -> C-ABI -> kernels) gives the same.
scripts/knossos_crosscheck.clj feeds the same files to stock jepsen / Knossos on an outside JVM (expected.json's provenance)."""
import json
import os

import pytest

from helpers import GOLDEN
from jepsen_tigerbeetle_amd.jepsen import checker as jc, edn, set_full as sf
from jepsen_tigerbeetle_amd.knossos import model as M
from oracle import set_full as osf

DIR = os.path.join(GOLDEN, "edn_checkers")
CASES = json.load(open(os.path.join(DIR, "expected.json")))["cases"]
KEYS = ("valid?", "attempt-count", "stable-count", "lost-count", "lost", "never-read-count", "never-read", "stale-count", "stale", "duplicated-count")


def _norm(d):
    return {str(k): v for k, v in d.items()}


def test_checker_goldens_are_committed_and_reproducible():
    assert sum(c["checker"] == "set-full" for c in CASES) >= 12 and sum(c["checker"] == "linearizable" for c in CASES) >= 7
    outcomes = set()
    for c in CASES:
        h = edn.read_history(os.path.join(DIR, c["file"]))
        assert all(op["index"] == i for i, op in enumerate(h)), c["file"]
        if c["checker"] == "set-full":
            r = osf.check(h, linearizable=c["opts"]["linearizable?"])
            for k in KEYS:
                assert r[k] == c["expect"][k], (c["file"], k)
            assert _norm(r["duplicated"]) == c["expect"]["duplicated"], c["file"]
            outcomes |= {k for k in ("lost", "stale", "never-read") if r[k]} | ({"duplicated"} if r["duplicated"] else set()) | ({"valid"} if r["valid?"] is True else set())
        else:
            assert (c["op-index"] is None) == (c["valid?"] is True), c["file"]
    assert outcomes >= {"lost", "stale", "never-read", "duplicated", "valid"}


@pytest.mark.gpu
def test_checker_goldens_through_the_device(native):
    for c in CASES:
        h = edn.read_history(os.path.join(DIR, c["file"]))
        if c["checker"] == "set-full":
            got = sf.check(h, linearizable=c["opts"]["linearizable?"])
            for k in KEYS:
                assert got[k] == c["expect"][k], (c["file"], k, got[k], c["expect"][k])
            assert _norm(got["duplicated"]) == c["expect"]["duplicated"], c["file"]
        else:
            model = M.set() if c["model"] == "set" else M.bank(range(1, 9))
            for algorithm in ("linear", None):
                a = jc.linearizable({"model": model, "algorithm": algorithm}).check(None, h, None)
                assert a["valid?"] is c["valid?"], (c["file"], algorithm)
                if c["valid?"] is False:
                    assert a["op"]["index"] == c["op-index"], (c["file"], algorithm)
