"""The Clojure / JNA shim (clj/src/tigerbeetle/checker/gpu_linear.clj) cannot be executed here (no JVM: SURVEY.md section 0 F4).  What a
machine without a JVM CAN check, it checks: the file reads as Clojure forms (balanced brackets, strings, comments), its `abi` table
-- every struct size, field offset and enum code the shim uses -- equals the library's ctypes binding (which tests/test_abi.py pins to
include/tbcheck.h with gcc), every (o :struct :field) in the code names an entry of that table, no .getX / .setX takes a bare numeric
offset, the entry points it invokes are exported by libtbcheck.so, and the call-site patches apply to the reference where it is on disk."""
import ctypes as C
import os
import re
import shutil
import subprocess

import pytest

from jepsen_tigerbeetle_amd.jepsen import edn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "clj", "src", "tigerbeetle", "checker", "gpu_linear.clj")
REF = "/root/reference"


def top_level_forms(src):
    """Clojure reader, brackets only: returns the (start, end) spans of the top-level forms; raises on imbalance."""
    close = {"(": ")", "[": "]", "{": "}"}
    stack, spans, i, n, start = [], [], 0, len(src), None
    while i < n:
        c = src[i]
        if c == ";":
            while i < n and src[i] != "\n":
                i += 1
            continue
        if c == "\\":                      # character literal: \( \newline \a
            m = re.match(r"\\(newline|space|tab|return|backspace|formfeed|u[0-9a-fA-F]{4}|.)", src[i:], re.S)
            i += len(m.group(0))
            continue
        if c == '"':
            if not stack and start is None:
                start = i
            i += 1
            while src[i] != '"':
                i += 2 if src[i] == "\\" else 1
            i += 1
            if not stack:
                spans.append((start, i)); start = None
            continue
        if c in close:
            if not stack:
                start = i if start is None else start
            stack.append((c, i))
        elif c in ")]}":
            assert stack, f"unbalanced {c!r} at offset {i} (line {src.count(chr(10), 0, i) + 1})"
            o, at = stack.pop()
            assert close[o] == c, f"{o!r} opened on line {src.count(chr(10), 0, at) + 1} closed by {c!r} on line {src.count(chr(10), 0, i) + 1}"
            if not stack:
                spans.append((start, i + 1)); start = None
        elif not stack and c in "#'^`~@" and start is None:
            start = i                       # reader prefix of the next top-level form
        i += 1
    assert not stack, f"{stack[-1][0]!r} opened on line {src.count(chr(10), 0, stack[-1][1]) + 1} is never closed"
    return spans


@pytest.fixture(scope="module")
def shim():
    src = open(SHIM).read()
    forms = [src[a:b] for a, b in top_level_forms(src)]
    return src, forms


@pytest.fixture(scope="module")
def abi(shim):
    _, forms = shim
    (f,) = [x for x in forms if x.startswith("(def abi")]
    return edn.loads(f[len("(def abi"):-1])


def test_the_shim_reads_as_clojure_forms(shim):
    src, forms = shim
    assert forms[0].startswith("(ns tigerbeetle.checker.gpu-linear")
    names = set(re.findall(r"^\((?:defn-?|def|defrecord)\s+(?:\^\S+\s+)?([^\s\]\[()]+)", src, re.M))
    assert {"abi", "analysis", "linearizable", "check-batch", "memo-table", "set-full-indices", "bank-model", "Bank",
            "open-stream", "check-next!", "close-stream!", "comm-unique-id", "comm-init", "sharded-analysis"} <= names, names
    for clj in ("scripts/knossos_crosscheck.clj",):
        top_level_forms(open(os.path.join(ROOT, clj)).read())


def test_brackets_checker_catches_imbalance():
    top_level_forms('(a [b {c "d)" \\( ; )\n}])')
    for bad in ("(a [b)]", "(a", "a)", '(a "b)'):
        with pytest.raises((AssertionError, IndexError)):
            top_level_forms(bad)


def test_abi_table_equals_the_ctypes_binding(native, abi):
    N = native
    structs = {"events": N.Events, "ops": N.Ops, "model": N.Model, "opts": N.Opts, "config": N.Config, "result": N.Result,
               "batch_desc": N.BatchDesc, "setfull_in": N.SetFullIn, "setfull_out": N.SetFullOut, "setfull_rows": N.SetFullRows,
               "batch_input": N.BatchInput, "input_info": N.InputInfo, "progress": N.Progress}
    assert abi["version"] == N.lib().tbc_version() == 2
    assert set(abi) == set(structs) | {"version", "enums"}
    for name, cls in structs.items():
        tab = dict(abi[name])
        assert tab.pop("size") == C.sizeof(cls), name
        assert tab, name
        for field, off in tab.items():
            assert getattr(cls, field).offset == off, (name, field, off, getattr(cls, field).offset)
    # the shim writes every field of the structs it fills: nothing of those may be missing from the table
    for name in ("events", "ops", "model", "opts", "setfull_in", "setfull_out", "setfull_rows", "batch_desc"):
        assert set(abi[name]) - {"size"} == {f[0] for f in structs[name]._fields_}, name
    e = abi["enums"]
    assert e["type"] == {"invoke": 0, "ok": 1, "fail": 2, "info": 3}
    assert e["f"] == {"read": N.F_READ, "write": N.F_WRITE, "cas": N.F_CAS, "acquire": N.F_ACQUIRE, "release": N.F_RELEASE, "add": N.F_ADD, "txn": N.F_TXN,
                      "transfer": N.F_TRANSFER, "class": N.F_CLASS}
    assert e["model"] == {"register": N.MODEL_REGISTER, "cas-register": N.MODEL_CAS_REGISTER, "mutex": N.MODEL_MUTEX, "table": N.MODEL_TABLE,
                          "multi-register": N.MODEL_MULTI_REGISTER, "set": N.MODEL_SET, "bank": N.MODEL_BANK}
    assert e["alg"] == {"competition": N.ALG_COMPETITION, "wgl": N.ALG_WGL, "linear": N.ALG_LINEAR}
    assert e["valid"] == {"valid": N.VALID, "invalid": N.INVALID, "unknown": N.UNKNOWN}
    assert e["cause"] == {"none": N.CAUSE_NONE, "time-limit": N.CAUSE_TIME_LIMIT, "step-limit": N.CAUSE_STEP_LIMIT, "memory": N.CAUSE_VISITED_FULL}
    assert e["max-final-configs"] == len(N.Result().configs) and e["config-pending"] == len(N.Config().pending)
    assert e["wire-nil"] == N.WIRE_NIL and e["comm-id-bytes"] == N.COMM_ID_BYTES
    # ... and against the header itself, for the enum spellings the binding takes on trust
    hdr = open(os.path.join(ROOT, "include", "tbcheck.h")).read()
    for k, v in e["f"].items():
        assert re.search(rf"TBC_F_{k.upper()}\s*=\s*{v}\b", hdr), k
    for k, v in e["model"].items():
        assert re.search(rf"TBC_MODEL_{k.upper().replace('-', '_')}\s*=\s*{v}\b", hdr), k


def test_every_offset_in_the_code_comes_from_the_table(shim, abi):
    src, forms = shim
    code = "\n".join(f for f in forms if not f.startswith("(def abi"))
    code = re.sub(r";[^\n]*", "", code)
    uses = re.findall(r"\(o\s+:([a-z_]+)\s+:([a-z_0-9]+)\)", code)
    assert len(uses) > 50
    for s, k in uses:
        assert s in abi and k in abi[s], (s, k)
    result_uses = re.findall(r"\(R\s+:([a-z_0-9]+)\)", code)          # result-map's (R :field) = base + (o :result :field)
    assert len(result_uses) >= 8 and all(k in abi["result"] for k in result_uses), result_uses
    # no accessor takes a bare non-zero integer offset: (.getInt res 12) is how INTEGRATION.md's inline version drifted from ABI v1 to v2
    for m in re.finditer(r"\(\.(?:get|set)(?:Int|Long|Byte|Pointer|Short)\s+\S+\s+(\d+)[\s)]", code):
        assert m.group(1) == "0", m.group(0)
    assert not re.search(r"\(Memory\.\s+\d{2,}\)", code), "a struct allocated with a literal size"


def test_entry_points_the_shim_invokes_are_exported(native, shim):
    src, _ = shim
    called = set(re.findall(r'\(f "(tbc_[a-z_]+)"\)', src)) | set(re.findall(r'getFunction l "(tbc_[a-z_]+)"', src))
    assert {"tbc_version", "tbc_pair_events", "tbc_check", "tbc_result_free", "tbc_batch_create", "tbc_batch_run", "tbc_batch_destroy",
            "tbc_memo_build", "tbc_setfull_create", "tbc_setfull_run", "tbc_setfull_destroy",
            "tbc_batch_map_input", "tbc_batch_submit_input", "tbc_comm_unique_id", "tbc_comm_init", "tbc_comm_destroy", "tbc_batch_sweep_allgather"} <= called
    for name in called:
        assert hasattr(native.lib(), name), name


def test_integration_md_points_at_the_files():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for path in ("clj/src/tigerbeetle/checker/gpu_linear.clj", "clj/patches/project.patch", "clj/patches/set_full.patch", "clj/patches/ledger.patch"):
        assert path in doc and os.path.exists(os.path.join(ROOT, path)), path


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None, reason="the reference is only on disk in the build container")
def test_call_site_patches_apply_to_the_reference(tmp_path):
    work = tmp_path / "ref"
    for rel in ("project.clj", "src/tigerbeetle/workloads/set_full.clj", "src/tigerbeetle/tests/ledger.clj"):
        os.makedirs(os.path.dirname(work / rel), exist_ok=True)
        shutil.copy(os.path.join(REF, rel), work / rel)
    for p in ("project", "set_full", "ledger"):
        subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "clj", "patches", p + ".patch")], cwd=work)
    for rel in ("project.clj", "src/tigerbeetle/workloads/set_full.clj", "src/tigerbeetle/tests/ledger.clj"):
        top_level_forms(open(work / rel).read())          # still balanced after the patch
    assert "gpu-linear/linearizable" in open(work / "src/tigerbeetle/workloads/set_full.clj").read()
    assert "net.java.dev.jna/jna" in open(work / "project.clj").read()


def test_the_crosscheck_recipe_still_covers_every_golden_file():
    """`make -C clj crosscheck` is the only route to "parity pinned" (DESIGN.md): stock Knossos / jepsen.checker over tests/golden/edn*/.
    Nobody here can run it, so what can rot silently is checked: the script reads as Clojure, takes its cases from expected.json, knows every
    model and checker those cases name, the Makefile runs it over both directories and hands the output to the comparer -- and every
    .edn file on disk IS a case (a golden file nobody lists would never be cross-checked)."""
    import json
    script = open(os.path.join(ROOT, "scripts", "knossos_crosscheck.clj")).read()
    top_level_forms(script)
    assert '"expected.json"' in script and "(:cases expected)" in script
    known_models = set(re.findall(r'"([a-z-]+)"\s+#\(model/', script))
    mk = open(os.path.join(ROOT, "clj", "Makefile")).read()
    for d in ("edn", "edn_checkers"):
        gold = os.path.join(ROOT, "tests", "golden", d)
        cases = json.load(open(os.path.join(gold, "expected.json")))["cases"]
        listed = {c["file"] for c in cases}
        on_disk = {f for f in os.listdir(gold) if f.endswith(".edn")}
        assert listed == on_disk, (d, sorted(listed ^ on_disk))
        for c in cases:
            if "model" in c and c["model"] != "bank":          # (Knossos ships no bank model: those files pin this repository's own)
                assert c["model"] in known_models, c
            if "checker" in c:
                assert c["checker"] in ("set-full", "linearizable"), c
        assert f"tests/golden/{d} > tests/golden/{d}/expected_knossos.json" in mk
    assert "scripts/compare_crosscheck.py" in mk and os.path.exists(os.path.join(ROOT, "scripts", "compare_crosscheck.py"))
