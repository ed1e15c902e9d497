"""Compares the output of scripts/knossos_crosscheck.clj (stock Knossos, one JSON object per line) with
tests/golden/edn/expected.json and prints every disagreement; exit code 1 if there is one.
With --adopt it rewrites expected.json from the stock-Knossos answers (provenance "stock-knossos").
With --checkers the file is compared with tests/golden/edn_checkers/expected.json instead (jepsen.checker/set-full result keys,
knossos.model/set verdicts; the bank files are this repository's own model and are skipped by the Clojure side)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECKERS = "--checkers" in sys.argv
path = os.path.join(ROOT, "tests", "golden", "edn_checkers" if CHECKERS else "edn", "expected.json")
exp = json.load(open(path))
stock = {}
stock_sf = {}
for line in open([a for a in sys.argv[1:] if not a.startswith("--")][0]):
    line = line.strip()
    if line.startswith("{"):
        o = json.loads(line)
        if o.get("checker") == "set-full":
            stock_sf[(o["file"], bool(o["opts"]["linearizable?"]))] = o
        else:
            stock[o["file"]] = o
bad = 0
for c in exp["cases"]:
    if c.get("checker") == "set-full":
        s = stock_sf.get((c["file"], bool(c["opts"]["linearizable?"])))
        if s is None:
            print("missing from the stock-jepsen run:", c["file"], c["opts"]); bad += 1; continue
        for k, want in c["expect"].items():
            got = s["result"].get(k)
            if k == "duplicated":
                got = {str(a): b for a, b in (got or {}).items()}
            if k == "valid?" and got == "unknown":
                got = "unknown"
            if got != want:
                print(f"{c['file']} {c['opts']}: jepsen.checker/set-full :{k} = {got}, expected {want}"); bad += 1
        if "--adopt" in sys.argv:
            c["provenance"] = "stock-jepsen"
        continue
    if c.get("model") == "bank":
        continue
    s = stock.get(c["file"])
    if s is None:
        print("missing from the stock-Knossos run:", c["file"]); bad += 1; continue
    for who in ("wgl", "linear"):
        if s[who]["valid?"] != c["valid?"]:
            print(f"{c['file']}: {who} says valid? {s[who]['valid?']}, expected {c['valid?']}"); bad += 1
    if c["valid?"] is False and s["linear"]["op-index"] != c["op-index"]:
        print(f"{c['file']}: knossos.linear reports :op index {s['linear']['op-index']}, expected {c['op-index']}"); bad += 1
    if "--adopt" in sys.argv:
        c["valid?"], c["op-index"], c["provenance"] = s["linear"]["valid?"], s["linear"]["op-index"], "stock-knossos"
if "--adopt" in sys.argv:
    json.dump(exp, open(path, "w"), indent=1)
print(f"{len(exp['cases'])} files, {bad} disagreement(s)")
sys.exit(1 if bad else 0)
