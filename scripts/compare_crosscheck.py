"""Compares the output of scripts/knossos_crosscheck.clj (stock Knossos, one JSON object per line) with
tests/golden/edn/expected.json and prints every disagreement; exit code 1 if there is one.
With --adopt it rewrites expected.json from the stock-Knossos answers (provenance "stock-knossos")."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(ROOT, "tests", "golden", "edn", "expected.json")
exp = json.load(open(path))
stock = {}
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("{"):
        o = json.loads(line)
        stock[o["file"]] = o
bad = 0
for c in exp["cases"]:
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
