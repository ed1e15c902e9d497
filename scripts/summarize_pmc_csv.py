"""Summarise rocprofv3 --pmc --output-format csv passes: per kernel and counter, the mean over that kernel's
largest-grid dispatches (the batch launches).  usage: summarize_pmc_csv.py DIR [DIR ...]"""
import csv, glob, os, sys
from collections import defaultdict
for d in sys.argv[1:]:
    for p in glob.glob(os.path.join(d, "*counter_collection.csv")):
        rows = defaultdict(list)
        for r in csv.DictReader(open(p)):
            rows[(r["Kernel_Name"], r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"]), int(r["Dispatch_Id"]),
                                                               int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"]))
        print("==", p)
        for (k, c), l in sorted(rows.items()):
            if "tbc" not in k and "setfull" not in k:
                continue
            gmax = max(g for g, *_ in l)
            by = defaultdict(float); dur = {}
            for g, v, did, dt, *_ in l:
                if g == gmax:
                    by[did] += v; dur[did] = dt
            vals = list(by.values())
            print("%-58s %-22s launches=%d grid=%-9d per_launch=%.6g  (kernel %.3f ms; vgpr %s sgpr %s lds %s scratch %s)" %
                  (k[:58], c, len(vals), gmax, sum(vals) / len(vals), sum(dur.values()) / len(dur) / 1e6, l[0][4], l[0][5], l[0][6], l[0][7]))
