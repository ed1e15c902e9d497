"""FETCH_SIZE / WRITE_SIZE of the known-byte microbenchmarks (scripts/hbm_calib.hip) against the bytes they really
move.  usage: summarize_calib.py PROF_DIR   (expects calib_fetch/, calib_write/, calib_plain.log inside)"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
known, times = {}, {}
for line in open(os.path.join(d, "calib_plain.log")):
    p = line.split()
    if len(p) >= 7 and p[1] == "known_bytes":
        known[p[0]] = float(p[2]); times[p[0]] = float(p[4])
print("kernel    known bytes   time ms  useful GB/s   FETCH_SIZE (KB->B)  ratio   WRITE_SIZE (KB->B)  ratio")
ctr = {}
for which, sub in (("FETCH_SIZE", "calib_fetch"), ("WRITE_SIZE", "calib_write")):
    acc = defaultdict(list)
    for p in glob.glob(os.path.join(d, sub, "*counter_collection.csv")):
        rows = defaultdict(float)
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == which:
                rows[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (k, _), v in rows.items():
            acc[k].append(v)
    ctr[which] = {k: sum(v) / len(v) * 1024.0 for k, v in acc.items()}      # the counters are in KB
for k in ("read64", "write16", "write8", "cas8", "stream"):
    f, w = ctr["FETCH_SIZE"].get(k, 0.0), ctr["WRITE_SIZE"].get(k, 0.0)
    print("%-8s %12.4g %9.3f %12.1f %20.4g %6.2f %20.4g %6.2f" % (k, known.get(k, 0), times.get(k, 0), known.get(k, 0) / max(times.get(k, 1), 1e-9) / 1e6,
                                                                  f, f / max(known.get(k, 1), 1), w, w / max(known.get(k, 1), 1)))
