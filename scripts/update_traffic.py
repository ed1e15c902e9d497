"""Rewrite profiles/r05_traffic.json from the PMC passes of scripts/gpu_profile_r05.sh (run on the GPU box right after them).

usage: update_traffic.py <prof dir> <batch> <lanes per history> <visited per op>
Reads <dir>/pmc_summary.txt (FETCH_SIZE / WRITE_SIZE per launch of the search kernel, KB) and <dir>/trace.log (probes and new
configs of the same launches), stamps the entry with the kernel_sha of the sources it is run from -- bench.py only
copies an entry whose sha and configuration match the build it measures.
"""
import importlib.util, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, B, L, VPO = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
KERNEL = "wgl_narrow_kernel" if L < 64 else "wgl_beam_kernel"
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
vals, ms = {}, None
for line in open(os.path.join(d, "pmc_summary.txt")):
    m = re.search(KERNEL + r".*?(FETCH_SIZE|WRITE_SIZE)\s+launches=\d+\s+grid=\d+\s+per_launch=([0-9.e+]+)\s+\(kernel ([0-9.]+) ms", line)
    if m:
        vals[m.group(1)] = float(m.group(2)) * 1024.0
        ms = float(m.group(3))
run = [l for l in open(os.path.join(d, "trace.log")) if " run" in l and "probes=" in l][-1]
probes, visited = int(re.search(r"probes=(\d+)", run).group(1)), int(re.search(r"visited=(\d+)", run).group(1))
path = os.path.join(ROOT, "profiles", os.environ.get("TBC_TRAFFIC_FILE", "r05_traffic.json"))       # (a later round: TBC_TRAFFIC_FILE=r05_traffic.json)
doc = {"_comment": "HBM bytes per batch launch of the dominant kernel from rocprofv3 PMC passes: FETCH_SIZE and WRITE_SIZE in separate --pmc runs, "
                   "values in KB (x1024), used as counted: calibrated for this access pattern in round 2 (scripts/hbm_calib.hip, "
                   "profiles/r02_hbm_counter_calibration.txt: random 64 B bucket reads are counted 1.16x, a 16 B or 8 B store costs a 32 B sector; the 2x "
                   "under-report applies to wide coalesced streams only).  bench.py copies an entry into roofline.traffic only if config AND kernel_sha match "
                   "the build it runs."}
fetch, write = int(vals["FETCH_SIZE"]), int(vals["WRITE_SIZE"])
alg = 16 * (probes - visited) + 32 * visited
doc["entries"] = [{
    "kernel": KERNEL, "kernel_sha": bench.kernel_sha(), "histories_per_gpu": B, "search_width": 1 if L < 64 else 2, "lanes_per_history": L,
    "visited_per_op": VPO, "ops": 10000, "procs": 64, "busy": 0.1, "info": 0.0,
    "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "kernel_ms": ms,
    "algorithmic_bytes": alg, "traffic_over_algorithmic": round((fetch + write) / alg, 2),
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (the bench's own batch: seeds 0 .. B-1 through scripts/gpu_narrow_ab.py) of " +
              os.environ.get("TBC_TRAFFIC_SOURCE", "scripts/gpu_profile_r05.sh; summary committed as profiles/r05_pmc_final.txt"),
    "per_new_config_write_bytes": round(write / visited, 1), "per_probe_fetch_bytes": round(fetch / probes, 1)}]
json.dump(doc, open(path, "w"), indent=1)
print("traffic", fetch + write, "sha", bench.kernel_sha(), "kernel_ms", ms)
