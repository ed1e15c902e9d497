"""Rewrite profiles/r02_traffic.json from the PMC passes of scripts/gpu_profile_r02.sh (run on the GPU box right after them).

usage: update_traffic.py <prof dir> <batch> <width>
Reads <dir>/pmc_summary.txt (FETCH_SIZE / WRITE_SIZE per launch of wgl_beam_kernel, KB) and <dir>/trace.log (probes and new
configs of the same launches), stamps the entry with the kernel_sha of the sources it is run from -- bench.py only
copies an entry whose sha and configuration match the build it measures.
"""
import importlib.util, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, B, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
vals, ms = {}, None
for line in open(os.path.join(d, "pmc_summary.txt")):
    m = re.search(r"wgl_beam_kernel.*?(FETCH_SIZE|WRITE_SIZE)\s+launches=\d+\s+grid=\d+\s+per_launch=([0-9.e+]+)\s+\(kernel ([0-9.]+) ms", line)
    if m:
        vals[m.group(1)] = float(m.group(2)) * 1024.0
        ms = float(m.group(3))
run = [l for l in open(os.path.join(d, "trace.log")) if " run" in l and "steps=" in l][-1]
probes, visited = int(re.search(r"steps=(\d+)", run).group(1)), int(re.search(r"visited=(\d+)", run).group(1))
path = os.path.join(ROOT, "profiles", "r02_traffic.json")
doc = json.load(open(path))
fetch, write = int(vals["FETCH_SIZE"]), int(vals["WRITE_SIZE"])
doc["entries"] = [{
    "kernel": "wgl_beam_kernel<1,false,true>", "kernel_sha": bench.kernel_sha(), "histories_per_gpu": B, "search_width": W,
    "visited_per_op": 8, "ops": 10000, "procs": 64, "busy": 0.1, "info": 0.0,
    "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "kernel_ms": ms,
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_profile_r02.sh (64 seeds x 512 = 32,768 histories through "
              "scripts/gpu_quick_bench.py at this width; the bench's 32,768 distinct seeds do the same work within 0.5 %); summary "
              "committed as profiles/r02_pmc_w2_b32768.txt",
    "per_new_config_write_bytes": round(write / visited, 1), "per_probe_fetch_bytes": round(fetch / probes, 1)}]
json.dump(doc, open(path, "w"), indent=1)
print("traffic", fetch + write, "sha", bench.kernel_sha(), "kernel_ms", ms)
