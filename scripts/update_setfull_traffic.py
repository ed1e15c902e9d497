"""profiles/r06_setfull_traffic.json from the counter passes of scripts/gpu_profile_setfull.sh (run on the GPU box, in the same call):
usage: python scripts/update_setfull_traffic.py gpurun_out/prof_<tag>/pmc_summary.txt"""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
txt = open(sys.argv[1]).read()
def val(kern, ctr):
    for l in txt.splitlines():
        if kern in l and (" " + ctr + " ") in l:
            return float(re.search(r"per_launch=([0-9.e+]+)", l).group(1))
    raise SystemExit(f"no {ctr} of {kern} in {sys.argv[1]}")
f_any, f_res = val("setfull_any", "FETCH_SIZE"), val("setfull_resolve", "FETCH_SIZE")
w_any, w_res = val("setfull_any", "WRITE_SIZE"), val("setfull_resolve", "WRITE_SIZE")
with open(os.path.join(ROOT, "jepsen-tigerbeetle_amd", "csrc", "set_full.hip"), "rb") as fh:
    sha = hashlib.sha256(fh.read()).hexdigest()[:16]
d = {"source": "scripts/gpu_profile_setfull.sh (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in passes of their own, the set-full leg of bench.py); summary committed as profiles/r06_setfull_pmc.txt",
     "set_full_sha": sha, "elements": 262144, "reads": 32768,
     "counters_KB": {"setfull_any_kernel": {"FETCH_SIZE": f_any, "WRITE_SIZE": w_any}, "setfull_resolve_kernel": {"FETCH_SIZE": f_res, "WRITE_SIZE": w_res}},
     "correction": "MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (16 B a lane) -- setfull_any_kernel's loads are those, its FETCH_SIZE is doubled; setfull_resolve_kernel's 4 B loads of 64 lines and both WRITE_SIZEs are taken as counted (uncalibrated)",
     "traffic_bytes_per_scan": int((2 * f_any + f_res + w_any + w_res) * 1024)}
out = os.path.join(ROOT, "profiles", "r06_setfull_traffic.json")
json.dump(d, open(out, "w"), indent=1)
print(out, d["traffic_bytes_per_scan"], sha)
