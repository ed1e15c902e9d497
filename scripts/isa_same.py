#!/usr/bin/env python3
"""isa_same.py REV_A REV_B [file.hip ...] -- is the DEVICE code of every kernel that exists at REV_A the same at REV_B (REV_B may be WORKTREE)?
hipcc --cuda-device-only -S of each file at both revisions; per kernel symbol the instructions between its label and .Lfunc_end are
compared (comments, .loc / .file directives, block-label numbers, the kernarg size and the mangled names of inlined templates that
gained a trailing defaulted parameter are normalised).  Default files: the sources bench.py's kernel_sha() hashes that hold kernels, plus the level
sweep's.  A committed PMC traffic figure (profiles/*_traffic.json) is keyed by a hash of SOURCES; when an edit leaves the measured
kernels' instructions alone (new template instances beside them, host code) the entry is re-keyed, and this script is the evidence.
No GPU needed."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A, B = sys.argv[1], sys.argv[2]
files = sys.argv[3:] or ["wgl_beam.hip", "wgl_narrow.hip", "pack_open.hip", "jit_sweep_wg.hip", "jit_sweep.hip", "pack.hip"]


def checkout(rev, d):
    c = os.path.join(d, "jepsen-tigerbeetle_amd", "csrc"); os.makedirs(c); os.makedirs(os.path.join(d, "include"))
    if rev == "WORKTREE":
        for sub in ("jepsen-tigerbeetle_amd/csrc", "include"):
            for f in os.listdir(os.path.join(ROOT, sub)):
                if f.endswith((".h", ".hip")):
                    open(os.path.join(d, sub, f), "wb").write(open(os.path.join(ROOT, sub, f), "rb").read())
    else:
        for f in subprocess.check_output(["git", "-C", ROOT, "ls-tree", "--name-only", rev, "jepsen-tigerbeetle_amd/csrc/", "include/"]).decode().split():
            if f.endswith((".h", ".hip")):
                open(os.path.join(d, f), "wb").write(subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:{f}"]))
    return c


def kernels(c, f):
    s_path = os.path.join(c, f + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S", f, "-o", s_path],
                   cwd=c, check=True, stderr=subprocess.DEVNULL)
    s = open(s_path).read()
    out = {}
    for name in re.findall(r"\.amdhsa_kernel (\S+)", s):
        i = s.index("\n" + name + ":"); j = s.index(".Lfunc_end", i)
        body = [re.sub(r";.*$", "", l).rstrip() for l in s[i:j].splitlines() if not re.match(r"^\s*(;|\.loc|\.file)", l)]
        body = [re.sub(r"\.LBB\d+_", ".LBB_", l) for l in body]
        body = [re.sub(r"(ELb[01]|ELi\d+)+EEE", "EEE", l) for l in body]        # (template arguments of inlined bodies' constants)
        out[name] = [l for l in body if l and ".amdhsa_kernarg_size" not in l]      # (an argument struct that grew at its end: metadata, no instruction)
    return out


rc = 0
with tempfile.TemporaryDirectory() as ta, tempfile.TemporaryDirectory() as tb:
    ca, cb = checkout(A, ta), checkout(B, tb)
    for f in files:
        if not os.path.exists(os.path.join(ca, f)):
            continue
        ka, kb = kernels(ca, f), kernels(cb, f)
        for name, body in sorted(ka.items()):
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110] or name[:110]
            if name not in kb:
                print(f"{f}: {short}: NOT AT {B}"); rc = 1
            elif kb[name] == body:
                print(f"{f}: {short}: identical ({len(body)} lines)")
            else:
                print(f"{f}: {short}: DIFFERS"); rc = 1
sys.exit(rc)
