#!/usr/bin/env python3
"""Where a gfx950 kernel moves scalar registers to and from VGPR lanes (SGPR spills), block by block.

usage: scripts/spill_map.py <source.hip> <mangled-kernel-name-substring> [hipcc flags...]

Compiles the file for gfx950 with line tables, then prints every basic block of the kernel with its loop depth,
instruction count, v_readlane / v_writelane count and the source lines it comes from -- enough to see which of the
compiler's spills sit on the path a round actually takes (profiles/r02_k5_spills.txt).
"""
import re, subprocess, sys, tempfile, os

def main():
    src, pat = sys.argv[1], sys.argv[2]
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only",
                           "-gline-tables-only", "-S", "-o", out, src] + sys.argv[3:], stderr=subprocess.DEVNULL)
    L = open(out).read().split("\n")
    start = [i for i, l in enumerate(L) if re.match(r"^_Z\S*%s\S*:" % re.escape(pat), l)][0]
    end = [i for i, l in enumerate(L) if i > start and ".end_amdhsa_kernel" in l][0]
    name = L[start].split(":")[0]
    print("kernel", name)
    for l in L[start:end]:
        if re.match(r"^; (TotalNumSgprs|NumVgprs|ScratchSize|Occupancy)", l): print(" ", l[2:])
    for l in L[end:]:
        pass
    cur, order, info, loc = None, [], {}, 0
    for i in range(start, end):
        l = L[i]
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            loc = int(m.group(2)) if m.group(1) in ("0", "1") else -int(m.group(2)); continue
        m = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):", l)
        if m:
            cur = m.group(1); order.append(cur)
            d = re.search(r"Depth=(\d)", l)
            info[cur] = dict(n=0, sp=0, lines=set(), br=[], depth=d.group(1) if d else "0")
            continue
        if cur is None: continue
        ins = l.strip()
        if not ins or ins[0] in ";." : continue
        b = info[cur]; b["n"] += 1
        if ins.startswith(("v_readlane", "v_writelane")): b["sp"] += 1
        if loc > 0: b["lines"].add(loc)
        m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", ins)
        if m: b["br"].append(m.group(1))
    tot = {}
    for b in order:
        x = info[b]; ls = sorted(x["lines"])
        tot.setdefault(x["depth"], [0, 0]); tot[x["depth"]][0] += x["n"]; tot[x["depth"]][1] += x["sp"]
        if x["sp"] or "-a" in sys.argv:
            print("%-12s depth %s  instrs %3d  lane moves %2d  src %s..%s  -> %s" % (b, x["depth"], x["n"], x["sp"], ls[0] if ls else "-", ls[-1] if ls else "-", ",".join(x["br"])))
    for d in sorted(tot): print("depth %s: %d instructions, %d lane moves" % (d, tot[d][0], tot[d][1]))

if __name__ == "__main__":
    main()
