#!/bin/bash
# Register / LDS / spill figures of every gfx950 kernel in the built objects, read from the code objects'
# metadata notes (what the hardware dispatcher is told) -- not from compiler remarks.
# usage: scripts/kernel_resources.sh [object ...]   (default: jepsen-tigerbeetle_amd/csrc/build/*.o)
set -e
LLVM=/opt/rocm/lib/llvm/bin
cd "$(dirname "$0")/.."
objs=("$@"); [ ${#objs[@]} -eq 0 ] && objs=(jepsen-tigerbeetle_amd/csrc/build/*.o)
tmp=$(mktemp -d)
for o in "${objs[@]}"; do
  cp "$o" "$tmp/x.o"
  (cd "$tmp" && $LLVM/llvm-objdump --offloading x.o >/dev/null 2>&1) || continue
  co=$(ls "$tmp"/x.o.*gfx950* 2>/dev/null | head -1)
  [ -n "$co" ] || { rm -f "$tmp"/x.o*; continue; }
  $LLVM/llvm-readelf --notes "$co" 2>/dev/null | python3 -c '
import sys, re
txt = sys.stdin.read()
name = sys.argv[1]
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    kn = g("name")
    print("%-16s %-80s vgpr %3s sgpr %3s vgpr_spill %3s sgpr_spill %3s lds %6s scratch %5s" % (name, kn[:80], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
' "$(basename "$o")"
  rm -f "$tmp"/x.o*
done
rm -rf "$tmp"
