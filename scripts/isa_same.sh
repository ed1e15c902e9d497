#!/bin/bash
# isa_same.sh REV_A REV_B -- is the DEVICE code of the sources bench.py's kernel_sha() hashes the same at two revisions?
# (hipcc --cuda-device-only -S of wgl_beam.hip, wgl_narrow.hip and pack_open.hip at each revision, directives and comments dropped,
# diffed).  REV_B may be WORKTREE.  A committed PMC traffic figure (profiles/*_traffic.json) is keyed by a hash of the SOURCES; when an
# edit changes host code only (a launcher's signature) the entry is re-keyed, and this script is the evidence that the kernels the
# figure was measured on are byte for byte the ones that run.  No GPU needed.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
A=$1; B=$2
T=$(mktemp -d)
for rev in "$A" "$B"; do
  d=$T/$rev/jepsen-tigerbeetle_amd/csrc
  mkdir -p "$d" "$T/$rev/include"
  if [ "$rev" = WORKTREE ]; then
    cp "$ROOT"/jepsen-tigerbeetle_amd/csrc/*.h "$ROOT"/jepsen-tigerbeetle_amd/csrc/*.hip "$d"/; cp "$ROOT"/include/*.h "$T/$rev/include/"
  else
    for f in $(git -C "$ROOT" ls-tree --name-only "$rev" jepsen-tigerbeetle_amd/csrc/ include/ | grep -E '\.(h|hip)$'); do git -C "$ROOT" show "$rev:$f" > "$T/$rev/$f"; done
  fi
  for k in wgl_beam wgl_narrow pack_open; do
    (cd "$d" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC --cuda-device-only -S $k.hip -o $k.s 2>/dev/null)
    grep -v -E '^\s*\.(file|ident|loc)|^\s*;' "$d/$k.s" > "$d/$k.clean"
  done
done
rc=0
for k in wgl_beam wgl_narrow pack_open; do
  if diff -q "$T/$A/jepsen-tigerbeetle_amd/csrc/$k.clean" "$T/$B/jepsen-tigerbeetle_amd/csrc/$k.clean" > /dev/null; then echo "$k: device code identical ($(wc -l < "$T/$A/jepsen-tigerbeetle_amd/csrc/$k.clean") lines)"; else echo "$k: DEVICE CODE DIFFERS"; rc=1; fi
done
rm -rf "$T"
exit $rc
