"""Timing probe of the checker/set-full scan on the bench's synthetic matrix (see bench.py, extra.set_full)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N
from jepsen_tigerbeetle_amd.jepsen import set_full as sf
rng = np.random.default_rng(7)
E, R = int(sys.argv[1]) if len(sys.argv) > 1 else 262144, int(sys.argv[2]) if len(sys.argv) > 2 else 32768
add_invoke = (np.sort(rng.choice(4 * (E + R), E, replace=False)) * 2).astype(np.uint32)
read_invoke = (np.sort(rng.choice(4 * (E + R), R, replace=False)) * 2 + 1).astype(np.uint32)
read_ok = read_invoke + (rng.integers(1, 2000, R) * 2).astype(np.uint32)
add_ok = (add_invoke + 2001).astype(np.uint32)
p = np.searchsorted(add_ok, read_invoke).astype(np.int64)
w = np.arange(E // 32, dtype=np.int64)
M = np.where(32 * (w + 1)[None, :] <= p[:, None], 0xFFFFFFFF, np.where(32 * w[None, :] >= p[:, None], 0, (1 << np.clip(p[:, None] - 32 * w[None, :], 0, 31)) - 1)).astype(np.uint32)
for e in rng.choice(E // 2, 300, replace=False):
    M[R * 3 // 4:, e // 32] &= np.uint32(~(1 << (e % 32)) & 0xFFFFFFFF)
class A: pass
a = A(); a.E, a.R, a.wpr = E, R, E // 32
a.add_invoke, a.add_ok, a.read_invoke, a.read_ok, a.present = add_invoke, add_ok, read_invoke, read_ok, np.ascontiguousarray(M)
with sf.Scan(a) as sc:
    sc.run()
    runs = [sc.run() for _ in range(5)]
ms = statistics.mean(r["ns_scan"] for r in runs) / 1e6
print(f"E={E} R={R} matrix {runs[0]['bytes_matrix']/1e9:.3f} GB scanned {runs[0]['bytes_scanned']/1e9:.3f} GB  scan {ms:.3f} ms  {runs[0]['bytes_scanned']/ms/1e6:.1f} GB/s")
