"""Where one history's level sweep (K6, jit_sweep.hip) spends its time, on the CPU restatement (oracle/sweep_ref.c): the probes of the most
expensive wavefront ("critical") against the segment length, the number of calls a cut may leave open (sweep_set_max_ids: more origins per
segment than the kernel's 128) and the origins per wavefront (sweep_set_slice).  usage: sweep_cut_study.py [histories]
Round 4, 16 bench histories (10k invocations / 64 processes at 10 % duty), T = 32, m = 4, 32 origins per wavefront: 340 wavefronts,
398k probes per history, the critical wavefront 21.5k of them (max 41.8k) -- 50x the mean.  Cuts at fronts with 5 / 6 calls open (256 /
512 origins), windows of 8 / 16: critical 21k - 29k.  16 .. 1 origins per wavefront: critical 21.4k .. 20.8k while the total grows to 4.3M.
The critical wavefront is ONE burst of concurrency (a level of 240 - 1,440 configs) that no cut divides and that a single origin reaches
whole: neither more segments nor fewer origins per wavefront shorten it; only more lanes on that level's sub-rounds do."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jepsen_tigerbeetle_amd  # noqa: F401
from jepsen_tigerbeetle_amd import _native as N, synth
from oracle import wgl

nh = int(sys.argv[1]) if len(sys.argv) > 1 else 16
om = {"kind": 1, "init": N.NIL}
hs = [h.as_dict() for h in synth.register_ops_many(range(nh), n_ops=10000, n_procs=64, busy=0.1, info=0.0)]
L = wgl.lib()


def row(tag, T, m):
    rs = [wgl.check_sweep(h, om, seg_target=T, max_cut_open=m, n_dom=6) for h in hs]
    assert all(r["valid"] == 1 for r in rs)
    print(tag, "| wavefronts", int(np.mean([r["n_waves"] for r in rs])), "longest segment", max(r["longest_segment"] for r in rs),
          "probes", int(np.mean([r["probes"] for r in rs])), "critical: mean", int(np.mean([r["max_segment_probes"] for r in rs])),
          "max", max(r["max_segment_probes"] for r in rs), "| widest level", max(r["max_level"] for r in rs), flush=True)


for m, ids in ((4, 128), (5, 256), (6, 512)):
    L.sweep_set_max_ids(C.c_uint32(ids))
    for T in (8, 16, 32):
        row(f"cut at <= {m} open, window {T}", T, m)
L.sweep_set_max_ids(C.c_uint32(128))
for G in (32, 16, 8, 4, 2, 1):
    L.sweep_set_slice(C.c_uint32(G))
    row(f"{G} origins per wavefront", 32, 4)
L.sweep_set_slice(C.c_uint32(32))
