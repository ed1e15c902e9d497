"""Oracle counts behind the race of list orders: probes of wgl_beam.c on N bench histories at a given duty cycle, once per list
order; prints the tail of min-over-the-first-k-orders.  usage: python scripts/order_study.py <busy> <histories> <probe cap>   (profiles/r06_order_study_160_histories.txt: 0.5 160 3e6)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multiprocessing import Pool
busy = float(sys.argv[1]); n = int(sys.argv[2]); cap = int(float(sys.argv[3]))
orders = [16+24, 16+48, 4, 1, 16+8, 0, 16+4, 16+12, 16+16, 16+32, 16+64, 16+128, 2, 3]
def work(i):
    import jepsen_tigerbeetle_amd
    from jepsen_tigerbeetle_amd import _native as N, synth
    from oracle import wgl
    om = {"kind": 1, "init": N.NIL}
    h = synth.register_ops_many([10_000_000 + i], n_ops=10000, n_procs=64, busy=busy, info=0.0)[0].as_dict()
    out = []
    for lo in orders:
        r = wgl.check_beam(h, om, 4, want_witness=False, list_order=lo, max_probes=cap)
        out.append(r["probes"] if r["valid"] != -1 else cap * 2)
    return out
if __name__ == "__main__":
    with Pool(8) as p:
        res = np.array(p.map(work, range(n)))
    pass
    for k in (1, 6, 10, 14):
        m = res[:, :k].min(axis=1)
        print(f"first {k:2d} orders: max of min {m.max():>9d}  p99 {int(np.percentile(m, 99)):>9d}  median {int(np.median(m)):>8d}  unfinished {(m >= cap).sum()}")
    print("per order unfinished:", (res >= cap).sum(axis=0))
