"""Two batches in flight on one GPU, each driven by its own host thread on its own stream: does one batch's pack beside the
other's search (the library chains the searches, tbc_api.hip SearchTurn) beat running them back to back?
usage: gpu_two_in_flight.py <histories per batch> <waves per SIMD of the narrow kernel, comma list; 0 = the build's> [runs]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
WPS = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 3]
RUNS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 2
t = time.time()
hists = [synth.register_ops_many(range(k * B, (k + 1) * B), n_ops=10000, n_procs=64, busy=0.1, info=0.0) for k in range(NB)]
print(f"gen {time.time()-t:.2f}s", flush=True)
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
opts = core.make_opts(time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=4, lanes_per_history=8)
batches = [core.Batch(h, gm, opts) for h in hists]
print("device GB", [round(b.device_bytes() / 1e9, 1) for b in batches], flush=True)
for b in batches:
    b.run()
ref = [b.verdicts().copy() for b in batches]
for w in WPS:
    os.environ["TBC_NARROW_WAVES_PER_SIMD"] = str(w)
    for it in range(2):
        t = time.time(); batches[0].run(); dt = time.time() - t
        tm = batches[0].timing_ns()
        print(f"wps {w} alone: {dt*1e3:.1f} ms  hist/s={B/dt:.0f}  ms={ {k: round(x/1e6,1) for k,x in tm.items()} }", flush=True)
    log = [[] for _ in batches]
    go = threading.Barrier(len(batches) + 1)
    def work(i):
        go.wait()
        for r in range(RUNS):
            t0 = time.time(); batches[i].run(); t1 = time.time()
            log[i].append((t0, t1, {k: round(x / 1e6, 1) for k, x in batches[i].timing_ns().items()}))
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(batches))]
    for x in th: x.start()
    go.wait(); t = time.time()
    for x in th: x.join()
    dt = time.time() - t
    print(f"wps {w} {len(batches)} in flight: {len(batches)*RUNS} runs in {dt*1e3:.1f} ms  hist/s={len(batches)*RUNS*B/dt:.0f}", flush=True)
    for i, l in enumerate(log):
        for t0, t1, tm in l:
            print(f"    batch {i}: {1e3*(t0-t):7.1f} .. {1e3*(t1-t):7.1f}  {tm}", flush=True)
    for b, r in zip(batches, ref):
        assert np.array_equal(b.verdicts(), r), "verdicts changed"
