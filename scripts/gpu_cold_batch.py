"""A cold batch: the bench's 32,768 histories from host columns to verdicts -- tbc_batch_create (TBC_DEBUG=2 prints its phases)
and the first passes, with the device memory the batch holds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
t = time.time()
hists = synth.register_ops_many(range(B), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
print(f"gen {time.time()-t:.2f}s", flush=True)
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
opts = core.make_opts(time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=4)
for rep in range(2):
    t = time.perf_counter()
    b = core.Batch(hists, gm, opts)
    tc = time.perf_counter() - t
    t = time.perf_counter(); b.run(); t1 = time.perf_counter() - t
    print(f"create {tc:.3f} s (tbc_batch_create {b.create_s:.3f} s, the rest is numpy concatenating the histories' columns), first pass {t1 * 1e3:.1f} ms, cold rate {B / (tc + t1):.0f} histories/s ({B / (b.create_s + t1):.0f} from concatenated host columns), device GB {b.device_bytes() / 1e9:.2f}, lanes {b.lanes_per_history()}", flush=True)
    for it in range(3):
        t = time.perf_counter(); b.run(); dt = time.perf_counter() - t
        tm = b.timing_ns()
        print(f"   pass {it + 2}: {dt * 1e3:.1f} ms  { {k: round(x / 1e6, 3) for k, x in tm.items()} } valid {int((b.verdicts() == 1).sum())}", flush=True)
    b.close()
