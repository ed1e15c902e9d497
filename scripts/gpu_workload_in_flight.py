"""workload 3 / 2 with TWO batch objects in flight (each on its own host thread, as the headline's): one batch's stragglers -- the race of
list orders runs a few hundred wavefronts for as long as its hardest history takes -- beside the other batch's first pass."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jepsen_tigerbeetle_amd import _native as N, core, synth

busy = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
vpo = int(sys.argv[3]) if len(sys.argv) > 3 else 32
F = int(sys.argv[4]) if len(sys.argv) > 4 else 2
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
model = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
opts = core.make_opts(time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=vpo)
sets = [synth.register_ops_many(range(20_000_000 + k * B, 20_000_000 + (k + 1) * B), n_ops=10000, n_procs=64, busy=busy, info=0.0) for k in range(F)]
batches = [core.Batch(h, model, opts) for h in sets]
print("device GB per batch", round(batches[0].device_bytes() / 1e9, 1), flush=True)
for b in batches: b.run()          # warm-up
def loop(b):
    for _ in range(steps): b.run()
t = time.perf_counter()
th = [threading.Thread(target=loop, args=(b,)) for b in batches]
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t
print(f"busy {busy} B {B} x {F} in flight, {steps} steps each: {F * steps * B / dt:.0f} histories/s ({dt / (F * steps) * 1e3:.0f} ms a step); unknown {sum(int((b.verdicts() == N.UNKNOWN).sum()) for b in batches)} raced {[b.last_raced() for b in batches]}", flush=True)
t = time.perf_counter()
for _ in range(steps): batches[0].run()
dt = time.perf_counter() - t
print(f"one batch at a time: {steps * B / dt:.0f} histories/s ({dt / steps * 1e3:.0f} ms a step)", flush=True)
for b in batches: b.close() if hasattr(b, "close") else None
