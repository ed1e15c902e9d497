"""The cliff between 19 and 32 calls in flight (bench.py's workloads 3 and 2): B histories of 10k ops / 64 processes at several duty
cycles through the library's default path (a wavefront per history, 4 configs per round), one pass each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, core, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
for busy in (0.3, 0.35, 0.4, 0.45, 0.5):
    hists = synth.register_ops_many(range(40_000_000, 40_000_000 + B), n_ops=10000, n_procs=64, busy=busy, info=0.0)
    with core.Batch(hists, gm, core.make_opts(time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=256 if busy > 0.3 else 32)) as b:
        t = time.perf_counter(); b.run(); dt = time.perf_counter() - t
        v = b.verdicts(); c = b.counters(); tm = b.timing_ns()
        print(f"busy {busy:.2f} (~{64 * busy:.0f} calls in flight): {B} histories in {dt:.2f} s = {B / dt:.0f} histories/s, valid {int((v == 1).sum())} unknown {int((v == -1).sum())}, "
              f"probes per history {c['probes'] // B}, new configs per history {c['visited'] // B}, width {b.search_width()}, search {tm['search'] / 1e6:.0f} ms retries {tm['retries'] / 1e6:.0f} ms", flush=True)
