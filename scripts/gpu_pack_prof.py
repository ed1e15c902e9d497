"""Where pack_wg64_kernel's time goes (a TBC_PACK_PROF build: scripts/build_variant.sh): per phase, the 100 MHz ticks the first thread of
every workgroup spent, summed over the batch.  usage: TBC_LIB_PATH=.../libtbcheck_packprof.so python scripts/gpu_pack_prof.py [B]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, core, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
hists = synth.register_ops_many(range(B), n_ops=10000, n_procs=64, busy=0.1, info=0.0)
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
def peek():
    w = (C.c_uint32 * 64)()
    N.lib().tbc_pack_prof_read.restype = C.c_int
    assert N.lib().tbc_pack_prof_read(w) == 1
    return list(w)
with core.Batch(hists, gm, core.make_opts(time_limit_ms=600000, want_witness=False, algorithm=N.ALG_COMPETITION, visited_per_op=4)) as b:
    b.run()
    b.run()
    before = peek()
    b.run()
    after = peek()
    tm = b.timing_ns()
d = [(after[32 + i] - before[32 + i]) & 0xFFFFFFFF for i in range(8)]
tot = sum(d)
names = ["-", "0+1 zero, validate, count", "2 bitmap prefix", "3 list starts", "4 walk", "sentinels + fence", "5 one open op per process", "6 open counts"]
print(f"pack {tm['pack'] / 1e6:.2f} ms; ticks of 10 ns, summed over {B} workgroups' first threads")
for i in range(1, 8):
    print(f"  phase {names[i]:32s} {d[i]:>12d}  {100.0 * d[i] / max(tot, 1):5.1f} %   mean {d[i] * 10 / B / 1e3:8.1f} us per workgroup")
print(f"  total mean {tot * 10 / B / 1e3:.1f} us per workgroup")
