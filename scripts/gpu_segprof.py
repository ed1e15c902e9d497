import ctypes as C, os, sys
os.environ["TBC_DEBUG"] = "1"
sys.path.insert(0, ".")
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
lib = N.lib(); gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
for K in (8, 4, 16):
    for seed in (1, 3):
        ops = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=seed, busy=0.1))
        r = core.check_ops(ops, gm, core.make_opts(search_width=K, algorithm=N.ALG_COMPETITION, want_witness=False))
        buf = (C.c_uint32 * 40)(); lib.tbc_debug_peek(buf, 40)
        seg = [buf[20 + 2 * i] | (buf[21 + 2 * i] << 32) for i in range(5)]
        rounds = buf[30]; tot = sum(seg)
        print(f"K={K} seed={seed} search={r['ns_search']/1e6:.1f}ms rounds={rounds} probes={r['probes']} cycles/round: " +
              " ".join(f"{n}={s/rounds:.0f}" for n, s in zip(("pop", "load", "child", "probe", "push"), seg)) + f" total={tot/rounds:.0f}", flush=True)
