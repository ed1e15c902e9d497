"""Staged GPU diagnostic for the wide (beam) schedule vs oracle/wgl_beam.c."""
import ctypes as C, os, sys, threading, time
os.environ["TBC_DEBUG"] = "1"
sys.path.insert(0, ".")
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import wgl

lib = N.lib()
stop = False
def watch():
    buf = (C.c_uint32 * 24)(); last = None
    while not stop:
        time.sleep(1.0)
        if lib.tbc_debug_peek(buf, 24):
            cur = list(buf)
            if cur != last:
                print("   [dbg] pack=%#x | tag=%#x | hidx=%d iters=%d sp=%d probes=%d visited=%d T=%d np=%d epoch=%d | end verdict=%d iters=%d" % (
                    cur[0], cur[4], cur[8], cur[9], cur[10], cur[11], cur[12], cur[13], cur[14], cur[15], C.c_int32(cur[16]).value, cur[17]), flush=True)
                last = cur
threading.Thread(target=watch, daemon=True).start()
print("devices", lib.tbc_device_count(), flush=True)
model = {"kind": 1, "init": N.NIL}
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
stages = [(8, 3, 0.0, 0.0, 0.5), (8, 3, 0.1, 0.5, 0.8), (40, 4, 0.0, 0.0, 0.5), (200, 8, 0.02, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3),
          (1000, 16, 0.02, 0.0, 0.5), (1000, 16, 0.0, 0.6, 0.2), (3000, 64, 0.02, 0.0, 0.1), (10000, 64, 0.0, 0.0, 0.1), (10000, 64, 0.0, 0.7, 0.05)]
allok = True
for (n, p, info, corrupt, busy) in stages:
    ops = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=1, busy=busy, info=info, corrupt=corrupt))
    exp = wgl.check_beam(ops.as_dict(), model, K, round_pairs=(256 if K > 16 else 64))
    print(f"stage n={n} p={p} info={info} corrupt={corrupt}: ops={len(ops)} W={ops.n_process} oracle valid={exp['valid']} iters={exp['iterations']} rounds={exp['rounds']} probes={exp['probes']} visited={exp['visited']}", flush=True)
    t = time.time()
    try:
        got = core.check_ops(ops, gm, core.make_opts(time_limit_ms=8000, search_width=K, algorithm=N.ALG_COMPETITION))
    except Exception as e:
        print("   EXC", e, flush=True); allok = False; continue
    dt = time.time() - t
    same = got["valid"] == exp["valid"] and got["probes"] == exp["probes"] and got["visited"] == exp["visited"] and got["backtracks"] == exp["expanded"] and got["max_depth"] == exp["max_stack"]
    if exp["valid"] == 0:
        same = same and got["fail_op"] == exp["fail_op"]
    wit = (exp["valid"] != 1) or (got["witness"] is not None and np.array_equal(got["witness"], exp["witness"]))
    allok = allok and same and wit
    print(f"   gpu valid={got['valid']} cause={got['cause']} probes={got['probes']} visited={got['visited']} expanded={got['backtracks']} maxstack={got['max_depth']} fail={got['fail_op']} (oracle {exp['fail_op']}) same={same} witness_same={wit} wall={dt*1e3:.1f}ms search={got['ns_search']/1e6:.2f}ms pack={got['ns_pack']/1e6:.3f}ms", flush=True)
stop = True
print("ALL OK" if allok else "MISMATCHES", flush=True)
