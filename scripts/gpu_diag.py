"""Staged GPU diagnostic with a watchdog thread printing the library's progress words."""
import ctypes as C, os, sys, threading, time
os.environ["TBC_DEBUG"] = "1"; os.environ["TBC_SYNC_EACH"] = "1"
sys.path.insert(0, ".")
import numpy as np
import jepsen_tigerbeetle_amd
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import wgl

lib = N.lib()
stop = False
def watch():
    buf = (C.c_uint32 * 24)()
    last = None
    while not stop:
        time.sleep(1.0)
        if lib.tbc_debug_peek(buf, 24):
            cur = list(buf)
            if cur != last:
                print("   [dbg] pack=%#x h=%d | search tag=%#x n_ret=%d status=%d W=%d | hidx=%d fi=%d depth=%d steps=%d visited=%d best=%d any=%#x | end verdict=%d steps=%d" % (
                    cur[0], cur[1], cur[4], cur[5], cur[6], cur[7], cur[8], cur[9], cur[10], cur[11], cur[12], cur[13], cur[14] | (cur[15] << 32), C.c_int32(cur[16]).value, cur[17]), flush=True)
                last = cur
threading.Thread(target=watch, daemon=True).start()

print("devices", lib.tbc_device_count(), flush=True)
model = {"kind": 1, "init": N.NIL}
gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
stages = [(8, 3, 0.0, 0.0, 0.5), (8, 3, 0.1, 0.5, 0.8), (40, 4, 0.0, 0.0, 0.5), (200, 8, 0.02, 0.0, 0.5), (200, 8, 0.0, 0.6, 0.3),
          (1000, 16, 0.0, 0.0, 0.5), (1000, 16, 0.0, 0.6, 0.2), (3000, 64, 0.0, 0.0, 0.1), (10000, 64, 0.0, 0.0, 0.1)]
for (n, p, info, corrupt, busy) in stages:
    ops = columns.pair_events(synth.register_events(n_ops=n, n_procs=p, seed=1, busy=busy, info=info, corrupt=corrupt))
    exp = wgl.check(ops.as_dict(), model, "window", max_steps=3_000_000)
    print(f"stage n={n} p={p} info={info} corrupt={corrupt}: ops={len(ops)} W={ops.n_process} oracle valid={exp['valid']} steps={exp['steps']} visited={exp['visited']}", flush=True)
    t = time.time()
    try:
        got = core.check_ops(ops, gm, core.make_opts(time_limit_ms=5000, max_steps=4_000_000))
    except Exception as e:
        print("   EXC", e, flush=True); continue
    dt = time.time() - t
    same = all(got[k] == exp[k] for k in ("valid", "steps", "visited", "backtracks", "max_depth"))
    wit = (exp["valid"] != 1) or np.array_equal(got["witness"], exp["witness"])
    print(f"   gpu valid={got['valid']} cause={got['cause']} steps={got['steps']} visited={got['visited']} fail={got['fail_op']} (oracle {exp['fail_op']}) same={same} witness_same={wit} wall={dt*1e3:.1f}ms search={got['ns_search']/1e6:.2f}ms pack={got['ns_pack']/1e6:.3f}ms", flush=True)
stop = True
