"""Count form on the device, quick look: the crashed tiers through tbc_check, timed, beside the oracle's pipeline."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jepsen_tigerbeetle_amd  # noqa
from jepsen_tigerbeetle_amd import _native as N, columns, core, synth
from oracle import wgl
def oracle_pipeline(w, ops, width):
    return w.check_count_pipeline(ops, om, width=width)

gm = core.make_model(N.MODEL_CAS_REGISTER, N.NIL)
om = {"kind": 1, "init": N.NIL}
for info in (0.01, 0.05):
    for corrupt in (0.0, 0.5):
        h = columns.pair_events(synth.register_events(n_ops=10000, n_procs=64, seed=4242, busy=0.1, info=info, corrupt=corrupt))
        for cf in (True, False):
            o = core.make_opts(time_limit_ms=3000, algorithm=N.ALG_COMPETITION, want_witness=False, count_form=cf)
            core.check_ops(h, gm, o)
            t = time.perf_counter(); g = core.check_ops(h, gm, o); t = (time.perf_counter() - t) * 1e3
            print(f"info {info} corrupt {corrupt} count_form {cf}: valid {g['valid']} fail {g['fail_op']} width {g['search_width']} probes {g['probes']} visited {g['visited']} "
                  f"{t:.2f} ms (pack {g['ns_pack'] / 1e6:.2f} search {g['ns_search'] / 1e6:.2f})", flush=True)
        t = time.perf_counter(); v, fo, last, tot, how = oracle_pipeline(wgl, h.as_dict(), 4); t = (time.perf_counter() - t) * 1e3
        print(f"   oracle pipeline (width 4): valid {v} fail {fo} {how} probes {tot['probes']} visited {tot['visited']} {t:.1f} ms")
        t = time.perf_counter(); e = wgl.check(h.as_dict(), om, "window", want_witness=False, max_steps=20_000_000); t = (time.perf_counter() - t) * 1e3
        print(f"   plain port: valid {e['valid']} {t:.1f} ms", flush=True)
