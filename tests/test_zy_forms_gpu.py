"""The single-history path under each of its experimental forms, ON THE DEVICE: the level sweep's ring / fingerprint / sixteen-wavefront
forms (jit_sweep_wg.hip) and pack by a workgroup's sixteen wavefronts (pack_one.hip) -- and the batch form of that pack (four
wavefronts per history, pack + open counts in one pass; TBC_PACK_WG=1) under the batch parity tests.  Each form is selected by an environment switch
the library reads once per process, so each runs the sweep's own GPU tests (tests/test_sweep.py: every record against
oracle/sweep_ref.c) -- and, for the pack, the histories pack must refuse -- in a process of its own.

STANDING, and why these are xfail(strict=False): the forms are verified under the wavefront emulator (tests/test_sweep_wg_emu.py,
tests/test_pack_one_emu.py) and had no GPU minutes left to run on when they were committed; nothing takes them by default.  A form
that passes here reports XPASS and may become a default; one that fails reports xfail and stays a switch.  (Last but one in
collection order: nothing else waits behind them.)"""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

# (the ring and the fingerprint forms one by one: bench.py's extra.single_history_forms compares their counters with the default's)
FORMS = [("compact+solo+fingerprint", {"TBC_SWEEP_WG_COMPACT": "2", "TBC_SWEEP_WG_FP": "1"}),
         ("pack-one+counts+compact+solo+fingerprint", {"TBC_PACK_ONE": "2", "TBC_SWEEP_WG_COMPACT": "2", "TBC_SWEEP_WG_FP": "1"}),
         ("pack-wg", {"TBC_PACK_WG": "1"}), ("pack-wg-or-error", {"TBC_PACK_WG": "2"}), ("lean-tables", {"TBC_NARROW_LEAN": "1"}),
         ("lists-by-completion+lean+lazy-lookahead", {"TBC_NARROW_ORDER": "1", "TBC_NARROW_LEAN": "2"}),
         ("lists-by-completion-a-write-24-ranks-later", {"TBC_NARROW_ORDER": "40"})]       # (16 + W; 2 = writes last: the same code path)
# (the ring, the fingerprint alone, the compact walk alone, sixteen wavefronts, pack-one alone: timed and compared counter by counter by
# bench.py's extra.single_history_forms, not run through the test files here -- the GPU tier's minutes are the driver's)


# The GPU tier's minutes are the driver's: all the forms together get BUDGET_S of it (a form that hangs on the device costs its own
# process's timeout once; the forms after the budget is spent are skipped, not failed), so the tier ends in bounded time whatever they do.
BUDGET_S = float(os.environ.get("TBC_TEST_FORMS_BUDGET_S", "420"))
PER_FORM_S = 180
_spent = [0.0]


@pytest.mark.xfail(strict=False, reason="experimental form: emulator-verified, not yet run on the device when committed")
@pytest.mark.parametrize("name,env", FORMS, ids=[f[0] for f in FORMS])
def test_form_passes_the_sweeps_own_gpu_tests(native, name, env):
    if _spent[0] > BUDGET_S:
        pytest.skip(f"the forms' {BUDGET_S:.0f} s of the GPU tier were spent (TBC_TEST_FORMS_BUDGET_S)")
    t0 = time.time()
    targets = ["tests/test_sweep.py"]
    if "TBC_PACK_ONE" in env:        # what pack refuses, and one history through every engine
        targets += ["tests/test_gpu_parity.py::test_rejects_malformed_ops", "tests/test_gpu_parity.py::test_mutex_and_table_models",
                    "tests/test_gpu_parity.py::test_single_history_matches_oracle", "tests/test_gpu_parity.py::test_kat_through_knossos_surface"]
    if "TBC_PACK_WG" in env:         # the batch form of the pack (four wavefronts per history, pack + open counts): the batches of the wide and the
        # narrow schedule against their oracles (every counter depends on every list and count the pack leaves), what pack refuses
        targets = ["tests/test_gpu_parity.py::test_narrow_kernel_matches_its_oracle", "tests/test_gpu_parity.py::test_narrow_kernel_rules_lookahead_growth_and_limits",
                   "tests/test_gpu_parity.py::test_batch_matches_oracle_and_single", "tests/test_gpu_parity.py::test_rejects_malformed_ops",
                   "tests/test_gpu_parity.py::test_lookahead_value_range_crashed_writers_and_plain_register", "tests/test_gpu_parity.py::test_wide_window_many_crashed_processes",
                   "tests/test_gpu_parity.py::test_front_walk_by_front_and_by_slot_build_the_same_tables",      # (the bench's own shape: bench.py extra.batch_forms)
                   "tests/test_count_form_gpu.py::test_count_form_several_histories_per_wavefront", "tests/test_count_form_gpu.py::test_count_form_agrees_with_the_mask_form"]
    if "TBC_NARROW_LEAN" in env or "TBC_NARROW_ORDER" in env:     # the lean formats of the per-front lists and the lookahead records under the narrow kernel: its own file
        targets = ["tests/test_lean_gpu.py"]         # (the schedule is the oracle's look_two, not the default's: the other files would compare with the wrong one)
    if env.get("TBC_PACK_WG") == "2":        # (2: a batch that does not take the workgroup pack is an error -- these all fit, so they ran it)
        targets = ["tests/test_gpu_parity.py::test_narrow_kernel_matches_its_oracle", "tests/test_gpu_parity.py::test_big_quiet_batches_take_the_narrow_kernel_by_default"]
    try:
        r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + targets,
                           cwd=ROOT, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=PER_FORM_S)
    finally:
        _spent[0] += time.time() - t0
    tail = r.stdout.decode(errors="replace")[-1500:]
    assert r.returncode == 0, f"{name}: {tail}"
