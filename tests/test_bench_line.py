"""bench.py's last stdout line must stay small: the driver keeps a 13.7 KB tail of stdout and parses the LAST line (round 4's line
outgrew the tail and the round went unmeasured).  No GPU: the full object is a worst-case payload shaped like profiles/r04_bench_final.json.log."""
import io
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LONG = "x" * 400


def worst_case_line():
    roof = {"bound": "hbm", "achieved": 123456.789, "peak": 8000.0, "unit": "GB/s", "frac": 0.123456, "traffic": 141223833600, "kernel": "wgl_narrow_kernel",
            "kernel_ms": 117.456, "probes_per_launch": 349245319, "new_configs_per_launch": 326675180, "algorithmic_bytes_per_launch": 10814727984, "note": LONG}
    wl = {"workload": LONG, "histories_per_gpu": 8192, "search_width": 4, "lanes_per_history": 64, "value": 123456.78, "unit": "histories/s", "ms_per_step": 25999.718,
          "valid": 8192, "unknown": 0, "roofline": dict(roof), "device_ms": {"init": 1.0, "pack": 2.0, "search": 3.0, "retries": 0.1, "turn_wait": 0.0},
          "cpu_baseline": {"value": 27.992, "unit": "histories/s", "cores": 16, "kind": "port", "sample": LONG}}
    forms = [{"form": LONG, "env": {"TBC_X": "1"}, "device_ms": {"search": 1.0}, "note": LONG} for _ in range(40)]
    return {
        "metric": "histories/sec, 10k-op/64-proc cas-register histories (time-to-verdict ms in extra)", "value": 270431.92, "unit": "histories/s", "n_gpus": 8,
        "steps": 32, "warmup": 2, "ms_per_step": 121.169, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": bench.workload_name(10000, 64, 0.1, 0.0), "histories_per_gpu": 32768, "ops_after_pairing": 7332, "processes": 64, "busy": 0.1, "info_rate": 0.0,
                   "batches_in_flight": 2, "search_width": 1, "search_width_asked": 0, "lanes_per_history": 8, "histories_per_wavefront": 8, "round_budget": 0,
                   "parallelism": "independent histories sharded over 8 GPU(s), no collective", "note": LONG},
        "roofline": roof,
        "cpu_baseline": {"value": 771.681, "unit": "histories/s", "cores": 16, "kind": "port", "sample": LONG, "single_thread": {"value": 55.682, "unit": "histories/s", "cores": 1,
                         "ms_per_history": 17.959, "sample": LONG}, "same_schedule_as_kernel": {"value": 223.069, "sample": LONG, "all_cores": {"value": 3897.896, "sample": LONG}},
                         "level_sweep_on_cpu": {"value": 66.7, "sample": LONG}},
        "extra": {"valid": 65534, "unknown": 0, "device_ms": {"init_memsets": 3.697, "pack": 66.201, "search": 117.456, "retries": 0.2, "search_waiting_for_its_turn": 9.506, "note": LONG},
                  "device_GB": 169.734, "device_GB_per_batch": 84.891, "h2d_inclusive_hist_per_s": 88330.93,
                  "fresh_input": {"histories_per_s": 287654.32, "unit": "histories/s", "steps": 4, "ms_per_step": 113.912, "what": LONG, "pcie_GB_per_input": 2.883,
                                  "pcie_GB_s_over_the_timed_region": 25.31, "pcie_GB_s_while_copying": 52.77, "pcie_peak_GB_s": 63.0,
                                  "re_uploaded": {"histories_per_s": 291234.56, "steps": 12, "ms_per_step": 112.5, "pcie_GB_s_over_the_timed_region": 25.6,
                                                  "pcie_GB_s_while_copying": 52.9, "what": LONG},
                                  "lists_regrown": 0, "host_gen_and_encode_s": 40.12, "host_encode_s_per_batch": 1.23, "device_GB_per_batch_after": 96.5},
                  "one_batch_at_a_time": {"value": 204793.11, "device_ms": {"init_memsets": 0.7, "pack": 59.23, "search": 98.417}, "roofline_frac": 0.013736, "note": LONG},
                  "same_batch_one_history_per_wavefront": dict(wl),
                  "time_to_verdict_ms": {"valid_median": 2.499, "valid_min": 2.023, "answered_by_sweep": 10, "of": 10, "depth_first_with_witness_median": 22.96,
                                         "invalid_example": 3.029, "vs_cpu_port_single_thread": 7.19, "vs_cpu_same_schedule_single_thread": 2.8,
                                         "split_us": {"h2d": 100.0, "pack": 280.0, "cuts": 10.0, "sweep": 920.0, "d2h": 50.0, "host": 100.0}, "note": LONG},
                  "tiers": [{"info_rate": 0.05, "history": "1 bad read", "process_slots": 552, "gpu_ms": 1980.123, "gpu_verdict": 0, "gpu_analyzer": "wgl", "cpu_port_ms": 13000.123,
                             "cpu_verdict": -1, "cpu_same_algorithm_ms": 208.123, "cpu_same_algorithm_passes": LONG} for _ in range(6)],
                  "workload_2": dict(wl), "workload_3": dict(wl), "workload_crashed": dict(wl),
                  "set_full": {"elements": 262144, "reads": 32768, "scan_ms": 0.415, "end_to_end_ms": 4.169, "end_to_end_note": LONG, "roofline": dict(roof)},
                  "single_history_forms": forms, "batch_forms": forms, "one_history_over_all_gpus": {"median_ms": 1.5, "valid": 1, "gpus": 8}},
    }


def test_compact_line_is_small_and_complete():
    small = bench.compact_line(worst_case_line())
    text = json.dumps(small, separators=(",", ":"))
    assert len(text) < 4096, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "extra"):
        assert k in small, k
    assert small["config"]["workload"].startswith("cas-register") and "model" not in small["config"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"} <= set(small["roofline"])
    assert {"value", "unit", "cores", "kind", "sample", "single_thread", "same_schedule"} <= set(small["cpu_baseline"])
    assert len(small["extra"]["tiers"]["rows"]) == 6 and "batch_forms" not in small["extra"] and "single_history_forms" not in small["extra"]
    assert small["extra"]["time_to_verdict_ms"]["vs_cpu_same_schedule_single_thread"] == 2.8
    assert small["extra"]["workload_2"]["value"] == 123456.78
    assert small["extra"]["fresh_input"]["histories_per_s"] == 287654.32 and small["extra"]["fresh_input"]["re_uploaded"]["steps"] == 12


def test_emit_prints_the_compact_line_last(monkeypatch, tmp_path, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(worst_case_line())
    cap = capsys.readouterr()
    last = cap.out.splitlines()[-1]
    assert len(last) < 8192
    d = json.loads(last)
    assert "roofline" in d and "cpu_baseline" in d
    full = json.loads(open(os.path.join(str(tmp_path), "gpurun_out", "bench_full.json")).read())
    assert len(full["extra"]["batch_forms"]) == 40       # nothing is lost: the full object is on file (and on stderr)
    assert cap.err.startswith("[bench full] ")


def test_a_leg_that_died_still_fits():
    line = worst_case_line()
    for w in ("workload_2", "workload_3", "workload_crashed", "set_full"):
        line["extra"][w] = {"error": LONG}
    line["extra"]["tiers"] = {"error": "the leg's process was killed by signal 6"}
    text = json.dumps(bench.compact_line(line), separators=(",", ":"))
    assert len(text) < 4096
