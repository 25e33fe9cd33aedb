"""The ONE line bench.py prints must stay short enough for the driver to parse (round 3's 21 KB line was not) and keep value,
ms_per_step and roofline on the same leg (one batch in flight)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_line_from_a_committed_full_record():
    import bench
    with open(os.path.join(ROOT, "profiles", "r03_final_bench20.json")) as fh:
        full = json.load(fh)
    names = ["full_alignment", "pileup", "full_alignment_dwell"]
    line = bench.short_line(full, names, os.path.join(ROOT, "gpurun_out", "bench_full.json"))
    text = json.dumps(line)
    assert len(text) <= 4096, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "host_inclusive", "gt_concordance"):
        assert key in line, key
    assert line["config"]["batches_in_flight"] == 1
    one = full["one_batch_in_flight"]
    assert abs(line["value"] / one["value"] - 1) < 1e-4 and abs(line["ms_per_step"] / one["ms_per_step"] - 1) < 1e-4
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "mfma_util", "avg_launch_us", "launches"):
        assert key in roof, key
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    # the step the roofline's kernels belong to is the step `value` is computed from
    assert abs(roof["step_us_one_batch_in_flight"] / (1e3 * line["ms_per_step"]) - 1) < 1e-3
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["cores"] >= 1
    assert line["pileup"]["batches_in_flight"] == 1 and "roofline" in line["pileup"]
    assert line["full_record"] == os.path.join("gpurun_out", "bench_full.json")
