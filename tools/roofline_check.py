#!/usr/bin/env python3
"""Recompute roofline.frac of a bench line from a rocprofv3 kernel-stats CSV of the same leg and compare (the
two must agree within 3 %).  usage: roofline_check.py <bench.json> <kernel_stats_fa.csv> [<kernel_stats_pileup.csv>]
The CSV comes from `tools/gpu_round.sh prof_fa` (rocprofv3 --kernel-trace --stats over ONLY the one-batch-in-flight leg), the
bench line from an un-traced run of the same build on the same box: HIP events taken UNDER the tracer are ~10 % long."""
import csv
import json
import sys

PEAK = 2500e12
FLOP_FA_CONV = 449_418_240  # algorithmic FLOP of the convolution family per full-alignment window (C = 8)


def stats(path):
    return {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(path))}


def main():
    line = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    fa = stats(sys.argv[2])
    roof = line["roofline"]
    step_ms = line.get("one_batch_in_flight", {}).get("ms_per_step", line["ms_per_step"])  # the short line: value / ms_per_step ARE the one-in-flight leg
    batch = line["config"].get("batch_per_gpu") or line["config"].get("batch") or 256
    conv = {k: v for k, v in fa.items() if "conv3x3_planes_kernel" in k or "conv3x3_wino_planes_kernel" in k or "conv3x3_s2_planes_kernel" in k or "conv1_i8" in k}
    steps = min(c for k, (c, _) in conv.items() if "conv3x3" in k)
    print(f"| kernel (full alignment, B = {batch}, one batch in flight) | launches per step | average us (rocprofv3) |\n|---|---:|---:|")
    total = 0.0
    for k, (c, us) in sorted(conv.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
        per = c / steps
        total += per * us
        print(f"| `{k.replace('void c3::', '').split('(')[0]}` | {per:g} | {us:.1f} |")
    frac = FLOP_FA_CONV * batch / (total * 1e-6) / PEAK
    print(f"\nconvolution launches of one step: {total:.1f} us -> {FLOP_FA_CONV * batch / total / 1e6:.0f} algorithmic TFLOP/s = frac "
          f"**{frac:.4f}** of the 2 500 TFLOP/s 16-bit MFMA peak from the CSV; the bench line (HIP events, no tracer) says "
          f"**{roof['frac']:.4f}** ({roof.get('kernel_us_per_step', float('nan')):.1f} us of kernels per {1e3 * step_ms:.1f} us step, "
          f"mfma_util {roof.get('mfma_util', float('nan')):.3f}): {100 * abs(frac / roof['frac'] - 1):.1f} % apart.")
    whole = sum(c * us for k, (c, us) in fa.items() if k.startswith(("void c3::", "c3::"))) / steps
    print(f"all c3 kernels of one step: {whole:.1f} us (rocprofv3) against the un-traced step of {1e3 * step_ms:.1f} us.")
    # A traced run is a slower run (MI355X_MICROARCH.md, DVFS: profiled passes clock 2 - 3 % lower -- on a fast box more): the like-for-like
    # comparison is the family's SHARE of the step, i.e. the CSV's fraction scaled to the un-traced step.  The raw event-based fraction of the
    # line (every launch bracketed: longer than either) stands on the other side.
    scaled = frac * whole / (1e3 * step_ms)
    ev = (roof.get("events") or {}).get("frac")
    print(f"scaled to the un-traced step (x {whole / (1e3 * step_ms):.3f}): **{scaled:.4f}** -- {100 * abs(scaled / roof['frac'] - 1):.1f} % from the line's share-based "
          f"{roof['frac']:.4f}" + (f"; the line's raw event-based figure: {ev:.4f}" if ev else "") + ".")
    if abs(frac / roof["frac"] - 1) > 0.03 and abs(scaled / roof["frac"] - 1) > 0.03:
        print("MISMATCH > 3 %")
        sys.exit(1)


if __name__ == "__main__":
    main()
