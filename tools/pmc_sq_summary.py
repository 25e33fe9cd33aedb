#!/usr/bin/env python3
"""Summarise the two SQ counter passes of tools/gpu_round.sh `sq_fa` / `sq_p` (gpurun_out/pmc_sq_a_<fa|p>, pmc_sq_b_<fa|p>) per kernel:
MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs)), wave occupancy, wait shares, and the
vector-instruction mix (SQ_INSTS_VALU counts MFMAs too).  usage: pmc_sq_summary.py <fa|p> > profiles/<run>_pmc_sq_<fa|p>.md"""
import collections
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        a[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return a


def mean(d, k):
    v = d.get(k, [])
    return sum(v) / len(v) if v else float("nan")


def main():
    sfx = sys.argv[1] if len(sys.argv) > 1 else "fa"
    A = agg(os.path.join(ROOT, f"gpurun_out/pmc_sq_a_{sfx}/c3_counter_collection.csv"))
    B = agg(os.path.join(ROOT, f"gpurun_out/pmc_sq_b_{sfx}/c3_counter_collection.csv"))
    print("| kernel | launches | us (under PMC) | MfmaUtil % | waves per SIMD | VALU instr per MFMA (excl. the MFMA) | LDS instr per MFMA | VMEM instr per MFMA |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for k, a in A.items():
        if not k.startswith(("void c3::", "c3::")):
            continue
        b = B.get(k, {})
        gui = mean(a, "GRBM_GUI_ACTIVE") / 8
        im = mean(b, "SQ_INSTS_MFMA") if b else float("nan")
        f = lambda name: (mean(b, name) / im) if b and im else float("nan")
        print(f"| `{k[:72]}` | {len(a['GRBM_GUI_ACTIVE'])} | {gui / 2.4e3:.1f} | {100 * mean(a, 'SQ_VALU_MFMA_BUSY_CYCLES') / (gui * 1024):.1f} | "
              f"{mean(a, 'SQ_WAVE_CYCLES') * 4 / (gui * 1024):.2f} | {f('SQ_INSTS_VALU') - 1:.2f} | {f('SQ_INSTS_LDS'):.2f} | {f('SQ_INSTS_VMEM'):.2f} |")


if __name__ == "__main__":
    main()
