#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes of tools/gpu_round.sh (stages `pmc_fa` / `pmc_p`: --pmc FETCH_SIZE and --pmc WRITE_SIZE in
separate runs over ONLY the one-batch-in-flight leg of one workload, as MI355X_MICROARCH.md prescribes) into
profiles/pmc_traffic.json / pmc_traffic_pileup.json, which bench.py reports as roofline.traffic.  gfx950 correction: FETCH_SIZE
of wide coalesced reads is doubled.    usage: pmc_traffic.py <tag>   |   pmc_traffic.py pileup <tag>"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def l2_hit_rate(tag_dir):
    """TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) over the c3 kernels of the pass tools/gpu_round.sh l2_fa / l2_p collected
    (MI355X_MICROARCH.md, L2 section); None when the pass was not run"""
    path = os.path.join(ROOT, "gpurun_out", tag_dir, "c3_counter_collection.csv")
    if not os.path.exists(path):
        return None
    a = agg(path)
    hit = sum(sum(v.get("TCC_HIT_sum", [])) for k, v in a.items() if k.startswith(("void c3::", "c3::")))
    miss = sum(sum(v.get("TCC_MISS_sum", [])) for k, v in a.items() if k.startswith(("void c3::", "c3::")))
    per = {k[:90]: round(sum(v["TCC_HIT_sum"]) / max(sum(v["TCC_HIT_sum"]) + sum(v["TCC_MISS_sum"]), 1.0), 4)
           for k, v in a.items() if k.startswith(("void c3::", "c3::")) and "TCC_HIT_sum" in v and "TCC_MISS_sum" in v}
    return {"all_c3_kernels": hit / (hit + miss) if hit + miss else None, "per_kernel": per}


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        a[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return a


def main(tag):
    f = agg(os.path.join(ROOT, "gpurun_out/pmc_fetch_fa/c3_counter_collection.csv"))
    w = agg(os.path.join(ROOT, "gpurun_out/pmc_write_fa/c3_counter_collection.csv"))
    conv = lambda k: ("gemm_mfma_kernel<c3::Conv" in k) or ("conv1_i8" in k) or ("conv3x3_planes_kernel" in k) or ("conv3x3_wino_planes_kernel" in k) or ("conv3x3_s2_planes_kernel" in k)
    tot_f = tot_w = n = 0
    per = {}
    all_f = all_w = 0.0
    # forward passes of each pass (the bench's clock-ramp loop is time-based, so the two passes differ): one tail launch per pass
    steps = sum(len(v["FETCH_SIZE"]) for k, v in f.items() if "fc_tail_mfma_kernel<256>" in k or "fc_tail_kernel<256>" in k)
    steps_w = sum(len(v["WRITE_SIZE"]) for k, v in w.items() if "fc_tail_mfma_kernel<256>" in k or "fc_tail_kernel<256>" in k)
    for k in f:
        if not k.startswith(("void c3::", "c3::")):
            continue
        all_f += sum(f[k]["FETCH_SIZE"]) / max(steps, 1)
        all_w += (sum(w[k]["WRITE_SIZE"]) / max(steps_w, 1)) if k in w else 0.0
    for k in f:
        if not conv(k):
            continue
        fs, ws = f[k]["FETCH_SIZE"], w[k]["WRITE_SIZE"]
        # the two passes run different numbers of steps: each is brought to the launches of the FETCH pass before the two are added
        tot_f += sum(fs)
        tot_w += sum(ws) * len(fs) / len(ws)
        n += len(fs)
        per[k[:90]] = {"launches": len(fs), "fetch_kb_avg_reported": sum(fs) / len(fs), "write_kb_avg": sum(ws) / len(ws)}
    B = 256
    a1, a2, a3 = 45 * 17 * 64 * 4, 23 * 9 * 128 * 4, 12 * 5 * 256 * 4
    shapes = [(89 * 33 * 8, a1, 0), (a1, a1, 0), (a1, a1, a1), (a1, a2, 0), (a2, a2, 0), (a2, a2, a2), (a2, a3, 0), (a3, a3, 0),
              (a3, a3, a3)]
    out = {
        "tag": tag,
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around "
                  "`bench.py --gpus 1 --workload full_alignment --streams 1 --no-host-leg --no-cpu-baseline --no-reference-gpu --no-profiled-pass "
                  "--steps 5 --warmup 5 --repeats 3` (B=256, only the one-batch-in-flight leg; tools/gpu_round.sh pmc_fa)",
        "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KB = 1024 B",
        "kernel_family": "the convolution launches of one full-alignment step (8 with conv1 computed inside res1a / res1b, else 9)",
        "launches_per_step": n / steps if steps else None,
        "launches": n,
        "fabric_bytes_per_launch": (2 * tot_f + tot_w) * 1024 / n,
        "fetch_bytes_per_launch_corrected": 2 * tot_f * 1024 / n,
        "write_bytes_per_launch": tot_w * 1024 / n,
        "unfused_layer_bytes_per_step": sum(B * (a + b + c) for a, b, c in shapes),  # every layer reading its inputs and writing its output once
        "steps": steps,
        "fabric_bytes_per_step": (2 * all_f + all_w) * 1024 if steps else None,
        "what_the_counters_count": "FETCH_SIZE / WRITE_SIZE derive from the L2's memory-side request counters (TCC_EA0_RDREQ / _WRREQ): "
                                   "Infinity-Cache hits are counted, not excluded -- fabric bytes, an upper bound of HBM bytes",
        "l2_hit_rate": (l2_hit_rate("pmc_l2_fa") or {}).get("all_c3_kernels"),
        "l2_hit_rate_per_kernel": (l2_hit_rate("pmc_l2_fa") or {}).get("per_kernel"),
        "per_kernel": per,
    }
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print({k: v for k, v in out.items() if k != "per_kernel"})


def main_pileup(tag):
    """the two BiLSTM recurrences of one pileup step (B = 1024): gpurun_out/pmcp_fetch, pmcp_write (stage `pmcp`)"""
    f = agg(os.path.join(ROOT, "gpurun_out/pmc_fetch_p/c3_counter_collection.csv"))
    w = agg(os.path.join(ROOT, "gpurun_out/pmc_write_p/c3_counter_collection.csv"))
    lstm = lambda k: ("lstm1_fused_kernel" in k) or ("lstm_recurrent_kernel" in k)
    tot_f = tot_w = n = 0
    per = {}
    all_f = all_w = 0.0
    # forward passes of each pass: one LSTM1 launch (of whichever variant: half or full tiles) per pass
    steps = sum(len(v["FETCH_SIZE"]) for k, v in f.items() if "lstm1_fused_kernel" in k)
    steps_w = sum(len(v["WRITE_SIZE"]) for k, v in w.items() if "lstm1_fused_kernel" in k)
    for k in f:
        fs, ws = f[k]["FETCH_SIZE"], w.get(k, {}).get("WRITE_SIZE", [0.0])
        per[k[:90]] = {"launches": len(fs), "fetch_kb_avg_reported": sum(fs) / len(fs), "write_kb_avg": sum(ws) / len(ws)}
        if k.startswith(("void c3::", "c3::")):
            all_f += sum(fs) / max(steps, 1)
            all_w += sum(ws) / max(steps_w, 1)
        if lstm(k):
            tot_f += sum(fs)
            tot_w += sum(ws) * len(fs) / len(ws)  # brought to the launches of the FETCH pass
            n += len(fs)
    B, T = 1024, 33
    # LSTM1: int8 windows in, h1 (256 fp32) out; LSTM2: gx2 (1280 fp32) in, h2 (320 fp32) out -- per (window, position)
    algorithmic = B * T * ((18 + 256 * 4) + (1280 * 4 + 320 * 4)) / 2
    out = {
        "tag": tag,
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) around "
                  "`bench.py --gpus 1 --workload pileup --streams 1 --no-host-leg --no-cpu-baseline --no-reference-gpu --no-profiled-pass "
                  "--steps 5 --warmup 5 --repeats 3` (B=1024, only the one-batch-in-flight leg; tools/gpu_round.sh pmc_p)",
        "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KB = 1024 B",
        "kernel_family": "the two BiLSTM recurrence launches of one pileup step (lstm1_fused_kernel + lstm_recurrent_kernel_v2<160>)",
        "launches": n,
        "fabric_bytes_per_launch": (2 * tot_f + tot_w) * 1024 / n,
        "fetch_bytes_per_launch_corrected": 2 * tot_f * 1024 / n,
        "write_bytes_per_launch": tot_w * 1024 / n,
        "algorithmic_bytes_per_launch": algorithmic,
        "steps": steps,
        "fabric_bytes_per_step": (2 * all_f + all_w) * 1024 if steps else None,
        "what_the_counters_count": "FETCH_SIZE / WRITE_SIZE derive from the L2's memory-side request counters (TCC_EA0_RDREQ / _WRREQ): "
                                   "Infinity-Cache hits are counted, not excluded -- fabric bytes, an upper bound of HBM bytes",
        "l2_hit_rate": (l2_hit_rate("pmc_l2_p") or {}).get("all_c3_kernels"),
        "l2_hit_rate_per_kernel": (l2_hit_rate("pmc_l2_p") or {}).get("per_kernel"),
        "per_kernel": per,
    }
    with open(os.path.join(ROOT, "profiles", "pmc_traffic_pileup.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print({k: v for k, v in out.items() if k != "per_kernel"})


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "pileup":
        main_pileup(sys.argv[2])
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else "round1")
