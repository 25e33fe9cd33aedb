#!/usr/bin/env python3
"""One-command parity check against a TRAINED checkpoint (BASELINE.json configs[0]: the ONT quick demo, HG003 chr20;
SURVEY 8c/8d "Config 1").  Nothing in this offline image can run it -- there is no BAM, FASTA or model directory -- so the
tool exists for the day the data does:

  1. run the reference pipeline once with `--use_gpu`-style tensor dumping so that stage A leaves its `.npy` / `.info`
     tensor files on disk (docs/quick_demo/ont_quick_demo.md:24-34; the list is `--output_tensor_can_fn_list`,
     clair3/CallVariantsFromCffi.py:106-133);
  2. `python tools/replay_demo.py --reference /path/to/Clair3 --model-dir /path/to/r1041_e82_400bps_sup_v500 \\
         --list /path/to/tensor_can_fn_list --pileup` (or `--full-alignment [--enable-dwell-time]`)

It replays the SAME tensors through both backends --
  (a) the reference's own modules on CPU (clair3/model.py imported from --reference, `_load_torch_checkpoint` +
      `_torch_predict`, clair3/CallVariantsFromCffi.py:19-28,48-52), and
  (b) libc3hip through clair3_amd (the drop-in path of INTEGRATION.md) --
and reports: max |dY| against the 1e-4 gate, arg-max concordance of the gt21 / zygosity (/ indel-length) heads with a
near-tie report, the VCF rows the reference's unmodified batch_output prints from either set of rows (diffed line by line),
and -- the one thing seeded-random weights could not tell us -- the per-layer activation maxima of the trained model next
to the range guard of the fp16x3 kernels (16 000; fp16 max 65 504).  Exit code 0 = identical calls and rows within the gate.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HEADS = (("gt21", 0, 21), ("zygosity", 21, 24), ("indel_1", 24, 57), ("indel_2", 57, 90))
GUARD = 16000.0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="a Clair3 checkout (the directory holding clair3/ and shared/)")
    ap.add_argument("--model-dir", required=True, help="directory with pileup.pt / full_alignment.pt")
    ap.add_argument("--list", required=True, help="tensor file list written by stage A (--output_tensor_can_fn_list)")
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--pileup", action="store_true")
    g.add_argument("--full-alignment", action="store_true")
    ap.add_argument("--enable-dwell-time", action="store_true")
    ap.add_argument("--batch", type=int, default=1000)
    ap.add_argument("--max-windows", type=int, default=200000, help="bound on the windows replayed through the CPU reference")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--vcf", action="store_true", help="also run the reference's batch_output on both row sets and diff the text")
    args = ap.parse_args(argv)

    sys.path.insert(0, args.reference)
    import torch
    from clair3.CallVariantsFromCffi import _load_torch_checkpoint, _torch_predict  # the reference's own loader / caller
    from clair3.model import Clair3_F, Clair3_P
    from clair3_amd import model as hip_model, predict as hip_predict, worker

    pileup = args.pileup
    indel = not pileup  # scripts/clair3_c_impl_pipeline.py: --add_indel_length only for the full-alignment step
    channels = 18 if pileup else (9 if args.enable_dwell_time else 8)
    ckpt = os.path.join(args.model_dir, "pileup" if pileup else "full_alignment")

    # (a) reference modules, CPU
    ref = (Clair3_P if pileup else Clair3_F)(add_indel_length=indel, predict=True, input_channels=channels)
    _load_torch_checkpoint(ref, ckpt, torch.device("cpu"))
    ref.eval()
    # (b) HIP path through the mirror of the same interface
    hip = (hip_model.Clair3_P if pileup else hip_model.Clair3_F)(add_indel_length=indel, predict=True, input_channels=channels)
    hip.keep_activations(True)
    hip.to(args.device)
    hip_predict._load_torch_checkpoint(hip, ckpt, "cpu")

    y_ref, y_hip, positions, alt_infos = [], [], [], []
    act_max = {}
    layers = ["lstm1_out", "lstm2_out", "l4_out"] if pileup else [f"act{l}" for l in range(9)] + ["spp", "l4_out"]
    seen = 0
    for X, pos, alt in worker.iter_batches(args.list, args.batch):
        X = np.ascontiguousarray(X)
        y_ref.append(_torch_predict(ref, torch.device("cpu"), X))
        y_hip.append(hip_predict._hip_predict(hip, args.device, X))
        positions.extend(pos)
        alt_infos.extend(alt)
        for name in layers:  # activation maxima of the last micro-batch of this call
            try:
                n = len(X)
                shape = None
                if pileup:
                    shape = {"lstm1_out": (n, 33, 256), "lstm2_out": (n, 33, 320), "l4_out": (n, 128)}[name]
                elif name.startswith("act"):
                    l = int(name[3])
                    h, w = X.shape[1], X.shape[2]
                    for s in (2, 1, 1, 2, 1, 1, 2, 1, 1)[: l + 1]:
                        h, w = (h - 1) // s + 1, (w - 1) // s + 1
                    shape = (n, h, w, (64, 64, 64, 128, 128, 128, 256, 256, 256)[l])
                else:
                    shape = {"spp": (n, 3584), "l4_out": (n, 256)}[name]
                a = hip.debug_fetch(name, shape)
                act_max[name] = max(act_max.get(name, 0.0), float(np.abs(a).max()))
            except Exception as e:  # the maxima are a report, not the gate
                act_max.setdefault(name, float("nan"))
                print(f"[replay] could not fetch {name}: {e}", file=sys.stderr)
        seen += len(X)
        if seen >= args.max_windows:
            break
    y_ref, y_hip = np.concatenate(y_ref), np.concatenate(y_hip)

    ok = True
    err = float(np.abs(y_ref.astype(np.float64) - y_hip).max())
    print(f"windows replayed: {len(y_ref)}   max |dY| = {err:.3e}  (gate 1e-4)")
    ok &= err <= 1e-4
    for name, lo, hi in HEADS:
        if lo >= y_ref.shape[1]:
            break
        a, b = y_hip[:, lo:hi].argmax(1), y_ref[:, lo:hi].argmax(1)
        top2 = np.sort(y_ref[:, lo:hi], axis=1)[:, -2:]
        diff = a != b
        hard = diff & ((top2[:, 1] - top2[:, 0]) > 1e-5)
        print(f"  {name:9s} identical arg-max {int((~diff).sum())} / {len(a)};  differ {int(diff.sum())}, of which outside near-ties (1e-5): {int(hard.sum())}")
        ok &= not hard.any()
    flag, on_fp32 = hip.range_status()
    print("per-layer activation maxima of this checkpoint (fp16x3 range guard at %.0f, fp16 max 65504):" % GUARD)
    for name in layers:
        v = act_max.get(name, float("nan"))
        print(f"  {name:10s} {v:12.4f}  {'<-- beyond the guard: the handle runs on fp32 matrix instructions' if v >= GUARD else ''}")
    print(f"range flag raised: {bool(flag)}   handle on the fp32 fallback: {on_fp32}")

    if args.vcf:
        import clair3.CallVariants as cv
        if pileup:
            import shared.param_p as param
        else:
            import shared.param_f as param
        cv.param = param
        cfg = cv.OutputConfig(
            is_show_reference=True, is_debug=False, is_haploid_precise_mode_enabled=False,
            is_haploid_sensitive_mode_enabled=False, is_output_for_ensemble=False, quality_score_for_pass=None,
            tensor_fn=None, input_probabilities=False, add_indel_length=indel, gvcf=False, pileup=pileup,
            enable_long_indel=False, maximum_variant_length_that_need_infer=param.maximum_variant_length_that_need_infer,
            keep_iupac_bases=False)
        n = len(y_ref)
        rows_ref = cv.batch_output(positions[:n], alt_infos[:n], y_ref, cfg, None)
        rows_hip = cv.batch_output(positions[:n], alt_infos[:n], y_hip, cfg, None)
        differ = [(a, b) for a, b in zip(rows_ref, rows_hip) if a != b]
        gt_differ = [(a, b) for a, b in differ if a.split("\t")[9].split(":")[0] != b.split("\t")[9].split(":")[0]] if differ else []
        print(f"VCF rows: {len(rows_ref)} printed, {len(differ)} differ as text (QUAL is printed with %.2f), {len(gt_differ)} differ in GT")
        for a, b in differ[:10]:
            print("   ref:", a.strip()[:200])
            print("   hip:", b.strip()[:200])
        ok &= not gt_differ
    print("RESULT:", "identical calls, rows within the gate" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
