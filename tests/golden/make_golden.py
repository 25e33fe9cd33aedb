#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference modules.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

For every case it rebuilds the seeded synthetic state_dict / windows (clair3_amd/synthetic.py), loads the
state_dict *strictly* into the reference ``clair3.model.Clair3_P`` / ``Clair3_F`` (so key names and shapes are
pinned), runs the reference forward exactly like ``_torch_predict`` (clair3/CallVariantsFromCffi.py:48-52) on
the CPU with one thread, and stores only what cannot be regenerated without the reference: the float32
probability rows ``Y_ref`` (+ digests of the inputs so recipe drift is caught).  For the decode goldens it also
runs the reference ``batch_output`` (clair3/CallVariants.py:1069) on ``Y_ref`` with synthetic .info strings and
stores the VCF rows.
"""
import hashlib
import json
import os
import sys

import numpy as np

REF = os.environ.get("CLAIR3_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import torch  # noqa: E402

from clair3_amd import synthetic as syn  # noqa: E402

# name, kind, channels, add_indel_length, weight seed, peaked, input seed, recipe, batch, x dtype
CASES = [
    ("pileup_realistic", syn.PILEUP, 18, False, 0, False, 0, "realistic", 48, "int8"),
    ("pileup_uniform_i32", syn.PILEUP, 18, False, 1, False, 1, "uniform", 32, "int32"),
    ("pileup_peaked", syn.PILEUP, 18, False, 2, True, 2, "realistic", 32, "int8"),
    ("pileup_indel_heads", syn.PILEUP, 18, True, 3, False, 3, "realistic", 16, "int8"),
    ("fa_realistic", syn.FULL_ALIGNMENT, 8, True, 0, False, 0, "realistic", 24, "int8"),
    ("fa_uniform", syn.FULL_ALIGNMENT, 8, True, 1, False, 1, "uniform", 8, "int8"),
    ("fa_peaked", syn.FULL_ALIGNMENT, 8, True, 2, True, 2, "realistic", 16, "int8"),
    ("fa_dwell", syn.FULL_ALIGNMENT, 9, True, 3, False, 3, "realistic", 16, "int8"),
    ("fa_no_indel_heads", syn.FULL_ALIGNMENT, 8, False, 4, False, 4, "realistic", 8, "int8"),
    ("fa_hifi_depth55", syn.FULL_ALIGNMENT, 8, True, 5, False, 5, "realistic", 12, "int8"),
    # weights re-parametrised the way training leaves them (clair3_amd/synthetic.py _trained_like): per-channel BatchNorm
    # gamma / sigma over 1e-3 .. 1e3, channel magnitudes over 2.8 decades, zero bias_hh, a few +-8 LSTM weights
    ("fa_trained_like", syn.FULL_ALIGNMENT, 8, True, 6, True, 6, "realistic", 16, "int8"),
    ("fa_dwell_trained_like", syn.FULL_ALIGNMENT, 9, True, 7, False, 7, "realistic", 8, "int8"),
    ("pileup_trained_like", syn.PILEUP, 18, False, 8, True, 8, "realistic", 32, "int8"),
    # the BASELINE.json sizes themselves (configs[1] / configs[2]: B = 1024 pileup, B = 256 full alignment): the rows of the real
    # reference modules for every window of the benchmark batches' shapes; Y_ref only (no .info / VCF strings for these)
    ("pileup_baseline_1024", syn.PILEUP, 18, False, 9, False, 9, "realistic", 1024, "int8"),
    ("fa_baseline_256", syn.FULL_ALIGNMENT, 8, True, 10, False, 10, "realistic", 256, "int8"),
]
BASELINE_SIZE = {"pileup_baseline_1024", "fa_baseline_256"}
TRAINED_LIKE = {"fa_trained_like", "fa_dwell_trained_like", "pileup_trained_like"}
# matrix depth of the full-alignment cases that are not ONT (shared/param_f.py:11: hifi / ilmn = 55 rows)
DEPTH = {"fa_hifi_depth55": 55}


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def sd_digest(sd):
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return h.hexdigest()[:16]


def case_inputs(case):
    name, kind, ch, indel, wseed, peaked, xseed, recipe, batch, xdt = case
    sd = syn.make_state_dict(kind, ch, indel, seed=wseed, peaked=peaked, trained_like=name in TRAINED_LIKE)
    if kind == syn.PILEUP:
        x = syn.make_pileup_windows(batch, xseed, recipe, dtype=np.dtype(xdt), channels=ch)
    else:
        x = syn.make_fa_windows(batch, xseed, recipe, channels=ch, depth=DEPTH.get(name, syn.FA_DEPTH_ONT))
    return sd, x


def reference_forward(kind, ch, indel, sd, x):
    from clair3.model import Clair3_F, Clair3_P
    cls = Clair3_P if kind == syn.PILEUP else Clair3_F
    m = cls(add_indel_length=indel, predict=True, input_channels=ch)
    m.eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    with torch.inference_mode():
        y = m(torch.from_numpy(x))
    return y.detach().cpu().numpy()


def synthetic_info(n, seed, ref_bases="ACGT"):
    """positions / alt_info strings in the grammar of src/clair3_full_alignment_dwell.c:957-1004 and
    clair3/CallVariantsFromCffi.py:116-122 ("<ctg>:<pos>:<ref>" and "<depth>-X<b> n I<ref><ins> n D<del> n R<ref> n ")."""
    rng = np.random.default_rng(seed)
    pos, alt = [], []
    for i in range(n):
        ref = ref_bases[rng.integers(0, 4)]
        seq = "".join(ref_bases[j] for j in rng.integers(0, 4, size=33))
        seq = seq[:16] + ref + seq[17:]
        depth = int(rng.integers(20, 80))
        parts = []
        left = depth
        others = [b for b in ref_bases if b != ref]
        k = int(rng.integers(0, min(left, 30) + 1))
        if k:
            parts.append(f"X{others[rng.integers(0, 3)]} {k}")
            left -= k
        if rng.random() < 0.4 and left > 2:
            k = int(rng.integers(1, left // 2 + 1))
            ins = "".join(ref_bases[j] for j in rng.integers(0, 4, size=int(rng.integers(1, 5))))
            parts.append(f"I{ref}{ins} {k}")
            left -= k
        if rng.random() < 0.4 and left > 2:
            k = int(rng.integers(1, left // 2 + 1))
            dele = seq[17:17 + int(rng.integers(1, 5))]
            parts.append(f"D{dele} {k}")
            left -= k
        if left > 0:
            parts.append(f"R{ref} {left}")
        pos.append(f"chr20:{100000 + 37 * i}:{seq}")
        alt.append(f"{depth}-" + " ".join(parts) + " ")
    return pos, alt


def reference_decode(kind, indel, y, pos, alt):
    import clair3.CallVariants as cv
    if kind == syn.PILEUP:
        import shared.param_p as param
    else:
        import shared.param_f as param
    cv.param = param
    cfg = cv.OutputConfig(
        is_show_reference=True, is_debug=False, is_haploid_precise_mode_enabled=False,
        is_haploid_sensitive_mode_enabled=False, is_output_for_ensemble=False, quality_score_for_pass=None,
        tensor_fn=None, input_probabilities=False, add_indel_length=indel, gvcf=False, pileup=(kind == syn.PILEUP),
        enable_long_indel=False, maximum_variant_length_that_need_infer=param.maximum_variant_length_that_need_infer,
        keep_iupac_bases=False)
    return cv.batch_output(pos, alt, y, cfg, None)


def main():
    torch.set_num_threads(1)
    manifest = {}
    only = None  # `--only name,name`: (re)generate just these cases and keep every other entry of the manifest as it is
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
        with open(os.path.join(HERE, "manifest.json")) as f:
            manifest = json.load(f)
    for case in CASES:
        name, kind, ch, indel, wseed, peaked, xseed, recipe, batch, xdt = case
        if only is not None and name not in only:
            continue
        sd, x = case_inputs(case)
        y = reference_forward(kind, ch, indel, sd, x)
        out = {"y_ref": y.astype(np.float32)}
        pos, alt, rows = [], [], ""
        if name not in BASELINE_SIZE:
            pos, alt = synthetic_info(batch, seed=1000 + xseed)
            try:
                rows = reference_decode(kind, indel, y, pos, alt)
            except Exception as e:  # decode goldens are a "next" row; never block the model goldens on them
                print(f"[warn] reference batch_output failed for {name}: {e!r}")
                rows = ""
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        manifest[name] = dict(kind=kind, channels=ch, add_indel_length=indel, weight_seed=wseed, peaked=peaked,
                              input_seed=xseed, recipe=recipe, batch=batch, x_dtype=xdt, x_sha=digest(x),
                              trained_like=name in TRAINED_LIKE, baseline_size=name in BASELINE_SIZE,
                              depth=(int(x.shape[1]) if kind == syn.FULL_ALIGNMENT else None),
                              sd_sha=sd_digest(sd), y_sha=digest(y), positions=pos, alt_info=alt, vcf_rows=rows,
                              torch=torch.__version__)
        print(f"{name}: x{x.shape} -> y{y.shape}  argmax21={np.bincount(y[:, :21].argmax(1), minlength=21).tolist()}")
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)


if __name__ == "__main__":
    main()
