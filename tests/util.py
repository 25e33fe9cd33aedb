"""Shared helpers of the test-suite: rebuild golden cases from their seeds, compare probability rows."""
import json
import os

import numpy as np

from clair3_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# parity gate of BASELINE.json: probabilities within 1e-4 (fp32), genotype / zygosity labels identical
PROB_TOL = 1e-4
# arg-max labels may legitimately differ only when the reference's own top-2 are closer than this (the reference
# moves its own probabilities by 1-2 ulp = 1.2e-7 with thread count or batch size, so "identical labels" is only meaningful outside ~1e-6)
NEAR_TIE = 1e-6
HEAD_SLICES = ((0, 21), (21, 24), (24, 57), (57, 90))


def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def case_inputs(meta):
    sd = syn.make_state_dict(meta["kind"], meta["channels"], meta["add_indel_length"], seed=meta["weight_seed"],
                             peaked=meta["peaked"], trained_like=meta.get("trained_like", False))
    if meta["kind"] == syn.PILEUP:
        x = syn.make_pileup_windows(meta["batch"], meta["input_seed"], meta["recipe"], dtype=np.dtype(meta["x_dtype"]),
                                    channels=meta["channels"])
    else:
        x = syn.make_fa_windows(meta["batch"], meta["input_seed"], meta["recipe"], channels=meta["channels"],
                                depth=meta.get("depth") or syn.FA_DEPTH_ONT)
    return sd, x


def golden_y(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))["y_ref"]


def label_mismatches(y, y_ref):
    """Per head: windows whose arg-max differs although the reference top-2 gap exceeds NEAR_TIE."""
    bad = []
    for lo, hi in HEAD_SLICES:
        if lo >= y_ref.shape[1]:
            break
        a, r = y[:, lo:hi], y_ref[:, lo:hi]
        diff = np.nonzero(a.argmax(1) != r.argmax(1))[0]
        for i in diff:
            top2 = np.sort(r[i])[-2:]
            if top2[1] - top2[0] > NEAR_TIE:
                bad.append((lo, int(i)))
    return bad


def assert_rows_match(y, y_ref, tol=PROB_TOL, what=""):
    assert y.shape == y_ref.shape, f"{what}: shape {y.shape} vs {y_ref.shape}"
    assert y.dtype == np.float32
    assert np.isfinite(y).all(), f"{what}: non-finite probabilities"
    err = float(np.abs(y.astype(np.float64) - y_ref.astype(np.float64)).max())
    assert err <= tol, f"{what}: max |dY| = {err:.3e} > {tol}"
    bad = label_mismatches(y, y_ref)
    assert not bad, f"{what}: arg-max label differs outside near-ties at (head_start, window) {bad[:8]}"
    return err
