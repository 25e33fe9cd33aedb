"""Bounded soak INSIDE the -m gpu suite: random batch sizes, window recipes and weight sets (plain /
trained-like / peaked) for the five model shapes through the C ABI, gated against the rows of the REFERENCE's own fp32 modules
(clair3/model.py from the staged copy, called as clair3/CallVariantsFromCffi.py:48-52 does on the CPU):
|Y_hip - Y_reference_fp32| <= 1e-4 on every checked row, labels identical outside the reference's own near-ties (1e-6); and the same
windows travelling as two batches give bit-identical rows.  Full-alignment batches (<= 330 windows) are checked against the reference on
EVERY row; of a pileup batch (<= 1300 windows, the reference's LSTM is the slow side) both ends and 16 random rows are -- the middle of a
pileup batch meets the reference only through that sample plus the split-batch bit-identity below, which is what makes this a bounded soak.  Fixed seeds 7 and 11 (the seeds of tests/diag/fuzz_parity.py, whose
open-ended form stays a diagnostic), a fixed number of batches each, so the run is reproducible and takes well under a minute."""
import numpy as np
import pytest

from clair3_amd import synthetic as syn
from tests import refmodels, util
from tests.test_parity_gpu import make_model

pytestmark = pytest.mark.gpu

SHAPES = [(syn.PILEUP, 18, False), (syn.PILEUP, 18, True), (syn.FULL_ALIGNMENT, 8, True), (syn.FULL_ALIGNMENT, 9, True),
          (syn.FULL_ALIGNMENT, 8, False)]
EDGE_SIZES = [1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 184, 185, 186, 255, 256, 257]


@pytest.mark.parametrize("seed,batches", [(7, 36), (11, 36)])
def test_random_batches_and_weight_sets_against_the_reference_rows(seed, batches):
    root = refmodels.reference_root_or_skip()
    import torch
    torch.set_num_threads(8)
    rng = np.random.default_rng(seed)
    models = {}
    worst, rows_checked, windows, weight_sets = 0.0, 0, 0, 0
    worst_case = None
    for _ in range(batches):
        kind, ch, indel = SHAPES[int(rng.integers(len(SHAPES)))]
        s = int(rng.integers(1 << 30))
        key = (kind, ch, indel)
        if key not in models or rng.random() < 0.25:
            flags = dict(seed=s, peaked=bool(rng.random() < 0.25), trained_like=bool(rng.random() < 0.5))
            sd = syn.make_state_dict(kind, ch, indel, **flags)
            models[key] = (make_model(kind, ch, indel, sd), refmodels.reference_model(root, kind, sd, indel, ch), flags)
            weight_sets += 1
        m, m_ref, flags = models[key]
        hi = 1300 if kind == syn.PILEUP else 330
        n = int(rng.integers(1, hi)) if rng.random() < 0.7 else int(rng.choice(EDGE_SIZES))
        recipe = "uniform" if rng.random() < 0.3 else "realistic"
        x = syn.make_pileup_windows(n, seed=s, recipe=recipe) if kind == syn.PILEUP else \
            syn.make_fa_windows(n, seed=s, recipe=recipe, channels=ch)
        y = m.predict_numpy(x)
        assert y.dtype == np.float32 and np.isfinite(y).all()
        # full alignment: every row; pileup (the reference is the slow side): both ends and a random middle sample
        sample = np.unique(np.r_[0:min(n, 8), max(0, n - 8):n, rng.integers(0, n, size=min(n, 16))])  # (drawn for both kinds: the random stream stays round 4's)
        idx = np.arange(n) if kind == syn.FULL_ALIGNMENT else sample
        y_ref = refmodels.reference_rows(m_ref, x[idx])
        what = f"seed {seed}: kind={kind} ch={ch} indel={indel} n={n} recipe={recipe} input_seed={s} weights={flags}"
        err = util.assert_rows_match(y[idx], y_ref, tol=util.PROB_TOL, what=what)  # 1e-4 + labels outside near-ties
        if err > worst:
            worst, worst_case = err, what
        rows_checked += len(idx)
        windows += n
        if n > 3:  # the same windows in another batch composition: bit-identical rows
            k = int(rng.integers(1, n))
            y2 = np.concatenate([m.predict_numpy(x[:k]), m.predict_numpy(x[k:])])
            assert np.array_equal(y, y2), f"rows depend on the batch they travel in ({what}, split at {k})"
    print(f"fuzz seed {seed}: {batches} batches, {windows} windows, {weight_sets} weight sets, {rows_checked} rows against the "
          f"reference's fp32 modules: worst |dY| = {worst:.2e} ({worst_case})")
