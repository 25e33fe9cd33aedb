"""The library on weights that went through an optimizer (tests/sgd_trained.py): the closest thing to a trained checkpoint this offline image
can make.  The reference's own modules are trained for a few hundred AdamW steps on teacher-labelled synthetic windows; the library loads the
resulting state dict (strictly, as clair3/CallVariantsFromCffi.py:19-28 does) and its rows are held to the reference's fp32 CPU rows
(north_star: 1e-4, labels identical outside the reference's near-ties) and, layer by layer, to the fp64 oracle; the range guard stays quiet
and the headroom of every convolution stage to the fp16 range is printed.  VERDICT r5 'what's missing' item 1 (reference side:
clair3/Train.py:87-107,386-388; docs/quick_demo/ont_quick_demo.md for what stays blocked on data)."""
import numpy as np
import pytest

from clair3_amd import synthetic as syn
from tests import refmodels, sgd_trained, util
from tests.test_parity_gpu import make_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,channels,steps,batch,n", [(syn.PILEUP, 18, 200, 64, 1100), (syn.FULL_ALIGNMENT, 8, 120, 16, 300),
                                                         (syn.FULL_ALIGNMENT, 9, 60, 16, 130)])
def test_rows_on_weights_that_went_through_an_optimizer(kind, channels, steps, batch, n, monkeypatch):
    from oracle import oracle
    monkeypatch.delenv("C3HIP_FP32", raising=False)
    root = refmodels.reference_root_or_skip()
    sd, losses = sgd_trained.train_reference(root, kind, channels, True, steps=steps, batch=batch, seed=3)
    assert losses[-1] < 0.9 * losses[0], losses
    x = syn.make_windows(kind, n, seed=4242, channels=channels)
    m = make_model(kind, channels, True, sd)  # the product path (conv1 inside res1a / res1b, the pooling inside res3b)
    y = m.predict_numpy(x)
    y_ref = refmodels.reference_rows(refmodels.reference_model(root, kind, sd, True, channels), x)
    what = f"{kind} C={channels} after {steps} AdamW steps"
    err = util.assert_rows_match(y, y_ref, tol=util.PROB_TOL, what=what)  # 1e-4 + labels outside near-ties, every row
    assert m.range_status() == (0, False) and "on_fp32=0" in m.describe(), m.describe()
    # against the exact rows, layer by layer, on the first windows (the oracle is the slow side)
    k = 24
    y_o, d = oracle.forward(kind, sd, x[:k], True, debug=True)
    mk = make_model(kind, channels, True, sd, keep=True)
    yk = mk.predict_numpy(x[:k])
    worst = {}
    names = ("lstm1_out", "lstm2_out", "l4_out") if kind == syn.PILEUP else tuple(f"act{i}" for i in range(9)) + ("l4_out",)
    for name in names:
        if name not in d:
            continue
        a = mk.debug_fetch(name, d[name].shape)
        scale = max(float(np.abs(d[name]).max()), 1e-30)
        worst[name] = (float(np.abs(a - d[name]).max()) / scale, scale)
    print(f"{what}: loss {losses[0]:.2f} -> {losses[-1]:.2f}, peak probability {float(y_ref.max()):.3f}; max |dY| vs the reference's fp32 rows {err:.2e} "
          f"({n} windows), vs the exact rows {float(np.abs(yk - y_o).max()):.2e}; {m.describe()}")
    print("   layer: error relative to the layer's range (range): " + ", ".join(f"{k_} {e:.1e} ({s:.3g})" for k_, (e, s) in worst.items()))
    assert err < 2e-5 and float(np.abs(yk - y_o).max()) < 2e-5
    assert all(e < 2e-5 for e, _ in worst.values()), worst
    if kind == syn.FULL_ALIGNMENT:  # headroom of the activations to the range guard's 16000 (fp16 pieces need |x| < 65504)
        assert max(s for k_, (_, s) in worst.items() if k_.startswith("act")) < 16000
    # the same windows in another batch composition: bit-identical rows
    assert np.array_equal(np.concatenate([m.predict_numpy(x[:97]), m.predict_numpy(x[97:])]), y)
