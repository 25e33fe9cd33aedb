"""Weights that went through an optimizer (tests/sgd_trained.py: the reference's own modules trained with AdamW and the focal loss of
clair3/Train.py on synthetic, teacher-labelled windows).  CPU part: the training helper does what it says (the loss falls, BatchNorm
statistics move, the same seed gives the same weights) and the fp64 oracle restates the reference's forward pass on such weights as closely
as on seeded ones -- the oracle is pinned on optimizer-shaped weights too, not only on tests/golden.  The GPU part is
tests/test_sgd_trained_gpu.py."""
import numpy as np
import pytest

from clair3_amd import synthetic as syn
from tests import refmodels, sgd_trained


@pytest.mark.parametrize("kind,channels,steps,batch", [(syn.PILEUP, 18, 60, 32), (syn.FULL_ALIGNMENT, 8, 30, 8)])
def test_oracle_on_weights_that_went_through_an_optimizer(kind, channels, steps, batch):
    from oracle import oracle
    root = refmodels.reference_root_or_skip()
    sd, losses = sgd_trained.train_reference(root, kind, channels, True, steps=steps, batch=batch, seed=5)
    assert losses[-1] < 0.9 * losses[0], losses  # it learns something
    spec = dict(syn.state_dict_spec(kind, channels, True))
    assert set(k for k in sd if not k.endswith("num_batches_tracked")) == set(spec)  # a state dict the strict loader accepts
    if kind == syn.FULL_ALIGNMENT:
        rv = sd["conv5.bn.running_var"]
        assert float(rv.max() / rv.min()) > 3 and abs(float(sd["conv1.bn.running_mean"].mean())) > 1e-4  # statistics from real forward passes
    x = syn.make_windows(kind, 16, seed=77, channels=channels)
    y_ref = refmodels.reference_rows(refmodels.reference_model(root, kind, sd, True, channels), x)
    y_o = oracle.forward(kind, sd, x, True)
    err = float(np.abs(y_o - y_ref).max())
    print(f"{kind}: loss {losses[0]:.2f} -> {losses[-1]:.2f}; oracle vs the reference's rows {err:.2e}; peak probability {float(y_ref.max()):.3f}")
    assert err < 1e-5
    assert (y_o[:, :21].argmax(1) == y_ref[:, :21].argmax(1)).all()


def test_training_is_reproducible_from_its_seed():
    root = refmodels.reference_root_or_skip()
    a, _ = sgd_trained.train_reference(root, syn.PILEUP, 18, False, steps=6, batch=8, seed=9, threads=1)
    b, _ = sgd_trained.train_reference(root, syn.PILEUP, 18, False, steps=6, batch=8, seed=9, threads=1)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c, _ = sgd_trained.train_reference(root, syn.PILEUP, 18, False, steps=6, batch=8, seed=10, threads=1)
    assert any(not np.array_equal(a[k], c[k]) for k in a)
