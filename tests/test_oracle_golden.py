"""The CPU oracle (oracle/c3_oracle.c) against the golden vectors produced by the real reference modules
(tests/golden/make_golden.py -> clair3/model.py Clair3_P / Clair3_F run on CPU).  Runs without a GPU."""
import hashlib

import numpy as np
import pytest

from oracle import oracle
from tests import util

CASES = sorted(util.manifest().keys())


def _digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("name", CASES)
def test_recipe_is_stable(name):
    """The seeded weights/windows rebuilt here are byte-identical to what the reference consumed."""
    meta = util.manifest()[name]
    sd, x = util.case_inputs(meta)
    assert _digest(x) == meta["x_sha"]
    h = hashlib.sha256()
    for k in sd:
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    assert h.hexdigest()[:16] == meta["sd_sha"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    meta = util.manifest()[name]
    sd, x = util.case_inputs(meta)
    y = oracle.forward(meta["kind"], sd, x, meta["add_indel_length"])
    y_ref = util.golden_y(name)
    # fp64 restatement vs the fp32 reference: observed <= 3.7e-6 (peaked heads), gate at 1e-5
    err = util.assert_rows_match(y, y_ref, tol=1e-5, what=name)
    assert err < 1e-5


def test_rows_are_probabilities():
    meta = util.manifest()["fa_realistic"]
    y = util.golden_y("fa_realistic")
    for lo, hi in util.HEAD_SLICES:
        np.testing.assert_allclose(y[:, lo:hi].sum(1), 1.0, atol=1e-5)
    assert meta["add_indel_length"]


def test_oracle_edge_cases():
    from clair3_amd import synthetic as syn
    sd = syn.make_state_dict(syn.PILEUP, seed=5)
    # empty batch
    y = oracle.pileup_forward(sd, np.zeros((0, 33, 18), np.int8))
    assert y.shape == (0, 24)
    # int8 and int32 views of the same counts give the same rows
    x8 = syn.make_pileup_windows(5, seed=9)
    y8 = oracle.pileup_forward(sd, x8)
    y32 = oracle.pileup_forward(sd, x8.astype(np.int32))
    assert np.array_equal(y8, y32)
    # windows are independent: a batch equals its rows run one by one
    y1 = np.concatenate([oracle.pileup_forward(sd, x8[i:i + 1]) for i in range(5)])
    assert np.array_equal(y8, y1)
    with pytest.raises(TypeError):
        oracle.pileup_forward(sd, x8.astype(np.float32))


def test_oracle_debug_dumps_shapes():
    from clair3_amd import synthetic as syn
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, add_indel_length=True, seed=3)
    x = syn.make_fa_windows(2, seed=1)
    y, d = oracle.fa_forward(sd, x, True, debug=True)
    assert d["act0"].shape == (2, 45, 17, 64) and d["act5"].shape == (2, 23, 9, 128)
    assert d["act8"].shape == (2, 12, 5, 256) and d["spp"].shape == (2, 3584)
    assert (d["act8"] >= 0).all() and y.shape == (2, 90)


@pytest.mark.parametrize("name", CASES)
def test_torch_port(name):
    """bench.py's cpu_baseline port (oracle/torch_port.py, same ATen operators as the reference modules)
    reproduces the reference rows: identical operators => differences only from operator-call grouping."""
    import torch
    from oracle import torch_port
    meta = util.manifest()[name]
    sd, x = util.case_inputs(meta)
    torch.set_num_threads(1)
    y = torch_port.forward(meta["kind"], torch_port.to_torch(sd), x, meta["add_indel_length"]).numpy()
    util.assert_rows_match(y, util.golden_y(name), tol=2e-6, what=name)
