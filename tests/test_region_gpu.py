"""On-device pileup window gathering against explicit host slicing."""
import numpy as np
import pytest

from clair3_amd import _lib, synthetic as syn
from clair3_amd.model import Clair3_P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.int8, np.int32])
def test_region_gather_equals_sliced_windows(dtype):
    sd = syn.make_state_dict(syn.PILEUP, seed=201)
    m = Clair3_P(predict=True).to("cuda:0")
    m.load_state_dict(sd)
    rng = np.random.default_rng(202)
    n_cols = 5000
    # a region-wide matrix as calculate_clair3_pileup returns it: one row of 18 counts per reference position
    region = syn.make_pileup_windows(n_cols // 33 + 1, seed=203, dtype=dtype).reshape(-1, 18)[:n_cols]
    starts = np.sort(rng.choice(n_cols - 33, size=700, replace=False)).astype(np.int32)
    starts[:3] = [0, 1, n_cols - 33]  # both ends and overlapping neighbours
    windows = np.stack([region[s:s + 33] for s in starts])  # the reference's host-side slicing (:362-364)
    y = m.predict_region(region, starts)
    # parity: against the CPU oracle on the host-sliced windows (not against the HIP path itself)
    from oracle import oracle
    from tests import util
    err = util.assert_rows_match(y, oracle.pileup_forward(sd, windows, False), what="region gather vs oracle")
    assert err < 2e-5
    # and the gather changes nothing: bit-identical to the same windows sent pre-sliced
    assert np.array_equal(y, m.predict_numpy(windows))
    with pytest.raises(_lib.C3Error, match="outside"):
        m.predict_region(region, np.array([n_cols - 32], np.int32))
    assert m.predict_region(region, np.zeros(0, np.int32)).shape == (0, 24)


def test_region_from_the_size_t_matrix():
    """plp_data.matrix itself (size_t counts, viewed with np.frombuffer as CreateTensorPileupFromCffi.py:140-146 does, without
    the copy): same rows as the int32 windows the reference's PIPE mode feeds the model."""
    sd = syn.make_state_dict(syn.PILEUP, seed=211)
    m = Clair3_P(predict=True).to("cuda:0")
    m.load_state_dict(sd)
    n_cols = 3000
    region32 = syn.make_pileup_windows(n_cols // 33 + 1, seed=212, dtype=np.int32).reshape(-1, 18)[:n_cols]
    region32[7, 3] = 300  # a count that int8 files would wrap and the size_t / int32 path must not
    buf = region32.astype(np.int64).tobytes()  # the C buffer
    region64 = np.frombuffer(buf, dtype=np.int64).reshape(n_cols, 18)
    starts = np.arange(0, n_cols - 33, 7, dtype=np.int32)
    y = m.predict_region(region64, starts)
    assert np.array_equal(y, m.predict_region(region32, starts))
    from oracle import oracle
    from tests import util
    windows = np.stack([region32[s:s + 33] for s in starts])
    assert util.assert_rows_match(y, oracle.pileup_forward(sd, windows, False), what="size_t region vs oracle") < 2e-5
