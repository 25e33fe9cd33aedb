"""SURVEY 8f N1 (decode slice): the class maxima the reference's decoder derives from a probability row.

CPU: the numpy restatement (oracle/decode_oracle.py) against goldens produced by the real reference function
(tests/golden/make_golden_decode.py), bit for bit; the host-side tables of clair3_amd.decode against the enumeration.
GPU: c3_outcome_maxima through the C ABI against the same goldens and against the oracle on larger seeded sets.
"""
import os

import numpy as np
import pytest

from oracle import decode_oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def golden(indel):
    return np.load(os.path.join(HERE, "golden", "decode_indel.npz" if indel else "decode_noindel.npz"))


def check_against_golden(maxp, arg, early, g):
    assert np.array_equal(early, g["early"])
    live = ~g["early"]  # the reference returns before enumerating on an early exit: only class 0 is defined there
    assert np.array_equal(maxp[live].view(np.uint32), g["maxp"][live].view(np.uint32)), "class maxima differ from the reference"
    assert np.array_equal(arg[live], g["argmax"][live]), "positions of the maxima differ from the reference"
    assert np.array_equal(maxp[~live, 0].view(np.uint32), g["maxp"][~live, 0].view(np.uint32))


@pytest.mark.parametrize("indel", [True, False])
def test_oracle_matches_reference_goldens(indel):
    g = golden(indel)
    maxp, arg, early = decode_oracle.outcome_maxima(g["y"], g["ref21"], indel)
    assert early.sum() >= 5 and (~early).sum() >= 80  # both branches are exercised
    check_against_golden(maxp, arg, early, g)


def test_class_entry_tables_follow_the_reference_enumeration():
    from clair3_amd import decode
    assert len(decode._INSINS) == 136 and len(decode._DELDEL) == 241 and len(decode._INSDEL) == 256
    assert decode.class_entry(6, 0) == (1, 1) and decode.class_entry(6, 135) == (16, 16)
    assert decode.class_entry(8, 0) == (1, 2) and decode.class_entry(8, 240) == (16, 16)
    assert decode.class_entry(8, 15) == (1, 2)  # (2, 1) is stored as (1, 2): CallVariants.py:354
    assert decode.class_entry(5, 5) == ("C", 2) and decode.class_entry(3, 15) == 16
    assert decode.class_entry(5, 2, add_indel_length=False) == "G" and decode.class_entry(1, 3) == 9
    assert np.array_equal(decode.ref_gt21_indices("ACGT"), [0, 4, 7, 9])


def _model(indel):
    from clair3_amd import synthetic as syn
    from clair3_amd.model import Clair3_F, Clair3_P
    if indel:
        m = Clair3_F(add_indel_length=True, predict=True, input_channels=8).to("cuda:0")
        m.load_state_dict(syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=0))
    else:
        m = Clair3_P(add_indel_length=False, predict=True).to("cuda:0")
        m.load_state_dict(syn.make_state_dict(syn.PILEUP, 18, False, seed=0))
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("indel", [True, False])
def test_kernel_matches_reference_goldens(indel):
    from clair3_amd import decode
    g = golden(indel)
    maxp, arg, early = decode.outcome_maxima(_model(indel), g["y"], g["ref21"])
    check_against_golden(maxp, arg, early, g)


@pytest.mark.gpu
@pytest.mark.parametrize("indel", [True, False])
def test_kernel_matches_oracle_on_model_rows(indel):
    """Rows produced by the networks themselves (incl. a ragged count), reference bases cycling through ACGT."""
    from clair3_amd import decode, synthetic as syn
    m = _model(indel)
    x = syn.make_windows(syn.FULL_ALIGNMENT if indel else syn.PILEUP, 203, seed=5, channels=8 if indel else 18)
    y = m.predict_numpy(x)
    ref = "".join("ACGT"[(3 * i) % 4] for i in range(len(y)))
    maxp, arg, early = decode.outcome_maxima(m, y, ref)
    o_maxp, o_arg, o_early = decode_oracle.outcome_maxima(y, decode.ref_gt21_indices(ref), indel)
    assert np.array_equal(early, o_early)
    assert np.array_equal(maxp.view(np.uint32), o_maxp.view(np.uint32))
    assert np.array_equal(arg, o_arg)
    # the overall maximum sits in at least one class, and the kernel agrees with the oracle on which
    best = maxp.max(axis=1, keepdims=True)
    assert np.array_equal(maxp == best, o_maxp == o_maxp.max(axis=1, keepdims=True))
    # the same handle keeps predicting after the decoder has grown its scratch buffer, twice (the growth
    # path used to free the handle's range flag, which every fp16x3 epilogue writes)
    assert np.array_equal(m.predict_numpy(x), y)
    decode.outcome_maxima(m, np.concatenate([y, y]), ref + ref)
    assert np.array_equal(m.predict_numpy(x), y)
    assert m.range_status() == (0, False)


@pytest.mark.gpu
def test_outcome_maxima_rejects_bad_arguments():
    from clair3_amd import _lib, decode
    m = _model(False)
    y = np.full((3, 24), 1.0 / 24, dtype=np.float32)
    with pytest.raises(_lib.C3Error):
        decode.outcome_maxima(m, y, np.array([0, 1, 4], dtype=np.uint8))  # 1 = AC is not a reference base pair
    with pytest.raises(_lib.C3Error):
        decode.outcome_maxima(m, y[:, :20], "ACG")
    maxp, arg, early = decode.outcome_maxima(m, y[:0], "")
    assert maxp.shape == (0, 10) and early.shape == (0,)


def test_oracle_columns_agree_with_the_per_base_maxima():
    """decode_oracle.decode_columns is outcome_maxima evaluated for the four bases: pinned by the same goldens"""
    g = golden(True)
    cols = decode_oracle.decode_columns(g["y"], True)
    assert cols.shape == (len(g["y"]), 31) and cols.dtype == np.float32
    base = np.array([{0: 0, 4: 1, 7: 2, 9: 3}[int(k)] for k in g["ref21"]])
    rows = np.arange(len(base))
    assert np.array_equal(((cols[:, 22].astype(np.int32) >> base) & 1).astype(bool), g["early"])
    assert np.array_equal(cols[rows, 9 + base].view(np.uint32), g["maxp"][:, 0].view(np.uint32))
    live = ~g["early"]
    assert np.array_equal(cols[live, 0:9].view(np.uint32), g["maxp"][live, 1:].view(np.uint32))
    assert np.array_equal(cols[live, 13:22].astype(np.int32), g["argmax"][live, 1:])


@pytest.mark.parametrize("indel", [True, False])
def test_first_decision_and_qual_match_the_real_reference(indel):
    """winner class (output_from :722-751 on the real lists) and QUAL (the real quality_score_from) stored by
    make_golden_decode.py == the oracle's restatement, exactly; and the host read-out of the oracle's columns returns them"""
    from clair3_amd import decode
    g = golden(indel)
    winner, qual = decode_oracle.first_decision(g["y"], g["ref21"], indel)
    assert np.array_equal(winner, g["winner"]) and np.array_equal(qual, g["qual"])
    assert len(set(g["winner"].tolist())) >= 8 and (g["qual"] > 0).sum() > 20  # the goldens exercise most classes
    wide = np.concatenate([g["y"], decode_oracle.decode_columns(g["y"], indel)], axis=1)
    letters = "".join({0: "A", 4: "C", 7: "G", 9: "T"}[int(k)] for k in g["ref21"])
    d = decode.first_decisions(wide, letters, g["y"].shape[1])
    assert np.array_equal(d["cls"], g["winner"]) and np.array_equal(d["qual"], g["qual"]) and np.array_equal(d["early"], g["early"])
    live = ~g["early"]
    rows = np.arange(len(wide))[live]
    assert np.array_equal(d["pos"][live], g["argmax"][rows, g["winner"][live].astype(int)] * (g["winner"][live] > 0))
    assert np.array_equal(d["prob"][live].view(np.uint32), g["maxp"][rows, g["winner"][live].astype(int)].view(np.uint32))
    # IUPAC reference bases resolve as output_from resolves them (clair3/CallVariants.py:690 -> shared/utils.py:42-45):
    # U -> T, R -> A, Y / S / B -> C, K -> G ...; anything else (lower case included: a KeyError there) is an error
    alias = {"A": "RWMDHVN", "C": "YSB", "G": "K", "T": "U"}
    iupac = "".join(alias[c][i % len(alias[c])] for i, c in enumerate(letters))
    d2 = decode.first_decisions(wide, iupac, g["y"].shape[1])
    assert all(np.array_equal(d[k], d2[k]) for k in d)
    from clair3_amd import _lib
    with pytest.raises(_lib.C3Error, match="IUPAC"):
        decode.first_decisions(wide[:2], "a?", g["y"].shape[1])


@pytest.mark.gpu
@pytest.mark.parametrize("indel", [True, False])
def test_decode_columns_of_golden_rows(indel):
    """c3_decode_columns on the hand-made rows (ties, 0.5 boundaries, early exits) == the oracle's columns, bit for bit"""
    from clair3_amd import decode
    g = golden(indel)
    rows = decode.decode_columns(_model(indel), g["y"])
    assert rows.shape == (len(g["y"]), g["y"].shape[1] + decode.DECODE_COLS)
    assert np.array_equal(rows[:, :g["y"].shape[1]].view(np.uint32), g["y"].view(np.uint32))
    want = decode_oracle.decode_columns(g["y"], indel)
    assert np.array_equal(rows[:, g["y"].shape[1]:].view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("indel", [True, False])
def test_predict_appends_the_decode_columns(indel):
    """model.decode_columns(): every predict entry point returns [probabilities | columns]; the probabilities are the
    bits of the plain call, the columns are the oracle's; switching it off restores the plain rows"""
    from clair3_amd import decode, synthetic as syn
    m = _model(indel)
    kind = syn.FULL_ALIGNMENT if indel else syn.PILEUP
    x = syn.make_windows(kind, 77, seed=9, channels=8 if indel else 18)
    y = m.predict_numpy(x)
    m.decode_columns(True)
    assert m.row_size == m.output_size + decode.DECODE_COLS
    wide = m.predict_numpy(x)
    assert wide.shape == (len(x), m.row_size)
    assert np.array_equal(wide[:, :m.output_size].view(np.uint32), y.view(np.uint32))
    want = decode_oracle.decode_columns(y, indel)
    assert np.array_equal(wide[:, m.output_size:].view(np.uint32), want.view(np.uint32))
    t0, t1 = m.submit(x[:40], 0), m.submit(x[40:], 1)   # the asynchronous pair, both slots
    both = np.concatenate([m.wait(t0), m.wait(t1)])
    assert np.array_equal(both.view(np.uint32), wide.view(np.uint32))
    import torch
    xd = torch.from_numpy(x).to("cuda:0")
    yd = m(xd)                                          # device-to-device entry
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy().view(np.uint32), wide.view(np.uint32))
    if not indel:
        region = np.ascontiguousarray(np.concatenate([x[0], x[1]]))
        r = m.predict_region(region, [0, 33])
        assert np.array_equal(r.view(np.uint32), wide[:2].view(np.uint32))
    m.decode_columns(False)
    assert np.array_equal(m.predict_numpy(x).view(np.uint32), y.view(np.uint32))
