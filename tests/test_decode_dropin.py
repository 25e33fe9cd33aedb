"""the reference's OWN decoder (clair3/CallVariants.py batch_output -> output_with -> output_from) driven
by decoder columns instead of its Python enumeration must print the same VCF rows, character for character.

Runs only where the reference checkout is mounted (the build container): the columns come from oracle/decode_oracle.py
here; tests/test_decode.py::test_decode_columns_* (GPU) checks that the device produces the same columns bit for bit.
"""
import os
import sys

import numpy as np
import pytest

from oracle import decode_oracle
from tests import util

REF = os.environ.get("CLAIR3_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "clair3")),
                                reason="needs the reference checkout (build container only)")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, REF)
    import clair3.CallVariants as cv
    from clair3_amd import decode
    unpatched = cv.batch_output if not getattr(cv, "_c3hip_decoder", None) else None
    assert unpatched is not None
    decode.install_decoder()
    assert decode.install_decoder() is cv._c3hip_decoder  # idempotent
    yield cv, unpatched
    sys.path.remove(REF)
    for k in [k for k in sys.modules if k == "clair3" or k.startswith("clair3.") or k.startswith("shared")]:
        del sys.modules[k]  # the next test that imports the reference gets unpatched modules


def config(cv, pileup, indel):
    if pileup:
        import shared.param_p as param
    else:
        import shared.param_f as param
    cv.param = param
    return cv.OutputConfig(
        is_show_reference=True, is_debug=False, is_haploid_precise_mode_enabled=False,
        is_haploid_sensitive_mode_enabled=False, is_output_for_ensemble=False, quality_score_for_pass=None,
        tensor_fn=None, input_probabilities=False, add_indel_length=indel, gvcf=False, pileup=pileup,
        enable_long_indel=False, maximum_variant_length_that_need_infer=param.maximum_variant_length_that_need_infer,
        keep_iupac_bases=False)


def widen(y, indel):
    return np.ascontiguousarray(np.concatenate([y, decode_oracle.decode_columns(y, indel)], axis=1), dtype=np.float32)


@pytest.mark.parametrize("name", sorted(util.manifest().keys()))
def test_golden_vcf_rows(name, ref):
    """rows of the reference networks (tests/golden/*.npz) -> the VCF text the unpatched reference printed for them"""
    cv, _ = ref
    meta = util.manifest()[name]
    if not meta["vcf_rows"]:
        pytest.skip("no decode golden for this case")
    y = util.golden_y(name)
    cfg = config(cv, meta["kind"] == "pileup", meta["add_indel_length"])
    rows = cv.batch_output(meta["positions"], meta["alt_info"], widen(y, meta["add_indel_length"]), cfg, None)
    assert rows == meta["vcf_rows"]


def alt_infos(n, seed):
    """alt_info strings that make output_from work for its living: candidates of every kind, rows that offer nothing
    for the winning class (rejected -> zeroed -> next best) and rows that offer nothing at all."""
    rng = np.random.default_rng(seed)
    pos, alt = [], []
    for i in range(n):
        seq = "".join("ACGT"[j] for j in rng.integers(0, 4, size=33))
        ref_base = "ACGT"[i % 4]
        seq = seq[:16] + ref_base + seq[17:]
        others = [b for b in "ACGT" if b != ref_base]
        depth = int(rng.integers(12, 90))
        kind = i % 6
        parts = []
        if kind in (0, 1, 5):
            parts.append(f"X{others[int(rng.integers(0, 3))]} {int(rng.integers(1, depth // 2))}")
        if kind in (1, 5):
            parts.append(f"X{others[int(rng.integers(0, 3))]} {int(rng.integers(1, depth // 3))}")
        if kind in (2, 5):
            for _ in range(int(rng.integers(1, 4))):
                ins = "".join("ACGT"[j] for j in rng.integers(0, 4, size=int(rng.integers(1, 20))))
                parts.append(f"I{ref_base}{ins} {int(rng.integers(1, depth // 3))}")
        if kind in (3, 5):
            for _ in range(int(rng.integers(1, 4))):
                parts.append(f"D{seq[17:17 + int(rng.integers(1, 16))]} {int(rng.integers(1, depth // 3))}")
        parts.append(f"R{ref_base} {int(rng.integers(1, depth))}")  # kind 4: reference reads only
        pos.append(f"chr{1 + i % 3}:{5000 + 41 * i}:{seq}")
        alt.append(f"{depth}-" + " ".join(parts) + " ")
    return pos, alt


@pytest.mark.parametrize("indel", [True, False])
def test_retry_loop_is_the_reference_s(indel, ref):
    """flat, peaked, tied and early-exit rows (tests/golden/decode_*.npz) x adversarial alt_info: identical text, and
    the fallback to the reference's enumeration happens only where a candidate was rejected"""
    cv, unpatched = ref
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "decode_indel.npz" if indel else "decode_noindel.npz"))
    y = np.concatenate([g["y"]] * 3)
    pos, alt = alt_infos(len(y), seed=11 + indel)
    cfg = config(cv, not indel, indel)
    want = unpatched(pos, alt, y, cfg, None)
    from clair3_amd import decode
    calls = {"rows": 0, "lists": 0}
    real_array, real_outcome = decode._ClassProbs._array, decode.outcome_from_columns

    def counting_array(self):
        if self.arr is None:
            calls["lists"] += 1
        return real_array(self)

    def counting_outcome(*a, **k):
        calls["rows"] += 1
        return real_outcome(*a, **k)
    decode._ClassProbs._array, decode.outcome_from_columns = counting_array, counting_outcome
    try:
        got = cv.batch_output(pos, alt, widen(y, indel), cfg, None)
    finally:
        decode._ClassProbs._array, decode.outcome_from_columns = real_array, real_outcome
    assert got == want
    assert want.count("\n") >= len(y) // 2
    if not indel:  # 24-column rows: columns dropped, the reference's own enumeration (nothing to save there)
        assert calls["rows"] == 0
        return
    assert calls["rows"] == len(y)
    assert 0 < calls["lists"] < 9 * calls["rows"]
    print(f"indel={indel}: {calls['rows']} rows, {calls['lists']} class lists had to be formed")


@pytest.mark.parametrize("indel", [True, False])
def test_class_lists_are_the_reference_s_bit_for_bit(indel, ref):
    """clair3_amd.decode.class_list (vectorised float32) against the lists of the unpatched enumeration"""
    cv, _ = ref
    from clair3_amd import decode
    enumerate_rows = cv._c3hip_decoder[0].__closure__  # the original function lives in the closure
    original = [c.cell_contents for c in enumerate_rows if callable(c.cell_contents)
                and getattr(c.cell_contents, "__name__", "") == "possible_outcome_probabilites_from"][0]
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "decode_indel.npz" if indel else "decode_noindel.npz"))
    tuple_pos = {1: 1, 2: 2, 3: 4, 4: 11, 5: 9, 6: 6, 7: 16, 8: 13, 9: 18}
    checked = 0
    for row, early in zip(g["y"], g["early"]):
        if early:
            continue
        p1, p2 = (row[24:57], row[57:90]) if indel else (0, 0)
        want = original(row[:21], row[21:24], p1, p2, reference_base="A", alt_info_dict={}, add_indel_length=indel)
        if len(want) == 1:
            continue
        for cls, pos in tuple_pos.items():
            got = decode.class_list(cls, row[:21], row[21:24], p1, p2, indel)
            ref_list = np.array(want[pos], dtype=np.float32)
            assert got.dtype == np.float32 and got.shape == ref_list.shape, (cls, got.shape, ref_list.shape)
            assert np.array_equal(got.view(np.uint32), ref_list.view(np.uint32)), f"class {cls}"
            if indel and cls in (3, 4, 6, 8, 9):
                assert decode._ENTRIES[cls] == list(want[pos - 1])
        if indel:
            assert decode._ENTRIES[50] == list(want[7]) and decode._ENTRIES[51] == list(want[8])
            assert decode._ENTRIES[70] == list(want[14]) and decode._ENTRIES[71] == list(want[15])
        checked += 1
    assert checked >= 60


def test_rows_without_columns_take_the_reference_path(ref):
    cv, unpatched = ref
    meta = util.manifest()["fa_realistic"]
    y = util.golden_y("fa_realistic")
    cfg = config(cv, False, True)
    assert cv.batch_output(meta["positions"], meta["alt_info"], y, cfg, None) == meta["vcf_rows"]
    assert unpatched(meta["positions"], meta["alt_info"], y, cfg, None) == meta["vcf_rows"]
