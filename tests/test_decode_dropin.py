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


def config(cv, pileup, indel, **changes):
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
        keep_iupac_bases=False)._replace(**changes)


def printer_of(cv, cfg):
    """the RowPrinter (clair3_amd/vcf_rows.py) the rebound batch_output keeps for this configuration, counters reset"""
    pr = cv._c3hip_row_printers[(cfg, id(cv.param))]
    pr.taken = pr.retried = pr.handed_back = 0
    return pr


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
    cv.batch_output(pos[:1], alt[:1], widen(y[:1], indel), cfg, None)
    pr = printer_of(cv, cfg)
    try:
        got = cv.batch_output(pos, alt, widen(y, indel), cfg, None)
    finally:
        decode._ClassProbs._array, decode.outcome_from_columns = real_array, real_outcome
    assert got == want
    assert want.count("\n") >= len(y) // 2
    # rows are printed from the columns (vcf_rows.py), also after rejected candidates; only rows whose maximum two classes
    # share go back to the reference's output_with -- on the look-alike lists when they carry indel lengths
    assert pr.taken + pr.handed_back == len(y) and pr.retried > len(y) // 10
    assert pr.handed_back <= len(y) // 4
    assert calls["rows"] == (pr.handed_back if indel else 0)
    print(f"indel={indel}: {pr.taken} rows printed from the columns ({pr.retried} after rejected candidates), "
          f"{pr.handed_back} handed back, {calls['lists']} class lists formed for those")


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


@pytest.mark.parametrize("indel", [True, False])
@pytest.mark.parametrize("noise", [0.0, 0.35])
def test_rows_with_a_story_print_the_reference_s_text(indel, noise, ref):
    """rows peaked on each of the ten outcome classes with alt_info that offers (noise 0) or partly lacks (0.35) the alleles
    the class needs -- what a trained model and a real pileup hand the decoder: every row is printed from the columns
    (clair3_amd/vcf_rows.py), none goes back to the reference, and the text is the unpatched batch_output's"""
    from tests.decode_rows import consistent_rows
    cv, unpatched = ref
    pos, alt, y, classes = consistent_rows(1500, seed=7 + indel, indel=indel, noise=noise)
    cfg = config(cv, not indel, indel)
    want = unpatched(pos, alt, y, cfg, None)
    cv.batch_output(pos[:1], alt[:1], widen(y[:1], indel), cfg, None)
    pr = printer_of(cv, cfg)
    assert cv.batch_output(pos, alt, widen(y, indel), cfg, None) == want
    assert pr.taken == len(y) and pr.handed_back == 0
    assert (pr.retried == 0) if noise == 0 else (pr.retried > 100)
    if noise == 0:  # the story holds: one row per candidate, of the class it was peaked on
        rows = want.splitlines()
        assert len(rows) == len(y)
        gts = [r.rsplit("\t", 1)[1].split(":")[0] for r in rows]
        assert all(g == "0/0" for g, c in zip(gts, classes) if c == 0)
        assert all(g == "1/1" for g, c in zip(gts, classes) if c in (1, 3, 4))
        assert all(g in ("0/1", "1/2") for g, c in zip(gts, classes) if c in (2, 5, 6, 7, 8, 9))


@pytest.mark.parametrize("indel", [True, False])
def test_output_modes_and_odd_inputs(indel, ref):
    """the switches of OutputConfig and the input forms output_with accepts (clair3/CallVariants.py:1127-1154): bytes and
    numpy.bytes_ strings, contig names with colons, IUPAC and lower-case bases, depth 0, a QUAL threshold, no reference rows,
    kept IUPAC bases, gVCF rows (the PL field) and the two haploid modes (round 6); --enable_long_indel (round 6); the mode the row printer leaves to the reference (ensemble output) goes through it unchanged"""
    from tests.decode_rows import consistent_rows
    cv, unpatched = ref
    pos, alt, y, classes = consistent_rows(400, seed=19, indel=indel, noise=0.2)
    rng = np.random.default_rng(5)
    for i in range(len(pos)):
        chrom, p, seq = pos[i].split(":")
        kind = i % 8
        if kind == 1:
            chrom = "HLA-A*01:01:01:01"
        elif kind == 2:
            seq = seq[:16] + "NRYKM"[i % 5] + seq[17:]            # IUPAC centre base (BASE2ACGT resolves it, :690)
        elif kind == 3:
            seq = seq[:17] + "N" + seq[18:]                        # IUPAC base inside a deletion
        pos[i] = f"{chrom}:{p}:{seq}" + ("\n" if kind == 4 else "")
        if kind == 5 and classes[i] == 0:
            alt[i] = "0-" + alt[i].split("-", 1)[1]               # depth 0 (:1324; with two alternative alleles the reference divides by it, :1357)
        elif kind == 6:
            alt[i] = alt[i].split("-")[0] + "-"                   # no reads of any kind
        if kind == 7:
            pos[i], alt[i] = pos[i].encode(), np.bytes_(alt[i].encode())
    yw = widen(y, indel)
    for changes in ({}, {"is_show_reference": False}, {"quality_score_for_pass": 12}, {"keep_iupac_bases": True},
                    {"gvcf": True}, {"gvcf": True, "keep_iupac_bases": True, "is_show_reference": False},
                    {"is_haploid_precise_mode_enabled": True}, {"is_haploid_sensitive_mode_enabled": True},
                    {"is_haploid_precise_mode_enabled": True, "is_haploid_sensitive_mode_enabled": True, "gvcf": True},
                    {"enable_long_indel": True}, {"enable_long_indel": True, "maximum_variant_length_that_need_infer": 100000, "gvcf": True},
                    {"is_output_for_ensemble": True}):
        cfg = config(cv, not indel, indel, **changes)
        want = unpatched(pos, alt, y, cfg, None)
        got = cv.batch_output(pos, alt, yw, cfg, None)
        assert got == want, changes
        pr = cv._c3hip_row_printers[(cfg, id(cv.param))]
        assert pr.usable == (not changes.get("is_output_for_ensemble"))
        if any(k.startswith("is_haploid") for k in changes):
            gts = {r.split("\t")[9].split(":")[0] for r in got.splitlines()}
            assert gts and gts <= {"0", "1"}, gts
        if changes.get("gvcf"):
            assert got.count("GT:GQ:DP:AD:AF:PL") == got.count("\n") > 0
    # a centre base outside the IUPAC table is the reference's KeyError (:690), with or without the columns
    bad = pos[0].decode() if isinstance(pos[0], bytes) else pos[0]
    chrom, p, seq = bad.rsplit(":", 2)
    bad = f"{chrom}:{p}:{seq[:16]}x{seq[17:]}"
    cfg = config(cv, not indel, indel)
    with pytest.raises(KeyError):
        unpatched([bad], alt[:1], y[:1], cfg, None)
    with pytest.raises(KeyError):
        cv.batch_output([bad], alt[:1], yw[:1], cfg, None)
    # the file-writing form (args.output_file, :1108-1110)
    import io
    import types
    sink = types.SimpleNamespace(output_file=io.StringIO())
    assert cv.batch_output(pos, alt, yw, cfg, None, sink) == ""
    assert sink.output_file.getvalue() == unpatched(pos, alt, y, cfg, None)


@pytest.mark.parametrize("indel", [True, False])
def test_class_lists_of_a_whole_batch_are_the_per_row_lists(indel):
    """vcf_rows.class_lists_of_rows (every class list of every row of a batch in one pass) against decode.class_list row by row,
    which the test above pins to the reference's lists: the same bits, laid end to end in class order"""
    from clair3_amd import decode, vcf_rows
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "decode_indel.npz" if indel else "decode_noindel.npz"))
    rng = np.random.default_rng(5)
    extra = rng.random((64, g["y"].shape[1])).astype(np.float32) ** 6  # peaky rows, many near-zero products (subnormals included)
    y = np.concatenate([g["y"], extra])
    got = vcf_rows.class_lists_of_rows(y, indel)
    assert got.dtype == np.float32 and got.shape == (len(y), len(vcf_rows._KLASS[indel]))
    for r, row in enumerate(y):
        p1, p2 = (row[24:57], row[57:90]) if indel else (0, 0)
        want = np.concatenate([np.asarray(decode.class_list(k, row[:21], row[21:24], p1, p2, indel), dtype=np.float32) for k in range(1, 10)])
        assert np.array_equal(got[r].view(np.uint32), want.view(np.uint32)), r
    # and the chain order is the (chain rank, index) order the walk's stable sort relies on
    c = vcf_rows._CHAIN[indel]
    keys = list(zip(vcf_rows._RANK[indel][c].tolist(), vcf_rows._INDEX[indel][c].tolist()))
    assert keys == sorted(keys)


def test_dead_entries_are_exactly_the_ones_the_lookup_rejects(ref):
    """RowPrinter._dead (the entries a rejection walk steps over without asking, read off the key lengths of the row's alt_info
    dictionary) against RowPrinter._alleles -- i.e. the reference's own lookups -- entry by entry, on alt_info
    dictionaries with no / one / several insertions and deletions of every length, lengths beyond the inference range included:
    an entry of classes 3-9 is marked dead exactly when the per-entry lookup rejects it"""
    from clair3_amd import decode, vcf_rows
    cv, _ = ref
    cfg = config(cv, False, True)
    pr = vcf_rows.RowPrinter(cv, cfg)
    rng = np.random.default_rng(3)
    n_entries = len(vcf_rows._KLASS[True])
    checked = rejected = 0
    for trial in range(60):
        d = {}
        for kind in "ID":
            for _ in range(int(rng.integers(0, 4)) if trial % 5 else 0):
                length = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 30, 49, 50, 51, 60]))
                bases = "".join("ACGT"[j] for j in rng.integers(0, 4, size=length))
                d[kind + ("A" + bases if kind == "I" else bases)] = int(rng.integers(1, 30))
        if trial % 3 == 0:
            d["XC"] = 5
        d["RA"] = 9
        dead = pr._dead(d)[0]
        assert np.array_equal(pr._dead(d)[1], dead[vcf_rows._CHAIN[True]])
        for e in range(n_entries):
            k, pos = int(vcf_rows._KLASS[True][e]), int(vcf_rows._INDEX[True][e])
            if k <= 2:
                assert not dead[e]  # SNP classes are always left to the lookup
                continue
            alleles = pr._alleles(k, pos, "A", vcf_rows._Lookups(cv, d, cfg.maximum_variant_length_that_need_infer))
            assert dead[e] == (alleles is None), (trial, d, decode.CLASS_NAMES[k], pos, alleles)
            checked += 1
            rejected += alleles is None
    assert checked > 40000 and 0.2 < rejected / checked < 0.9
