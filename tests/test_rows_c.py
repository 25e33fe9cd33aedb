"""c3_vcf_rows (csrc/c3_rows.h): the one-pass host printer of the rows whose first decision stands, held entry for entry to the Python
path it accelerates (clair3_amd/vcf_rows.py RowPrinter._rows_py -- which calls the REFERENCE's own find_alt_base /
insertion_bases_using_alt_info_from / deletion_bases_using_alt_info_from / quality_score_from / filtration_value_from) and, through
the rebound batch_output, to the unpatched reference decoder.  No GPU: the function is plain host code of libc3hip.so and the decoder
columns come from oracle/decode_oracle.py or are made up (every class and entry can be asked for directly through the columns).
Runs only where the reference checkout is mounted (the build container)."""
import os
import sys
import time

import numpy as np
import pytest

from tests.decode_rows import consistent_rows

REF = os.environ.get("CLAIR3_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "clair3")), reason="needs the reference checkout (build container only)")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, REF)
    import clair3.CallVariants as cv
    from clair3_amd import decode
    unpatched = cv.batch_output if not getattr(cv, "_c3hip_decoder", None) else None
    assert unpatched is not None
    decode.install_decoder()
    yield cv, unpatched
    sys.path.remove(REF)
    for k in [k for k in sys.modules if k == "clair3" or k.startswith("clair3.") or k.startswith("shared")]:
        del sys.modules[k]


def _printer(cv, pileup, indel, **changes):
    from clair3_amd.vcf_rows import RowPrinter
    from tests.test_decode_dropin import config
    pr = RowPrinter(cv, config(cv, pileup, indel, **changes))
    assert pr.usable and pr._c is not None, "the C pass is not active"
    return pr


def _random_alt_info(rng, ref_base):
    """alt_info strings of every shape the lookups branch on: several alleles per kind, equal counts (the tie rules of max() and of
    the two sorts), lengths inside and outside 1..maximum_variant_length_that_need_infer, repeated keys (dict semantics), keys of
    other kinds, an empty dictionary"""
    depth = int(rng.integers(1, 120))
    if rng.random() < 0.05:
        return f"{depth}-" if rng.random() < 0.5 else f"{depth}"
    parts = []
    count = lambda: int(rng.integers(1, 6)) if rng.random() < 0.5 else int(rng.integers(1, 60))  # noqa: E731 (small range: ties happen)
    bases = lambda n: "".join("ACGT"[j] for j in rng.integers(0, 4, size=n))  # noqa: E731
    for _ in range(int(rng.integers(0, 9))):
        kind = "XIDR"[int(rng.integers(0, 4))] if rng.random() < 0.95 else "Q"
        if kind == "X":
            key = "X" + "ACGTN"[int(rng.integers(0, 5))] + ("" if rng.random() < 0.9 else bases(2))
        elif kind == "I":
            key = "I" + ref_base + bases(int(rng.integers(0, 22)) if rng.random() < 0.9 else int(rng.integers(40, 70)))
        elif kind == "D":
            key = "D" + bases(int(rng.integers(1, 22)) if rng.random() < 0.9 else int(rng.integers(40, 70)))
        else:
            key = kind + ref_base
        parts.append(f"{key} {count()}")
        if kind == "I" and len(key) > 45 and rng.random() < 0.7:  # neighbours of a long insertion: other alleles within ~10 % of its length (--enable_long_indel)
            for _ in range(int(rng.integers(1, 4))):
                near = max(2, int((len(key) - 1) * (1 + rng.uniform(-0.14, 0.14))))
                parts.append(f"I{ref_base}{bases(near - 1)} {count()}")
        if rng.random() < 0.08:
            parts.append(f"{key} {count()}")  # the same key again: the last count wins, the key keeps its place
    return f"{depth}-" + " ".join(parts) + (" " if rng.random() < 0.7 else "")


def _asked_rows(rng, n, indel, cv):
    """rows whose decoder columns ASK for a class / entry directly (the device derives them from the probabilities; here every class and
    entry is drawn at random so that every branch of every lookup is met), with random alt_info strings"""
    from clair3_amd import decode as dec
    width = 90 if indel else 24
    y = np.zeros((n, width + dec.DECODE_COLS), np.float32)
    y[:, :width] = rng.random((n, width), dtype=np.float32)
    pos, alt = [], []
    for i in range(n):
        seq = "".join("ACGT"[j] for j in rng.integers(0, 4, size=33))
        if rng.random() < 0.06:  # IUPAC / lower-case / unknown centre bases
            seq = seq[:16] + "RYKMNacgtX"[int(rng.integers(0, 10))] + seq[17:]
        contig = ("chr%d" % rng.integers(1, 23)) if rng.random() < 0.9 else "HLA-A*01:01:01:01"  # a contig name with colons
        pos.append(f"{contig}:{int(rng.integers(1, 10 ** 8))}:{seq}")
        alt.append(_random_alt_info(rng, seq[16]))
        cls = int(rng.integers(0, 10))
        cols = y[i, width:]
        cols[0:9] = rng.permutation(9).astype(np.float32) / 16 + rng.random(dtype=np.float32) / 64  # nine distinct maxima
        if rng.random() < 0.03:
            cols[int(rng.integers(0, 9))] = cols[int(rng.integers(0, 9))]  # now and then a shared maximum
        cols[9:13] = rng.random(4, dtype=np.float32)
        cols[13:22] = [int(rng.integers(0, dec._CLASS_LEN[indel][k + 1])) for k in range(9)]
        cols[23:27] = cls
        q = rng.random()
        if q < 0.1:  # probabilities at the ends of the range (QUAL 0 and the 1e-10 guard of quality_score_from)
            cols[0:13] = np.float32(1.0) if q < 0.05 else np.float32(1e-12)
            cols[0:9] += (-np.arange(9) * 1e-7).astype(np.float32) if q < 0.05 else (np.arange(9) * 1e-13).astype(np.float32)
    return pos, alt, y


@pytest.mark.parametrize("indel", [True, False])
@pytest.mark.parametrize("changes", [{}, {"quality_score_for_pass": 12}, {"is_show_reference": False, "keep_iupac_bases": True}, {"gvcf": True},
                                     {"gvcf": True, "keep_iupac_bases": True, "quality_score_for_pass": 8}, {"is_haploid_precise_mode_enabled": True},
                                     {"is_haploid_sensitive_mode_enabled": True, "gvcf": True},
                                     {"enable_long_indel": True, "maximum_variant_length_that_need_infer": 100000},
                                     {"enable_long_indel": True, "maximum_variant_length_that_need_infer": 100000, "gvcf": True, "quality_score_for_pass": 5}])
def test_every_class_and_entry_against_the_python_path(indel, changes, ref):
    """rows that ask for every class / entry over random alt_info dictionaries: wherever the C pass prints a row (or says the reference
    prints nothing), the per-row Python path -- the reference's own lookup functions -- gives the same text"""
    cv, _ = ref
    rng = np.random.default_rng(20 + indel)
    pr = _printer(cv, not indel, indel, **changes)
    n = 6000
    pos, alt, y = _asked_rows(rng, n, indel, cv)
    got = pr._rows_c(pos, alt, y)
    assert got is not None
    texts, todo = got
    want = pr._rows_py(pos, alt, y)
    back = set(todo)
    by_class = {}
    for i in range(n):
        if i in back:
            continue
        assert texts[i] == want[i], (i, pos[i], alt[i], y[i, -31:].tolist(), texts[i], want[i])
        c = int(y[i, -8])
        by_class[c] = by_class.get(c, 0) + 1
    assert set(by_class) == set(range(10)) and min(by_class.values()) > 50, by_class  # every class is printed by the C pass
    if any(k.startswith("is_haploid") for k in changes):  # (:1191-1199, :1327-1329)
        printed = [t for i, t in enumerate(texts) if i not in back and t]
        assert printed and all(t.split("\t")[9].split(":")[0] in ("0", "1") for t in printed)
        if changes.get("is_haploid_precise_mode_enabled"):  # the heterozygous classes are not printed (a row whose first candidate was
            quiet = sum(texts[i] is None for i in range(n) if i not in back)  # rejected may end in another class: counted, not matched)
            assert quiet > 0.2 * (n - len(back)), quiet
    assert len(todo) < 0.7 * n  # (random dictionaries rarely offer what a random entry asks for: rejected first candidates go back)
    # and the public entry gives the same list
    assert pr.rows(pos, alt, y) == want


@pytest.mark.parametrize("indel", [True, False])
@pytest.mark.parametrize("noise", [0.0, 0.3])
@pytest.mark.parametrize("gvcf", [False, True])
def test_rows_with_a_story_through_the_whole_decoder(indel, noise, gvcf, ref):
    """consistent rows (tests/decode_rows.py: what a trained model and a real pileup hand the decoder): the rebound batch_output with
    the C pass == without it == the UNPATCHED reference decoder, character for character; and nearly every row is printed in C"""
    from tests.test_decode_dropin import config, widen
    cv, unpatched = ref
    n = 3000
    pos, alt, y, _ = consistent_rows(n, seed=5, indel=indel, noise=noise)
    cfg = config(cv, not indel, indel, gvcf=gvcf)  # gvcf: rows with the PL field (compute_PL :1397-1454), printed by the same pass
    yw = widen(y, indel)
    cv.batch_output(pos[:1], alt[:1], yw[:1], cfg, None)  # (makes the configuration's printer)
    pr = cv._c3hip_row_printers[(cfg, id(cv.param))]
    assert pr._c is not None
    pr.by_c = pr.taken = pr.retried = pr.handed_back = 0
    text = cv.batch_output(pos, alt, yw, cfg, None)
    share = pr.by_c / max(pr.taken + pr.handed_back, 1)
    assert pr.taken + pr.handed_back == n
    assert text == unpatched(pos, alt, y, cfg, None)
    keep, pr._c = pr._c, None
    try:
        assert cv.batch_output(pos, alt, yw, cfg, None) == text
    finally:
        pr._c = keep
    assert (text.count("GT:GQ:DP:AD:AF:PL") == text.count("\n")) if gvcf else "PL" not in text
    print(f"indel={indel} noise={noise} gvcf={gvcf}: {100 * share:.1f} % of the rows printed by c3_vcf_rows")
    assert share > (0.95 if noise == 0 else 0.6)


@pytest.mark.parametrize("indel", [True, False])
@pytest.mark.parametrize("sharp", [0.0, 2.0, 6.0])
def test_the_walk_over_rejected_candidates(indel, sharp, ref):
    """flat and half-peaked probability rows over alt_info strings that offer little (tests/test_decode_dropin.alt_infos): most rows reject
    their first candidate, many reject dozens.  The loop's later passes run in C (class lists in chain order, stable walk by falling
    probability, the tie rule): the text equals the per-row Python path's and the UNPATCHED reference decoder's, and the walk's share
    is reported.  C3HIP_ROWS_C_WALK=0 hands those rows back instead."""
    from tests.test_decode_dropin import alt_infos, config, widen
    cv, unpatched = ref
    rng = np.random.default_rng(int(10 * sharp) + indel)
    n = 1500
    width = 90 if indel else 24
    y = np.zeros((n, width), np.float32)
    for lo, hi in ((0, 21), (21, 24)) + (((24, 57), (57, 90)) if indel else ()):
        logits = rng.normal(0.0, 1.0, size=(n, hi - lo))
        logits[np.arange(n), rng.integers(0, hi - lo, size=n)] += sharp
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        y[:, lo:hi] = (e / e.sum(axis=1, keepdims=True)).astype(np.float32)
    pos, alt = alt_infos(n, seed=int(sharp) + 3)
    cfg = config(cv, not indel, indel)
    pr = _printer(cv, not indel, indel)
    yw = widen(y, indel)
    texts = pr.rows(pos, alt, yw)
    walked, back = pr.retried, pr.handed_back
    assert texts == pr._rows_py(pos, alt, yw)
    from clair3_amd.vcf_rows import FALLBACK
    mine = "".join(t for t in texts if t is not None and t is not FALLBACK)
    if not any(t is FALLBACK for t in texts):
        assert mine == unpatched(pos, alt, y, cfg, None)
    assert cv.batch_output(pos, alt, yw, cfg, None) == unpatched(pos, alt, y, cfg, None)
    print(f"indel={indel} sharp={sharp}: {pr.by_c} of {n} rows printed in C, {walked} of them after rejected candidates; {back} left to the reference")
    assert walked > (0.2 * n if sharp < 6 else 0) and pr.by_c > 0.9 * n
    keep = pr._c.walk
    pr._c.walk = 0
    try:
        got = pr._rows_c(pos, alt, yw)
        assert len(got[1]) >= walked and pr.rows(pos, alt, yw) == texts
    finally:
        pr._c.walk = keep


def test_texts_of_other_kinds_and_odd_inputs(ref):
    """bytes / numpy.bytes_ / str texts, trailing white space, a NUL or a non-ASCII byte in a text, numbers that are not plain digits,
    a depth of zero, an empty batch: printed identically or handed back, never wrong and never an exception"""
    cv, _ = ref
    rng = np.random.default_rng(7)
    pr = _printer(cv, False, True)
    pos, alt, y = _asked_rows(rng, 400, True, cv)
    want = pr._rows_py(pos, alt, y)
    as_bytes = [p.encode() for p in pos], [np.bytes_(a.encode()) for a in alt]
    assert pr.rows(as_bytes[0], as_bytes[1], y) == want
    assert pr.rows([p + "\n" for p in pos], [a + " \n" for a in alt], y) == want
    odd_alt = list(alt)
    odd_alt[0], odd_alt[1], odd_alt[2], odd_alt[3], odd_alt[4] = "0-XA 3", "12-XA +3", "1_0-XA 3", "20-XÄ 3 ", "20-XA  3"
    want_odd = None
    try:
        want_odd = pr._rows_py(pos, odd_alt, y)
    except Exception:
        pass  # (the reference itself may refuse such a text: then the C pass must have handed the row back, which rows() shows by raising too)
    if want_odd is not None:
        assert pr.rows(pos, odd_alt, y) == want_odd
    texts, todo = pr._rows_c(pos, odd_alt, y)
    assert {0, 1, 2, 3, 4} <= set(todo)
    assert pr._rows_c([pos[0], "a\0b:1:" + "A" * 33], [alt[0], alt[1]], y[:2]) is None  # a NUL inside a text
    # more distinct alleles than the C pass keeps per row (192): handed back, printed by the per-row path, same text as without the pass
    many = "900-" + " ".join(f"I{pos[5].split(':')[-1][16]}{'ACGT'[i % 4]}{'A' * (i // 4)} {1 + i % 7}" for i in range(250)) + " XA 40 XC 7 "
    big_alt = list(alt)
    big_alt[5] = many
    texts, todo = pr._rows_c(pos, big_alt, y)
    assert 5 in todo and pr.rows(pos, big_alt, y) == pr._rows_py(pos, big_alt, y)
    # a very long contig name and a very long allele: the output buffer is sized from the texts
    long_pos = list(pos)
    long_pos[7] = "contig_" + "x" * 5000 + ":" + ":".join(pos[7].split(":")[-2:])
    long_alt = list(alt)
    long_alt[7] = "50-I" + pos[7].split(":")[-1][16] + "ACGT" * 40 + " 30 D" + "GATTACA" * 30 + " 12 XT 5 "
    assert pr.rows(long_pos, long_alt, y) == pr._rows_py(long_pos, long_alt, y)
    assert pr.rows([], [], y[:0]) == []
    # a strided view of the rows (the loop hands slices of its shared-memory block)
    big = np.zeros((len(y), y.shape[1] + 7), np.float32)
    big[:, :y.shape[1]] = y
    assert pr.rows(pos, alt, big[:, :y.shape[1]]) == want


def test_quality_follows_the_numpy_in_use(ref):
    """QUAL is quality_score_from (:375-381) of the float32 maximum: float32 arithmetic under numpy >= 2, double before.  The C pass is told
    which rule the running numpy follows and must print what the reference's function returns for probabilities across the range,
    including the ones right at the rounding boundaries of two decimals"""
    cv, _ = ref
    pr = _printer(cv, False, True)
    rng = np.random.default_rng(11)
    p = np.concatenate([rng.random(20000, dtype=np.float32), np.float32(1.0) - rng.random(5000, dtype=np.float32) * np.float32(1e-3),
                        rng.random(5000, dtype=np.float32) * np.float32(1e-4), np.array([0.0, 1.0, 0.5, 1e-10, 1e-38], np.float32)])
    n = len(p)
    y = np.zeros((n, 90 + 31), np.float32)
    cols = y[:, 90:]
    cols[:, 0] = p            # class 1 holds the maximum p ...
    cols[:, 1:9] = -1.0       # (... alone)
    cols[:, 23:27] = 1
    cols[:, 13] = 0           # homo SNP "AA"
    seq = "C" * 33
    pos = [f"chr1:{i + 1}:{seq}" for i in range(n)]
    alt = ["30-XA 12 RC 18"] * n
    texts, todo = pr._rows_c(pos, alt, y)
    assert not todo
    for i in range(n):
        q = cv.quality_score_from(p[i])
        f = texts[i].split("\t")
        assert f[5] == "%.2f" % q and f[9].split(":")[1] == "%d" % q, (i, float(p[i]), q, texts[i])


def test_genotype_likelihoods_follow_the_numpy_in_use(ref):
    """--gvcf: the PL field is compute_PL (:1397-1454) of the row's 21-genotype and zygosity probabilities -- float32 products; sum, division and
    the 1e-8 guard in float32 under numpy >= 2, in double before (Python's sum() starts from the int 0).  The C pass under the rule of the
    running numpy must print what the reference's function returns -- on rows whose likelihoods sit at the ceil() boundaries too --, and
    under the other rule what a restatement of that rule gives"""
    import math
    cv, _ = ref
    pr = _printer(cv, False, True, gvcf=True)
    rng = np.random.default_rng(17)
    n = 20000
    y = np.zeros((n, 90 + 31), np.float32)
    y[:, :24] = rng.random((n, 24), dtype=np.float32) ** np.float32(4.0)   # peaked, like soft-max rows
    y[: n // 4, :24] = np.float32(2.0) ** -rng.integers(0, 30, size=(n // 4, 24)).astype(np.float32)  # powers of two: exact ratios, ceil() of integers
    cols = y[:, 90:]
    kinds = rng.integers(0, 4, size=n)
    pos, alt = [], []
    for i in range(n):
        cols[i, 0:9] = -1.0
        if kinds[i] == 0:    # homo SNP C -> A: "AA" of the homo list
            cols[i, 0], cols[i, 23:27], cols[i, 13] = 0.5, 1, 0
            a = "30-XA 12 RC 18"
        elif kinds[i] == 1:  # hetero SNP with two new bases: A,G -> six genotypes
            cols[i, 1], cols[i, 23:27], cols[i, 14] = 0.5, 2, 1
            a = "30-XA 12 XG 9"
        elif kinds[i] == 2:  # homo insertion
            cols[i, 2], cols[i, 23:27], cols[i, 15] = 0.5, 3, 1
            a = "30-ICTT 12 RC 18"
        else:                # homo deletion
            cols[i, 3], cols[i, 23:27], cols[i, 16] = 0.5, 4, 1
            a = "30-DGG 12 RC 18"
        pos.append(f"chr1:{i + 1}:" + "C" * 16 + "C" + "GG" + "C" * 14)
        alt.append(a)
    texts, todo = pr._rows_c(pos, alt, y)
    assert not todo
    six = 0
    for i in range(n):
        f = texts[i].rstrip("\n").split("\t")
        assert f[8] == "GT:GQ:DP:AD:AF:PL"
        want = cv.compute_PL(None, y[i, 21:24], y[i, :21], f[3], f[4])
        assert f[9].split(":")[-1] == ",".join(str(v) for v in want), (i, f, want)
        six += len(want) == 6
    assert six > n // 8
    # the other rule, restated: float32 products, everything else in double
    keep = pr._c.f32_arith
    pr._c.f32_arith = 0 if keep else 1
    try:
        texts, todo = pr._rows_c(pos, alt, y)
    finally:
        pr._c.f32_arith = keep
    f32 = np.float32
    for i in range(0, n, 7):
        f = texts[i].rstrip("\n").split("\t")
        alts = f[4].split(",")
        ref_b = f[3]
        all_base = [ref_b] + alts
        genos = [[0, 0], [0, 1], [1, 1]] if len(alts) == 1 else [[0, 0], [0, 1], [1, 1], [0, 2], [1, 2], [2, 2]]
        like = []
        for g0, g1 in genos:
            lab = cv.mix_two_partial_labels(cv.partial_label_from(ref_b, all_base[g0]), cv.partial_label_from(ref_b, all_base[g1]))
            z = 0 if g0 == g1 == 0 else 1 if g0 == g1 else 2
            like.append(f32(y[i, cv.gt21_enum_from_label(lab)]) * f32(y[i, 21 + z]))
        if keep:   # the running numpy is >= 2: the other rule is the double one
            total = 0.0
            for v in like:
                total += float(v)
            xs = [float(v) / total + 1e-8 for v in like]
        else:
            total = f32(0)
            for v in like:
                total = f32(total + v)
            xs = [float(f32(f32(v / total) + f32(1e-8))) for v in like]
        pls = [-10 * (math.log(v) / math.log(10.0)) for v in xs]
        want = [int(math.ceil(v - min(pls))) for v in pls]
        assert f[9].split(":")[-1] == ",".join(str(v) for v in want), (i, f, want)


def test_rows_per_second_of_one_decode_core(ref):
    """not a gate on speed (a loaded CI host may be slow) beyond "clearly faster": the rate of the row printer with and without the C
    pass on consistent full-alignment rows, printed for the record"""
    from tests.test_decode_dropin import config, widen
    cv, _ = ref
    n = 4000
    pos, alt, y, _ = consistent_rows(n, seed=3, indel=True, noise=0.05)
    cfg = config(cv, False, True)
    yw = widen(y, True)
    cv.batch_output(pos, alt, yw, cfg, None)
    pr = cv._c3hip_row_printers[(cfg, id(cv.param))]

    def best(reps=5):
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            cv.batch_output(pos, alt, yw, cfg, None)
            t.append(time.perf_counter() - t0)
        return n / min(t)

    with_c = best()
    keep, pr._c = pr._c, None
    try:
        without = best()
    finally:
        pr._c = keep
    print(f"row printer, one core, 5 % of the rows reject their first candidate: {without:,.0f} rows/s per-row Python, {with_c:,.0f} rows/s with c3_vcf_rows")
    assert with_c > 1.5 * without
