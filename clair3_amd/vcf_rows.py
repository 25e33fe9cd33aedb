"""SURVEY 8f N1, the rest of the decoder: the VCF row of a candidate whose FIRST decision stands, printed without the
reference's per-row machinery.

With decoder columns behind every probability row (include/c3hip.h, ``model.decode_columns()``) the first pass of
``output_from`` (clair3/CallVariants.py:722-751) is known for the whole batch at once (``decode.first_decisions``): the
class it settles on, the winning entry of that class, QUAL, the homo-reference early exit.  What is left per row is the
allele lookup in the row's alt_info string and the row's text.  ``RowPrinter.rows`` does exactly that.  Where the
reference rejects the candidate and loops (``probabilities[idx] = 0; continue``, :760 ff.: the reads do not offer the
allele) it walks the same sequence of candidates the loop would meet -- every entry of the nine class lists above the
homo-reference probability, by falling probability, ties in the order of the loop's if / elif chain and of ``.index()`` --
instead of re-scanning ten lists per rejected candidate.  What it does not fully understand goes to the reference's own
``output_with`` (on the same decoder columns, through ``decode.install_decoder``'s list look-alikes): a maximum shared by
two classes, a reference base outside the IUPAC table, any output mode other than the plain VCF row (debug, gVCF,
haploid modes, long-indel read counting, ensemble output).  The allele lookups themselves ARE
the reference's functions (``find_alt_base``, ``insertion_bases_using_alt_info_from``, ``deletion_bases_using_alt_info_from``,
called through the module the drop-in is installed into), as are ``filtration_value_from``, ``genotype_string_from`` and
``convert_iupac_to_n`` and ``quality_score_from`` (called on the float32 scalar the reference would hold, so QUAL follows
the reference under either NumPy promotion rule); what is restated here is the control flow of ``output_from``
(:722-1016) and the read-count / formatting tail of ``output_with`` (:1176-1394), each block citing the lines it follows.

tests/test_decode_dropin.py: identical text to the unpatched ``batch_output`` on golden rows, on rows peaked on every
class with matching, partially matching and empty alt_info, and on adversarial rows; the share of rows that took the fast
path is reported.
"""
import numpy as np

from . import decode as dec

FALLBACK = object()  # "print this row with the reference's output_with"

_HETERO_CLASSES = frozenset((2, 5, 6, 7, 8, 9))  # the classes a haploid-precise run drops (:1191-1196)
_GENOTYPE_OF_CLASS = (0, 1, 2, 1, 1, 2, 2, 2, 2, None)  # homo_reference / homo_variant / hetero_variant per class (:1204-1209)
# rank of a class in output_from's if / elif chain (:748, :758, :778, :800, :830, :879, :897, :924, :977): among classes that
# hold the same maximum the first of the chain is the one whose candidate is looked up (and zeroed on rejection)
_CHAIN_RANK = np.array([0, 0, 1, 2, 5, 3, 4, 6, 7, 8])
# class and index of every entry of the nine lists laid end to end, and its rank in the chain
_KLASS = {indel: np.concatenate([np.full(n, k) for k, n in enumerate(dec._CLASS_LEN[indel]) if k]) for indel in (True, False)}
_INDEX = {indel: np.concatenate([np.arange(n) for k, n in enumerate(dec._CLASS_LEN[indel]) if k]) for indel in (True, False)}
_RANK = {indel: _CHAIN_RANK[_KLASS[indel]] for indel in (True, False)}
# rows without indel lengths (clair3/CallVariants.py:526-566): every entry is ONE float32 product zygosity x gt21 (commutative,
# so one gather-multiply gives the 23 of them bit for bit): the zygosity / gt21 index of every entry in list order
_Z24 = np.array([1] * 4 + [2] * 6 + [1, 1] + [2] * 4 + [2] + [2] * 4 + [2, 2])
_G24 = np.array(list(dec.HOMO_SNP_GT21) + list(dec.HETERO_SNP_GT21) + [15, 10] + [16, 17, 18, 19] + [15] + [11, 12, 13, 14] + [10, 20])
# the entries of the nine lists in the order the loop's if / elif chain and .index() break ties: (chain rank, index).  A STABLE
# sort of a row's values, taken in this order, by falling probability is the walk of output_from's loop
_CHAIN = {indel: np.lexsort((_INDEX[indel], _RANK[indel])) for indel in (True, False)}
_KLASS_C = {indel: _KLASS[indel][_CHAIN[indel]] for indel in (True, False)}
_INDEX_C = {indel: _INDEX[indel][_CHAIN[indel]] for indel in (True, False)}


def class_lists_of_rows(y, indel):
    """dec.class_list for classes 1..9 laid end to end (the order of _KLASS / _INDEX), for ALL rows of ``y`` at once: (R, 804)
    float32 with the indel-length heads, (R, 23) without.  The same float32 products in the same order and association as
    clair3/CallVariants.py:526-659 (element-wise float32 multiplication is the same IEEE operation on a column as on a scalar), so
    row r equals the concatenation of dec.class_list(k, ...) of that row bit for bit (tests/test_decode_dropin.py)."""
    y = np.asarray(y, dtype=np.float32)
    g, z = y[:, :21], y[:, 21:24]
    if not indel:
        return z[:, _Z24] * g[:, _G24]
    p1, p2 = y[:, 24:57], y[:, 57:90]
    hv, ht = z[:, 1], z[:, 2]
    o = 16  # VariantLength.index_offset
    c = lambda v: v[:, None]  # noqa: E731
    v0 = p1[:, o] * p2[:, o]
    return np.concatenate([
        c(v0 * hv) * g[:, dec._HS],                                                  # 1 :579-581
        c(v0 * ht) * g[:, dec._TS],                                                  # 2 :582-584
        p1[:, o + dec._I16] * p2[:, o + dec._I16] * c(hv * g[:, 15]),                # 3 :303-308, :587-590
        p1[:, o - dec._I16] * p2[:, o - dec._I16] * c(hv * g[:, 10]),                # 4 :331-336, :613-616
        (c(p1[:, o]) * p2[:, o + dec._L4]) * g[:, 16 + dec._B4] * c(ht),             # 5 :311-316, :600-606
        p1[:, o + dec._II] * p2[:, o + dec._IJ] * c(ht * g[:, 15]),                  # 6 :318-328
        (p1[:, o - dec._L4] * c(p2[:, o])) * g[:, 11 + dec._B4] * c(ht),             # 7 :339-345, :627-633
        p1[:, o - dec._DI] * p2[:, o - dec._DJ] * c(ht * g[:, 10]),                  # 8 :348-359
        p1[:, o - dec._XI] * p2[:, o + dec._XJ] * c(ht * g[:, 20]),                  # 9 :362-371
    ], axis=1)


_CLASS_START = np.cumsum((0,) + dec._CLASS_LEN[True][1:])  # start of class k's list, laid end to end, at [k - 1]
_PLAIN = {ord(ch): None for ch in "ACGTNacgtn,."}  # what convert_iupac_to_n leaves alone (shared/utils.py:27-40)


def _text(x):
    """chr_pos_seq / alt_info as str (:1127-1131, :1145-1148)"""
    if type(x) == np.memmap:
        return x[0].decode()
    if type(x) == np.bytes_ or type(x) == bytes:
        return x.decode()
    return x


def _bytes(x):
    """chr_pos_seq / alt_info as bytes for the C pass (the same three kinds _text accepts)"""
    if type(x) == np.memmap:
        return bytes(x[0])
    if type(x) == np.bytes_ or type(x) == bytes:
        return bytes(x)
    return x.encode()


def _joined(texts, n):
    """the first n texts NUL-separated as bytes: one join when they are all str or all bytes (what a worker's lists are), else per item"""
    seq = texts if len(texts) == n and type(texts) is list else [texts[i] for i in range(n)]
    if n and type(seq[0]) is str:
        try:
            return "\0".join(seq).encode()
        except TypeError:
            pass
    elif n and type(seq[0]) is bytes:
        try:
            return b"\0".join(seq)
        except TypeError:
            pass
    return b"\0".join([_bytes(x) for x in seq])


class _Lookups:
    """The allele lookups of one row: the reference's own functions on the row's alt_info dictionary, each distinct question
    asked once (they depend on the dictionary and a length / a base only, and a row that rejects many candidates asks the
    same few again and again)."""
    __slots__ = ("cv", "d", "infer", "memo")

    def __init__(self, cv, d, infer):
        self.cv, self.d, self.infer, self.memo = cv, d, infer, {}

    def find_alt_base(self, alt=None):
        k = ("x", alt)
        if k not in self.memo:
            self.memo[k] = self.cv.find_alt_base(self.d, alt)
        return self.memo[k]

    def ins(self, propose, ignore="", multi=False):
        k = ("i", propose, ignore, multi)
        if k not in self.memo:
            self.memo[k] = self.cv.insertion_bases_using_alt_info_from(
                alt_info_dict=self.d, propose_insertion_length=propose, maximum_insertion_length=self.infer,
                insertion_bases_to_ignore=ignore, return_multi=multi)
        return self.memo[k]

    def dele(self, propose, ignore="", multi=False):
        k = ("d", propose, ignore, multi)
        if k not in self.memo:
            self.memo[k] = self.cv.deletion_bases_using_alt_info_from(
                alt_info_dict=self.d, propose_deletion_length=propose, maximum_deletion_length=self.infer,
                deletion_bases_to_ignore=ignore, return_multi=multi)
        return self.memo[k]


class RowPrinter:
    """One per (reference module, output_config).  ``usable`` is False when the configuration asks for anything but the
    plain VCF row; ``rows`` then must not be called."""

    def __init__(self, cv, output_config):
        c = output_config
        self.cv, self.cfg = cv, c
        self.usable = not (c.is_debug or c.is_output_for_ensemble or c.input_probabilities)
        self.width = 90 if c.add_indel_length else 24
        self.flank = cv.param.flankingBaseNum
        G = cv.Genotype
        self.gt = [cv.genotype_string_from(g) for g in (G.homo_reference, G.homo_variant, G.hetero_variant)]
        self.gt_multi = cv.genotype_string_from(G.hetero_variant_multi)
        self.info = "P" if c.pileup else "F"
        self.max_len = cv.VariantLength.max
        self.taken = self.retried = self.handed_back = 0  # rows printed here / of those after rejections / rows left to output_with
        self.by_c = 0  # rows whose text came from the one-pass C printer (c3_vcf_rows)
        self._dead_cache = {}
        self._c = self._c_config() if self.usable else None

    # ------------------------------------------------------------------------------------------------ the one-pass C printer
    def _c_config(self):
        """c3_rows_config for libc3hip's c3_vcf_rows (csrc/c3_rows.h: the common rows of a batch printed in one pass of plain host
        code), or None when the library is not there / C3HIP_ROWS_C=0 / a constant of the reference is not what the C side was
        written for -- the per-row Python below then prints every row, as it did until round 5."""
        import os
        if os.environ.get("C3HIP_ROWS_C", "1").strip().lower() in ("0", "false", "no", "off"):
            return None
        try:
            from . import _lib
            L = _lib.lib()
        except Exception:
            return None
        cv, c = self.cv, self.cfg
        if self.max_len != dec.MAX_LEN or list(cv.HOMO_SNP_LABELS) != ["AA", "CC", "GG", "TT"] or \
                list(cv.HETERO_SNP_LABELS) != ["AC", "AG", "AT", "CG", "CT", "GT"] or any(len(g) > 7 for g in self.gt + [self.gt_multi]):
            return None
        qs = c.quality_score_for_pass
        if qs is not None and not isinstance(qs, (int, float)):
            return None
        cf = _lib.RowsConfig()
        cf.width, cf.flank, cf.show_reference, cf.keep_iupac = self.width, int(self.flank), int(bool(c.is_show_reference)), int(bool(c.keep_iupac_bases))
        cf.has_qs_pass, cf.qs_pass = int(qs is not None), float(qs if qs is not None else 0.0)
        cf.pileup, cf.max_len, cf.infer = int(bool(c.pileup)), int(self.max_len), int(c.maximum_variant_length_that_need_infer)
        cf.phred_trans = float(cv.Phred_Trans)
        # quality_score_from (:375-381) computes (1.0 - p) on a numpy float32 scalar: float32 arithmetic under numpy >= 2, double before
        cf.f32_arith = int((1.0 - np.float32(0.25)).dtype == np.float32)
        cf.walk = int(os.environ.get("C3HIP_ROWS_C_WALK", "1").strip().lower() not in ("0", "false", "no", "off"))
        if c.gvcf:  # the PL field (compute_PL :1397-1454): the label tables and the base dictionary the C side was written for
            labels = ["AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT", "DelDel", "ADel", "CDel", "GDel", "TDel", "InsIns", "AIns", "CIns",
                      "GIns", "TIns", "InsDel"]
            try:
                same = [cv.gt21_enum_from_label(x) for x in labels] == list(range(21)) and \
                    "".join(cv.BASE2ACGT[b] for b in "ACGTURYSWKMBDHVN") == "ACGTTACCAGACAAAA" and len(cv.BASE2ACGT) == 16
            except (KeyError, AttributeError):
                same = False
            if not same:
                return None
        cf.gvcf = int(bool(c.gvcf))
        cf.haploid = int(bool(c.is_haploid_precise_mode_enabled)) | (int(bool(c.is_haploid_sensitive_mode_enabled)) << 1)
        if c.enable_long_indel:  # get_long_indel_read_count reads three constants of the module's param at call time (:390-395)
            prm = cv.param
            if not (isinstance(prm.maximum_variant_length_that_need_infer, int) and isinstance(prm.long_indel_distance_proportion, float)):
                return None
            cf.long_indel = int(not prm.cal_precise_long_indel_af)
            cf.long_infer, cf.long_prop = int(prm.maximum_variant_length_that_need_infer), float(prm.long_indel_distance_proportion)
        for k, g in enumerate(self.gt + [self.gt_multi]):
            cf.gt[k].value = g.encode()
        self._lib = L
        return cf

    def _c_pass(self, batch_chr_pos_seq, alt_info_list, batch_Y):
        """One call of c3_vcf_rows over the batch -> (text of all printed rows in row order, offsets (n + 1), status (n)) or None when
        the pass could not run at all."""
        n = len(batch_chr_pos_seq)
        y = batch_Y
        if not (isinstance(y, np.ndarray) and y.dtype == np.float32 and y.ndim == 2 and y.strides[1] == 4 and y.strides[0] % 4 == 0
                and len(alt_info_list) >= n):
            return None
        try:
            pos_b = _joined(batch_chr_pos_seq, n)
            alt_b = _joined(alt_info_list, n)
        except (TypeError, AttributeError, UnicodeError):
            return None
        if pos_b.count(b"\0") != n - 1 or alt_b.count(b"\0") != n - 1:
            return None  # a NUL inside a text: not this path's business
        cap = 192 * n + 2 * (len(pos_b) + len(alt_b)) + 4096  # (a row is its contig, two alleles out of its alt_info and ~60 bytes of fields)
        out = np.empty(cap, np.uint8)
        off = np.empty(n + 1, np.int64)
        status = np.empty(n, np.uint8)
        rc = self._lib.c3_vcf_rows(self._c, n, pos_b, len(pos_b), alt_b, len(alt_b), y.ctypes.data, y.strides[0] // 4, out.ctypes.data, cap,
                                   off.ctypes.data, status.ctypes.data)
        if rc != 0:
            return None
        return out[: int(off[n])].tobytes().decode("ascii"), off, status

    def _rows_c(self, batch_chr_pos_seq, alt_info_list, batch_Y):
        """-> (texts, todo): texts[i] = the row's text / None (prints nothing) where the C pass printed it, todo = the rows it handed
        back; None when the pass could not run at all."""
        got = self._c_pass(batch_chr_pos_seq, alt_info_list, batch_Y)
        if got is None:
            return None
        text, off, status = got
        n = len(status)
        o = off.tolist()
        st = status.tolist()
        texts = [None] * n
        todo = []
        for i in range(n):
            if st[i] == 1:
                todo.append(i)
            elif o[i + 1] > o[i]:
                texts[i] = text[o[i]:o[i + 1]]
        self.retried += st.count(2)
        return texts, todo

    def batch_text(self, batch_chr_pos_seq, alt_info_list, batch_Y, print_with_reference):
        """The VCF text of the whole batch as ONE string, rows in order: the slice of the C pass's buffer between two rows it handed
        back is taken as it is (no per-row strings), handed-back rows go through the per-row path, and what that leaves
        (FALLBACK) through ``print_with_reference(i)`` (the reference's own output_with).  What batch_output writes."""
        n = len(batch_chr_pos_seq)
        got = self._c_pass(batch_chr_pos_seq, alt_info_list, batch_Y) if self._c is not None and n else None
        if got is None:
            texts = self._rows_py(batch_chr_pos_seq, alt_info_list, batch_Y)
            parts = [print_with_reference(i) if t is FALLBACK else t for i, t in enumerate(texts)]
            return "".join([t for t in parts if t is not None])
        text, off, status = got
        todo = np.flatnonzero(status == 1)
        self.retried += int(np.count_nonzero(status == 2))
        self.by_c += n - len(todo)
        self.taken += n - len(todo)
        if len(todo) == 0:
            return text
        rest = self._rows_py([batch_chr_pos_seq[i] for i in todo], [alt_info_list[i] for i in todo], batch_Y[todo])
        parts, prev = [], 0
        for i, t in zip(todo.tolist(), rest):
            parts.append(text[int(off[prev]):int(off[i])])
            if t is FALLBACK:
                t = print_with_reference(i)
            if t is not None:
                parts.append(t)
            prev = i + 1
        parts.append(text[int(off[prev]):int(off[n])])
        return "".join(parts)

    # ------------------------------------------------------------------------------------------------ one pass of output_from
    def _alleles(self, cls, pos, ref, look):
        """(reference_base, alternate_base) the loop of output_from is left with after the pass for class ``cls`` / entry
        ``pos``, or None where it goes on to the next candidate.  The loop runs ``while reference_base is None or
        alternate_base is None`` (:721), so a ``continue`` that comes AFTER both were assigned ends it just like an accepted
        candidate does: a SNP whose looked-up base is the reference base (:754, :773), an ACGT+insertion whose SNP has no
        reads (:822-825), two equal insertions (:875-877), two deletions that fail the allele check (:973-975) all leave
        the loop with the alleles assigned so far -- restated here as the reference behaves, not as it reads.
        ``look``: the row's _Lookups."""
        cv = self.cv
        indel, cap = self.cfg.add_indel_length, self.max_len
        find_alt_base, ins, dele = look.find_alt_base, look.ins, look.dele
        if cls == 1:  # homo SNP (:748-756)
            lab = cv.HOMO_SNP_LABELS[pos]
            alt = lab[0] if lab[0] != ref else lab[1]
            _, alt = find_alt_base(alt)
            return None if alt is None else (ref, alt)  # alt == ref leaves the loop too (and output_with prints nothing, :1178)
        if cls == 2:  # hetero SNP (:758-775)
            lab = cv.HETERO_SNP_LABELS[pos]
            if lab[0] != ref and lab[1] != ref:
                bases, _ = find_alt_base()
                return None if len(bases) < 2 else (ref, ",".join(bases[:2]))
            alt = lab[0] if lab[0] != ref else lab[1]
            _, alt = find_alt_base(alt)
            return None if alt is None else (ref, alt)
        entry = dec.class_entry(cls, pos, indel)
        if cls == 3:  # homo insertion (:778-796)
            bases = ins((entry if entry and entry < cap else None))
            return None if len(bases) == 0 else (ref, bases)
        if cls == 5:  # hetero ACGT + insertion (:800-828)
            base, length = entry if indel else (entry, None)
            bases = ins((length if length and length < cap else None))
            if len(bases) == 0:
                return None
            if base != ref:
                snps, _ = find_alt_base()
                if len(snps) == 0:
                    return ref, bases  # :822-825: zeroed and `continue`d with both alleles assigned -- the loop ends here
                return ref, "{},{}".format(snps[0], bases)
            return ref, bases
        if cls == 6:  # two insertions (:830-877)
            pair = []
            if indel:
                l1, l2 = entry
                b1 = ins((l1 if l1 and l1 < cap else None))
                if len(b1):
                    b2 = ins((l2 if l2 and l2 < cap else None), b1)
                    if len(b2):
                        pair = [b1, b2]
            if len(pair) < 2:
                pair = ins(None, "", True)
            if len(pair) < 2:
                return None
            first, other = pair
            return (ref, first) if other == first else (ref, "{},{}".format(other, first))  # :869-877
        if cls == 4:  # homo deletion (:879-895)
            bases = dele((entry if entry and entry < cap else None))
            if len(bases) == 0:
                return None
            r = ref + bases
            return r, r[0]
        if cls == 7:  # hetero ACGT + deletion (:897-922)
            base, length = entry if indel else (entry, None)
            bases = dele((length if length and length < cap else None))
            if len(bases) == 0:
                return None
            r = ref + bases
            if base != r[0]:
                return r, "{},{}".format(r[0], base + r[1:])
            return r, r[0]
        if cls == 8:  # two deletions (:924-975)
            pair = []
            if indel:
                l1, l2 = sorted(entry, reverse=True)
                b1 = dele((l1 if l1 and l1 < cap else None))
                if len(b1) > 0:
                    b2 = dele((l2 if l2 and l2 < cap else None), b1)
                    if len(b2) > 0:
                        pair = [b1, b2] if len(b1) > len(b2) else [b2, b1]
            if len(pair) < 2:
                pair = dele(None, "", True)
            if len(pair) < 2:
                return None
            longer, other = pair
            r = ref + longer
            a1, a2 = r[0], r[0] + r[len(other) + 1:]
            if a1 != a2 and r != a1 and r != a2:
                return r, "{},{}".format(a1, a2)
            return r, a1  # :973-975
        # cls == 9: insertion and deletion (:977-1008)
        l1, l2 = entry if indel else (None, None)
        ibases = ins((l2 if l2 and l2 < cap else None))
        dbases = dele((l1 if l1 and l1 < cap else None))
        if len(ibases) == 0 or len(dbases) == 0:
            return None
        r = ref + dbases
        return r, "{},{}".format(r[0], ibases + r[1:])

    # ------------------------------------------------------------------------------------------------ the tail of output_with
    def _row(self, cls, ref, alt, prob, chromosome, position, depth, d, y_row=None):
        """the text of output_with (:1176-1394) for a row of class ``cls`` with alleles (ref, alt) and maximum probability
        ``prob`` (the float32 scalar), or None where it prints nothing"""
        cv, c = self.cv, self.cfg
        is_ref = cls == 0
        if (not c.is_show_reference and is_ref) or (not is_ref and ref == alt):  # :1176-1180
            return None
        multi = "," in str(alt)
        if c.is_haploid_precise_mode_enabled and cls in _HETERO_CLASSES:  # :1191-1196 (hetero SNP, ACGT + Ins, InsIns, ACGT + Del, DelDel, Ins and Del)
            return None
        elif c.is_haploid_sensitive_mode_enabled and multi:  # :1197-1199
            return None
        gt = self.gt_multi if multi else self.gt[_GENOTYPE_OF_CLASS[cls]]  # :1203-1211 (class 9 always carries two alleles)
        if c.is_haploid_precise_mode_enabled or c.is_haploid_sensitive_mode_enabled:  # :1327-1329 (placed here: nothing in between reads it)
            gt = "1" if "1" in gt else "0"
        snp, insd, deld, ref_count = {}, {}, {}, 0  # decode_alt_info (:1215-1230)
        for k, n in d.items():
            n = int(n)
            t = k[0]
            if t == "X":
                snp[k[1]] = n
            elif t == "I":
                insd[k[1:]] = n
            elif t == "D":
                deld[k[1:]] = n
            elif t == "R":
                ref_count = n
        ref_count = max(0, ref_count)
        supported, counts = 0, []
        # --enable_long_indel: the reads of alleles within 10 % of a long allele's length are counted with it (get_long_indel_read_count,
        # :383-402 -- the reference's own function; as the reference calls it, i.e. for a deletion WITHOUT is_del, where it finds nothing)
        long_ins = (lambda bases: cv.get_long_indel_read_count(alt_info=insd, proposed_ins_base=bases, is_del=False)) if c.enable_long_indel else (lambda bases: 0)
        long_del = (lambda n: cv.get_long_indel_read_count(alt_info=deld, propose_del_base_length=n)) if c.enable_long_indel else (lambda n: 0)
        if is_ref:  # :1236-1238
            supported, alt = ref_count, "."
        elif cls <= 2:  # SNPs (:1240-1246)
            for base in str(alt):
                if base == ",":
                    continue
                n = snp[base] if base in snp else 0
                supported += n
                counts.append(n)
        elif cls == 3 or cls == 6:  # insertions (:1247-1255)
            for bases in alt.split(","):
                n = (insd[bases] if bases in insd else 0) + long_ins(bases)
                supported += n
                counts.append(n)
        elif cls == 5:  # SNP + insertion (:1256-1270)
            snp_base = alt.split(",")[0][0] if multi else None
            bases = alt.split(",")[1] if multi else alt
            n_snp = (snp[snp_base] if snp_base in snp else 0) if multi else 0
            n_ins = (insd[bases] if bases in insd else 0) + long_ins(bases)
            supported = n_ins + n_snp
            if snp_base:
                counts.append(n_snp)
            counts.append(n_ins)
        elif cls == 4 or cls == 8:  # deletions (:1271-1288)
            if len(deld) > 0:
                if cls == 4:
                    bases = ref[1:] if len(ref) > 1 else None
                    supported = (deld[bases] if bases in deld else 0) + (long_del(len(bases)) if c.enable_long_indel else 0)
                    counts.append(supported)
                elif len(deld) > 1:
                    for bases in alt.split(","):
                        n_del = len(ref) - len(bases)
                        hit = [deld[k] for k in deld if len(k) == n_del]
                        n = (hit[0] if len(hit) > 0 else 0) + long_del(n_del)
                        counts.append(n)
                        supported += n
        elif cls == 7:  # SNP + deletion (:1289-1305)
            alts = alt.split(",")
            snp_base = (alts[1][0] if len(alts) > 1 else None) if multi else None
            n_snp = (snp[snp_base] if snp_base in snp else 0) if multi else 0
            bases = ref[1:] if len(ref) > 1 else None
            n_del = (deld[bases] if bases in deld else 0) + (long_del(len(bases)) if c.enable_long_indel else 0)
            supported = n_del + n_snp
            if snp_base:
                counts.append(n_snp)
            counts.append(n_del)
        else:  # insertion and deletion (:1306-1322)
            for bases in alt.split(","):
                n_del = len(ref) - len(bases)
                if n_del < 0:
                    ibases = bases[:-(len(ref) - 1)] if len(ref) > 1 else bases
                    n = (insd[ibases] if ibases in insd else 0) + long_ins(ibases)
                else:
                    hit = [deld[k] for k in deld if len(k) == n_del]
                    n = (hit[0] if len(hit) > 0 else 0) + long_del(n_del)
                counts.append(n)
                supported += n
        af = ((supported + 0.0) / depth) if depth != 0 else 0.0  # :1324-1326
        if af > 1:
            af = 1
        qual = cv.quality_score_from(prob)  # :1329
        filt = cv.filtration_value_from(quality_score_for_pass=c.quality_score_for_pass, quality_score=qual, is_reference=is_ref)
        if not c.keep_iupac_bases:  # :1342-1344; the function returns strings of A/C/G/T/N , . unchanged
            if ref.translate(_PLAIN):
                ref = cv.convert_iupac_to_n(ref)
            if alt.translate(_PLAIN):
                alt = cv.convert_iupac_to_n(alt)
        ad = str(ref_count) + (("," + ",".join([str(n) for n in counts])) if len(counts) else "")  # :1357-1360
        afs = "%.4f" % af if len(counts) <= 1 else ",".join(["%.4f" % (min(1.0, 1.0 * n / depth)) for n in counts])
        if c.gvcf:  # :1360-1378 -- the PL field; compute_PL (:1397-1454) is the reference's own, on the row's own float32 slices
            pls = ",".join([str(x) for x in cv.compute_PL(gt, y_row[21:24], y_row[:21], ref, alt)])
            return "%s\t%d\t.\t%s\t%s\t%.2f\t%s\t%s\tGT:GQ:DP:AD:AF:PL\t%s:%d:%d:%s:%s:%s\n" % (
                chromosome, position, ref, alt, qual, filt, self.info, gt, qual, depth, ad, afs, pls)
        return "%s\t%d\t.\t%s\t%s\t%.2f\t%s\t%s\tGT:GQ:DP:AD:AF\t%s:%d:%d:%s:%s\n" % (
            chromosome, position, ref, alt, qual, filt, self.info, gt, qual, depth, ad, afs)

    # ------------------------------------------------------------------------------------------------ a batch
    def rows(self, batch_chr_pos_seq, alt_info_list, batch_Y):
        """batch_Y: (B, 24|90 + DECODE_COLS) float32.  Returns a list with, per row, its VCF text, None (the reference prints
        nothing for it) or FALLBACK."""
        n = len(batch_chr_pos_seq)
        done = self._rows_c(batch_chr_pos_seq, alt_info_list, batch_Y) if self._c is not None and n else None
        if done is not None:
            # the common rows are printed; what was handed back -- rejected first candidates, shared maxima, odd texts -- goes through
            # the per-row path below as a batch of its own
            texts, todo = done
            self.by_c += n - len(todo)
            self.taken += n - len(todo)
            if todo:
                idx = np.asarray(todo)
                rest = self._rows_py([batch_chr_pos_seq[i] for i in todo], [alt_info_list[i] for i in todo], batch_Y[idx])
                for i, t in zip(todo, rest):
                    texts[i] = t
            return texts
        return self._rows_py(batch_chr_pos_seq, alt_info_list, batch_Y)

    def _rows_py(self, batch_chr_pos_seq, alt_info_list, batch_Y):
        """the per-row path (round 5's ``rows``): every row of the batch it is given"""
        n = len(batch_chr_pos_seq)
        parsed = [None] * n
        centre = bytearray(n)
        for i in range(n):  # :1127-1143
            info = _text(batch_chr_pos_seq[i]).rstrip().split(":")
            if len(info) == 3:
                chromosome, position, seq = info
            else:
                position, seq = info[-2], info[-1]
                chromosome = ":".join(info[:-2])
            ref = seq[self.flank if len(seq) > 1 else 0]
            parsed[i] = (chromosome, int(position), ref)
            o = ord(ref)
            centre[i] = o if o < 256 else 0
        cols = batch_Y[:, self.width:]
        b = np.frombuffer(bytes(centre), dtype=np.uint8)
        bi = dec._REF_BASE_INDEX[b].astype(np.int64)
        known = bi >= 0
        bi = np.where(known, bi, 0)
        r = np.arange(n)
        cls = cols[r, 23 + bi].astype(np.int64)
        k = np.maximum(cls, 1) - 1
        pos = cols[r, 13 + k].astype(np.int64).tolist()
        prob = np.where(cls > 0, cols[r, k], cols[r, 9 + bi])  # float32: the maximum of the first pass (:722-733) / all_pro[0] (:702-707)
        # a maximum that two classes share: output_with's flag chains are not output_from's (:1203-1209 against :748 ff.)
        shared = (cols[:, 0:9] == cols[r, k][:, None]).sum(axis=1) > 1
        fast = (known & ~((cls > 0) & shared)).tolist()
        cls = cls.tolist()
        out = [FALLBACK] * n
        walks = []
        for i in range(n):
            if not fast[i]:
                continue
            chromosome, position, ref = parsed[i]
            a = _text(alt_info_list[i]).rstrip().split("-")  # :1150-1154
            depth = int(a[0])
            seqs = (a[1] if len(a) > 1 else "").split(" ")
            d = dict(zip(seqs[::2], [int(item) for item in seqs[1::2]])) if len(seqs) else {}
            c = cls[i]
            acgt = "ACGT"[bi[i]]
            if c == 0:  # homo reference, early exit or not (:702-707, :735-740): both alleles are the A/C/G/T form of the base
                out[i] = self._row(0, acgt, acgt, prob[i], chromosome, position, depth, d, batch_Y[i])
                continue
            look = _Lookups(self.cv, d, self.cfg.maximum_variant_length_that_need_infer)
            alleles = self._alleles(c, pos[i], ref, look)
            if alleles is not None:
                out[i] = self._row(c, alleles[0], alleles[1], prob[i], chromosome, position, depth, d, batch_Y[i])
                continue
            walks.append((i, c, ref, look, chromosome, position, depth, d, acgt))
        if walks:
            # the rows whose first candidate the reads do not offer: their class lists in ONE pass over the batch, in chain order
            indel = self.cfg.add_indel_length
            idx = np.fromiter((w[0] for w in walks), dtype=np.int64, count=len(walks))
            values = class_lists_of_rows(batch_Y[idx, :self.width], indel)[:, _CHAIN[indel]]
            homo = cols[idx, 9 + bi[idx]]
            above = values > homo[:, None]
            # every row's walk order in ONE stable sort: entries above the homo-reference probability by falling probability (ties in
            # chain order), the others -- keyed +inf -- behind them; row j walks order[j, :count[j]].  (Rows that reject their first
            # candidate have most of their entries above it: sorting only those with one lexsort over the batch was twice as slow.)
            order = np.argsort(np.where(above, -values, np.float32(np.inf)), axis=1, kind="stable")
            count = above.sum(axis=1).tolist()
            for j, (i, c, ref, look, chromosome, position, depth, d, acgt) in enumerate(walks):
                found = self._next_candidate(values[j], order[j, :count[j]], c, pos[i], ref, look)
                if found is FALLBACK:
                    continue
                self.retried += 1
                if found is None:  # nothing above the homo-reference probability is offered by the reads (:735-740)
                    out[i] = self._row(0, acgt, acgt, homo[j], chromosome, position, depth, d, batch_Y[i])
                else:
                    c, alleles, p = found
                    out[i] = self._row(c, alleles[0], alleles[1], p, chromosome, position, depth, d, batch_Y[i])
        back = sum(1 for v in out if v is FALLBACK)
        self.handed_back += back
        self.taken += n - back
        return out

    def _dead(self, d):
        """Which entries of the nine lists laid end to end (indel-length rows) _alleles is CERTAIN to reject for a row whose
        alt_info dictionary is ``d``, from one pass over its keys.  What _alleles does per class (the line numbers there): a
        homo insertion / deletion and an ACGT + insertion / deletion are rejected exactly when the lookup for their length comes
        back empty; an insertion-and-deletion when either does; two insertions / two deletions are rejected exactly when the
        reads offer fewer than two alleles of that kind; SNPs are always left to _alleles.  And a lookup
        (clair3/CallVariants.py:117-201, no bases to ignore) comes back empty exactly when the reads hold neither an allele of the
        proposed length (+ the reference base for insertions; the last length, >= 16, proposes none: :783, :884) nor any allele of
        1 .. maximum_variant_length_that_need_infer bases.  Returns (dead in list order, dead in chain order)."""
        cap, infer = self.max_len, self.cfg.maximum_variant_length_that_need_infer
        ilen, dlen = [], []
        for key in d:
            t = key[0]
            if t == "I":
                ilen.append(len(key) - 1)
            elif t == "D":
                dlen.append(len(key) - 1)
        n_ins = sum(1 for L in ilen if 1 <= L <= infer)
        n_del = sum(1 for L in dlen if 1 <= L <= infer)
        # the answer depends on the dictionary only through "0, 1 or more alleles in range" per kind and -- with none in range -- the
        # lengths of the out-of-range ones: a handful of distinct cases per job, each worked out once
        key = (min(n_ins, 2), min(n_del, 2), tuple(sorted(set(ilen))) if n_ins == 0 else (), tuple(sorted(set(dlen))) if n_del == 0 else ())
        hit = self._dead_cache.get(key)
        if hit is not None:
            return hit
        lengths = np.arange(1, cap + 1)
        no_ins = np.full(cap, n_ins == 0)
        no_del = np.full(cap, n_del == 0)
        if n_ins == 0 and ilen:
            no_ins &= ~np.isin(lengths + 1, ilen)
            no_ins[cap - 1] = True
        if n_del == 0 and dlen:
            no_del &= ~np.isin(lengths, dlen)
            no_del[cap - 1] = True
        o = _CLASS_START
        dead = np.zeros(o[9], dtype=bool)
        dead[o[2]:o[3]] = no_ins                                      # 3 homo_Ins: entry = length - 1
        dead[o[3]:o[4]] = no_del                                      # 4 homo_Del
        dead[o[4]:o[5]] = np.repeat(no_ins, 4)                        # 5 hetero_ACGT_Ins: entry = 4 (length - 1) + base
        dead[o[6]:o[7]] = np.repeat(no_del, 4)                        # 7 hetero_ACGT_Del
        dead[o[8]:o[9]] = np.logical_or.outer(no_del, no_ins).ravel()  # 9 hetero_InsDel: entry = 16 (deletion - 1) + insertion - 1
        if infer >= cap + 1:
            # every proposed length (< 16, + the reference base for insertions) lies inside the general range of the lookups, so a
            # second allele for a proposed pair can only be one that return_multi would offer as well: with fewer than two alleles
            # of the kind in range both the proposals (:838-848, :929-944) and return_multi (:849-855, :945-950) come back short
            if n_ins < 2:
                dead[o[5]:o[6]] = True                                # 6 hetero_InsIns
            if n_del < 2:
                dead[o[7]:o[8]] = True                                # 8 hetero_DelDel
        hit = self._dead_cache[key] = (dead, dead[_CHAIN[True]])
        return hit

    def _next_candidate(self, values, keep, cls0, pos0, ref, look):
        """The passes of output_from's loop after its first candidate (class cls0, entry pos0) was rejected: -> (class,
        alleles, maximum probability) of the first candidate the reads offer, None when the loop ends on the homo-reference
        probability, FALLBACK when the accepted maximum is shared by two classes.
        ``values``: the row's nine lists laid end to end IN CHAIN ORDER (_CHAIN), ``keep``: the entries above the homo-reference
        probability in walk order (stable sort by falling probability; formed for the whole batch at once in ``rows``).
        Each pass of the loop takes the maximum over homo_Ref and the nine lists, returns the reference call when that is
        homo_Ref (:735), else looks up the first class of the chain that holds it at its first index and zeroes that entry
        on rejection -- i.e. it walks the entries above homo_Ref by (probability falling, chain rank, index): a stable sort of
        the chain-ordered entries by falling probability."""
        indel = self.cfg.add_indel_length
        v = values[keep]
        klass, index = _KLASS_C[indel][keep], _INDEX_C[indel][keep]
        if len(keep) == 0 or klass[0] != cls0 or index[0] != pos0:
            return FALLBACK  # the device's first decision is not the head of the walk: leave the row to the reference
        if indel:  # entries no lookup can satisfy are stepped over without asking (SNP entries are never among them)
            dead = self._dead(look.d)[1][keep]  # (in chain order, like `keep`)
            todo = np.flatnonzero(~dead[1:]) + 1
        else:
            todo = range(1, len(keep))
        klass, index = klass.tolist(), index.tolist()
        for j in todo:
            k = klass[j]
            alleles = self._alleles(k, index[j], ref, look)
            if alleles is None:
                continue
            p = v[j]
            t = j + 1
            while t < len(klass) and v[t] == p:  # untried entries with the same probability: flags of other classes (:742-750)
                if klass[t] != k:
                    return FALLBACK
                t += 1
            return k, alleles, p
        return None
