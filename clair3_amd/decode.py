"""the arithmetic of the reference decoder on the device.

``outcome_maxima(model, Y, ref_bases)`` returns, for every probability row, what the reference's ``output_from``
(clair3/CallVariants.py:676-741) looks at in the lists that ``possible_outcome_probabilites_from`` (:510-659) builds in
pure Python: per outcome class the maximum and the position of its first occurrence in the reference's enumeration
order, and whether the row takes the homo-reference early exit.  ``CLASS_NAMES`` / ``class_entry`` translate a
(class, position) pair back into the lengths / bases the reference attaches to that list entry, so a caller can skip the
enumeration (~800 float32 products per full-alignment row, 3.8 k rows/s/core) and keep only the allele-string logic.

``install_decoder()`` wires the same results into an UNMODIFIED reference decoder: with ``model.decode_columns()`` on,
``_hip_predict`` returns rows of ``output_size + DECODE_COLS`` floats (the columns described in include/c3hip.h);
``batch_output`` (clair3/CallVariants.py:1069) ignores the surplus columns by construction, and the rebound
``possible_outcome_probabilites_from`` hands ``output_from`` (:676-1025) list look-alikes that answer ``max()``,
``in``, ``.index()`` and ``np.argmax`` from the device columns.  The first time ``output_from`` rejects a candidate and
zeroes its entry (``probabilities[idx] = 0``, :760 ff.) the list of THAT class is formed -- one vectorised float32
product, bit-identical to the reference's scalar loop -- and its maximum is tracked from then on, so the retry loop
meets the values it would have met.  tests/test_decode_dropin.py: identical VCF text from the reference's own
batch_output with and without the columns, including rows that reject hundreds of candidates.
"""
import ctypes as C

import numpy as np

from . import _lib

# order of the reference's max(...) call, CallVariants.py:722-733
CLASS_NAMES = ("homo_Ref", "homo_SNP", "hetero_SNP", "homo_Ins", "homo_Del", "hetero_ACGT_Ins", "hetero_InsIns",
               "hetero_ACGT_Del", "hetero_DelDel", "hetero_InsDel")
REF_GT21 = {"A": 0, "C": 4, "G": 7, "T": 9}  # gt21_enum_from_label(base + base), clair3/task/gt21.py:3-27
HOMO_SNP_GT21 = (0, 4, 7, 9)                 # AA CC GG TT   (clair3/task/gt21.py:111)
HETERO_SNP_GT21 = (1, 2, 3, 5, 6, 8)         # AC AG AT CG CT GT (:114)
MAX_LEN = 16                                  # VariantLength.max (clair3/task/variant_length.py:6-12)

_INSINS = [(i, j) for i in range(1, MAX_LEN + 1) for j in range(i, MAX_LEN + 1)]                       # :318-328
_DELDEL = [((i, j) if i < j else (j, i)) for i in range(1, MAX_LEN + 1) for j in range(1, MAX_LEN + 1)
           if not (i == j and i != MAX_LEN)]                                                               # :348-359
_INSDEL = [(i, j) for i in range(1, MAX_LEN + 1) for j in range(1, MAX_LEN + 1)]                       # :362-371


def class_entry(cls, position, add_indel_length=True):
    """What the reference stores beside probability number ``position`` of class ``cls`` (lengths, length tuples, bases)."""
    name = CLASS_NAMES[cls]
    if name == "homo_Ref":
        return None
    if name == "homo_SNP":
        return HOMO_SNP_GT21[position]
    if name == "hetero_SNP":
        return HETERO_SNP_GT21[position]
    if not add_indel_length:
        return "ACGT"[position] if name in ("hetero_ACGT_Ins", "hetero_ACGT_Del") else None
    if name in ("homo_Ins", "homo_Del"):
        return position + 1
    if name in ("hetero_ACGT_Ins", "hetero_ACGT_Del"):
        return "ACGT"[position % 4], position // 4 + 1
    if name == "hetero_InsIns":
        return _INSINS[position]
    if name == "hetero_DelDel":
        return _DELDEL[position]
    return _INSDEL[position]


def ref_gt21_indices(ref_bases):
    """Centre reference bases (str / bytes / sequence of single letters, already ACGT as BASE2ACGT leaves them) -> uint8 gt21 indices."""
    if isinstance(ref_bases, (bytes, bytearray)):
        ref_bases = ref_bases.decode()
    return np.fromiter((REF_GT21[b] for b in ref_bases), dtype=np.uint8, count=len(ref_bases))


def outcome_maxima(model, y, ref_bases):
    """y: (B, 24|90) float32 rows of ``model``; ref_bases: B letters or a uint8 array of gt21 indices.
    Returns (maxp (B,10) float32, argmax (B,10) int32, early_exit (B,) bool)."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    if y.ndim != 2 or y.shape[1] != model.output_size:
        raise _lib.C3Error(f"rows must be (B, {model.output_size}) float32, got {y.shape}")
    ref = ref_bases if isinstance(ref_bases, np.ndarray) and ref_bases.dtype == np.uint8 else ref_gt21_indices(ref_bases)
    ref = np.ascontiguousarray(ref)
    if len(ref) != len(y):
        raise _lib.C3Error(f"{len(ref)} reference bases for {len(y)} rows")
    maxp = np.empty((len(y), 10), dtype=np.float32)
    arg = np.empty((len(y), 10), dtype=np.int32)
    early = np.empty(len(y), dtype=np.uint8)
    _lib.check(_lib.lib().c3_outcome_maxima(model._handle, y.ctypes.data, len(y), ref.ctypes.data, maxp.ctypes.data,
                                             arg.ctypes.data, early.ctypes.data), "c3_outcome_maxima")
    return maxp, arg, early.astype(bool)


def decode_columns(model, y):
    """y: (B, 24|90) float32 rows of ``model`` -> (B, 24|90 + DECODE_COLS) rows with the decoder columns appended
    (c3_decode_columns); what ``model.decode_columns(True)`` makes the predict calls return directly."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    if y.ndim != 2 or y.shape[1] != model.output_size:
        raise _lib.C3Error(f"rows must be (B, {model.output_size}) float32, got {y.shape}")
    rows = np.empty((len(y), model.output_size + DECODE_COLS), dtype=np.float32)
    _lib.check(_lib.lib().c3_decode_columns(model._handle, y.ctypes.data, len(y), rows.ctypes.data), "c3_decode_columns")
    return rows


# reference base -> A/C/G/T index exactly as output_from resolves it (clair3/CallVariants.py:690: BASE2ACGT[...] =
# shared/utils.py:42-45 IUPAC_base_to_ACGT_base_dict, upper-case keys only -- a lower-case base is a KeyError there, an
# error here): U -> T, R/W/M/D/H/V/N -> A, Y/S/B -> C, K -> G
_REF_BASE_INDEX = np.full(256, -1, np.int8)
for _b, _t in zip("ACGTURYSWKMBDHVN", "ACGTTACCAGACAAAA"):
    _REF_BASE_INDEX[ord(_b)] = "ACGT".index(_t)


def first_decisions(rows, ref_bases, output_size):
    """Vectorised read-out of the decoder columns for rows whose reference base IS known (the worker knows it from the
    .info position string): what output_from (clair3/CallVariants.py:722-751) decides on its first pass and what
    output_with prints as QUAL (:1325) when no candidate is rejected --
        cls    0 homo_Ref ... 9 hetero_InsDel (CLASS_NAMES)      pos    position of the winning entry in its class list
        qual   quality_score_from(maximum) as the same double     prob   the maximum itself (float32)
        early  the homo-reference early exit
    ``class_entry(cls, pos)`` turns (cls, pos) into lengths / bases; only the allele-string lookup in alt_info is left to
    Python.  rows: (B, output_size + DECODE_COLS) as returned with model.decode_columns(True)."""
    rows = np.asarray(rows, dtype=np.float32)
    cols = rows[:, output_size:]
    if cols.shape[1] != DECODE_COLS:
        raise _lib.C3Error(f"rows carry {cols.shape[1]} decoder columns, expected {DECODE_COLS}")
    if isinstance(ref_bases, (bytes, bytearray)):
        ref_bases = ref_bases.decode()
    b = np.frombuffer(ref_bases.encode(), dtype=np.uint8) if isinstance(ref_bases, str) else np.asarray(ref_bases, dtype=np.uint8)
    bi = _REF_BASE_INDEX[b].astype(np.int64)
    if (bi < 0).any():
        bad = sorted({chr(v) for v in np.asarray(b)[bi < 0]})
        raise _lib.C3Error(f"reference bases {bad} are outside the IUPAC table of shared/utils.py:42-45")
    r = np.arange(len(rows))
    cls = cols[r, 23 + bi].astype(np.int8)
    pos = np.where(cls > 0, cols[r, 13 + np.maximum(cls.astype(np.int64), 1) - 1], 0).astype(np.int32)
    prob = np.where(cls > 0, cols[r, np.maximum(cls.astype(np.int64), 1) - 1], cols[r, 9 + bi]).astype(np.float32)
    early = ((cols[:, 22].astype(np.int32) >> bi) & 1).astype(bool)
    return {"cls": cls, "pos": pos, "qual": cols[r, 27 + bi].astype(np.float64) / 100.0, "prob": prob, "early": early}


# ------------------------------------------------------------------------------------------------------------------
# decoder columns -> the reference's output_from
DECODE_COLS = 31
_CLASS_LEN = {True: (1, 4, 6, 16, 16, 64, 136, 64, 241, 256), False: (1, 4, 6, 1, 1, 4, 1, 4, 1, 1)}


_I16 = np.arange(1, MAX_LEN + 1)
_II, _IJ = (np.array(v) for v in zip(*_INSINS))
_DI, _DJ = (np.array(v) for v in zip(*[(i, j) for i in range(1, MAX_LEN + 1) for j in range(1, MAX_LEN + 1)
                                      if not (i == j and i != MAX_LEN)]))   # enumeration order (i outer), before the (min, max) swap
_XI, _XJ = (np.array(v) for v in zip(*_INSDEL))
_L4, _B4 = np.repeat(_I16, 4), np.tile(np.arange(4), MAX_LEN)
_HS, _TS = np.array(HOMO_SNP_GT21), np.array(HETERO_SNP_GT21)


def class_list(cls, g, z, p1, p2, add_indel_length):
    """The probability list of class ``cls`` (1..9) as a float32 array: the same products, in the same order and
    association, as clair3/CallVariants.py:526-659 forms with float32 scalars -- element-wise float32 multiplication is
    the same IEEE operation, so the values are bit-identical (tests/test_decode_dropin.py checks them against the
    reference's lists)."""
    hv, ht = z[1], z[2]
    if not add_indel_length:  # :526-566
        if cls == 1: return hv * g[_HS]
        if cls == 2: return ht * g[_TS]
        if cls == 3: return np.array([hv * g[15]])
        if cls == 4: return np.array([hv * g[10]])
        if cls == 5: return g[16:20] * ht
        if cls == 6: return np.array([ht * g[15]])
        if cls == 7: return g[11:15] * ht
        if cls == 8: return np.array([ht * g[10]])
        return np.array([ht * g[20]])
    o = 16  # VariantLength.index_offset
    if cls == 1: return (p1[o] * p2[o] * hv) * g[_HS]                       # :579-581
    if cls == 2: return (p1[o] * p2[o] * ht) * g[_TS]                       # :582-584
    if cls == 3: return p1[o + _I16] * p2[o + _I16] * (hv * g[15])          # :303-308, :587-590
    if cls == 4: return p1[o - _I16] * p2[o - _I16] * (hv * g[10])          # :331-336, :613-616
    if cls == 5: return (p1[o] * p2[o + _L4]) * g[16 + _B4] * ht            # :311-316, :600-606
    if cls == 6: return p1[o + _II] * p2[o + _IJ] * (ht * g[15])            # :318-328
    if cls == 7: return (p1[o - _L4] * p2[o]) * g[11 + _B4] * ht            # :339-345, :627-633
    if cls == 8: return p1[o - _DI] * p2[o - _DJ] * (ht * g[10])            # :348-359
    return p1[o - _XI] * p2[o + _XJ] * (ht * g[20])                         # :362-371


class _Row:
    """One probability row in flight through output_from: its decoder columns and the heads the lists are made of."""
    __slots__ = ("cols", "g", "z", "p1", "p2", "indel")

    def __init__(self, cols, g, z, p1, p2, indel):
        self.cols, self.g, self.z, self.p1, self.p2, self.indel = cols, g, z, p1, p2, indel


class _ClassProbs:
    """Stands in for one of the reference's probability lists.  output_from (CallVariants.py:718-1010) asks a list for
    max(), `value in`, .index(value), np.argmax (:91,:96) and writes 0 over a rejected candidate (:760 ff.).  The
    maximum and its first position come from the device columns; the list itself (class_list) is only formed when a
    candidate of THIS class is rejected, and then the maximum is kept up to date instead of being searched for."""
    __slots__ = ("row", "cls", "n", "mx", "pos", "arr")

    def __init__(self, row, cls, n):
        self.row, self.cls, self.n = row, cls, n
        self.mx, self.pos, self.arr = row.cols[cls - 1], None, None

    def _first(self):
        if self.pos is None:
            self.pos = int(self.row.cols[12 + self.cls])
        return self.pos

    def _array(self):
        if self.arr is None:
            r = self.row
            self.arr = np.ascontiguousarray(class_list(self.cls, r.g, r.z, r.p1, r.p2, r.indel), dtype=np.float32)
        return self.arr

    def __len__(self):
        return self.n

    def __iter__(self):  # only max() iterates these lists: the maximum is all it can learn
        return iter((self.mx,))

    def __contains__(self, value):
        if value == self.mx:
            return True
        if value > self.mx:
            return False
        return bool((self._array() == value).any())

    def index(self, value):
        if value == self.mx:
            return self._first()
        hits = np.flatnonzero(self._array() == value)
        if not len(hits):
            raise ValueError(f"{value!r} is not in list")
        return int(hits[0])

    def __array__(self, dtype=None, copy=None):
        a = self._array()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, i):
        return self._array()[i]

    def __setitem__(self, i, value):
        a = self._array()
        a[i] = value
        self.pos = int(a.argmax())  # first occurrence, like list.index(max(list))
        self.mx = a[self.pos]


# the lengths / length tuples / bases the reference keeps beside each probability list (:587-659): the same for every
# row and only ever read by output_from, so one shared copy
_ENTRIES = {
    3: list(range(1, MAX_LEN + 1)), 4: list(range(1, MAX_LEN + 1)), 6: list(_INSINS), 8: list(_DELDEL), 9: list(_INSDEL),
    50: ["ACGT"[b] for _ in range(MAX_LEN) for b in range(4)], 51: [i for i in range(1, MAX_LEN + 1) for _ in range(4)],
}
_ENTRIES[70], _ENTRIES[71] = _ENTRIES[50], _ENTRIES[51]


def outcome_from_columns(cols, gt21_probabilities, genotype_probabilities, variant_length_probabilities_1,
                         variant_length_probabilities_2, reference_base, add_indel_length):
    """What possible_outcome_probabilites_from (CallVariants.py:510-659) returns for this row, answered from its decoder
    columns ``cols``: [homo_Ref_probability] on the early exit (:532-534 / :573-576), else the 19-tuple with
    _ClassProbs / _ClassEntries where the reference has lists."""
    b = "ACGT".index(reference_base)
    homo_ref = cols[9 + b]
    if (int(cols[22]) >> b) & 1:
        return [homo_ref]
    row = _Row(cols, gt21_probabilities, genotype_probabilities, variant_length_probabilities_1,
               variant_length_probabilities_2, bool(add_indel_length))
    n = _CLASS_LEN[row.indel]
    P = lambda cls: _ClassProbs(row, cls, n[cls])  # noqa: E731
    if row.indel:
        E = _ENTRIES
        return (homo_ref, P(1), P(2), E[3], P(3), E[6], P(6), E[50], E[51], P(5), E[4], P(4),
                E[8], P(8), E[70], E[71], P(7), E[9], P(9))
    return (homo_ref, P(1), P(2), [], P(3), [], P(6), [], [], P(5), [], P(4), [], P(8), [], [], P(7), [], P(9))


class _Batch:
    __slots__ = ("base", "stride", "nbytes", "cols")

    def __init__(self):
        self.base = self.stride = self.nbytes = 0
        self.cols = None


def install_decoder():
    """Rebind clair3.CallVariants.possible_outcome_probabilites_from / batch_output (and the copy of batch_output that
    clair3.CallVariantsFromCffi imported, :15) so that rows carrying decoder columns are decoded from them.  Rows
    without the columns -- and every call from elsewhere -- go through the reference's code untouched."""
    import clair3.CallVariants as cv
    if getattr(cv, "_c3hip_decoder", None):
        return cv._c3hip_decoder
    enumerate_rows = cv.possible_outcome_probabilites_from
    reference_batch_output = cv.batch_output
    cur = _Batch()

    def possible_outcome_probabilites_from(gt21_probabilities, genotype_probabilities, variant_length_probabilities_1,
                                           variant_length_probabilities_2, reference_base, alt_info_dict,
                                           add_indel_length=False):
        if cur.cols is not None and isinstance(gt21_probabilities, np.ndarray) and reference_base in ("A", "C", "G", "T"):
            off = gt21_probabilities.__array_interface__["data"][0] - cur.base
            if 0 <= off < cur.nbytes and off % cur.stride == 0:
                return outcome_from_columns(cur.cols[off // cur.stride], gt21_probabilities, genotype_probabilities,
                                            variant_length_probabilities_1, variant_length_probabilities_2,
                                            reference_base, add_indel_length)
        return enumerate_rows(gt21_probabilities, genotype_probabilities, variant_length_probabilities_1,
                              variant_length_probabilities_2, reference_base=reference_base,
                              alt_info_dict=alt_info_dict, add_indel_length=add_indel_length)

    printers = {}

    def batch_output(batch_chr_pos_seq, alt_info_list, batch_Y, output_config, output_utilities, args=None):
        width = 90 if output_config.add_indel_length else 24
        wide = isinstance(batch_Y, np.ndarray) and batch_Y.ndim == 2 and batch_Y.shape[1] == width + DECODE_COLS \
            and batch_Y.dtype == np.float32 and len(batch_Y) > 0
        if not wide:
            return reference_batch_output(batch_chr_pos_seq, alt_info_list, batch_Y, output_config, output_utilities, args)
        if len(batch_Y) != len(batch_chr_pos_seq):
            return reference_batch_output(batch_chr_pos_seq, alt_info_list, batch_Y[:, :width], output_config, output_utilities, args)  # its error message
        key = (output_config, id(cv.param))
        printer = printers.get(key)
        if printer is None:
            from .vcf_rows import RowPrinter
            printer = printers[key] = RowPrinter(cv, output_config)
            cv._c3hip_row_printers = printers
        if printer.usable:
            # rows whose first decision stands are printed from the columns (vcf_rows.py); every other row goes through the
            # reference's own output_with, on the look-alike lists when the rows carry indel lengths
            cum = cv.param.label_shape_cum
            if output_config.add_indel_length:
                cur.base = batch_Y.__array_interface__["data"][0]
                cur.stride = batch_Y.strides[0]
                cur.nbytes = cur.stride * len(batch_Y)
                cur.cols = batch_Y[:, width:]
            def print_with_reference(i):
                y = batch_Y[i]
                p1, p2 = (y[cum[1]:cum[2]], y[cum[2]:cum[3]]) if output_config.add_indel_length else (0, 0)
                return cv.output_with(batch_chr_pos_seq[i], alt_info_list[i], y[:cum[0]], y[cum[0]:cum[1]], p1, p2, output_config, output_utilities)

            try:
                # one string for the batch (vcf_rows.RowPrinter.batch_text: the rows c3_vcf_rows printed are never cut into per-row
                # strings), written once -- the reference writes row by row to the same file object, in the same order
                text = printer.batch_text(batch_chr_pos_seq, alt_info_list, batch_Y, print_with_reference)
                if args is not None:
                    if text:
                        args.output_file.write(text)
                    return ""
                return text
            finally:
                cur.cols = None
        if not output_config.add_indel_length:
            # 24-column rows: the reference's lists hold 1-6 products each, nothing to save (measured: 33 k rows/s/core
            # either way); the columns are dropped and the reference decodes as it always does
            return reference_batch_output(batch_chr_pos_seq, alt_info_list, batch_Y[:, :width], output_config,
                                          output_utilities, args)
        cur.base = batch_Y.__array_interface__["data"][0]
        cur.stride = batch_Y.strides[0]
        cur.nbytes = cur.stride * len(batch_Y)
        cur.cols = batch_Y[:, width:]
        try:
            return reference_batch_output(batch_chr_pos_seq, alt_info_list, batch_Y[:, :width], output_config,
                                          output_utilities, args)
        finally:
            cur.cols = None

    cv.possible_outcome_probabilites_from = possible_outcome_probabilites_from
    cv.batch_output = batch_output
    cv._c3hip_decoder = (possible_outcome_probabilites_from, batch_output)
    import sys
    w = sys.modules.get("clair3.CallVariantsFromCffi")
    if w is not None and hasattr(w, "batch_output"):
        w.batch_output = batch_output
    return cv._c3hip_decoder
