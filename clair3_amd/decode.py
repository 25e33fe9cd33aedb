"""SURVEY 8f N1, first slice: the arithmetic of the reference decoder on the device.

``outcome_maxima(model, Y, ref_bases)`` returns, for every probability row, what the reference's ``output_from``
(clair3/CallVariants.py:676-741) looks at in the lists that ``possible_outcome_probabilites_from`` (:510-659) builds in
pure Python: per outcome class the maximum and the position of its first occurrence in the reference's enumeration
order, and whether the row takes the homo-reference early exit.  ``CLASS_NAMES`` / ``class_entry`` translate a
(class, position) pair back into the lengths / bases the reference attaches to that list entry, so a caller can skip the
enumeration (~800 float32 products per full-alignment row, 3.8 k rows/s/core) and keep only the allele-string logic.
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
