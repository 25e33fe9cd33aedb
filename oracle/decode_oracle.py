"""CPU restatement of the reference decoder's probability enumeration -- TEST INFRASTRUCTURE ONLY.

Follows clair3/CallVariants.py:510-659 (possible_outcome_probabilites_from) and the tuple builders :303-372 with
numpy float32 scalars, i.e. one float32 rounding per product in the reference's multiplication order, and reduces every
class list the way output_from (:722-749) uses it: max() (first maximal element) and list.index().  Pinned: the golden
fixture tests/golden/decode_*.npz holds the lists' maxima as computed by the REAL reference function
(tests/golden/make_golden_decode.py); tests/test_decode.py checks this file against it bit for bit.
"""
import numpy as np

OFF = 16  # VariantLength.index_offset


def _first_max(values):
    best, pos = values[0], 0
    for i, v in enumerate(values):
        if v > best:
            best, pos = v, i
    return best, pos


def outcome_lists(row, ref21, add_indel_length):
    """The ten class lists of one row, in the order of the reference's max(...) call."""
    f = np.float32
    g = [f(v) for v in row[:21]]
    hr, hv, ht = f(row[21]), f(row[22]), f(row[23])
    hs, ts = (0, 4, 7, 9), (1, 2, 3, 5, 6, 8)
    if not add_indel_length:  # :526-566
        return [[hr * g[ref21]], [hv * g[k] for k in hs], [ht * g[k] for k in ts], [hv * g[15]], [hv * g[10]],
                [g[16 + b] * ht for b in range(4)], [ht * g[15]], [g[11 + b] * ht for b in range(4)], [ht * g[10]],
                [ht * g[20]]]
    p1 = [f(v) for v in row[24:57]]
    p2 = [f(v) for v in row[57:90]]
    v0 = p1[OFF] * p2[OFF]
    xi, xd = hv * g[15], hv * g[10]
    yi, yd, yx = ht * g[15], ht * g[10], ht * g[20]
    R = range(1, 17)
    return [
        [v0 * hr * g[ref21]],                                                                   # :571-573
        [v0 * hv * g[k] for k in hs],                                                           # :579-581
        [v0 * ht * g[k] for k in ts],                                                           # :582-584
        [p1[OFF + i] * p2[OFF + i] * xi for i in R],                                            # :303-308
        [p1[OFF - i] * p2[OFF - i] * xd for i in R],                                            # :331-336
        [p1[OFF] * p2[OFF + i] * g[16 + b] * ht for i in R for b in range(4)],                  # :311-316, :600-607
        [p1[OFF + i] * p2[OFF + j] * yi for i in R for j in range(i, 17)],                      # :318-328
        [p1[OFF - i] * p2[OFF] * g[11 + b] * ht for i in R for b in range(4)],                  # :339-345, :627-634
        [p1[OFF - i] * p2[OFF - j] * yd for i in R for j in R if not (i == j and i != 16)],     # :348-359
        [p1[OFF - i] * p2[OFF + j] * yx for i in R for j in R],                                 # :362-371
    ]


def outcome_maxima(y, ref21, add_indel_length):
    y = np.asarray(y, dtype=np.float32)
    maxp = np.zeros((len(y), 10), dtype=np.float32)
    arg = np.zeros((len(y), 10), dtype=np.int32)
    early = np.zeros(len(y), dtype=bool)
    for r, row in enumerate(y):
        k = int(ref21[r])
        for c, lst in enumerate(outcome_lists(row, k, add_indel_length)):
            maxp[r, c], arg[r, c] = _first_max(lst)
        if add_indel_length:  # :573-576
            early[r] = row[24 + OFF] >= 0.5 and row[57 + OFF] >= 0.5 and row[21] >= 0.5 and row[k] >= 0.5
        else:                 # :532-534
            early[r] = row[21] >= 0.5 and row[k] >= 0.5
    return maxp, arg, early


ELIF_ORDER = (1, 2, 3, 5, 6, 4, 7, 8, 9)  # output_from's if / elif chain, clair3/CallVariants.py:753-978


def quality_score(p):
    """clair3/CallVariants.py:375-381 on a numpy float32 scalar (numpy >= 2: the python floats are weak, the quotient is
    formed in float32; math.log and the rest in double)"""
    from math import e, log
    p = np.float32(p)
    phred_trans = -10 * log(e, 10)
    tmp = max(phred_trans * log(((1.0 - p) + 1e-10) / (p + 1e-10)) + 10, 0)
    return float(round(tmp, 2))


def first_decision(y, ref21, add_indel_length):
    """(winner class, QUAL) of output_from's first pass (:722-751) per row: 0 when the overall maximum is the homo_Ref
    probability or the row takes the early exit, else the first class of the if / elif chain holding the maximum."""
    maxp, arg, early = outcome_maxima(y, ref21, add_indel_length)
    winner = np.zeros(len(maxp), dtype=np.int8)
    qual = np.zeros(len(maxp), dtype=np.float64)
    for r in range(len(maxp)):
        if early[r]:
            qual[r] = quality_score(maxp[r, 0])
            continue
        m = maxp[r].max()
        if m != maxp[r, 0]:
            winner[r] = next(c for c in ELIF_ORDER if maxp[r, c] == m)
        qual[r] = quality_score(m)
    return winner, qual


def decode_columns(y, add_indel_length):
    """The 31 decoder columns of include/c3hip.h (C3_DECODE_COLS) for every row: maxima and first positions of classes
    1..9 (independent of the reference base); homo_Ref probability, early-exit bit, first-decision class and 100 x QUAL for
    each base A, C, G, T."""
    y = np.asarray(y, dtype=np.float32)
    out = np.zeros((len(y), 31), dtype=np.float32)
    for b, k in enumerate((0, 4, 7, 9)):
        maxp, arg, early = outcome_maxima(y, np.full(len(y), k), add_indel_length)
        out[:, 9 + b] = maxp[:, 0]
        out[:, 22] += early.astype(np.float32) * (1 << b)
        winner, qual = first_decision(y, np.full(len(y), k), add_indel_length)
        out[:, 23 + b] = winner
        out[:, 27 + b] = np.rint(qual * 100.0)
        if b == 0:
            out[:, 0:9] = maxp[:, 1:]
            out[:, 13:22] = arg[:, 1:]
    return out
