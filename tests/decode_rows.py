"""Synthetic decoder inputs with a known story: probability rows peaked on one outcome class / entry of the reference's
enumeration (clair3/CallVariants.py:510-659) and alt_info strings that do (or do not quite) support it -- what a trained
model and a real pileup give the decoder most of the time, next to the adversarial rows of tests/test_decode_dropin.py."""
import numpy as np

from clair3_amd import decode as dec

GT21_OF_REF = {"A": 0, "C": 4, "G": 7, "T": 9}


def _softmax_peak(n, idx, rng, sharp):
    logits = rng.normal(0.0, 1.0, size=n)
    logits[idx] += sharp
    e = np.exp(logits - logits.max())
    return (e / e.sum()).astype(np.float32)


def consistent_rows(n, seed, indel=True, noise=0.0, sharp=9.0):
    """-> (positions, alt_infos, Y (n, 24|90) float32, classes).  Row i is peaked on class i % 10 (entry drawn at random) and
    its alt_info carries the alleles that class needs; with ``noise`` > 0 that share of rows gets an alt_info that lacks them
    (the reference then rejects the candidate and tries the next best)."""
    rng = np.random.default_rng(seed)
    pos, alt, rows, classes = [], [], [], []
    for i in range(n):
        cls = i % 10
        seq = "".join("ACGT"[j] for j in rng.integers(0, 4, size=33))
        ref = seq[16]
        others = [b for b in "ACGT" if b != ref]
        depth = int(rng.integers(20, 90))
        g_idx, z_idx, l1, l2 = GT21_OF_REF[ref], 0, 0, 0
        parts = []
        cnt = lambda: int(rng.integers(4, max(5, depth // 2)))  # noqa: E731
        ins_bases = lambda L: ref + "".join("ACGT"[j] for j in rng.integers(0, 4, size=L))  # noqa: E731
        if cls == 1:
            b = others[int(rng.integers(0, 3))]
            g_idx, z_idx = GT21_OF_REF[b], 1
            parts.append(f"X{b} {cnt()}")
        elif cls == 2:
            k = int(rng.integers(0, 6))
            pair = ("AC", "AG", "AT", "CG", "CT", "GT")[k]
            g_idx, z_idx = dec.HETERO_SNP_GT21[k], 2
            for b in pair:
                if b != ref:
                    parts.append(f"X{b} {cnt()}")
        elif cls == 3:
            L = int(rng.integers(1, 17))
            g_idx, z_idx, l1, l2 = 15, 1, L, L
            parts.append(f"I{ins_bases(L)} {cnt()}")
        elif cls == 4:
            L = int(rng.integers(1, 16))
            g_idx, z_idx, l1, l2 = 10, 1, -L, -L
            parts.append(f"D{seq[17:17 + L]} {cnt()}")
        elif cls == 5:
            L, b = int(rng.integers(1, 17)), int(rng.integers(0, 4))
            g_idx, z_idx, l1, l2 = 16 + b, 2, 0, L
            parts.append(f"I{ins_bases(L)} {cnt()}")
            if "ACGT"[b] != ref:
                parts.append(f"X{'ACGT'[b]} {cnt()}")
        elif cls == 6:
            i1, i2 = sorted(int(v) for v in rng.integers(1, 17, size=2))
            g_idx, z_idx, l1, l2 = 15, 2, i1, i2
            a, b2 = ins_bases(i1), ins_bases(i2)
            if a == b2:
                b2 = b2[:-1] + ("A" if b2[-1] != "A" else "C")
            parts += [f"I{a} {cnt()}", f"I{b2} {cnt()}"]
        elif cls == 7:
            L, b = int(rng.integers(1, 16)), int(rng.integers(0, 4))
            g_idx, z_idx, l1, l2 = 11 + b, 2, -L, 0
            parts.append(f"D{seq[17:17 + L]} {cnt()}")
            if "ACGT"[b] != ref:
                parts.append(f"X{'ACGT'[b]} {cnt()}")
        elif cls == 8:
            i1, i2 = (int(v) for v in rng.choice(np.arange(1, 16), size=2, replace=False))
            g_idx, z_idx, l1, l2 = 10, 2, -i1, -i2
            parts += [f"D{seq[17:17 + i1]} {cnt()}", f"D{seq[17:17 + i2]} {cnt()}"]
        elif cls == 9:
            dl, il = int(rng.integers(1, 16)), int(rng.integers(1, 17))
            g_idx, z_idx, l1, l2 = 20, 2, -dl, il
            parts += [f"D{seq[17:17 + dl]} {cnt()}", f"I{ins_bases(il)} {cnt()}"]
        if noise and rng.random() < noise and parts:
            parts = parts[1:] if rng.random() < 0.5 else []
        parts.append(f"R{ref} {cnt()}")
        order = rng.permutation(len(parts))
        row = [_softmax_peak(21, g_idx, rng, sharp), _softmax_peak(3, z_idx, rng, sharp)]
        if indel:
            row += [_softmax_peak(33, 16 + l1, rng, sharp), _softmax_peak(33, 16 + l2, rng, sharp)]
        rows.append(np.concatenate(row))
        pos.append(f"chr{1 + i % 4}:{9000 + 37 * i}:{seq}")
        alt.append(f"{depth}-" + " ".join(parts[j] for j in order) + " ")
        classes.append(cls)
    return pos, alt, np.stack(rows).astype(np.float32), np.array(classes)
