#!/usr/bin/env python3
"""Numerics study for DESIGN.md section 7 (not part of the product or the tests): what would the full-alignment
probabilities look like if the convolutions ran on bf16 MFMAs with operands split into 1 / 2 / 3 bf16 pieces
(bf16x1 / x3 / x6 / x9 products, fp32 accumulation) instead of fp32 MFMAs?  CPU emulation with torch: every
piece-product is an fp32 convolution of bf16-representable operands, i.e. exactly what v_mfma_f32_32x32x16_bf16
accumulates.  Compares with the committed reference rows (tests/golden/*.npz).

    python tests/diag/bf16x_study.py

TEST INFRASTRUCTURE (lives under tests/ because it calls the oracle as its checker; never imported by the product).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import torch_port  # noqa: E402
from tests import util  # noqa: E402

PAIRS = {1: [(0, 0)], 3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)],
         9: [(i, j) for i in range(3) for j in range(3)]}


def pieces(t, n, dtype=torch.bfloat16):
    out, r = [], t
    for _ in range(n):
        p = r.to(dtype).float()
        out.append(p)
        r = r - p
    return out


def split_conv(x, w, b, stride, products):
    if products == 0:
        return F.conv2d(x, w, b, stride=stride, padding=1)
    if products < 0:  # fp16 pieces: -1 = single fp16, -3 = two pieces each, x0w0 + x0w1 + x1w0, -4 = + x1w1
        n = 1 if products == -1 else 2
        xs, ws = pieces(x, n, torch.float16), pieces(w, n, torch.float16)
        pairs = {-1: [(0, 0)], -3: [(0, 0), (0, 1), (1, 0)], -4: [(0, 0), (0, 1), (1, 0), (1, 1)]}[products]
    else:
        n = {1: 1, 3: 2, 6: 3, 9: 3}[products]
        xs, ws = pieces(x, n), pieces(w, n)
        pairs = PAIRS[products]
    acc = None
    for i, j in sorted(pairs, key=lambda p: -(p[0] + p[1])):  # small terms first
        y = F.conv2d(xs[i], ws[j], None, stride=stride, padding=1)
        acc = y if acc is None else acc + y
    return acc + b.view(1, -1, 1, 1)


@torch.inference_mode()
def fa_forward(sd, x, indel, products):
    x = (torch.as_tensor(x).float() / 100).permute(0, 3, 1, 2)
    for s in range(3):
        for k, (conv, bn, stride) in enumerate(torch_port._CONVS[3 * s: 3 * s + 3]):
            g = sd[f"{bn}.weight"].double() / torch.sqrt(sd[f"{bn}.running_var"].double() + 1e-3)
            w = (sd[f"{conv}.weight"].double() * g.view(-1, 1, 1, 1)).float()  # BatchNorm folded, as libc3hip packs it
            b = ((sd[f"{conv}.bias"].double() - sd[f"{bn}.running_mean"].double()) * g + sd[f"{bn}.bias"].double()).float()
            y = split_conv(x if k == 0 else (a if k == 1 else t), w, b, stride, products)
            if k == 0:
                a = F.relu(y)
            elif k == 1:
                t = F.relu(y)
            else:
                x = F.relu(a + y)
    return torch_port._tail(sd, torch_port._spp(x), indel).numpy()


def main():
    torch.set_num_threads(8)
    print(f"{'case':22s} {'products':>9s} {'max|dY|':>10s} {'labels differing':>17s}")
    for name in ("fa_realistic", "fa_peaked", "fa_uniform", "fa_dwell"):
        meta = util.manifest()[name]
        sd, x = util.case_inputs(meta)
        sd = torch_port.to_torch(sd)
        ref = util.golden_y(name)
        for products in (0, 9, 6, 3, 1, -4, -3, -1):
            y = fa_forward(sd, x, meta["add_indel_length"], products)
            bad = sum(len(v) for v in util.label_mismatches(y, ref).values()) if isinstance(util.label_mismatches(y, ref), dict) else util.label_mismatches(y, ref)
            print(f"{name:22s} {('fp32' if products == 0 else 'bf16x%d' % products if products > 0 else 'fp16x%d' % -products):>9s} {np.abs(y - ref).max():10.2e} {str(bad):>17s}")


if __name__ == "__main__":
    main()
