"""Weights that went through an OPTIMIZER (test infrastructure; nothing here is on the product path).

No trained checkpoint exists offline (BASELINE configs[0] is blocked on data), and ``synthetic._trained_like`` only re-parametrises seeded
weights.  This module trains the REFERENCE's own modules (clair3/model.py Clair3_P / Clair3_F from the staged copy or the checkout) for a few
hundred steps the way clair3/Train.py does -- AdamW at param.initialLearningRate with param.l2RegularizationLambda (:386-388), the focal loss
of :87-107 on one-hot labels of the four heads, BatchNorm and dropout in training mode -- on synthetic windows whose labels are a fixed random
"teacher" function of the window, so that the gradients mean something and the loss falls.  What comes out is what gradient descent leaves:
BatchNorm running statistics gathered from real forward passes, convolution / LSTM / FC weights moved by Adam, heads that have started to
peak.  Deterministic for a given seed on one torch build (one thread, fixed generators); nothing is stored: the tests train, then compare."""
import importlib

import numpy as np

from clair3_amd import synthetic as syn
from tests import refmodels

HEADS = ((0, 21), (21, 24), (24, 57), (57, 90))


def _teacher_labels(x, seed, n_heads):
    """one-hot labels (B, 24 | 90) from a fixed random projection of a cheap summary of the window (the centre columns' channel sums):
    a deterministic function of the window, so the network has something to learn"""
    rng = np.random.default_rng(seed)
    xf = x.astype(np.float32)
    feat = xf[:, 12:21].reshape(len(x), -1) if x.ndim == 3 else xf[:, :, 12:21].sum(axis=1).reshape(len(x), -1)  # pileup (B,33,18) / FA (B,89,33,C)
    feat = (feat - feat.mean(axis=0)) / (feat.std(axis=0) + 1.0)
    proj = rng.normal(0.0, 1.0, size=(feat.shape[1], HEADS[n_heads - 1][1])).astype(np.float32)
    logits = feat @ proj
    y = np.zeros_like(logits)
    for lo, hi in HEADS[:n_heads]:
        y[np.arange(len(x)), lo + logits[:, lo:hi].argmax(axis=1)] = 1.0
    return y


def train_reference(root, kind, channels, indel, steps=150, batch=24, seed=0, threads=8, device="cpu", lr_scale=1.0):
    """-> (state_dict as numpy arrays, list of the loss every 10 steps).  The reference's module in train() mode, AdamW, focal loss.
    device="cuda": the same training through PyTorch-ROCm (tests/diag/long_training_parity.py: thousands of steps in a minute)."""
    import torch
    torch.manual_seed(seed)
    torch.set_num_threads(threads)
    n_heads = 4 if indel else 2
    with refmodels._reference_on_path(root):
        model_py = importlib.import_module("clair3.model")
        param = importlib.import_module("shared.param_p" if kind == syn.PILEUP else "shared.param_f")
        cls = model_py.Clair3_P if kind == syn.PILEUP else model_py.Clair3_F
        m = cls(add_indel_length=indel, predict=False, input_channels=channels)
        lr, wd = float(param.initialLearningRate) * lr_scale, float(param.l2RegularizationLambda)
    m.to(device)
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=lr, weight_decay=wd)  # clair3/Train.py:386-388
    losses = []
    for step in range(steps):
        x = syn.make_windows(kind, batch, seed=seed * 100003 + step, channels=channels)
        y_true = torch.from_numpy(_teacher_labels(x, seed, n_heads)).to(device)
        heads = m(torch.from_numpy(x).to(device))
        loss = 0.0
        for (lo, hi), y_pred in zip(HEADS[:n_heads], heads):  # FocalLoss.forward, clair3/Train.py:100-107 (gamma 2, no class weights)
            p = torch.clamp(y_pred, min=1e-9, max=1 - 1e-9)
            t = y_true[:, lo:hi]
            loss = loss + ((-t * torch.log(p)) * (((1 - p) ** 2) * t)).sum(dim=-1).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 10 == 0 or step == steps - 1:
            losses.append(float(loss.detach()))
    m.eval()
    m.to("cpu")
    sd = {k: np.ascontiguousarray(v.detach().cpu().numpy()) for k, v in m.state_dict().items()}
    return sd, losses
