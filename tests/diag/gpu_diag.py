#!/usr/bin/env python3
"""Layer-by-layer diagnostic of the HIP path against the CPU oracle (run on the GPU box).
Never asserts: prints one line per tensor so a single gpurun call localises a bug.
TEST INFRASTRUCTURE (lives under tests/ because it calls the oracle as its checker; never imported by the product).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from clair3_amd import synthetic as syn  # noqa: E402
from clair3_amd.model import Clair3_F, Clair3_P  # noqa: E402
from oracle import oracle  # noqa: E402


def report(tag, a, ref):
    d = np.abs(a.astype(np.float64) - ref.astype(np.float64))
    scale = max(1.0, float(np.abs(ref).max()))
    bad = d > 1e-4 * scale
    msg = f"  {tag:12s} shape={str(a.shape):22s} max_abs_err={d.max():.3e} ref_max={np.abs(ref).max():.3e} bad={bad.mean():.4f}"
    if bad.any():
        idx = np.argwhere(bad)
        msg += f" first_bad={idx[0].tolist()} got={a[tuple(idx[0])]:.5f} want={ref[tuple(idx[0])]:.5f}"
        # which trailing-dim / row positions are wrong (helps to spot tile/lane mapping bugs)
        flat = bad.reshape(-1, bad.shape[-1])
        msg += f" bad_cols={np.nonzero(flat.any(0))[0][:12].tolist()} bad_rows={np.nonzero(flat.any(1))[0][:12].tolist()}"
    print(msg, flush=True)


def fa(channels=8, n=6, seed=0):
    print(f"== full alignment C={channels} n={n}")
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, channels, True, seed=seed)
    x = syn.make_fa_windows(n, seed=seed, channels=channels)
    m = Clair3_F(add_indel_length=True, predict=True, input_channels=channels).keep_activations(True).to("cuda:0")
    m.load_state_dict(sd)
    y = m.predict_numpy(x)
    y_o, d = oracle.fa_forward(sd, x, True, debug=True)
    for l in range(9):
        report(f"act{l}", m.debug_fetch(f"act{l}", d[f"act{l}"].shape), d[f"act{l}"])
    report("spp", m.debug_fetch("spp", d["spp"].shape), d["spp"])
    report("l4_out", m.debug_fetch("l4_out", d["l4_out"].shape), d["l4_out"])
    report("y", y, y_o)


def pileup(n=48, seed=0, dtype=np.int8):
    print(f"== pileup n={n} {np.dtype(dtype).name}")
    sd = syn.make_state_dict(syn.PILEUP, 18, False, seed=seed)
    x = syn.make_pileup_windows(n, seed=seed, dtype=dtype)
    m = Clair3_P(add_indel_length=False, predict=True).keep_activations(True).to("cuda:0")
    m.load_state_dict(sd)
    y = m.predict_numpy(x)
    y_o, d = oracle.pileup_forward(sd, x, False, debug=True)
    # gx1 = x W_ih^T + b in the kernel's permuted column order: check through lstm1_out instead
    for key in ("lstm1_out", "lstm2_out", "l4_out"):
        report(key, m.debug_fetch(key, d[key].shape), d[key])
    report("y", y, y_o)


def quick_timing():
    import torch
    print("== quick timing (device-resident, includes all kernels)")
    for kind, B, ch, indel in ((syn.FULL_ALIGNMENT, 256, 8, True), (syn.PILEUP, 1024, 18, False)):
        sd = syn.make_state_dict(kind, ch, indel, seed=0)
        cls = Clair3_P if kind == syn.PILEUP else Clair3_F
        m = cls(add_indel_length=indel, predict=True, input_channels=ch).to("cuda:0")
        m.load_state_dict(sd)
        x = torch.from_numpy(syn.make_windows(kind, B, seed=1)).cuda()
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t = time.time()
        it = 20
        for _ in range(it):
            m(x)
        torch.cuda.synchronize()
        dt = (time.time() - t) / it
        print(f"  {kind:15s} B={B}: {dt * 1e3:.3f} ms/batch -> {B / dt:,.0f} windows/s", flush=True)
        m.profile(True)
        m.profile_reset()
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        for r in m.profile_read():
            ms = r["total_ms"] / r["launches"]
            print(f"    {r['name']:10s} {ms * 1e3:9.1f} us/launch  {r['flops'] / r['launches'] / ms / 1e9:8.1f} TFLOP/s"
                  f"  {r['bytes'] / r['launches'] / ms / 1e6:8.1f} GB/s", flush=True)
        m.profile(False)


def host_path():
    """PCIe-inclusive rates of the host-buffer entry points (pageable numpy in / out), never bench.py's `value`."""
    from clair3_amd import worker
    print("== host-buffer path (H2D + kernels + D2H, pageable numpy buffers)")
    for kind, B, ch, indel, nb in ((syn.FULL_ALIGNMENT, 256, 8, True, 24), (syn.PILEUP, 1024, 18, False, 24)):
        sd = syn.make_state_dict(kind, ch, indel, seed=0)
        cls = Clair3_P if kind == syn.PILEUP else Clair3_F
        ms = []
        for _ in range(2):
            m = cls(add_indel_length=indel, predict=True, input_channels=ch).to("cuda:0")
            m.load_state_dict(sd)
            ms.append(m)
        x = syn.make_windows(kind, B, seed=1)
        batches = [(x, ["p"] * B, ["a"] * B)] * nb
        ms[0].predict_numpy(x), ms[1].predict_numpy(x)
        t = time.time()
        for _ in range(nb):
            ms[0].predict_numpy(x)
        sync = nb * B / (time.time() - t)
        t = time.time()
        worker.predict_batches(ms[0], iter(batches), lambda p, a, y: None)
        one = nb * B / (time.time() - t)
        t = time.time()
        worker.predict_batches(ms, iter(batches), lambda p, a, y: None)
        two = nb * B / (time.time() - t)
        print(f"  {kind:15s} B={B}: c3_predict (sync) {sync:,.0f} | submit/wait, 1 handle {one:,.0f} | submit/wait, 2 handles {two:,.0f} windows/s", flush=True)


def decode_rate():
    """Rows/s of c3_outcome_maxima (host rows in, class maxima out, copies included) next to the numpy restatement of the
    reference's Python enumeration (oracle/decode_oracle.py, one core) on the same rows."""
    from clair3_amd import decode
    from oracle import decode_oracle
    print("== decode slice (SURVEY 8f N1): class maxima per probability row")
    for kind, ch, indel, n in ((syn.FULL_ALIGNMENT, 8, True, 16384), (syn.PILEUP, 18, False, 16384)):
        cls = Clair3_P if kind == syn.PILEUP else Clair3_F
        m = cls(add_indel_length=indel, predict=True, input_channels=ch).to("cuda:0")
        m.load_state_dict(syn.make_state_dict(kind, ch, indel, seed=0))
        y = np.tile(m.predict_numpy(syn.make_windows(kind, 512, seed=2)), (n // 512, 1))
        ref = np.resize(np.array([0, 4, 7, 9], dtype=np.uint8), n)
        decode.outcome_maxima(m, y, ref)
        t = time.time()
        for _ in range(10):
            decode.outcome_maxima(m, y, ref)
        gpu = 10 * n / (time.time() - t)
        t = time.time()
        decode_oracle.outcome_maxima(y[:256], ref[:256], indel)
        cpu = 256 / (time.time() - t)
        print(f"  {kind:15s} {n} rows x {y.shape[1]} floats: c3_outcome_maxima {gpu:,.0f} rows/s | python enumeration {cpu:,.0f} rows/s/core", flush=True)
    # host read-out of the first decision (class, position, QUAL) from rows that carry the columns: what is left for one core
    for kind, ch, indel, n in ((syn.FULL_ALIGNMENT, 8, True, 65536), (syn.PILEUP, 18, False, 65536)):
        cls = Clair3_P if kind == syn.PILEUP else Clair3_F
        m = cls(add_indel_length=indel, predict=True, input_channels=ch).to("cuda:0")
        m.load_state_dict(syn.make_state_dict(kind, ch, indel, seed=0))
        m.decode_columns(True)
        wide = np.tile(m.predict_numpy(syn.make_windows(kind, 512, seed=2)), (n // 512, 1))
        letters = "ACGT" * (n // 4)
        decode.first_decisions(wide, letters, m.output_size)
        t = time.time()
        for _ in range(5):
            d = decode.first_decisions(wide, letters, m.output_size)
        rate = 5 * n / (time.time() - t)
        print(f"  {kind:15s} first_decisions (class, position, QUAL from the device columns): {rate:,.0f} rows/s/core; "
              f"classes seen {np.bincount(d['cls'], minlength=10).tolist()}", flush=True)
    # what the decoder columns cost on the prediction path (host windows in, host rows out, one batch at a time)
    print("== c3_predict with / without the decoder columns (c3_model_set_decode_columns)")
    for kind, ch, indel, B in ((syn.FULL_ALIGNMENT, 8, True, 256), (syn.FULL_ALIGNMENT, 8, True, 1000), (syn.PILEUP, 18, False, 1024)):
        cls = Clair3_P if kind == syn.PILEUP else Clair3_F
        m = cls(add_indel_length=indel, predict=True, input_channels=ch).to("cuda:0")
        m.load_state_dict(syn.make_state_dict(kind, ch, indel, seed=0))
        x = syn.make_windows(kind, B, seed=3)
        rates = []
        for on in (False, True, False, True):
            m.decode_columns(on)
            m.predict_numpy(x)
            reps = 40
            t = time.time()
            for _ in range(reps):
                m.predict_numpy(x)
            rates.append(reps * B / (time.time() - t))
        print(f"  {kind:15s} B={B}: plain {rates[0]:,.0f} / {rates[2]:,.0f}  with columns {rates[1]:,.0f} / {rates[3]:,.0f} windows/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["fa", "fa9", "pileup", "pileup32", "time"]
    for w in what:
        try:
            if w == "fa":
                fa(8)
            elif w == "fa9":
                fa(9, n=3, seed=3)
            elif w == "pileup":
                pileup()
            elif w == "pileup32":
                pileup(n=20, seed=1, dtype=np.int32)
            elif w == "time":
                quick_timing()
            elif w == "host":
                host_path()
            elif w == "decode":
                decode_rate()
        except Exception as e:  # keep going: the point is to collect as much as possible per GPU call
            print(f"!! {w} failed: {type(e).__name__}: {e}", flush=True)
