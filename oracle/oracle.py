"""ctypes front-end of the CPU oracle (oracle/c3_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package ``clair3_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libc3oracle.so")

_fp = C.POINTER(C.c_float)


class _LstmDir(C.Structure):
    _fields_ = [("w_ih", _fp), ("w_hh", _fp), ("b_ih", _fp), ("b_hh", _fp)]


class _Linear(C.Structure):
    _fields_ = [("w", _fp), ("b", _fp)]


class _ConvBn(C.Structure):
    _fields_ = [("w", _fp), ("b", _fp), ("bn_w", _fp), ("bn_b", _fp), ("bn_mean", _fp), ("bn_var", _fp)]


class _PileupWeights(C.Structure):
    _fields_ = [("lstm1", _LstmDir * 2), ("lstm2", _LstmDir * 2), ("L4", _Linear), ("L5", _Linear * 4),
                ("head", _Linear * 4)]


class _FaWeights(C.Structure):
    _fields_ = [("conv", _ConvBn * 9), ("L4", _Linear), ("L5", _Linear * 4), ("head", _Linear * 4)]


class _PileupDebug(C.Structure):
    _fields_ = [("lstm1_out", _fp), ("lstm2_out", _fp), ("l4_out", _fp)]


class _FaDebug(C.Structure):
    _fields_ = [("act", _fp * 9), ("spp", _fp), ("l4_out", _fp)]


_HEADS = ("Y_gt21_logits", "Y_genotype_logits", "Y_indel_length_logits_1", "Y_indel_length_logits_2")
_FA_CONVS = (("conv1.conv", "conv1.bn"), ("res_block1.0.conv1", "res_block1.0.bn1"),
             ("res_block1.0.conv2", "res_block1.0.bn2"), ("conv3.conv", "conv3.bn"),
             ("res_block2.0.conv1", "res_block2.0.bn1"), ("res_block2.0.conv2", "res_block2.0.bn2"),
             ("conv5.conv", "conv5.bn"), ("res_block3.0.conv1", "res_block3.0.bn1"),
             ("res_block3.0.conv2", "res_block3.0.bn2"))
_FA_COUT = (64, 64, 64, 128, 128, 128, 256, 256, 256)
_FA_STRIDE = (2, 1, 1, 2, 1, 1, 2, 1, 1)

_lib = None


def build(force=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "c3_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.c3o_pileup_forward.restype = C.c_int
        _lib.c3o_pileup_forward.argtypes = [C.POINTER(_PileupWeights), C.c_void_p, C.c_int, C.c_long, C.c_int,
                                            C.c_int, C.c_int, _fp, C.POINTER(_PileupDebug), C.c_int]
        _lib.c3o_fa_forward.restype = C.c_int
        _lib.c3o_fa_forward.argtypes = [C.POINTER(_FaWeights), C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int,
                                        C.c_int, _fp, C.POINTER(_FaDebug), C.c_int]
    return _lib


def _as_numpy_sd(state_dict):
    out = {}
    for k, v in state_dict.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        if k.endswith("num_batches_tracked"):
            continue
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def _p(a):
    return a.ctypes.data_as(_fp)


def _fill_fc(w, sd, nb):
    w.L4.w, w.L4.b = _p(sd["L4.weight"]), _p(sd["L4.bias"])
    for i in range(nb):
        w.L5[i].w, w.L5[i].b = _p(sd[f"L5_{i + 1}.weight"]), _p(sd[f"L5_{i + 1}.bias"])
        w.head[i].w, w.head[i].b = _p(sd[f"{_HEADS[i]}.weight"]), _p(sd[f"{_HEADS[i]}.bias"])


def pileup_forward(state_dict, x, add_indel_length=False, debug=False, n_threads=0):
    """Oracle for Clair3_P(predict=True).forward (clair3/model.py:130-161). x: (B,33,C) int8|int32."""
    sd = _as_numpy_sd(state_dict)
    x = np.ascontiguousarray(x)
    if x.dtype not in (np.int8, np.int32):
        raise TypeError(f"pileup oracle takes int8/int32 windows, got {x.dtype}")
    B, T, Cn = x.shape
    w = _PileupWeights()
    for layer, dst in (("LSTM1", w.lstm1), ("LSTM2", w.lstm2)):
        for d, sfx in enumerate(("", "_reverse")):
            dst[d].w_ih = _p(sd[f"{layer}.weight_ih_l0{sfx}"])
            dst[d].w_hh = _p(sd[f"{layer}.weight_hh_l0{sfx}"])
            dst[d].b_ih = _p(sd[f"{layer}.bias_ih_l0{sfx}"])
            dst[d].b_hh = _p(sd[f"{layer}.bias_hh_l0{sfx}"])
    nb = 4 if add_indel_length else 2
    _fill_fc(w, sd, nb)
    y = np.empty((B, 90 if add_indel_length else 24), dtype=np.float32)
    dbg, dump = None, {}
    if debug:
        dump = {"lstm1_out": np.empty((B, T, 256), np.float32), "lstm2_out": np.empty((B, T, 320), np.float32),
                "l4_out": np.empty((B, 128), np.float32)}
        dbg = _PileupDebug(_p(dump["lstm1_out"]), _p(dump["lstm2_out"]), _p(dump["l4_out"]))
    rc = lib().c3o_pileup_forward(C.byref(w), x.ctypes.data, x.dtype.itemsize, B, T, Cn, int(add_indel_length),
                                  _p(y), C.byref(dbg) if dbg else None, n_threads)
    if rc != 0:
        raise RuntimeError(f"c3o_pileup_forward failed rc={rc}")
    return (y, dump) if debug else y


def fa_forward(state_dict, x, add_indel_length=True, debug=False, n_threads=0):
    """Oracle for Clair3_F(predict=True).forward (clair3/model.py:377-416). x: (B,H,W,C) int8 NHWC."""
    sd = _as_numpy_sd(state_dict)
    x = np.ascontiguousarray(x)
    if x.dtype != np.int8:
        raise TypeError(f"full-alignment oracle takes int8 windows, got {x.dtype}")
    B, H, W, Cn = x.shape
    w = _FaWeights()
    for i, (cv, bn) in enumerate(_FA_CONVS):
        w.conv[i].w, w.conv[i].b = _p(sd[f"{cv}.weight"]), _p(sd[f"{cv}.bias"])
        w.conv[i].bn_w, w.conv[i].bn_b = _p(sd[f"{bn}.weight"]), _p(sd[f"{bn}.bias"])
        w.conv[i].bn_mean, w.conv[i].bn_var = _p(sd[f"{bn}.running_mean"]), _p(sd[f"{bn}.running_var"])
    nb = 4 if add_indel_length else 2
    _fill_fc(w, sd, nb)
    y = np.empty((B, 90 if add_indel_length else 24), dtype=np.float32)
    dbg, dump = None, {}
    if debug:
        dbg = _FaDebug()
        h, wd = H, W
        for i in range(9):
            h, wd = (h - 1) // _FA_STRIDE[i] + 1, (wd - 1) // _FA_STRIDE[i] + 1
            dump[f"act{i}"] = np.empty((B, h, wd, _FA_COUT[i]), np.float32)
            dbg.act[i] = _p(dump[f"act{i}"])
        dump["spp"] = np.empty((B, 3584), np.float32)
        dump["l4_out"] = np.empty((B, 256), np.float32)
        dbg.spp, dbg.l4_out = _p(dump["spp"]), _p(dump["l4_out"])
    rc = lib().c3o_fa_forward(C.byref(w), x.ctypes.data, B, H, W, Cn, int(add_indel_length), _p(y),
                              C.byref(dbg) if dbg else None, n_threads)
    if rc != 0:
        raise RuntimeError(f"c3o_fa_forward failed rc={rc}")
    return (y, dump) if debug else y


def forward(kind, state_dict, x, add_indel_length, **kw):
    if kind == "pileup":
        return pileup_forward(state_dict, x, add_indel_length, **kw)
    return fa_forward(state_dict, x, add_indel_length, **kw)
