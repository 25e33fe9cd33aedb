"""Drop-in replacements for the model-call helpers of the reference worker
(/root/reference/clair3/CallVariantsFromCffi.py:19-52) and the GPU-slot probe of the GPU wrapper
(/root/reference/clair3/CallVariantsFromCffiGPU.py:13-43).  Same names, same argument meaning, same
error behaviour (loaders raise; callers keep their sys.exit handling).
"""
import numpy as np

from . import _lib
from .model import Clair3_F, Clair3_P, _device_index


def _load_torch_checkpoint(model, checkpoint_path, device=None):
    """clair3/CallVariantsFromCffi.py:19-28: append '.pt' when missing, torch.load, accept a bare state_dict or
    {"state_dict": ...}, strict load.  torch is only used to deserialise the file (map_location='cpu')."""
    import torch
    if not checkpoint_path.endswith('.pt'):
        checkpoint_path = checkpoint_path + '.pt'
    checkpoint = torch.load(checkpoint_path, map_location="cpu")
    if isinstance(checkpoint, dict) and "state_dict" in checkpoint:
        state_dict = checkpoint["state_dict"]
    else:
        state_dict = checkpoint
    model.load_state_dict(state_dict)


def _select_device(use_gpu=True):
    """clair3/CallVariantsFromCffi.py:31-34 returns cpu when no GPU is usable; this path has no CPU
    implementation, so an unusable GPU is an error instead of a silent fallback."""
    if not use_gpu:
        raise _lib.C3Error("clair3_amd is the GPU path; run the reference for CPU inference")
    if _lib.device_count() < 1:
        raise _lib.C3Error("no MI355X / HIP device visible")
    return "cuda:0"


def _hip_predict(model, device, X):
    """_torch_predict(model, device, X) (clair3/CallVariantsFromCffi.py:48-52): numpy windows in, numpy
    float32 (B, 24|90) probabilities out; H2D, forward and D2H are done by libc3hip (pinned staging)."""
    if device is not None and model._device is not None and _device_index(device) != model._device:
        model.to(device)
    return model.predict_numpy(np.asarray(X))


_torch_predict = _hip_predict  # the name the reference call sites use


def build_model(pileup, add_indel_length, platform="ont", enable_dwell_time=False, device=0, chkpnt_fn=None):
    """Model factory block of call_variants_from_cffi (clair3/CallVariantsFromCffi.py:223-248)."""
    if platform != "ont":
        # hifi/ilmn use a 55-row matrix (shared/param_f.py:11); supported by geometry, but only ONT is validated
        depth = 55
    else:
        depth = 89
    if pileup:
        m = Clair3_P(add_indel_length=add_indel_length, predict=True, input_channels=18)
    else:
        m = Clair3_F(add_indel_length=add_indel_length, predict=True, input_channels=9 if enable_dwell_time else 8)
        m.set_geometry(depth, 33)
    m.to(device)
    m.eval()
    if chkpnt_fn is not None:
        _load_torch_checkpoint(m, chkpnt_fn, device)
    return m


def get_gpu_memory():
    """[(device_index, free_MB)] like the nvidia-smi parser it replaces
    (clair3/CallVariantsFromCffiGPU.py:13-19) but through hipMemGetInfo."""
    out = []
    for d in range(_lib.device_count()):
        free_b, _ = _lib.mem_info(d)
        out.append((d, free_b // (1024 * 1024)))
    return out


def check_gpu_memory(min_memory_mb, device_list=None):
    """Slots per device = free_MB // min_memory_mb (clair3/CallVariantsFromCffiGPU.py:21-43; the reference
    uses 5000 MB per pileup worker, 8000 MB per full-alignment worker, :55-56)."""
    gpu_thread_dict = {}
    total = 0
    for d, free_mb in get_gpu_memory():
        if device_list is not None and d not in device_list:
            continue
        n = int(free_mb // min_memory_mb)
        gpu_thread_dict[d] = n
        total += n
    if total == 0:
        raise _lib.C3Error(f"No GPU has {min_memory_mb} MB free")
    return gpu_thread_dict, total
