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


# set by callvar.install(decoder=True): rows of models with the indel-length heads also carry the decoder columns
# (clair3_amd/decode.py) that the rebound batch_output consumes
DECODER_COLUMNS = False


def _hip_predict(model, device, X):
    """_torch_predict(model, device, X) (clair3/CallVariantsFromCffi.py:48-52): numpy windows in, numpy
    float32 (B, 24|90) probabilities out; H2D, forward and D2H are done by libc3hip (pinned staging).
    With DECODER_COLUMNS the 90-column rows are followed by model.DECODE_COLS decoder columns."""
    if device is not None and model._device is not None and _device_index(device) != model._device:
        model.to(device)
    want = bool(DECODER_COLUMNS and model.add_indel_length)
    if want != model._decode_cols:
        model.decode_columns(want)
    return model.predict_numpy(np.asarray(X))


_torch_predict = _hip_predict  # the name the reference call sites use


def build_model(pileup, add_indel_length, platform="ont", enable_dwell_time=False, device=0, chkpnt_fn=None):
    """Model factory block of call_variants_from_cffi (clair3/CallVariantsFromCffi.py:223-248)."""
    if platform != "ont":
        # hifi/ilmn use a 55-row matrix (shared/param_f.py:11); golden case fa_hifi_depth55
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


def get_gpu_memory(gpu_id=None):
    """Free device memory in MB, one entry per queried device -- same return shape as the nvidia-smi parser it
    replaces (clair3/CallVariantsFromCffiGPU.py:13-19) but through hipMemGetInfo (c3_mem_info)."""
    ids = range(_lib.device_count()) if gpu_id is None else [int(gpu_id)]
    return [int(_lib.mem_info(d)[0] // (1024 * 1024)) for d in ids]


def check_gpu_memory(memory, device_ids=None, print_log=True):
    """clair3/CallVariantsFromCffiGPU.py:21-43: one "GPU thread" (worker slot) per `memory` MB of free device
    memory; returns the device id repeated once per slot.  The reference sys.exit(1)s when nothing is usable;
    this raises C3Error and the installed wrapper (callvar.install) converts it back into the exit."""
    all_device_ids = list(range(_lib.device_count()))
    if device_ids is None:
        device_ids = all_device_ids
    if not all_device_ids:
        return
    gpu_id_list = []
    for device_id in all_device_ids:
        free_mem = get_gpu_memory(gpu_id=device_id)[0]
        gpu_threads = free_mem // memory
        gpu_id_list += [device_id] * int(gpu_threads)
        if print_log:
            print(f"GPU {device_id} free memory: {free_mem} MB, assigning {memory} MB per thread, "
                  f"{gpu_threads} threads available")
    if len(device_ids) == 0:
        raise _lib.C3Error("No GPU available, Please disabling --use_gpu for variant calling, exiting.")
    if len(gpu_id_list) == 0:
        raise _lib.C3Error("No memory in GPU, Please assign GPU memory first, exiting.")
    return gpu_id_list
