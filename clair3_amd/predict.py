"""Drop-in replacements for the model-call helpers of the reference worker
(/root/reference/clair3/CallVariantsFromCffi.py:19-52) and the GPU-slot probe of the GPU wrapper
(/root/reference/clair3/CallVariantsFromCffiGPU.py:13-43).  Same names, same argument meaning, same
error behaviour (loaders raise; callers keep their sys.exit handling).
"""
import numpy as np

from . import _lib
from .model import Clair3_F, Clair3_P, _device_index


def _load_torch_checkpoint(model, checkpoint_path, device=None):
    """clair3/CallVariantsFromCffi.py:19-28: append '.pt' when missing, deserialise, accept a bare state_dict or {"state_dict": ...},
    strict load.  The file is read by clair3_amd/ptfile.py (numpy only: a worker process that never needs torch never imports it,
    clair3_amd/lazy_torch.py); what that reader does not handle -- the legacy non-zip format, exotic dtypes, pickled modules -- goes to
    torch.load(map_location='cpu') as before (C3HIP_PTFILE=0: always)."""
    import os
    from . import ptfile
    if not checkpoint_path.endswith('.pt'):
        checkpoint_path = checkpoint_path + '.pt'
    checkpoint = None
    if os.environ.get("C3HIP_PTFILE", "1").strip().lower() not in ("0", "false", "no", "off"):
        try:
            checkpoint = ptfile.load(checkpoint_path)
        except ptfile.Unsupported:
            checkpoint = None
    if checkpoint is None:
        import torch
        checkpoint = torch.load(checkpoint_path, map_location="cpu")
    if isinstance(checkpoint, dict) and "state_dict" in checkpoint:
        state_dict = checkpoint["state_dict"]
    else:
        state_dict = checkpoint
    model.load_state_dict(state_dict)
    _register_current(model)


# The model the worker process is calling variants with: the reference loop creates ONE model, loads it here, and only then
# creates its batch generator (clair3/CallVariantsFromCffi.py:246-273), so the rebound generator (callvar.install: the
# transport of clair3_amd/worker.py behind tensor_generator_for_chunk) can find the handle it should run ahead on.
_CURRENT_MODEL = None
# id(X) -> (model, group, X, lo, hi) of batches the rebound generator has already submitted (worker.lookahead_batches)
_PENDING = {}


def _register_current(model):
    global _CURRENT_MODEL
    import weakref
    _CURRENT_MODEL = weakref.ref(model) if hasattr(model, "submit") else None


def current_model():
    return _CURRENT_MODEL() if _CURRENT_MODEL is not None else None


def _select_device(use_gpu=True):
    """clair3/CallVariantsFromCffi.py:31-34 returns cpu when no GPU is usable; this path has no CPU
    implementation, so an unusable GPU is an error instead of a silent fallback."""
    if not use_gpu:
        raise _lib.C3Error("clair3_amd is the GPU path; run the reference for CPU inference")
    if _lib.device_count() < 1:
        raise _lib.C3Error("no MI355X / HIP device visible")
    return "cuda:0"


# set by callvar.install(decoder=True): the rows (24 or 90 probabilities) also carry the decoder columns
# (clair3_amd/decode.py) that the rebound batch_output consumes
DECODER_COLUMNS = False


def _hip_predict(model, device, X):
    """_torch_predict(model, device, X) (clair3/CallVariantsFromCffi.py:48-52): numpy windows in, numpy
    float32 (B, 24|90) probabilities out; H2D, forward and D2H are done by libc3hip (pinned staging).
    With DECODER_COLUMNS the rows are followed by model.DECODE_COLS decoder columns."""
    if device is not None and model._device is not None and _device_index(device) != model._device:
        model.to(device)
    ent = _PENDING.pop(id(X), None)
    if ent is not None and ent[0] is model and ent[2] is X:
        # submitted ahead by the rebound batch generator (worker.lookahead_batches), alone or in a group of consecutive
        # batches: the rows are on their way or here
        return ent[1].take(ent[3], ent[4])
    want = bool(DECODER_COLUMNS)
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


# Worker processes ("GPU threads") per MI355X.  The reference sizes this for 16-80 GB CUDA cards: free_MB // 8000 (full
# alignment) or // 5000 (pileup) processes per device (clair3/CallVariantsFromCffiGPU.py:33-34,55-56) -- 36 / 57 on a 288 GB
# MI355X, each with its own HIP context, workspace (up to ~3 GB for pileup), staging threads and `cpu_threads` decode
# processes, all behind one PCIe link.  One libc3hip handle fed through its submit / wait ring already fills the chip
# (bench.py: host-inclusive rate 0.9-0.97 of the device-resident one; three handles side by side add < 10 % on the device and
# LOSE when fed from the host), so the slot count is 1 per device unless C3HIP_SLOTS_PER_GPU says otherwise
# (capped by the reference's own memory rule).
def slots_per_gpu():
    import os
    try:
        return max(1, int(os.environ.get("C3HIP_SLOTS_PER_GPU", "1")))
    except ValueError:
        return 1


def check_gpu_memory(memory, device_ids=None, print_log=True):
    """clair3/CallVariantsFromCffiGPU.py:21-43: returns the device id repeated once per worker slot.  The reference gives
    every `memory` MB of free device memory a slot; here a device gets min(that, slots_per_gpu()) slots -- see above.
    ``device_ids`` (from ``--device=cuda:2,3``, :60-65) are the PHYSICAL ids the caller exported as CUDA_VISIBLE_DEVICES
    before asking: visible ordinal i is physical device_ids[i], and that is what a slot must carry, because the slot's
    worker sets CUDA_VISIBLE_DEVICES to it (clair3/CallVariantsFromCffi.py:216; the reference returns the ordinal there).
    The reference sys.exit(1)s when nothing is usable; this raises C3Error and the installed wrapper
    (callvar.install) converts it back into the exit."""
    all_device_ids = list(range(_lib.device_count()))
    if device_ids is None:
        device_ids = all_device_ids
    if not all_device_ids:
        return
    gpu_id_list = []
    cap = slots_per_gpu()
    for device_id in all_device_ids:
        free_mem = get_gpu_memory(gpu_id=device_id)[0]
        by_memory = int(free_mem // memory)
        gpu_threads = min(by_memory, cap)
        physical = device_ids[device_id] if device_id < len(device_ids) else device_id
        gpu_id_list += [physical] * gpu_threads
        if print_log:
            print(f"GPU {physical} free memory: {free_mem} MB, assigning {memory} MB per thread, "
                  f"{gpu_threads} threads available (one libc3hip worker fills an MI355X; the reference's "
                  f"free // {memory} rule would start {by_memory}; C3HIP_SLOTS_PER_GPU overrides)")
    if len(device_ids) == 0:
        raise _lib.C3Error("No GPU available, Please disabling --use_gpu for variant calling, exiting.")
    if len(gpu_id_list) == 0:
        raise _lib.C3Error("No memory in GPU, Please assign GPU memory first, exiting.")
    return gpu_id_list
