"""Host-side mirror of the reference's model interface for the hot path.

``Clair3_P`` / ``Clair3_F`` take the same constructor arguments as the reference modules
(/root/reference/clair3/model.py:61, :285), load the same state_dict / ``.pt`` checkpoints
(clair3/CallVariantsFromCffi.py:19-28) and are called the same way -- ``Y = model(X)`` -- but the forward
pass runs in hand-written HIP kernels (libc3hip.so) instead of ATen.  PyTorch is used only to read
checkpoints and, optionally, to own device tensors handed to ``model(X)``.
"""
import ctypes as C

import numpy as np

from . import _lib

_NP_DTYPE = {np.dtype(np.int8): _lib.DTYPE_I8, np.dtype(np.int32): _lib.DTYPE_I32}


def _device_index(device):
    """int | 'cuda' | 'cuda:1' | torch.device -> HIP device ordinal (a CPU device is an error: no fallback)."""
    if device is None:
        return 0
    if isinstance(device, int):
        return device
    s = str(device)
    if s.startswith("cuda") or s.startswith("hip"):
        return int(s.split(":")[1]) if ":" in s else 0
    raise _lib.C3Error(f"clair3_amd models only run on an MI355X HIP device, not on {s!r} (there is no CPU path)")


class _HipModel:
    KIND = None
    DEFAULT_CHANNELS = None

    def __init__(self, add_indel_length=False, predict=False, input_channels=None, device=None):
        self.add_indel_length = bool(add_indel_length)
        self.predict = bool(predict)
        self.input_channels = self.DEFAULT_CHANNELS if input_channels is None else int(input_channels)
        self.output_size = 90 if self.add_indel_length else 24
        self._handle = None
        self._device = None
        self._pending_sd = None
        self._keep = False
        self._geometry = None
        self._decode_cols = False
        if device is not None:
            self.to(device)

    # ---- torch.nn.Module look-alikes used by the reference call sites ----
    def to(self, device):
        idx = _device_index(device)
        if self._handle is not None and idx == self._device:
            return self
        sd = self._pending_sd
        self._destroy()
        self._device = idx
        h = _lib.lib().c3_model_create(self.KIND, self.input_channels, int(self.add_indel_length), idx)
        if not h:
            raise _lib.C3Error(f"c3_model_create: {_lib.last_error()}")
        self._handle = C.c_void_p(h)
        if self._geometry is not None:
            _lib.check(_lib.lib().c3_model_set_geometry(self._handle, *self._geometry), "c3_model_set_geometry")
        if self._keep:
            _lib.check(_lib.lib().c3_debug_keep_activations(self._handle, 1), "c3_debug_keep_activations")
        if self._decode_cols:
            _lib.check(_lib.lib().c3_model_set_decode_columns(self._handle, 1), "c3_model_set_decode_columns")
        if sd is not None:
            self._load(sd)
        return self

    def cuda(self, device=0):
        return self.to(device)

    def eval(self):  # inference only: dropout is identity, BatchNorm uses running statistics
        return self

    def train(self, mode=True):
        if mode:
            raise _lib.C3Error("clair3_amd implements the inference path only")
        return self

    def set_geometry(self, depth=89, positions=33):
        self._geometry = (int(depth), int(positions))
        if self._handle is not None:
            _lib.check(_lib.lib().c3_model_set_geometry(self._handle, *self._geometry), "c3_model_set_geometry")
            if self._pending_sd is not None:
                self._load(self._pending_sd)
        return self

    DECODE_COLS = 31  # C3_DECODE_COLS (include/c3hip.h)

    @property
    def row_size(self):
        """floats per output row: output_size, + DECODE_COLS decoder columns when decode_columns() is on"""
        return self.output_size + (self.DECODE_COLS if self._decode_cols else 0)

    def decode_columns(self, enable=True):
        """Append the decoder columns (clair3_amd/decode.py) to every output row."""
        self._decode_cols = bool(enable)
        if self._handle is not None:
            _lib.check(_lib.lib().c3_model_set_decode_columns(self._handle, int(self._decode_cols)),
                       "c3_model_set_decode_columns")
        return self

    def load_state_dict(self, state_dict, strict=True):
        """Same contract as nn.Module.load_state_dict(strict=True): missing / unexpected / mis-shaped keys raise."""
        if not strict:
            raise _lib.C3Error("only strict loading is supported (as the reference inference loaders do)")
        sd = {}
        for k, v in state_dict.items():
            if hasattr(v, "detach"):
                v = v.detach().cpu().numpy()
            v = np.asarray(v)
            if k.endswith("num_batches_tracked"):
                continue
            sd[k] = np.ascontiguousarray(v, dtype=np.float32)
        self._pending_sd = sd
        if self._handle is None:
            self.to(0)
        else:
            self._load(sd)
        return self

    def _load(self, sd):
        n = len(sd)
        descs = (_lib.TensorDesc * n)()
        keep = []
        for i, (k, v) in enumerate(sd.items()):
            name = k.encode()
            keep.append(name)
            descs[i].name = name
            descs[i].dtype = _lib.DTYPE_F32
            descs[i].ndim = v.ndim
            for j, s in enumerate(v.shape):
                descs[i].shape[j] = s
            descs[i].data = v.ctypes.data
        rc = _lib.lib().c3_model_load(self._handle, descs, n)
        if rc != 0:
            raise _lib.C3Error(f"Error(s) in loading state_dict for {type(self).__name__}: {_lib.last_error()}")

    # ---- the forward pass ----
    def __call__(self, x):
        return self.forward(x)

    def forward(self, x, checked=False):
        """x: numpy (host) or torch tensor (host or cuda).  Returns the same container kind holding the
        (B, 24|90) float32 probabilities (``predict=True`` layout of the reference, model.py:152-159).
        Host inputs always carry the fp16-range guard (c3_predict_wait); for a cuda tensor ``checked=True`` selects
        c3_predict_device_checked (synchronises the stream), the default stays asynchronous and unchecked --
        ``range_status()`` tells afterwards whether any batch raised the flag."""
        if self._handle is None:
            raise _lib.C3Error("model has no device/weights yet: call .to(device) and .load_state_dict() first")
        if not self.predict:
            raise _lib.C3Error("only predict=True (concatenated heads) is implemented, as the call sites use "
                               "(clair3/CallVariantsFromCffi.py:232,243)")
        is_torch = hasattr(x, "data_ptr") and hasattr(x, "is_cuda")
        if is_torch and x.is_cuda:
            import torch
            if x.device.index not in (None, self._device) and x.device.index != self._device:
                raise _lib.C3Error(f"input on cuda:{x.device.index} but model on device {self._device}")
            x = x.contiguous()
            dt = {torch.int8: _lib.DTYPE_I8, torch.int32: _lib.DTYPE_I32}.get(x.dtype)
            if dt is None:
                raise _lib.C3Error(f"unsupported window dtype {x.dtype}")
            self._check_shape(tuple(x.shape), dt)
            y = torch.empty((x.shape[0], self.row_size), dtype=torch.float32, device=x.device)
            stream = torch.cuda.current_stream(x.device).cuda_stream
            fn = _lib.lib().c3_predict_device_checked if checked else _lib.lib().c3_predict_device
            _lib.check(fn(self._handle, x.data_ptr(), dt, x.shape[0], y.data_ptr(), C.c_void_p(stream)),
                       "c3_predict_device_checked" if checked else "c3_predict_device")
            return y
        xn = x.numpy() if is_torch else np.asarray(x)
        y = self.predict_numpy(xn)
        if is_torch:
            import torch
            return torch.from_numpy(y)
        return y

    def _check_shape(self, shape, dt):
        if self.KIND == _lib.KIND_FULL_ALIGNMENT and len(shape) == 4 and shape[-1] == self.input_channels:
            # the reference network is convolutional + pyramid pooling: the same module takes the 89-row ONT matrix
            # and the 55-row hifi / ilmn one (shared/param_f.py:11), and its call sites construct it without naming
            # the depth (clair3/CallVariantsFromCffi.py:239-243) -- follow the tensor
            geometry = (int(shape[1]), int(shape[2]))
            if geometry != (self._geometry or (89, 33)):
                self.set_geometry(*geometry)
        wbytes = _lib.lib().c3_model_window_bytes(self._handle, dt)
        item = 4 if dt == _lib.DTYPE_I32 else 1
        n = 1
        for s in shape[1:]:
            n *= s
        if n * item != wbytes or shape[-1] != self.input_channels:
            raise _lib.C3Error(f"window shape {shape[1:]} does not match the model "
                               f"({wbytes // item} elements per window, {self.input_channels} channels)")

    def predict_numpy(self, x):
        x = np.ascontiguousarray(x)
        dt = _NP_DTYPE.get(x.dtype)
        if dt is None:
            raise _lib.C3Error(f"unsupported window dtype {x.dtype} (int8 / int32 expected)")
        self._check_shape(x.shape, dt)
        y = np.empty((x.shape[0], self.row_size), dtype=np.float32)
        _lib.check(_lib.lib().c3_predict(self._handle, x.ctypes.data, dt, x.shape[0], y.ctypes.data), "c3_predict")
        return y

    def submit(self, x, slot=0):
        """Asynchronous half of predict_numpy (c3_predict_submit); returns a handle for wait()."""
        x = np.ascontiguousarray(x)
        dt = _NP_DTYPE.get(x.dtype)
        if dt is None:
            raise _lib.C3Error(f"unsupported window dtype {x.dtype} (int8 / int32 expected)")
        self._check_shape(x.shape, dt)
        y = np.empty((x.shape[0], self.row_size), dtype=np.float32)
        _lib.check(_lib.lib().c3_predict_submit(self._handle, x.ctypes.data, dt, x.shape[0], y.ctypes.data, slot),
                   "c3_predict_submit")
        return slot, y

    def submit_dev(self, x, y_dev_ptr, slot=0):
        """submit() with the rows left on the device: the forward pass writes them at device address ``y_dev_ptr``
        (len(x) * row_size floats; c3_predict_submit_dev) -- what a rank of a sharded job does with rows that go to the gather.
        wait() on the ticket returns None (it still runs the range guard)."""
        x = np.ascontiguousarray(x)
        dt = _NP_DTYPE.get(x.dtype)
        if dt is None:
            raise _lib.C3Error(f"unsupported window dtype {x.dtype} (int8 / int32 expected)")
        self._check_shape(x.shape, dt)
        _lib.check(_lib.lib().c3_predict_submit_dev(self._handle, x.ctypes.data, dt, x.shape[0], C.c_void_p(int(y_dev_ptr)), slot),
                   "c3_predict_submit_dev")
        return slot, None

    def wait(self, ticket):
        slot, y = ticket
        _lib.check(_lib.lib().c3_predict_wait(self._handle, slot), "c3_predict_wait")
        return y

    def sharing(self, handles=1):
        """Tell the handle how many handles feed its GPU side by side (c3_model_set_sharing; a speed hint only)."""
        _lib.check(_lib.lib().c3_model_set_sharing(self._handle, int(handles)), "c3_model_set_sharing")
        return self

    def describe(self):
        """which kernel forms the last forward pass took (c3_model_describe)"""
        buf = C.create_string_buffer(512)
        _lib.check(_lib.lib().c3_model_describe(self._handle, buf, 512), "c3_model_describe")
        return buf.value.decode()

    def range_status(self):
        """(flag, on_fp32): flag != 0 when an fp16x3 batch of this handle produced an activation near the fp16 range
        (or a non-finite row under the checked entry); on_fp32 when the handle has switched to fp32 matrix
        instructions.  Synchronises the device."""
        f, o = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().c3_model_range_status(self._handle, C.byref(f), C.byref(o)), "c3_model_range_status")
        return f.value, bool(o.value)

    def synchronize(self):
        _lib.check(_lib.lib().c3_model_synchronize(self._handle), "c3_model_synchronize")

    # ---- introspection (parity tests / bench) ----
    def keep_activations(self, enable=True):
        self._keep = bool(enable)
        if self._handle is not None:
            _lib.check(_lib.lib().c3_debug_keep_activations(self._handle, int(enable)), "c3_debug_keep_activations")
        return self

    def debug_fetch(self, name, shape):
        out = np.empty(shape, dtype=np.float32)
        _lib.check(_lib.lib().c3_debug_fetch(self._handle, name.encode(), out.ctypes.data, out.size), "c3_debug_fetch")
        return out

    def profile(self, enable=True):
        _lib.check(_lib.lib().c3_profile_enable(self._handle, int(enable)), "c3_profile_enable")

    def profile_reset(self):
        _lib.check(_lib.lib().c3_profile_reset(self._handle), "c3_profile_reset")

    def profile_read(self):
        buf = (_lib.KernelStat * 64)()
        n = _lib.lib().c3_profile_read(self._handle, buf, 64)
        if n < 0:
            raise _lib.C3Error(f"c3_profile_read: {_lib.last_error()}")
        return [dict(name=buf[i].name.decode(), launches=buf[i].launches, total_ms=buf[i].total_ms,
                     flops=buf[i].flops, bytes=buf[i].bytes, mfma_flops=buf[i].mfma_flops,
                     mfma_peak_tflops=buf[i].mfma_peak_tflops) for i in range(min(n, 64))]

    def _destroy(self):
        if self._handle is not None:
            _lib.lib().c3_model_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass


class Clair3_P(_HipModel):
    """Pileup network: 2 x bidirectional LSTM + FC heads (reference: clair3/model.py:58-161)."""
    KIND = _lib.KIND_PILEUP
    DEFAULT_CHANNELS = 18  # shared/param_p.py:32-36

    def predict_region(self, region, starts):
        """Rows for the windows region[starts[b] : starts[b] + 33] without materialising them: ``region`` is the
        (n_cols, 18) matrix of one pileup region, ``starts`` the per-candidate offsets the reference slices at
        (preprocess/CreateTensorPileupFromCffi.py:362-364).  Same rows, bit for bit, as predict_numpy on the slices."""
        region = np.ascontiguousarray(region)
        # int64 / uint64: the size_t matrix of plp_data viewed in place (np.frombuffer(ffi.buffer(plp_data.matrix, ...)),
        # CreateTensorPileupFromCffi.py:140-146) -- no .copy(), no astype
        dt = _lib.DTYPE_I64 if region.dtype in (np.dtype(np.int64), np.dtype(np.uint64)) else _NP_DTYPE.get(region.dtype)
        if dt is None or region.ndim != 2 or region.shape[1] != self.input_channels:
            raise _lib.C3Error(f"region must be (n_cols, {self.input_channels}) int8/int32/int64, got {region.dtype} {region.shape}")
        starts = np.ascontiguousarray(starts, dtype=np.int32)
        y = np.empty((len(starts), self.row_size), dtype=np.float32)
        _lib.check(_lib.lib().c3_predict_pileup_region(self._handle, region.ctypes.data, dt, region.shape[0],
                                                       starts.ctypes.data, len(starts), y.ctypes.data),
                   "c3_predict_pileup_region")
        return y


class Clair3_F(_HipModel):
    """Full-alignment network: residual 3x3-conv stack + pyramid pooling + FC heads (clair3/model.py:282-416)."""
    KIND = _lib.KIND_FULL_ALIGNMENT
    DEFAULT_CHANNELS = 8  # shared/param_f.py:24-31 (9 with --enable_dwell_time)
