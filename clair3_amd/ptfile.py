"""A checkpoint file of ``torch.save`` read without importing torch (numpy + zipfile + a restricted unpickler).

Why: the reference's stage-B worker is one PROCESS per GPU slot and stage (clair3/CallVariantsFromCffiGPU.py:138-199 builds one command per
group of tensor files), and with the model call in libc3hip a 120 000-window job is 0.2 s of loop inside a 1.7 s process -- 1.2 - 1.5 s of it ``import
torch``, whose only remaining job on this path is to deserialise the ``.pt`` file (clair3/CallVariantsFromCffi.py:19-28).  This module does
that job: ``load(path)`` returns what ``torch.load(path, map_location="cpu")`` returns, with every tensor as a numpy array.

The format (torch/serialization.py, the zip form every torch >= 1.6 writes): a zip archive ``<root>/data.pkl`` + ``<root>/data/<key>`` +
``<root>/byteorder``; the pickle names tensors as ``torch._utils._rebuild_tensor_v2(storage, offset, size, stride, requires_grad, hooks)``
with the storage a persistent id ``('storage', <storage class>, key, location, numel)``.  The unpickler below resolves exactly the globals
such a file needs (the rebuild functions, the storage classes, OrderedDict) and refuses everything else -- like torch's ``weights_only``
unpickler, it never executes code from the file.  Anything it does not understand (the legacy non-zip format, an unknown global, a
big-endian file, bfloat16 / complex storages) raises ``Unsupported``: the caller falls back to torch.load, which is what it used before.
"""
import collections
import pickle
import zipfile

import numpy as np


class Unsupported(Exception):
    """the file is not something this reader handles (the caller uses torch.load instead)"""


class _StorageType:
    def __init__(self, name, dtype):
        self.name, self.dtype = name, dtype


_STORAGES = {
    "FloatStorage": np.float32, "DoubleStorage": np.float64, "HalfStorage": np.float16, "LongStorage": np.int64, "IntStorage": np.int32,
    "ShortStorage": np.int16, "CharStorage": np.int8, "ByteStorage": np.uint8, "BoolStorage": np.bool_,
}
_DTYPES = {  # torch.<dtype> globals (files written with untyped storages name the dtype instead of a storage class)
    "float32": np.float32, "float": np.float32, "float64": np.float64, "double": np.float64, "float16": np.float16, "half": np.float16,
    "int64": np.int64, "long": np.int64, "int32": np.int32, "int": np.int32, "int16": np.int16, "short": np.int16, "int8": np.int8,
    "uint8": np.uint8, "bool": np.bool_,
}


class _Storage:
    """one ``<root>/data/<key>`` entry, read when the first tensor on it is rebuilt"""

    def __init__(self, archive, name, dtype, numel):
        self.archive, self.name, self.dtype, self.numel, self._data = archive, name, np.dtype(dtype), int(numel), None

    def data(self):
        if self._data is None:
            try:
                raw = self.archive.read(self.name)
            except KeyError:
                raise Unsupported(f"storage {self.name} is not in the archive")
            if len(raw) < self.numel * self.dtype.itemsize:
                raise Unsupported(f"storage {self.name}: {len(raw)} bytes for {self.numel} x {self.dtype}")
            self._data = np.frombuffer(raw, dtype=self.dtype, count=self.numel)
        return self._data


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    if not isinstance(storage, _Storage):
        raise Unsupported("a tensor on something that is not a storage of the archive")
    size, stride, storage_offset = tuple(int(s) for s in size), tuple(int(s) for s in stride), int(storage_offset)
    flat = storage.data()
    if any(s < 0 for s in size) or any(s < 0 for s in stride) or storage_offset < 0:
        raise Unsupported("negative size / stride / offset")
    last = storage_offset + sum((n - 1) * s for n, s in zip(size, stride)) if all(size) else storage_offset
    if all(size) and last >= flat.size:
        raise Unsupported(f"a tensor of size {size} / stride {stride} at {storage_offset} does not fit its storage of {flat.size}")
    if not all(size):
        return np.empty(size, dtype=flat.dtype)
    item = flat.dtype.itemsize
    view = np.lib.stride_tricks.as_strided(flat[storage_offset:], shape=size, strides=tuple(s * item for s in stride), writeable=False)
    return np.array(view, order="C")  # an owned, C-contiguous copy (nothing of the zip buffer stays referenced)


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None):
    return data


def _rebuild_parameter_with_state(data, requires_grad=False, backward_hooks=None, state=None):
    return data


_GLOBALS = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("torch._utils", "_rebuild_tensor_v2"): _rebuild_tensor_v2,
    ("torch._utils", "_rebuild_parameter"): _rebuild_parameter,
    ("torch._utils", "_rebuild_parameter_with_state"): _rebuild_parameter_with_state,
    ("torch", "Size"): tuple,
}


class _Unpickler(pickle.Unpickler):
    def __init__(self, fh, archive, root):
        super().__init__(fh)
        self.archive, self.root, self.storages = archive, root, {}

    def find_class(self, module, name):
        if (module, name) in _GLOBALS:
            return _GLOBALS[(module, name)]
        if module == "torch" and name in _STORAGES:
            return _StorageType(name, _STORAGES[name])
        if module == "torch" and name in _DTYPES:
            return _StorageType(name, _DTYPES[name])
        raise Unsupported(f"global {module}.{name}")

    def persistent_load(self, pid):
        if not (isinstance(pid, tuple) and len(pid) >= 5 and pid[0] == "storage" and isinstance(pid[1], _StorageType)):
            raise Unsupported(f"persistent id {pid!r:.80}")
        _, stype, key, _location, numel = pid[:5]
        key = str(key)
        if key not in self.storages:
            self.storages[key] = _Storage(self.archive, f"{self.root}/data/{key}", stype.dtype, numel)
        return self.storages[key]


def load(path):
    """-> the object ``torch.load(path, map_location='cpu')`` would return, tensors as C-contiguous numpy arrays of the same dtype and shape.
    Raises Unsupported for anything but a little-endian zip checkpoint of plain containers and dense tensors; OSError for a missing file."""
    with open(path, "rb") as probe:
        if probe.read(4) != b"PK\x03\x04":
            raise Unsupported("not a zip archive (the legacy torch.save format, or not a checkpoint)")
    try:
        archive = zipfile.ZipFile(path)
    except zipfile.BadZipFile as exc:
        raise Unsupported(f"unreadable zip archive: {exc}")
    with archive:
        pkls = [n for n in archive.namelist() if n.endswith("/data.pkl") and n.count("/") == 1]
        if len(pkls) != 1:
            raise Unsupported("no <root>/data.pkl in the archive")
        root = pkls[0].split("/")[0]
        if f"{root}/byteorder" in archive.namelist() and archive.read(f"{root}/byteorder").strip() != b"little":
            raise Unsupported("a big-endian checkpoint")
        with archive.open(pkls[0]) as fh:
            try:
                return _Unpickler(fh, archive, root).load()
            except Unsupported:
                raise
            except (pickle.UnpicklingError, AttributeError, EOFError, IndexError, TypeError, ValueError, KeyError) as exc:
                raise Unsupported(f"unreadable pickle: {type(exc).__name__}: {exc}")
