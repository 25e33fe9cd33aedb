"""``import torch`` postponed until something needs it.

The reference's stage-B worker is a process per GPU slot and stage (clair3/CallVariantsFromCffiGPU.py:138-199), and its modules ``import torch`` at the
top (clair3/CallVariantsFromCffi.py:5, clair3/CallVariants.py:4).  With the model call, the loader and the device selection rebound to
libc3hip (clair3_amd/callvar.py) and the checkpoint read by clair3_amd/ptfile.py, what the UNMODIFIED worker still asks torch for on the
GPU branch is

    torch.set_num_threads(n), torch.set_num_interop_threads(n)      (Run, clair3/CallVariantsFromCffi.py:62-63, CallVariants.py:210-211)
    torch.device("cpu")                                              (call_variants_from_cffi :203, overwritten on the GPU branch :217)

-- and the import is 1.2 - 1.9 s of a process whose loop over 120 000 windows takes 0.2 s.  ``install()`` puts a stand-in module into
``sys.modules['torch']`` that answers exactly those three by itself (the thread counts are remembered and handed to torch if it is ever
loaded) and imports the real package on ANY other attribute, forwarding to it from then on: nothing can behave differently, it can only
happen later or not at all.  ``callvar.install()`` uses it (C3HIP_LAZY_TORCH=0: never) when torch has not been imported yet.
``status()`` says whether the real package was loaded and what asked for it; C3HIP_LAZY_TORCH_REPORT=<file> writes that at exit.
"""
import importlib
import json
import os
import sys
import threading
import types

_LOCK = threading.RLock()


class _Device:
    """what torch.device(spec) has to be for code that only passes it around: .type, .index, str() like torch's"""

    def __init__(self, spec="cpu", index=None):
        if isinstance(spec, _Device):
            spec, index = spec.type, spec.index if index is None else index
        if isinstance(spec, int):
            spec, index = "cuda", spec
        spec = str(spec)
        if ":" in spec:
            spec, _, idx = spec.partition(":")
            index = int(idx)
        self.type, self.index = spec, index

    def __str__(self):
        return self.type if self.index is None else f"{self.type}:{self.index}"

    def __repr__(self):
        return f"device(type='{self.type}')" if self.index is None else f"device(type='{self.type}', index={self.index})"

    def __eq__(self, other):
        return str(self) == str(other)

    def __hash__(self):
        return hash(str(self))


class _LazyTorch(types.ModuleType):
    def __init__(self):
        super().__init__("torch", "stand-in of clair3_amd.lazy_torch: the real package is imported on first real use")
        d = self.__dict__
        d["_c3_real"], d["_c3_threads"], d["_c3_interop"], d["_c3_first_touch"] = None, None, None, None

    # ---- answered without the package
    def set_num_threads(self, n):
        d = self.__dict__
        d["_c3_threads"] = int(n)
        if d["_c3_real"] is not None:
            d["_c3_real"].set_num_threads(int(n))

    def set_num_interop_threads(self, n):
        d = self.__dict__
        d["_c3_interop"] = int(n)
        if d["_c3_real"] is not None:
            d["_c3_real"].set_num_interop_threads(int(n))

    def device(self, *args, **kwargs):
        real = self.__dict__["_c3_real"]
        if real is not None:
            return real.device(*args, **kwargs)
        return _Device(*args, **kwargs)

    # ---- everything else: the package
    def _c3_load(self, why):
        d = self.__dict__
        if d["_c3_real"] is None:
            with _LOCK:
                if d["_c3_real"] is None:
                    if sys.modules.get("torch") is self:
                        del sys.modules["torch"]
                    try:
                        real = importlib.import_module("torch")
                    except BaseException:
                        sys.modules.setdefault("torch", self)  # still importable on the next touch
                        raise
                    d["_c3_first_touch"] = why
                    if d["_c3_threads"] is not None:
                        real.set_num_threads(d["_c3_threads"])
                    if d["_c3_interop"] is not None:
                        try:
                            real.set_num_interop_threads(d["_c3_interop"])
                        except RuntimeError:  # torch allows it once, before any parallel work
                            pass
                    d["_c3_real"] = real
        return d["_c3_real"]

    def __getattr__(self, name):  # only reached for names the stand-in does not have
        return getattr(self._c3_load(name), name)

    def __dir__(self):
        return dir(self._c3_load("dir()"))


_STANDIN = None  # the one stand-in made in this process (once the real package is loaded it is no longer what sys.modules holds)


def install():
    """-> True when the stand-in is (now or already) what ``import torch`` finds; False when the real package is already imported."""
    global _STANDIN
    cur = sys.modules.get("torch")
    if isinstance(cur, _LazyTorch):
        return True
    if cur is not None:
        return False
    _STANDIN = sys.modules["torch"] = _LazyTorch()
    report = os.environ.get("C3HIP_LAZY_TORCH_REPORT", "").strip()
    if report:
        import atexit
        pid = os.getpid()

        def write():
            if os.getpid() == pid:  # not the forked decode processes
                with open(report, "w") as fh:
                    json.dump(status(), fh)
        atexit.register(write)
    return True


def standing_in():
    """the stand-in is what ``import torch`` finds right now (the real package has not been imported)"""
    return isinstance(sys.modules.get("torch"), _LazyTorch)


def wanted():
    return os.environ.get("C3HIP_LAZY_TORCH", "1").strip().lower() not in ("0", "false", "no", "off")


def status():
    """{"installed": a stand-in was put in place in this process, "real_loaded": torch itself has been imported,
    "first_touch": the attribute that made the stand-in import it}"""
    if _STANDIN is None:
        return {"installed": False, "real_loaded": "torch" in sys.modules, "first_touch": None}
    d = _STANDIN.__dict__
    return {"installed": True, "real_loaded": d["_c3_real"] is not None, "first_touch": d["_c3_first_touch"]}


def standin_module(name, attrs, load_real):
    """A module object ``name`` that has ``attrs`` and becomes the real module (``load_real()`` -> module) on any other attribute: used for
    clair3.model, whose two classes are rebound anyway and whose import needs torch.nn at class-definition time."""
    mod = types.ModuleType(name, "stand-in of clair3_amd.lazy_torch")
    mod.__dict__.update(attrs)
    state = {"real": None}

    def __getattr__(attr):
        if attr.startswith("__") and attr.endswith("__"):
            raise AttributeError(attr)
        if state["real"] is None:
            with _LOCK:
                if state["real"] is None:
                    state["real"] = load_real()
        return getattr(state["real"], attr)

    mod.__dict__["__getattr__"] = __getattr__
    return mod
