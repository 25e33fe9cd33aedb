"""clair3_amd/ptfile.py (a torch.save file read with numpy alone) against torch.load, and clair3_amd/lazy_torch.py (``import torch``
postponed until something needs it) in fresh interpreters: what the reference's worker still asks torch for after callvar.install()
must not import it; anything else must, and must behave as if it had been there all along."""
import collections
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from clair3_amd import ptfile
from clair3_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CLAIR3_REFERENCE", "/root/reference")


@pytest.mark.parametrize("kind,channels,indel", [(syn.PILEUP, 18, False), (syn.FULL_ALIGNMENT, 8, True), (syn.FULL_ALIGNMENT, 9, True)])
@pytest.mark.parametrize("wrapped", [False, True])
def test_model_checkpoints_read_like_torch_load(tmp_path, kind, channels, indel, wrapped):
    """the files the reference's loader accepts (clair3/CallVariantsFromCffi.py:19-28: a bare state_dict, or {"state_dict": ...}): same
    keys in the same order, same dtypes and shapes, same bits"""
    import torch
    sd = syn.make_state_dict(kind, channels, indel, seed=5)
    tensors = collections.OrderedDict((k, torch.from_numpy(np.asarray(v))) for k, v in sd.items())
    tensors["bn_counter.num_batches_tracked"] = torch.tensor(1234, dtype=torch.int64)
    path = str(tmp_path / "m.pt")
    torch.save({"state_dict": tensors, "epoch": 7, "lr": 1e-3} if wrapped else tensors, path)
    got, want = ptfile.load(path), torch.load(path, map_location="cpu")
    if wrapped:
        assert got["epoch"] == 7 and got["lr"] == 1e-3
        got, want = got["state_dict"], want["state_dict"]
    assert list(got) == list(want)
    for k, w in want.items():
        w = w.numpy()
        assert got[k].dtype == w.dtype and got[k].shape == w.shape and got[k].flags["C_CONTIGUOUS"] and np.array_equal(got[k], w), k


def test_views_dtypes_and_shared_storages(tmp_path):
    import torch
    g = torch.Generator().manual_seed(0)
    base = torch.randn(10, 12, generator=g)
    obj = collections.OrderedDict(
        t=torch.randn(6, 7, generator=g).t(), s=base[2:5, 1::2], row=base[3], same=base, scalar=torch.tensor(7, dtype=torch.int64),
        half=torch.randn(3, generator=g).half(), dbl=torch.randn(3, generator=g).double(), par=torch.nn.Parameter(torch.randn(2, 2, generator=g)),
        empty=torch.empty(0, 3), flag=torch.tensor([True, False]), u8=torch.arange(5, dtype=torch.uint8), i8=torch.arange(-3, 3, dtype=torch.int8),
        i32=torch.arange(4, dtype=torch.int32), nested={"a": [torch.ones(2), (torch.zeros(1), "text", 3.5, None)]})
    path = str(tmp_path / "v.pt")
    torch.save(obj, path)
    got, want = ptfile.load(path), torch.load(path, map_location="cpu")
    for k in ("t", "s", "row", "same", "scalar", "half", "dbl", "par", "empty", "flag", "u8", "i8", "i32"):
        w = want[k].detach().numpy()
        assert got[k].dtype == w.dtype and got[k].shape == w.shape and np.array_equal(got[k], w) and got[k].flags["C_CONTIGUOUS"], k
    assert got["s"].base is None or got["s"].base is not got["same"]  # owned copies: writing one does not change another
    got["same"][3, 0] = 99.0
    assert got["row"][0] != 99.0
    a = got["nested"]["a"]
    assert np.array_equal(a[0], np.ones(2, np.float32)) and a[1][1:] == ("text", 3.5, None) and np.array_equal(a[1][0], np.zeros(1, np.float32))


def test_what_the_reader_refuses(tmp_path):
    """never executes anything from the file: every global outside the tensor-rebuilding set is refused; the legacy format, bfloat16 and a
    truncated archive are Unsupported (the loader then uses torch.load); a missing file is the OSError torch.load raises too"""
    import pickle
    import zipfile
    import torch
    sd = {"w": torch.ones(3)}
    legacy = str(tmp_path / "legacy.pt")
    torch.save(sd, legacy, _use_new_zipfile_serialization=False)
    with pytest.raises(ptfile.Unsupported, match="not a zip"):
        ptfile.load(legacy)
    bf = str(tmp_path / "bf.pt")
    torch.save({"w": torch.ones(3).bfloat16()}, bf)
    with pytest.raises(ptfile.Unsupported, match="BFloat16Storage"):
        ptfile.load(bf)
    whole = str(tmp_path / "module.pt")
    torch.save(torch.nn.Linear(2, 2), whole)
    with pytest.raises(ptfile.Unsupported, match="global torch.nn"):
        ptfile.load(whole)
    evil = str(tmp_path / "evil.pt")

    class Boom:
        def __reduce__(self):
            return (os.system, ("echo executed > " + str(tmp_path / "executed"),))
    with zipfile.ZipFile(evil, "w") as z:
        z.writestr("archive/data.pkl", pickle.dumps({"x": Boom()}, protocol=2))
        z.writestr("archive/byteorder", "little")
    with pytest.raises(ptfile.Unsupported, match="global (posix|os|nt).system"):
        ptfile.load(evil)
    assert not (tmp_path / "executed").exists()
    good = str(tmp_path / "good.pt")
    torch.save({"w": torch.arange(100, dtype=torch.float32)}, good)
    short = str(tmp_path / "short.pt")
    with zipfile.ZipFile(good) as zin, zipfile.ZipFile(short, "w") as zout:
        for item in zin.namelist():
            data = zin.read(item)
            zout.writestr(item, data[:40] if item.endswith("data/0") else data)
    with pytest.raises(ptfile.Unsupported, match="bytes for 100"):
        ptfile.load(short)
    with pytest.raises(OSError):
        ptfile.load(str(tmp_path / "absent.pt"))


def test_the_loader_uses_the_reader_and_falls_back(tmp_path, monkeypatch):
    """predict._load_torch_checkpoint: '.pt' appended, {"state_dict": ...} unwrapped, the same state dict whichever reader answered"""
    import torch
    from clair3_amd import predict
    sd = syn.make_state_dict(syn.PILEUP, 18, False, seed=3)
    tensors = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
    torch.save({"state_dict": tensors}, str(tmp_path / "zip.pt"))
    torch.save(tensors, str(tmp_path / "legacy.pt"), _use_new_zipfile_serialization=False)

    class Recorder:
        def load_state_dict(self, state_dict):
            self.sd = state_dict
    seen = []
    real = ptfile.load
    monkeypatch.setattr(ptfile, "load", lambda p: (seen.append(p), real(p))[1])
    for name, by_reader in (("zip", True), ("legacy", False)):
        m = Recorder()
        predict._load_torch_checkpoint(m, str(tmp_path / name), "cuda:0")  # no extension: appended
        assert seen[-1].endswith(name + ".pt")
        assert isinstance(next(iter(m.sd.values())), np.ndarray) == by_reader
        assert list(m.sd) == list(sd)
        for k, v in sd.items():
            assert np.array_equal(np.asarray(m.sd[k]), np.asarray(v)), k
    monkeypatch.setenv("C3HIP_PTFILE", "0")
    n = len(seen)
    m = Recorder()
    predict._load_torch_checkpoint(m, str(tmp_path / "zip.pt"))
    assert len(seen) == n and not isinstance(next(iter(m.sd.values())), np.ndarray)
    with pytest.raises(OSError):
        predict._load_torch_checkpoint(Recorder(), str(tmp_path / "absent"))


def _fresh(code, *args, env=None):
    e = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests", "stubs")]))
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", code, *args], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_the_stand_in_answers_three_calls_and_becomes_torch_on_anything_else():
    out = _fresh("""
import json, sys
from clair3_amd import lazy_torch
assert lazy_torch.install() and lazy_torch.install()
import torch
torch.set_num_threads(3); torch.set_num_interop_threads(2)
d = [str(torch.device("cpu")), str(torch.device("cuda:1")), str(torch.device("cuda", 0)), repr(torch.device("cpu")), torch.device("cpu").type]
before = lazy_torch.status()
x = torch.zeros(2)                                    # anything else: the package
after = lazy_torch.status()
import torch.nn as nn
from torch import nn as nn2
print(json.dumps(dict(d=d, before=before, after=after, threads=[torch.get_num_threads(), torch.get_num_interop_threads()],
                      real=type(sys.modules["torch"]).__name__, same=nn is nn2, dev=str(torch.device("cpu")), is_tensor=isinstance(x, torch.Tensor))))
""")
    assert out["d"] == ["cpu", "cuda:1", "cuda:0", "device(type='cpu')", "cpu"]
    assert out["before"] == {"installed": True, "real_loaded": False, "first_touch": None}
    assert out["after"] == {"installed": True, "real_loaded": True, "first_touch": "zeros"}
    assert out["threads"] == [3, 2] and out["real"] == "module" and out["same"] and out["dev"] == "cpu" and out["is_tensor"]
    # with the package already imported nothing is put in its place
    out = _fresh("import json, torch; from clair3_amd import lazy_torch; print(json.dumps([lazy_torch.install(), lazy_torch.status()]))")
    assert out == [False, {"installed": False, "real_loaded": True, "first_touch": None}]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "clair3")), reason="needs the reference checkout (build container only)")
def test_install_and_the_workers_own_torch_calls_do_not_import_torch(tmp_path):
    """callvar.install() on the unmodified checkout, then exactly what the worker does with torch outside the rebound functions
    (clair3/CallVariantsFromCffi.py:62-63, :203) and the rebound loader on a checkpoint file: torch is never imported; the reference's
    clair3.model is -- together with torch -- as soon as anything but the two rebound classes is asked of it, and those stay rebound"""
    import torch
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=3)
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, str(tmp_path / "m.pt"))
    code = """
import json, sys, types
sys.path.insert(0, sys.argv[1])
sys.modules.setdefault("libclair3", types.ModuleType("libclair3"))
from clair3_amd import callvar, lazy_torch, predict
names = callvar.install(decoder=True)
assert callvar.install(decoder=True) == names
import clair3.CallVariantsFromCffi as w, clair3.CallVariants as legacy
w.torch.set_num_threads(4); w.torch.set_num_interop_threads(2); legacy.torch.set_num_threads(1)
device = w.torch.device("cpu")
from clair3.model import Clair3_F, Clair3_P
class Recorder:
    def load_state_dict(self, sd): self.n = len(sd)
m = Recorder(); w._load_torch_checkpoint(m, sys.argv[2], device)
s1 = lazy_torch.status()
import clair3.model
block = clair3.model.BasicBlock                       # the reference's module after all
s2 = lazy_torch.status()
print(json.dumps(dict(names=names, s1=s1, s2=s2, n=m.n, ours=[Clair3_F.__module__, clair3.model.Clair3_F.__module__, clair3.model.Clair3_P.__module__],
                      block=block.__module__, file=clair3.model.__file__, threads=w.torch.get_num_threads())))
"""
    out = _fresh(code, REF, str(tmp_path / "m"))
    assert out["s1"] == {"installed": True, "real_loaded": False, "first_touch": None}, out
    assert out["s2"]["real_loaded"] and out["n"] == len(sd)
    assert out["ours"] == ["clair3_amd.model"] * 3 and out["block"] == "clair3.model" and out["file"].endswith("clair3/model.py")
    assert out["threads"] == 1  # the last count asked for, applied when the package arrived
    assert "clair3.model.Clair3_F" in out["names"] and "clair3.CallVariantsFromCffi._torch_predict" in out["names"]
    # C3HIP_LAZY_TORCH=0: the package is imported by install() as before
    out = _fresh("import json, sys, types; sys.path.insert(0, sys.argv[1]); sys.modules.setdefault('libclair3', types.ModuleType('libclair3'));"
                 "from clair3_amd import callvar, lazy_torch; callvar.install(); print(json.dumps(lazy_torch.status()))", REF,
                 env={"C3HIP_LAZY_TORCH": "0"})
    assert out == {"installed": False, "real_loaded": True, "first_touch": None}
