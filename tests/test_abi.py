"""C-ABI checks that need no GPU: the library loads, exports every symbol include/c3hip.h declares, and
fails loudly (never silently falls back) when no HIP device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from clair3_amd import _lib, synthetic as syn
from tests.util import ROOT

HEADER = os.path.join(ROOT, "include", "c3hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(c3_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"libc3hip.so does not export {name}"
    assert b"gfx950" in lib_version()


def test_binary_corresponds_to_the_tree():
    """c3_version() embeds a hash of the sources it was compiled from: a stale libc3hip.so (they are git-ignored but
    travel to the GPU box) cannot be tested against newer source unnoticed"""
    from clair3_amd import build
    assert build.built_hash() == build.source_hash(), "libc3hip.so is stale: python -m clair3_amd.build"
    assert ("srchash:" + build.source_hash()).encode() in lib_version()


def lib_version():
    return _lib.lib().c3_version()


def test_no_torch_types_in_the_abi():
    """signatures use plain C types only (comments may mention PyTorch, declarations may not)"""
    code = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    assert "torch" not in code.lower() and "at::" not in code and "Tensor" not in code.replace("c3_tensor_desc", "")
    assert "#include <stddef.h>" in code and "#include <stdint.h>" in code and code.count("#include") == 2


def _has_gpu():
    try:
        return _lib.device_count() > 0
    except _lib.C3Error:
        return False


@pytest.mark.skipif(_has_gpu(), reason="only meaningful on a GPU-less host")
def test_fails_loudly_without_a_gpu():
    from clair3_amd.model import Clair3_F, Clair3_P
    m = Clair3_P(add_indel_length=False, predict=True)
    with pytest.raises(_lib.C3Error, match="no HIP device|no CPU"):
        m.load_state_dict(syn.make_state_dict(syn.PILEUP))
    with pytest.raises(_lib.C3Error):
        Clair3_F(add_indel_length=True, predict=True).to("cuda:0")
    with pytest.raises(_lib.C3Error, match="no CPU path"):
        Clair3_P(predict=True).to("cpu")
    with pytest.raises(_lib.C3Error):
        Clair3_P(predict=True)(np.zeros((1, 33, 18), np.int8))


def test_error_codes_not_aborts():
    L = _lib.lib()
    assert L.c3_model_create(7, 18, 0, 0) in (None, 0)
    assert b"kind" in L.c3_last_error()
    assert L.c3_model_create(_lib.KIND_PILEUP, 99, 0, 0) in (None, 0)
    assert b"input_channels" in L.c3_last_error()
    assert L.c3_model_destroy(None) == 0
    assert L.c3_predict(None, None, 0, 0, None) != 0
