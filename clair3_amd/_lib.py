"""ctypes binding of libc3hip.so (C ABI: include/c3hip.h).

The reference binds its native code with cffi in API mode (build.py:38-85 -> ``libclair3.lib.<fn>``); cffi is
not installed in this image, so the binding below uses ctypes against the same C ABI -- INTEGRATION.md
shows the equivalent cffi ``cdef``.  There is NO fallback: if the HIP library is missing or cannot be
loaded, importing a model fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("C3HIP_LIB", os.path.join(_HERE, "lib", "libc3hip.so"))

KIND_PILEUP = 0
KIND_FULL_ALIGNMENT = 1
DTYPE_I8, DTYPE_I32, DTYPE_F32, DTYPE_I64 = 0, 1, 2, 3

# every symbol include/c3hip.h declares (tests/test_abi.py checks the header against this list)
EXPORTS = (
    "c3_version", "c3_last_error", "c3_device_count", "c3_mem_info", "c3_device_pci_bus_id", "c3_model_create", "c3_model_set_geometry",
    "c3_model_load", "c3_model_output_size", "c3_model_row_size", "c3_model_set_decode_columns", "c3_model_window_bytes", "c3_predict", "c3_predict_submit", "c3_predict_submit_dev",
    "c3_predict_wait", "c3_comm_unique_id", "c3_comm_create", "c3_comm_destroy", "c3_gather_rows", "c3_comm_count", "c3_comm_abort", "c3_stream_wait", "c3_model_describe", "c3_model_set_sharing", "c3_predict_device", "c3_predict_device_checked", "c3_model_range_status", "c3_predict_pileup_region", "c3_outcome_maxima", "c3_decode_columns", "c3_vcf_rows", "c3_model_synchronize", "c3_model_destroy", "c3_debug_fetch",
    "c3_debug_keep_activations", "c3_profile_enable", "c3_profile_reset", "c3_profile_read",
)


class RowsConfig(C.Structure):
    """c3_rows_config (include/c3hip.h)"""
    _fields_ = [("width", C.c_int32), ("flank", C.c_int32), ("show_reference", C.c_int32), ("keep_iupac", C.c_int32),
                ("has_qs_pass", C.c_int32), ("pileup", C.c_int32), ("max_len", C.c_int32), ("infer", C.c_int32),
                ("f32_arith", C.c_int32), ("walk", C.c_int32), ("gvcf", C.c_int32), ("haploid", C.c_int32),
                ("long_indel", C.c_int32), ("long_infer", C.c_int32), ("qs_pass", C.c_double), ("phred_trans", C.c_double), ("long_prop", C.c_double),
                ("gt", (C.c_char * 8) * 4)]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("data", C.c_void_p)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double), ("mfma_flops", C.c_double), ("mfma_peak_tflops", C.c_double)]


class C3Error(RuntimeError):
    pass


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  The PyTorch-ROCm wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7, found through $ORIGIN) while libc3hip.so's RUNPATH points at /opt/rocm.  If libc3hip is
    loaded first, a later ``import torch`` brings a SECOND copy of the runtime whose HSA initialisation fails
    ("No HIP GPUs are available").  Pre-loading torch's copy (without importing torch) makes libc3hip's
    NEEDED libamdhip64.so.7 resolve to it, so both share one runtime whatever the import order.  Without
    torch installed nothing happens and /opt/rocm's runtime is used."""
    import importlib.machinery
    import sys
    from . import lazy_torch
    if "torch" in sys.modules and not lazy_torch.standing_in():
        return
    # (PathFinder, not importlib.util.find_spec: with lazy_torch's stand-in in sys.modules the latter looks at ITS __spec__; the package
    # may still be imported later in this process, and must then find the runtime it expects)
    try:
        spec = importlib.machinery.PathFinder.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libc3hip.so once; raise if it is not there (no CPU fallback by design)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise C3Error(f"{LIB_PATH} not found: build it with `python -m clair3_amd.build` "
                      "(hipcc --offload-arch=gfx950); clair3_amd has no CPU fallback")
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    L.c3_version.restype = C.c_char_p
    L.c3_last_error.restype = C.c_char_p
    L.c3_device_count.restype = C.c_int
    L.c3_mem_info.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.c3_vcf_rows.argtypes = [C.POINTER(RowsConfig), C.c_int64, C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                              C.c_void_p, C.c_void_p]
    L.c3_device_pci_bus_id.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.c3_model_create.restype = C.c_void_p
    L.c3_model_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.c3_model_set_geometry.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.c3_model_load.argtypes = [C.c_void_p, C.POINTER(TensorDesc), C.c_int]
    L.c3_model_output_size.argtypes = [C.c_void_p]
    L.c3_model_row_size.argtypes = [C.c_void_p]
    L.c3_model_set_decode_columns.argtypes = [C.c_void_p, C.c_int]
    L.c3_model_window_bytes.restype = C.c_int64
    L.c3_model_window_bytes.argtypes = [C.c_void_p, C.c_int]
    L.c3_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    L.c3_predict_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int]
    L.c3_predict_submit_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int]
    L.c3_predict_wait.argtypes = [C.c_void_p, C.c_int]
    L.c3_predict_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    L.c3_comm_unique_id.argtypes = [C.c_void_p]
    L.c3_comm_create.restype = C.c_void_p
    L.c3_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.c3_comm_destroy.argtypes = [C.c_void_p]
    L.c3_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.c_int, C.c_void_p]
    L.c3_comm_count.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.c3_comm_abort.argtypes = [C.c_void_p]
    L.c3_stream_wait.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.c3_model_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    L.c3_model_set_sharing.argtypes = [C.c_void_p, C.c_int]
    L.c3_predict_device_checked.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    L.c3_model_range_status.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.c3_predict_pileup_region.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    L.c3_outcome_maxima.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.c3_decode_columns.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.c3_model_synchronize.argtypes = [C.c_void_p]
    L.c3_model_destroy.argtypes = [C.c_void_p]
    L.c3_debug_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    L.c3_debug_keep_activations.argtypes = [C.c_void_p, C.c_int]
    L.c3_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.c3_profile_reset.argtypes = [C.c_void_p]
    L.c3_profile_read.argtypes = [C.c_void_p, C.POINTER(KernelStat), C.c_int]
    for name in EXPORTS:
        getattr(L, name)  # AttributeError here = the .so is older than the header
    _lib = L
    return L


def last_error():
    return lib().c3_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise C3Error(f"{what}: {last_error()}")


def device_count():
    n = lib().c3_device_count()
    if n < 0:
        raise C3Error(f"c3_device_count: {last_error()}")
    return n


def pci_bus_id(device=0):
    """PCI address of a visible device as sysfs spells it (c3_device_pci_bus_id)."""
    buf = C.create_string_buffer(32)
    check(lib().c3_device_pci_bus_id(int(device), buf, 32), "c3_device_pci_bus_id")
    return buf.value.decode()


def mem_info(device=0):
    """(free_bytes, total_bytes) of a device -- the ROCm replacement for the reference's nvidia-smi probe
    (clair3/CallVariantsFromCffiGPU.py:13-19)."""
    f, t = C.c_size_t(0), C.c_size_t(0)
    check(lib().c3_mem_info(device, C.byref(f), C.byref(t)), "c3_mem_info")
    return f.value, t.value
