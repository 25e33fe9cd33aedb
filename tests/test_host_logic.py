"""Host-side logic that needs no GPU: state_dict specs, synthetic recipes, sharding arithmetic."""
import os
import sys

import numpy as np
import pytest

from clair3_amd import synthetic as syn
from clair3_amd.dist import shard_range


def test_parameter_counts_match_the_reference():
    # Clair3_P 2,074,520 parameters; Clair3_F 2,986,522 (C=8) / 2,987,098 (C=9) + 2,688 BN buffers
    n = sum(int(np.prod(s)) for _, s in syn.state_dict_spec(syn.PILEUP))
    assert n == 2074520
    for c, want in ((8, 2986522), (9, 2987098)):
        spec = syn.state_dict_spec(syn.FULL_ALIGNMENT, c, True)
        params = sum(int(np.prod(s)) for k, s in spec if "running" not in k)
        bufs = sum(int(np.prod(s)) for k, s in spec if "running" in k)
        assert params == want and bufs == 2688


def test_window_recipes():
    x = syn.make_pileup_windows(64, seed=3)
    assert x.shape == (64, 33, 18) and x.dtype == np.int8
    # no all-zero column (preprocess/CreateTensorPileupFromCffi.py:365 drops them)
    assert (np.abs(x.astype(np.int32)).sum(axis=2) > 0).all()
    f = syn.make_fa_windows(4, seed=1, channels=9)
    assert f.shape == (4, 89, 33, 9) and f.dtype == np.int8
    assert set(np.unique(f[..., 0])) <= {0, 25, 50, 75, 100}
    assert ((f[..., 8] != 0) <= (f[..., 0] != 0)).all()  # dwell only where there is a base
    u = syn.make_fa_windows(2, seed=1, recipe="uniform")
    assert u.min() >= -100 and u.max() <= 100


@pytest.mark.parametrize("n,world", [(0, 1), (1, 8), (7, 8), (8, 8), (1000, 3), (1024, 8), (10007, 8)])
def test_shard_ranges_partition_the_windows(n, world):
    spans = [shard_range(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def test_install_rebinds_the_reference_call_sites():
    """callvar.install() against the real reference checkout (build container only: the GPU box has no
    /root/reference, where this test skips).  Checks that exactly the model-call names are rebound and that the
    worker's module-level helpers keep their signatures."""
    import inspect
    import os
    import sys
    ref = os.environ.get("CLAIR3_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "clair3")):
        pytest.skip("reference checkout not present")
    for k in [k for k in sys.modules if k == "clair3" or k.startswith("clair3.") or k.startswith("shared")]:
        del sys.modules[k]  # whatever an earlier test left rebound
    sys.path.insert(0, ref)
    try:
        import clair3.CallVariantsFromCffi as w
        import clair3.CallVariantsFromCffiGPU as g
        orig = {n: inspect.signature(getattr(w, n)) for n in ("_torch_predict", "_load_torch_checkpoint", "_select_device",
                                                              "_limit_gpu_memory", "tensor_generator_for_chunk")}
        orig_g = {n: inspect.signature(getattr(g, n)) for n in ("get_gpu_memory", "check_gpu_memory")}
        import clair3.CallVariants as legacy
        orig_l = {n: inspect.signature(getattr(legacy, n)) for n in ("_torch_predict", "_load_torch_checkpoint", "_select_device",
                                                                     "_limit_gpu_memory")}
        from clair3_amd import callvar, predict
        from clair3_amd.model import Clair3_F, Clair3_P
        names = callvar.install()
        assert len(names) == 14 and "clair3.CallVariantsFromCffi.ProcessPoolExecutor" in names
        assert w.tensor_generator_for_chunk._c3hip_original.__module__ == "clair3.CallVariantsFromCffi"
        assert callvar.install() == names  # idempotent: the generator is wrapped once
        assert not hasattr(w.tensor_generator_for_chunk._c3hip_original, "_c3hip_original")
        import clair3.model as ref_model
        assert ref_model.Clair3_P is Clair3_P and ref_model.Clair3_F is Clair3_F
        assert w._torch_predict is predict._hip_predict
        for n, sig in orig.items():
            assert list(inspect.signature(getattr(w, n)).parameters) == list(sig.parameters), n
        for n, sig in orig_g.items():
            assert list(inspect.signature(getattr(g, n)).parameters) == list(sig.parameters), n
        for n, sig in orig_l.items():  # the legacy stdin-pipe worker's twins (clair3/CallVariants.py:54-87)
            assert list(inspect.signature(getattr(legacy, n)).parameters) == list(sig.parameters), n
        assert legacy._torch_predict is predict._hip_predict
        # constructor keywords used at CallVariantsFromCffi.py:232,243
        m = ref_model.Clair3_F(add_indel_length=True, predict=True, input_channels=9)
        assert m.output_size == 90 and m.input_channels == 9
        # decoder=True: two more names, same parameter lists, and the worker's imported copy follows
        import clair3.CallVariants as cv
        sig = {n: list(inspect.signature(getattr(cv, n)).parameters) for n in ("batch_output", "possible_outcome_probabilites_from")}
        unpatched = cv.batch_output
        assert not predict.DECODER_COLUMNS
        names = callvar.install(decoder=True)
        assert len(names) == 16 and predict.DECODER_COLUMNS
        assert cv.batch_output is not unpatched and w.batch_output is cv.batch_output
        for n, params in sig.items():
            assert list(inspect.signature(getattr(cv, n)).parameters) == params, n
    finally:
        from clair3_amd import predict as _p
        _p.DECODER_COLUMNS = False
        sys.path.remove(ref)
        for k in [k for k in sys.modules if k == "clair3" or k.startswith("clair3.") or k.startswith("shared")]:
            del sys.modules[k]


def test_gpu_slots_on_a_288_gb_device(monkeypatch, capsys):
    """the GPU wrapper's slot rule (clair3/CallVariantsFromCffiGPU.py:21-43,55-56) on MI355X-sized memory.
    free_MB // 8000 would start 36 full-alignment (57 pileup) workers per device; one libc3hip worker fills the chip."""
    from clair3_amd import _lib, predict
    free = {0: 287_000 * 2**20, 1: 150_000 * 2**20, 2: 3_000 * 2**20}
    monkeypatch.setattr(_lib, "device_count", lambda: 3)
    monkeypatch.setattr(_lib, "mem_info", lambda d=0: (free[d], 288 * 2**30))
    monkeypatch.delenv("C3HIP_SLOTS_PER_GPU", raising=False)
    assert predict.get_gpu_memory() == [287_000, 150_000, 3_000]
    assert predict.get_gpu_memory(gpu_id=1) == [150_000]
    # one slot per device that has the memory at all; the 3 GB device gets none (the reference's rule still applies as a cap)
    assert predict.check_gpu_memory(8000, None, print_log=False) == [0, 1]
    assert predict.check_gpu_memory(5000, None, print_log=True) == [0, 1]
    out = capsys.readouterr().out
    assert "GPU 0 free memory: 287000 MB, assigning 5000 MB per thread, 1 threads available" in out
    assert "would start 57" in out
    # --device=cuda:4,5,6 -> CUDA_VISIBLE_DEVICES=4,5,6 (:60-65): the slot carries the PHYSICAL id its worker will export
    assert predict.check_gpu_memory(8000, [4, 5, 6], print_log=False) == [4, 5]
    monkeypatch.setenv("C3HIP_SLOTS_PER_GPU", "3")
    assert predict.check_gpu_memory(8000, None, print_log=False) == [0, 0, 0, 1, 1, 1]
    monkeypatch.setenv("C3HIP_SLOTS_PER_GPU", "100")
    assert predict.check_gpu_memory(8000, None, print_log=False) == [0] * 35 + [1] * 18
    # nothing usable: the reference's messages (the installed wrapper turns them back into sys.exit(1))
    free[0] = free[1] = 100 * 2**20
    with pytest.raises(_lib.C3Error, match="No memory in GPU"):
        predict.check_gpu_memory(8000, None, print_log=False)
    with pytest.raises(_lib.C3Error, match="No GPU available"):
        predict.check_gpu_memory(8000, [], print_log=False)
    monkeypatch.setattr(_lib, "device_count", lambda: 0)
    assert predict.check_gpu_memory(8000, None, print_log=False) is None


def _forked_pid():  # (module level: the pool pickles its task by name)
    return os.getpid()


def test_the_decode_pool_is_forked_before_the_device_is_in_use(monkeypatch):
    """callvar: on the GPU branch _select_device (the loop's first call into rebound code, before any HIP call) creates the
    ProcessPoolExecutor the loop is going to ask for -- every process forked -- and the module's name ProcessPoolExecutor hands
    exactly that executor to `ProcessPoolExecutor(max_workers=args.cpu_threads)`; any other request gets an ordinary executor
    (profiles/r05_l_fork_stall.txt: a fork of the interpreter's process with the device at work costs the loop ~0.3 s)."""
    import concurrent.futures as cf
    from clair3_amd import callvar
    monkeypatch.setattr(sys, "argv", ["clair3.py", "CallVariantsFromCffi", "--cpu_threads", "3", "--use_gpu", "True"])
    assert callvar._cpu_threads_from_argv() == 3
    monkeypatch.setattr(sys, "argv", ["clair3.py", "CallVariantsFromCffi", "--cpu_threads=2"])
    assert callvar._cpu_threads_from_argv() == 2
    monkeypatch.setattr(sys, "argv", ["clair3.py", "CallVariantsFromCffi"])
    assert callvar._cpu_threads_from_argv() == 4  # the reference's default (clair3/CallVariantsFromCffi.py:569)
    monkeypatch.setattr(sys, "argv", ["clair3.py", "CallVariantsFromCffi", "--cpu_t", "6", "--cpu", "x"])  # argparse's unambiguous prefixes
    assert callvar._cpu_threads_from_argv() == 6
    monkeypatch.setattr(sys, "argv", ["clair3.py", "CallVariantsFromCffi", "--cpu_thr=5"])
    assert callvar._cpu_threads_from_argv() == 5
    monkeypatch.setenv("C3HIP_DECODE_PROCS", "7")  # a launcher / a programmatic caller says it outright
    assert callvar._cpu_threads_from_argv() == 7
    monkeypatch.delenv("C3HIP_DECODE_PROCS")
    factory = callvar._make_pool_factory(cf.ProcessPoolExecutor)
    monkeypatch.setattr(callvar, "_lib", type("L", (), {"device_count": staticmethod(lambda: 1)}))
    monkeypatch.setattr(sys, "argv", ["clair3.py", "CallVariantsFromCffi", "--cpu_threads", "2"])
    callvar._drop_preforked()
    try:
        monkeypatch.setenv("C3HIP_PREFORK_POOL", "0")
        assert callvar._select_device_for_cffi_worker(True) == "cuda:0" and callvar._PREFORKED is None
        monkeypatch.setenv("C3HIP_PREFORK_POOL", "1")
        assert str(callvar._select_device_for_cffi_worker(False)) == "cpu" and callvar._PREFORKED is None  # the CPU branch has no pool
        assert callvar._select_device_for_cffi_worker(True) == "cuda:0"
        ex, n = callvar._PREFORKED
        assert n == 2 and len(ex._processes) == 2  # both processes exist already
        before = set(ex._processes)
        with factory(max_workers=2) as got:  # what the loop does (:302)
            assert got is ex and callvar._PREFORKED is None
            pids = {got.submit(_forked_pid).result() for _ in range(8)}
            assert pids <= before and os.getpid() not in pids  # no new fork: the tasks ran in the processes forked ahead
        with factory(max_workers=2) as other:  # a second pool is an ordinary one
            assert other is not ex and other.submit(_forked_pid).result() != os.getpid()
        # a request of another size: the pool forked ahead is dropped, the caller gets what it asked for
        assert callvar._select_device_for_cffi_worker(True) == "cuda:0"
        ex2 = callvar._PREFORKED[0]
        misses = callvar.STATS["prefork_mismatch"]
        with factory(max_workers=1) as small:
            assert small is not ex2 and callvar._PREFORKED is None
            assert small.submit(_forked_pid).result() != os.getpid()
        assert callvar.STATS["prefork_mismatch"] == misses + 1  # (and one line on stderr says which size to ask for)
    finally:
        callvar._drop_preforked()
