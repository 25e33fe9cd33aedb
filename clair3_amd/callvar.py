"""Drop-in installation into the reference's own entry points.

``install()`` rebinds, inside an *unmodified* checkout of the reference, exactly the names that make up the
model call of the GPU path -- nothing else of the pipeline is touched:

  clair3.CallVariantsFromCffi   _torch_predict (:48-52), _load_torch_checkpoint (:19-28), _select_device (:31-34),
                                _limit_gpu_memory (:37-45, a CUDA-caching-allocator knob: no-op here),
                                tensor_generator_for_chunk (:106-148; GPU branch with tensor files only: same batches,
                                submitted to the GPU two ahead of the loop -- see _make_batch_generator)
  clair3.model                  Clair3_P (:58), Clair3_F (:282)  -- imported lazily by the worker at :230/:239
  clair3.CallVariantsFromCffiGPU get_gpu_memory (:13-19), check_gpu_memory (:21-43)  (nvidia-smi -> hipMemGetInfo)

After it ``python clair3.py CallVariantsFromCffi --use_gpu True --gpu_id 0 ...`` and
``python clair3.py CallVariantsFromCffiGPU ...`` run their loops unchanged (clair3/CallVariantsFromCffi.py:299-353:
batches from .npy files, shared-memory hand-off of Y to the batch_output workers, VCF rows) with the forward
pass in libc3hip.  ``sitecustomize``-style use:

    PYTHONPATH=/path/to/clair3_amd_repo python -c "import clair3_amd.callvar as c; c.install(); import clair3; ..."

or add the two lines shown in INTEGRATION.md to clair3.py.
"""
import os
import sys

from . import _lib, predict
from . import lazy_torch as _lazy
from .model import Clair3_F, Clair3_P


def _select_device_for_worker(use_gpu):
    """CallVariantsFromCffi._select_device: the worker only uses the result as an opaque handle that is passed
    back to _torch_predict / model.to(), so a string is enough; CPU requests stay with the reference."""
    if use_gpu and _lib.device_count() > 0:
        return "cuda:0"
    import torch
    return torch.device("cpu")


# ---- the decode pool of the stage-B loop, forked BEFORE the device is in use.
# call_variants_from_cffi creates its ProcessPoolExecutor after the model is loaded, and the pool forks its processes on the loop's
# first submits -- right behind the first model call.  A fork of the interpreter's process with the device at work costs the loop
# ~0.3 s: the device does not answer for a quarter of a second and the loop thread and the fresh children crawl through
# copy-on-write faults (profiles/r05_l_fork_stall.txt; tests/diag/loop_timeline.py: 240 k full-alignment windows in 0.95 - 1.03 s,
# in 0.66 s = 365 k windows/s with the pool forked before the model load).  The loop's first call into rebound code on the GPU
# branch is _select_device(True) (clair3/CallVariantsFromCffi.py:217), before any HIP call of this process: the pool is created
# there -- the same concurrent.futures.ProcessPoolExecutor, the size the command line asks for, every process forked (one short
# task each) -- and the module's name ProcessPoolExecutor hands it to the loop's `with ProcessPoolExecutor(max_workers=
# args.cpu_threads)` (:302).  Any other request (another size, other arguments, a second pool) gets an ordinary executor.
# C3HIP_PREFORK_POOL=0 leaves the loop's pool alone.
_PREFORKED = None  # (executor, max_workers) waiting for the loop to ask for it
STATS = {"prefork_mismatch": 0}  # pools forked ahead with a size the loop did not ask for
PLACEMENT = None  # what dist.pin_to_device_numa did for this worker (a dict), once _select_device has run on the GPU branch


def _cpu_threads_from_argv(default=4):
    """--cpu_threads of the worker command line (clair3/CallVariantsFromCffi.py:569, default 4).  C3HIP_DECODE_PROCS (set by a
    launcher that knows better than the command line, or by a programmatic caller of call_variants_from_cffi) wins; on the command
    line argparse's unambiguous prefixes of --cpu_threads are recognised too (--cpu_t 8, --cpu_threads=8): the worker has no
    other option that starts with --cpu."""
    env = os.environ.get("C3HIP_DECODE_PROCS", "").strip()
    if env.isdigit() and int(env) > 0:
        return int(env)
    argv = sys.argv
    for i, a in enumerate(argv):
        name, eq, value = a.partition("=")
        if not (name.startswith("--cpu") and "--cpu_threads".startswith(name)):
            continue
        try:
            if eq:
                return int(value)
            if i + 1 < len(argv):
                return int(argv[i + 1])
        except ValueError:
            return default
    return default


def _prefork_decode_pool(n):
    """A ProcessPoolExecutor of n processes, all of them forked and idle; None if that is not possible."""
    global _PREFORKED
    import atexit
    import concurrent.futures as cf
    import time
    if n < 1:
        return None
    # the loop's first SharedMemory(create=True) starts multiprocessing's resource tracker BEFORE its first fork, so its decode
    # processes talk to the parent's tracker when they unlink a segment (:181); processes forked ahead must inherit it as well
    from multiprocessing import resource_tracker
    resource_tracker.ensure_running()
    ex = cf.ProcessPoolExecutor(max_workers=n)
    try:
        # a submit forks a new process when no idle one is waiting (concurrent/futures/process.py _adjust_process_count):
        # n tasks that outlast the n forks make n processes
        nap = 0.05
        for _ in range(3):  # (on a busy host a process may finish its task before the last fork: again, with longer tasks)
            for f in [ex.submit(time.sleep, nap) for _ in range(n)]:
                f.result()
            if len(getattr(ex, "_processes", None) or range(n)) >= n:
                break
            nap *= 3
    except Exception:  # noqa: BLE001  (no pool here is not an error: the loop creates its own)
        ex.shutdown(wait=False, cancel_futures=True)
        return None
    _PREFORKED = (ex, n)
    atexit.register(_drop_preforked)  # a loop that never asks for its pool must not leave processes behind
    return ex


def _drop_preforked():
    global _PREFORKED
    pre, _PREFORKED = _PREFORKED, None
    if pre is not None:
        pre[0].shutdown(wait=False, cancel_futures=True)


def _make_pool_factory(original):
    def ProcessPoolExecutor(max_workers=None, *args, **kwargs):
        global _PREFORKED
        pre, _PREFORKED = _PREFORKED, None
        if pre is not None:
            if max_workers == pre[1] and not args and not kwargs:
                return pre[0]
            # the size guessed from the command line is not the size the loop asks for: the idle processes are dropped and the loop
            # forks its own pool behind the first model call (the ~0.3 s stall this was meant to remove) -- said once, on stderr
            print(f"[clair3_amd] the decode pool forked ahead has {pre[1]} processes, the loop asks for {max_workers}: dropped "
                  f"(set C3HIP_DECODE_PROCS={max_workers} to fork the right size ahead, C3HIP_PREFORK_POOL=0 to fork none)", file=sys.stderr)
            STATS["prefork_mismatch"] += 1
            pre[0].shutdown(wait=False, cancel_futures=True)
        return original(max_workers, *args, **kwargs)

    ProcessPoolExecutor._c3hip_original = original
    return ProcessPoolExecutor


def _select_device_for_cffi_worker(use_gpu):
    """_select_device of the stage-B worker: on the GPU branch the decode pool is forked first (see above)."""
    if use_gpu:
        # the worker on the NUMA node of ITS GPU -- the reference has just set CUDA_VISIBLE_DEVICES to --gpu_id
        # (clair3/CallVariantsFromCffi.py:216), so ordinal 0 is that device -- from sysfs alone (no HIP call yet) and BEFORE the pool is
        # forked: the decode processes and the library's staging threads inherit the placement (C3HIP_NUMA_PIN=0 = off)
        global PLACEMENT
        from . import dist as c3dist
        PLACEMENT = c3dist.pin_to_device_numa(0, use_hip=False)
    if use_gpu and _PREFORKED is None and os.environ.get("C3HIP_PREFORK_POOL", "1").strip().lower() not in ("0", "false", "no", "off"):
        _prefork_decode_pool(_cpu_threads_from_argv())
    return _select_device_for_worker(use_gpu)


def _limit_gpu_memory(memory_mb, device):
    return  # libc3hip sizes its own workspace (c3_mem_info is the accounting hook); nothing to cap


def _limit_gpu_memory_legacy(memory_mb):  # clair3/CallVariants.py:72 has no device parameter
    return


def _check_gpu_memory_or_exit(memory, device_ids=None, print_log=True):
    from shared.utils import log_error  # reference helper, only available inside the reference tree
    try:
        return predict.check_gpu_memory(memory, device_ids, print_log)
    except _lib.C3Error as e:
        print(log_error(str(e)))
        sys.exit(1)


def _make_batch_generator(original):
    """tensor_generator_for_chunk (clair3/CallVariantsFromCffi.py:106-148) for the GPU branch: the same batches in the same
    order from the same ``--output_tensor_can_fn_list`` files, but memory-mapped and submitted to the GPU ahead of the loop
    (clair3_amd/worker.py: groups of consecutive batches per forward pass, ``PREFETCH_DEPTH`` groups in flight beyond the
    one being read), so that the loop's one blocking ``_torch_predict`` per batch finds its rows computed -- the unmodified
    loop then runs at the rate of the submit / wait ring instead of H2D -> forward -> D2H in sequence.  Every other use (in-process tensors, no model loaded through the rebound loader, CPU) is the
    reference's own generator."""
    from . import worker as transport

    def tensor_generator_for_chunk(gen_cls, args, batch_size=50):
        model = predict.current_model()
        if (getattr(args, "output_tensor_can_fn_list", None) is None or not getattr(args, "use_gpu", False)
                or model is None or PREFETCH_DEPTH <= 0):
            yield from original(gen_cls, args, batch_size=batch_size)
            return
        want = bool(predict.DECODER_COLUMNS)
        if want != model._decode_cols:
            model.decode_columns(want)
        yield from transport.lookahead_batches(model, transport.iter_tensor_files(args.output_tensor_can_fn_list), batch_size,
                                               predict._PENDING, depth=PREFETCH_DEPTH,
                                               group_windows=transport.group_windows_for(model))

    tensor_generator_for_chunk._c3hip_original = original
    return tensor_generator_for_chunk


# batches the rebound generator keeps submitted ahead of the reference loop (0 = the reference's generator, one blocking
# c3_predict per batch); env C3HIP_PREFETCH_DEPTH
PREFETCH_DEPTH = int(os.environ.get("C3HIP_PREFETCH_DEPTH", "2"))


def install(worker=True, gpu_wrapper=True, decoder=False, lazy_torch=None):
    """Patch the imported (or importable) reference modules in place.  Returns the list of rebound names.
    decoder=True additionally makes the rows of either network carry the decoder columns of libc3hip and
    rebinds clair3.CallVariants.possible_outcome_probabilites_from / batch_output to read them
    (clair3_amd/decode.py): same VCF text, ~6x the decode rate per host core on rows with the indel-length heads, ~2x on
    the 24-probability rows of the pileup network.  lazy_torch (default: C3HIP_LAZY_TORCH, on): see below."""
    done = []
    # ``import torch`` is 1.2 - 1.9 s of a worker process that, with these names rebound, never uses it (lazy_torch.py): when the package
    # has not been imported yet, a stand-in takes its place that imports it on first real use, and clair3.model -- whose classes are
    # replaced anyway, and whose import needs torch.nn -- is stood in for the same way
    if lazy_torch is None:
        lazy_torch = _lazy.wanted()
    if lazy_torch and _lazy.install() and "clair3.model" not in sys.modules:
        import importlib
        import clair3

        def load_reference_models():
            sys.modules.pop("clair3.model", None)
            real = importlib.import_module("clair3.model")
            real.Clair3_P, real.Clair3_F = Clair3_P, Clair3_F
            return real
        clair3.model = sys.modules["clair3.model"] = _lazy.standin_module("clair3.model", {"Clair3_P": Clair3_P, "Clair3_F": Clair3_F},
                                                                          load_reference_models)
    else:
        import clair3.model as ref_model
        ref_model.Clair3_P, ref_model.Clair3_F = Clair3_P, Clair3_F
    done += ["clair3.model.Clair3_P", "clair3.model.Clair3_F"]
    if worker:
        import clair3.CallVariantsFromCffi as w
        w._torch_predict = predict._hip_predict
        w._load_torch_checkpoint = predict._load_torch_checkpoint
        w._select_device = _select_device_for_cffi_worker
        w._limit_gpu_memory = _limit_gpu_memory
        if not hasattr(w.tensor_generator_for_chunk, "_c3hip_original"):
            w.tensor_generator_for_chunk = _make_batch_generator(w.tensor_generator_for_chunk)
        if not hasattr(w.ProcessPoolExecutor, "_c3hip_original"):
            w.ProcessPoolExecutor = _make_pool_factory(w.ProcessPoolExecutor)
        done += ["clair3.CallVariantsFromCffi." + n for n in
                 ("_torch_predict", "_load_torch_checkpoint", "_select_device", "_limit_gpu_memory",
                  "tensor_generator_for_chunk", "ProcessPoolExecutor")]
    if worker:
        # the twins of the same four functions in the legacy stdin-pipe worker (clair3/CallVariants.py:54-87; call_variants
        # :1456 ff. imports the model classes when it runs, i.e. it gets the rebound ones): same model call, its own transport
        import clair3.CallVariants as legacy
        legacy._torch_predict = predict._hip_predict
        legacy._load_torch_checkpoint = predict._load_torch_checkpoint
        legacy._select_device = _select_device_for_worker
        legacy._limit_gpu_memory = _limit_gpu_memory_legacy
        done += ["clair3.CallVariants." + n for n in ("_torch_predict", "_load_torch_checkpoint", "_select_device", "_limit_gpu_memory")]
    if decoder:
        from . import decode
        if worker:
            import clair3.CallVariantsFromCffi  # noqa: F401  (so that its imported copy of batch_output is rebound too)
        decode.install_decoder()
        predict.DECODER_COLUMNS = True
        done += ["clair3.CallVariants.possible_outcome_probabilites_from", "clair3.CallVariants.batch_output"]
    if gpu_wrapper:
        import clair3.CallVariantsFromCffiGPU as g
        g.get_gpu_memory = predict.get_gpu_memory
        g.check_gpu_memory = _check_gpu_memory_or_exit
        done += ["clair3.CallVariantsFromCffiGPU.get_gpu_memory", "clair3.CallVariantsFromCffiGPU.check_gpu_memory"]
    return done
