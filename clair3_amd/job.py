"""A whole sharded calling job (BASELINE.json configs[3]: "candidate windows sharded across 8 x MI355X").

The reference cuts its candidates into tensor files of at most 10 000 windows (preprocess/SelectCandidates.py:379), writes the
list of files (``--output_tensor_can_fn_list``), and splits that LIST over GPU slots -- file i goes to slot i % n_slots, each slot
writes its own VCF shard, SortVcf merges them (clair3/CallVariantsFromCffiGPU.py:138-156,163-199).  Here:

  * the file list is split into CONTIGUOUS runs of files, balanced by window count (``shard_files``), one run per rank -- or,
    when there are fewer than four files per rank, into contiguous WINDOW ranges that may cut a file (``shard_segments``) -- so
    the concatenation of the ranks' rows in rank order IS the reference's window order -- no sort step;
  * every rank drives its GPU with ``worker.predict_batches`` (memory-mapped files, the reference's batch boundaries, a ring of
    submit/wait slots);
  * the (n_r, 24|90) rows meet on rank 0 in one gather (``dist.RowExchange``): on RCCL directly (``c3_gather_rows``) on a GPU
    job -- through torch.distributed when RCCL's own rendezvous or first collective does not complete -- and through
    torch.distributed's gloo in the CPU tests.

``write_synthetic_job`` produces the stand-in for a 50x ONT genome the survey describes (no BAM is available offline): N windows of
the named shape in <= 10 000-window ``.npy`` / ``.info`` pairs plus the list file, positions numbered so that order is checkable.
"""
import os
import time

import numpy as np

from . import dist as c3dist, synthetic as syn, worker

MAX_WINDOWS_PER_FILE = 10000  # preprocess/SelectCandidates.py:379


def write_synthetic_job(directory, kind, n_windows, channels=None, per_file=MAX_WINDOWS_PER_FILE, seed=0, unique=2048):
    """Write ceil(n / per_file) tensor files + .info files + the list file; returns (list_fn, windows per file).
    Windows are ``unique`` seeded windows tiled (generation cost stays bounded for million-window jobs); the .info position of
    window g is ``chrS:<g+1>:<33 bases>`` so a consumer can verify global order."""
    os.makedirs(directory, exist_ok=True)
    channels = channels or (18 if kind == syn.PILEUP else 8)
    base = syn.make_windows(kind, min(unique, n_windows), seed=seed, channels=channels)
    names, counts = [], []
    g = 0
    ref = "ACGT" * 8 + "A"
    while g < n_windows:
        n = min(per_file, n_windows - g)
        idx = (np.arange(g, g + n) % len(base))
        name = f"tensor_{len(names):05d}"
        np.save(os.path.join(directory, name + ".npy"), base[idx])
        with open(os.path.join(directory, name + ".info"), "w") as f:
            f.write("\n".join(f"chrS:{g + i + 1}:{ref}\t30-RA 30 " for i in range(n)) + "\n")
        names.append(name)
        counts.append(n)
        g += n
    list_fn = os.path.join(directory, "tensor_can_fn_list")
    with open(list_fn, "w") as f:
        f.write("\n".join(names) + "\n")
    return list_fn, counts


def file_window_counts(list_fn):
    """windows per file of a list, read from the .npy headers only (no tensor is loaded)"""
    parent = os.path.dirname(list_fn)
    with open(list_fn) as f:
        names = [n for n in f.read().strip().split("\n") if n]
    return names, [int(np.load(os.path.join(parent, n + ".npy"), mmap_mode="r").shape[0]) for n in names]


def shard_files(counts, world):
    """Contiguous runs of files per rank, balanced by windows: rank r gets files [cut[r], cut[r + 1]).  A file is never split
    (files are the reference's unit of work); cuts are placed where the running window total crosses r / world of the job."""
    total = float(sum(counts))
    cuts, run, k = [0], 0, 1
    for i, c in enumerate(counts):
        run += c
        while k < world and run >= total * k / world - 1e-9:
            cuts.append(i + 1)
            k += 1
    while len(cuts) < world + 1:
        cuts.append(len(counts))
    cuts[-1] = len(counts)
    return cuts


def shard_segments(counts, world, min_files_per_rank=4):
    """The work of every rank as a list of (file index, first window, stop window) segments, in job order.  With plenty of
    files (>= ``min_files_per_rank`` x world) ranks own whole files (``shard_files``: the reference's unit of work, no file is
    opened twice); with fewer the job is cut into CONTIGUOUS WINDOW RANGES (``dist.shard_range`` over the job's windows, SURVEY
    8e), so two files on eight GPUs still fill eight GPUs.  Either way the ranks' rows concatenated in rank order are the
    job's windows in order."""
    n_files = len(counts)
    if n_files >= min_files_per_rank * world:
        cuts = shard_files(counts, world)
        return [[(f, 0, int(counts[f])) for f in range(cuts[r], cuts[r + 1]) if counts[f]] for r in range(world)]
    starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    out = []
    for r in range(world):
        lo, hi = c3dist.shard_range(int(starts[-1]), r, world)
        segs = []
        for f in range(n_files):
            a, b = max(lo, int(starts[f])), min(hi, int(starts[f + 1]))
            if a < b:
                segs.append((f, a - int(starts[f]), b - int(starts[f])))
        out.append(segs)
    return out


def _segment_files(list_fn, names, segments):
    """(tensor, positions, alt_infos) per segment: memory-mapped slices, the shape worker.lookahead_batches walks"""
    parent = os.path.dirname(list_fn)
    for f, lo, hi in segments:
        tensor, positions, alt_infos = worker._load_tensor_file(parent, names[f])
        yield tensor[lo:hi], positions[lo:hi], alt_infos[lo:hi]


def _rows_to_device(model, files, rows_dev, positions):
    """A rank whose rows are wanted ELSEWHERE (the gather): windows through the submit ring in groups of consecutive batches,
    rows written by the forward pass straight into ``rows_dev`` (a (n_local, W) float32 CUDA tensor) -- they never visit
    this host (c3_predict_submit_dev)."""
    from collections import deque
    group = max(1, worker.group_windows_for(model))
    width = rows_dev.shape[1]
    pending, off, i = deque(), 0, 0
    for tensor, pos, _ in files:
        positions.extend(pos)
        for g0 in range(0, len(tensor), group):
            x = np.ascontiguousarray(tensor[g0:g0 + group])
            if len(pending) == 3:
                model.wait(pending.popleft())
            pending.append(model.submit_dev(x, rows_dev.data_ptr() + off * width * 4, slot=i % 3))
            off += len(x)
            i += 1
    while pending:
        model.wait(pending.popleft())
    return off


def run_job(model, list_fn, rank=0, world=1, batch_size=1000, comm=None, consume=None, exchange=None):
    """Every rank calls this with the same list.  Returns on rank 0 a dict with the rows of the whole job in window order
    (numpy), the positions seen, and timings; on other ranks the timings only.
    ``model``: a loaded clair3_amd model (or a list of handles / any object with submit/wait, see worker.predict_batches).
    ``exchange``: a dist.RowExchange (RCCL directly, torch.distributed when that does not come up; made here when None);
    ``comm``: a dist.RcclComm to use as it is (no fallback).
    ``consume(positions, alt_infos, Y)``: a per-batch consumer on THIS rank's host (the decoder beside the job); without one,
    on a GPU job, the rank's rows stay on its device from the forward pass to the gather (``rows_path`` = "device") and cross
    PCIe once, on rank 0."""
    names, counts = file_window_counts(list_fn)
    segments = shard_segments(counts, world)
    mine = segments[rank]
    per_rank = [int(sum(hi - lo for _, lo, hi in segs)) for segs in segments]
    rows, positions = [], []

    def take(pos, alt, y):
        rows.append(y)
        positions.extend(pos)
        if consume is not None:
            consume(pos, alt, y)

    m0 = model[0] if isinstance(model, (list, tuple)) else model
    on_gpu = False
    # the rank's host side (staging copies, whatever `consume` forks) on the NUMA node of its GPU -- before the first staged copy
    # creates the library's helper threads; a stand-in model (CPU tests) has no device to ask about
    placement = c3dist.pin_to_device_numa(int(getattr(m0, "_device", 0) or 0)) if getattr(m0, "_handle", None) is not None else None
    t_setup = time.perf_counter()
    if world > 1 or comm is not None:
        import torch
    device = comm.device if comm is not None else (exchange.device if exchange is not None else int(getattr(m0, "_device", 0) or 0))
    if world > 1:
        import torch.distributed as dist
        on_gpu = dist.get_backend() == "nccl"  # then every tensor a control-plane collective touches lives on the rank's GPU
        # the exchange is made BEFORE the compute phase: whether the rows take the direct path (c3_gather_rows moves DEVICE
        # memory) decides where the forward pass leaves them.  (EVERY rank makes it here: its constructor holds collectives.)
        if comm is None and exchange is None:
            exchange = c3dist.RowExchange(rank, world, device=device)
    direct = comm is not None or (exchange is not None and exchange.mode == "rccl_direct")
    rows_on_device = on_gpu or direct  # the gather wants CUDA tensors: RCCL directly, or torch.distributed on nccl
    model_device = getattr(m0, "_device", None)
    same_device = model_device is None or int(model_device) == int(device)  # the forward pass writes the send buffer itself
    keep_on_device = (rows_on_device and consume is None and hasattr(m0, "submit_dev") and not isinstance(model, (list, tuple))
                      and same_device)
    t0 = time.perf_counter()
    # (making the exchange -- RCCL rendezvous, ncclCommInitRank, up to the gather timeout -- happens before t0: reported on its own
    # and counted in total_s, so that a job's wall time does not get better by what moved in front of the compute phase)
    exchange_setup_s = t0 - t_setup
    y_dev = None
    if keep_on_device:
        y_dev = torch.empty((per_rank[rank], int(m0.row_size)), dtype=torch.float32, device=f"cuda:{device}")
        # the block comes from torch's caching allocator and libc3hip writes it on its OWN stream: whatever torch still had
        # queued against a reused block must be done first
        torch.cuda.current_stream(device).synchronize()
        n_done = _rows_to_device(m0, _segment_files(list_fn, names, mine), y_dev, positions)
    elif not mine:
        n_done = 0
    elif isinstance(model, (list, tuple)):
        batches = ((X[lo:lo + batch_size], p[lo:lo + batch_size], a[lo:lo + batch_size])
                   for X, p, a in _segment_files(list_fn, names, mine) for lo in range(0, len(X), batch_size))
        n_done = worker.predict_batches(model, batches, take)
    else:
        pending = {}
        n_done = 0
        for X, pos, alt in worker.lookahead_batches(model, _segment_files(list_fn, names, mine), batch_size, pending, depth=2,
                                                    group_windows=worker.group_windows_for(model)):
            _, group, _, lo, hi = pending.pop(id(X))
            take(pos, alt, group.take(lo, hi))
            n_done += len(pos)
    t_compute = time.perf_counter() - t0
    assert n_done == per_rank[rank], (n_done, per_rank, rank)
    y_local = np.concatenate(rows) if rows else None
    out = {"rank": rank, "windows_local": n_done, "compute_s": t_compute, "segments_local": len(mine), "per_rank": per_rank,
           "files_local": len({f for f, _, _ in mine}), "rows_path": "device" if keep_on_device else "host",
           "exchange_setup_s": exchange_setup_s}
    if placement is not None:
        out["placement"] = placement
    if world == 1:
        if y_dev is not None:  # a one-rank job handed a communicator: the rows took the device path, one copy brings them home
            y_local = y_dev.cpu().numpy()
        out.update(rows=y_local, positions=positions, gather_s=0.0, total_s=time.perf_counter() - t_setup, gather="single")
        return out
    # the row width follows from the model; a stand-in without row_size (tests) asks the other ranks.  EVERY rank takes part,
    # each with the width it knows (0 if none): a conditional collective deadlocks the ranks that skip it
    local_width = int(getattr(m0, "row_size", 0) or (rows[0].shape[1] if rows else 0))
    w = torch.tensor([local_width], dtype=torch.int64, device=f"cuda:{device}" if on_gpu else "cpu")
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    width = int(w.item())
    if width <= 0:
        raise RuntimeError("no rank knows the row width (no windows anywhere and a model without row_size)")
    t1 = time.perf_counter()
    if y_dev is not None:
        y_t = y_dev
    else:
        if y_local is None:
            y_local = np.zeros((0, width), np.float32)
        y_t = torch.from_numpy(y_local)
        if rows_on_device:
            y_t = y_t.cuda(device)
    got = comm.gather(y_t, per_rank, dst=0) if comm is not None else exchange.gather(y_t, per_rank, dst=0)
    if y_t.is_cuda:
        torch.cuda.synchronize(device)
    y_all = got.cpu().numpy() if got is not None else None
    out.update(gather_s=time.perf_counter() - t1, total_s=time.perf_counter() - t_setup,
               gather="rccl_direct" if comm is not None else exchange.mode)
    if exchange is not None:
        out.update(exchange.report())
    if rank == 0:
        out["rows"] = y_all
    out["positions"] = positions
    return out
