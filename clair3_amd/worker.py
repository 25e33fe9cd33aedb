"""Tensor transport for one GPU worker.

The reference worker reads a list of ``<name>.npy`` / ``<name>.info`` pairs written by the tensor-extraction
stage, loads every file completely, slices it into batches and runs H2D -> forward -> D2H strictly in sequence
(clair3/CallVariantsFromCffi.py:106-133 ``tensor_generator_for_chunk`` and the loop at :302-353).  This module
keeps the file formats, the batch boundaries and the order of the rows exactly as the reference produces them,
but
  * memory-maps the ``.npy`` files instead of reading up to 235 MB per file up front,
  * keeps a ring of three ``c3_predict_submit`` / ``c3_predict_wait`` slots in flight: while batch *i* is on the GPU, batch
    *i+1* is being copied into the library's pinned staging buffer and its H2D transfer is already queued,
  * hands every ``(positions, alt_info, Y)`` batch to a caller-supplied consumer -- in the reference pipeline
    that is the unchanged ``batch_output`` (clair3/CallVariants.py:1069), run in the existing process pool.
One persistent process per GPU replaces the reference's ``free_MB // 8000`` short-lived workers per device
(clair3/CallVariantsFromCffiGPU.py:55-56,138-156).
"""
import os

import numpy as np


def read_info(path):
    """``<ctg>:<pos>:<ref seq>\\t<depth>-<alt info>`` per line -> (positions, alt_infos); the strings the reference's parsing
    yields (clair3/CallVariantsFromCffi.py:114-122: strip, split on newlines, first two tab-separated fields of every line)."""
    with open(path, "r") as f:
        text = f.read().strip()
    if not text:
        return [], []
    n_lines = text.count("\n") + 1
    cols = text.replace("\n", "\t").split("\t")
    if len(cols) == 2 * n_lines:  # two fields per line (what stage A writes): one split for the whole file ...
        positions, alt_infos = cols[0::2], cols[1::2]
        # ... unless a line with one field and a line with three cancelled out: "<ctg>:<pos>:<seq>" / "<depth>-..." tell
        if all(":" in p for p in positions) and all(a[:1].isdigit() for a in alt_infos):
            return positions, alt_infos
    positions, alt_infos = [], []
    for line in text.split("\n"):
        c = line.split("\t")
        positions.append(c[0])
        alt_infos.append(c[1])
    return positions, alt_infos


def _load_tensor_file(parent, name):
    tensor = np.load(os.path.join(parent, name + ".npy"), mmap_mode="r")
    positions, alt_infos = read_info(os.path.join(parent, name + ".info"))
    if len(tensor) != len(positions) or len(tensor) != len(alt_infos):
        raise ValueError(f"{name}: {len(tensor)} tensor rows but {len(positions)} .info rows")
    return tensor, positions, alt_infos


def iter_tensor_files(list_fn, first=0, stop=None, ahead=True):
    """Yield (tensor, positions, alt_infos) for every entry of an ``--output_tensor_can_fn_list`` file; tensors are
    memory-mapped.  Entries are relative to the list file's directory (CallVariantsFromCffi.py:112-113).
    ``first`` / ``stop`` restrict the walk to a contiguous run of entries (a rank's share of a sharded job).
    ``ahead``: the next file's .info (10 000 lines: milliseconds of Python string work, as long as the GPU needs for the
    file's pileup windows) is read on a helper thread while the caller works on the current file."""
    parent = os.path.dirname(list_fn)
    with open(list_fn, "r") as f:
        names = [n for n in f.read().strip().split("\n") if n != ""]
    names = names[first:stop]
    if not ahead or len(names) < 2:
        for name in names:
            yield _load_tensor_file(parent, name)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=1, thread_name_prefix="c3-info") as pool:
        nxt = pool.submit(_load_tensor_file, parent, names[0])
        for i in range(len(names)):
            cur = nxt.result()
            if i + 1 < len(names):
                nxt = pool.submit(_load_tensor_file, parent, names[i + 1])
            yield cur


def iter_batches(list_fn, batch_size, first=0, stop=None):
    """Same batch boundaries as the reference: batches never span files, the last batch of a file is short."""
    for tensor, positions, alt_infos in iter_tensor_files(list_fn, first, stop):
        n = len(tensor)
        for lo in range(0, n, batch_size):
            hi = min(lo + batch_size, n)
            yield tensor[lo:hi], positions[lo:hi], alt_infos[lo:hi]


HOST_SLOTS = 4  # C3_HOST_SLOTS (include/c3hip.h): submits in flight per handle


class _Group:
    """One forward pass over a run of consecutive batches of a file; its rows are waited for once, by the first batch that asks."""
    __slots__ = ("model", "ticket", "rows")

    def __init__(self, model, ticket):
        self.model, self.ticket, self.rows = model, ticket, None

    def take(self, lo, hi):
        if self.rows is None:
            self.rows = self.model.wait(self.ticket)
        return self.rows[lo:hi]

    def drain(self):
        if self.rows is None:
            try:
                self.rows = self.model.wait(self.ticket)
            except Exception:
                self.rows = ()


def group_windows_for(model):
    """How many windows the transport sends through ONE forward pass when the caller's batches are smaller: the pileup kernels
    are latency-bound at the reference's batch of 1000 (4.4 M windows/s device-resident at B=1000, 5.9 M from B=4000 on:
    profiles/r03_d_batch_sweep.txt), the full-alignment ones nearly flat (769 k at 1000, 783 k at 2000).  C3HIP_PREFETCH_GROUP
    overrides (0 / 1 = one forward pass per batch)."""
    env = os.environ.get("C3HIP_PREFETCH_GROUP")
    if env is not None:
        return max(0, int(env))
    return 4000 if getattr(model, "KIND", None) == 0 else 2000


def lookahead_batches(model, files, batch_size, pending, depth=2, group_windows=0):
    """The transport behind the reference's OWN loop.  ``call_variants_from_cffi`` pulls one batch from its generator and
    makes one blocking ``_torch_predict`` call on it (clair3/CallVariantsFromCffi.py:302-317); this generator keeps the
    loop's shape -- the same batches (never across files, the last batch of a file short), the same order -- and runs ahead
    of it: consecutive batches of a file travel in GROUPS of up to ``group_windows`` windows (one contiguous slice of the
    memory-mapped tensor, one ``model.submit``: staging copy, H2D, kernels and D2H queued), ``depth`` groups beyond the one the
    loop is reading are in flight, and ``pending[id(X)] = (model, group, X, lo, hi)`` tells the rebound ``_torch_predict``
    (predict._hip_predict) that the rows of batch ``X`` only need to be waited for and sliced.  Same rows as the blocking
    calls: a window's row does not depend on the batch it travels in.
    ``files``: an iterator of (tensor, positions, alt_infos) per tensor file (iter_tensor_files)."""
    from collections import deque
    slots = depth + 1
    if not 1 <= slots <= HOST_SLOTS:
        raise ValueError(f"depth must be in [0, {HOST_SLOTS - 1}], got {depth}")
    per_group = max(1, int(group_windows) // int(batch_size)) * int(batch_size)

    def groups():  # (tensor slice of the group, [(lo, hi, positions, alt_infos) per batch, offsets inside the group])
        for tensor, positions, alt_infos in files:
            n = len(tensor)
            for g0 in range(0, n, per_group):
                g1 = min(g0 + per_group, n)
                parts = [(lo - g0, min(lo + batch_size, g1) - g0, positions[lo:min(lo + batch_size, g1)], alt_infos[lo:min(lo + batch_size, g1)])
                         for lo in range(g0, g1, batch_size)]
                yield tensor[g0:g1], parts

    queue = deque()  # (group, Xg, parts), oldest first
    it = groups()
    n_submitted, exhausted = 0, False
    handed = []  # batches of the group being read that the loop has been given: (X, group)
    try:
        while True:
            while not exhausted and len(queue) < slots:
                try:
                    Xg, parts = next(it)
                except StopIteration:
                    exhausted = True
                    break
                Xg = np.ascontiguousarray(Xg)
                group = _Group(model, model.submit(Xg, slot=n_submitted % slots))
                n_submitted += 1
                queue.append((group, Xg, parts))
            if not queue:
                return
            group, Xg, parts = queue[0]
            for lo, hi, positions, alt_infos in parts:
                X = Xg[lo:hi]
                pending[id(X)] = (model, group, X, lo, hi)
                handed.append(X)
                yield X, positions, alt_infos
            # the loop has moved past this group: whatever it did not ask for must not stay registered, and the group's slot
            # must be free before it is used again
            for X in handed:
                pending.pop(id(X), None)
            handed = []
            group.drain()
            queue.popleft()
    finally:
        for X in handed:
            pending.pop(id(X), None)
        for group, _, _ in queue:  # abandoned or failed half way: nothing may stay in flight on the handle
            group.drain()


def predict_batches(model, batches, consume, slots=3):
    """Run ``model`` over an iterator of (X, positions, alt_infos) with a ring of ``slots`` submits in flight per handle
    (c3_predict_submit / c3_predict_wait, at most C3_HOST_SLOTS = 4) and call ``consume(positions, alt_infos, Y)`` for
    every batch, in order.  ``model`` needs ``submit(X, slot)`` / ``wait(ticket)`` (clair3_amd.model._HipModel).  Returns the
    number of windows processed.  Three slots keep the staging copy, the H2D transfer, the kernels and the D2H transfer of
    consecutive batches overlapped: 0.93 - 0.975 of the device-resident rate at the reference's batch of 1000 (bench.py
    host_inclusive.batch_1000).

    ``model`` may also be a list of handles loaded with the same weights: batch i then runs on handle i % len (own
    workspace and HIP streams each), so the kernels of consecutive batches overlap on the GPU as well."""
    from collections import deque
    if not 1 <= int(slots) <= HOST_SLOTS:
        raise ValueError(f"slots must be in [1, {HOST_SLOTS}] (C3_HOST_SLOTS of include/c3hip.h), got {slots}")
    models = list(model) if isinstance(model, (list, tuple)) else [model]
    total = 0
    pending = deque()  # (model, ticket, positions, alt_infos), oldest first
    depth = slots * len(models)
    i = 0
    for X, positions, alt_infos in batches:
        if len(pending) == depth:
            m0, t0, p0, a0 = pending.popleft()
            consume(p0, a0, m0.wait(t0))
        m = models[i % len(models)]
        slot = (i // len(models)) % slots
        ticket = m.submit(np.ascontiguousarray(X), slot=slot)  # staging copy + H2D + kernels + D2H enqueued
        pending.append((m, ticket, positions, alt_infos))
        total += len(positions)
        i += 1
    while pending:
        m0, t0, p0, a0 = pending.popleft()
        consume(p0, a0, m0.wait(t0))
    return total


def predict_file_list(model, list_fn, consume, batch_size=1000, first=0, stop=None):
    """Every window of a tensor-file list through ``model``; ``consume(positions, alt_infos, Y)`` per batch of ``batch_size`` in
    the reference's order (its GPU batch size is predictBatchSize * 5 = 1000, CallVariantsFromCffi.py:265-269; batches never
    span files).  One handle: the transport of the drop-in loop (lookahead_batches: consecutive batches of a file in one forward
    pass, two groups ahead, the next file read on a helper thread -- 4.0 - 4.8 M pileup windows/s against 3.0 M with one forward
    pass per batch).  A list of handles: one forward pass per batch, round-robin (predict_batches)."""
    if isinstance(model, (list, tuple)):
        return predict_batches(model, iter_batches(list_fn, batch_size, first, stop), consume)
    pending, total = {}, 0
    for X, positions, alt_infos in lookahead_batches(model, iter_tensor_files(list_fn, first, stop), batch_size, pending, depth=2,
                                                     group_windows=group_windows_for(model)):
        _, group, _, lo, hi = pending.pop(id(X))
        consume(positions, alt_infos, group.take(lo, hi))
        total += len(positions)
    return total
