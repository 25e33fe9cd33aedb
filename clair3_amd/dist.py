"""Multi-GPU sharding of candidate windows: one process per GPU, no data-path collective, one gather of the
per-window probability rows to rank 0.

The reference shards by *files* across GPU slots with GNU parallel and meets on disk in SortVcf
(clair3/CallVariantsFromCffiGPU.py:138-156,163-199; preprocess/SortVcf.py:290-362).  Windows are independent
samples (BatchNorm in eval mode, LSTM state per window), so here every rank takes a contiguous, near-equal
range of the window list -- which keeps VCF order trivially -- runs the HIP forward on its own GPU and the
(n_r, 24|90) float32 rows are gathered to rank 0 (RCCL over xGMI: backend "nccl" on ROCm; "gloo" on CPU for
the tests) for the unchanged reference decoder / MergeVcf.  Payload is 96 B (pileup) / 360 B (full alignment)
per window: even at 8 x 200 k windows/s that is < 0.6 GB/s into rank 0, three orders of magnitude below one
xGMI link, so one padded gather per super-batch is all the communication there is.
"""
import os


def shard_range(n_windows, rank, world_size):
    """[start, stop) of rank's contiguous share; the first n % world ranks get one extra window."""
    base, extra = divmod(int(n_windows), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


# ---------------------------------------------------------------------------------------------- placement of a rank's host side
# The reference starts one worker process per GPU slot with GNU parallel and leaves CPU placement to the OS
# (clair3/CallVariantsFromCffiGPU.py:138-156).  On an 8-GPU node the host side of a rank -- the staging copies into pinned memory,
# the forked decode workers that read the rows -- wants the NUMA node its GPU hangs off (SURVEY 8e names it as a scaling limiter).
# Everything here reads sysfs and the environment only, so it is testable without a GPU (tests/test_dist_cpu.py).

def visible_device_ids(env=None):
    """The physical device ids behind the ordinals 0..n-1 this process sees, or None when no variable restricts them.
    HIP honours ROCR_VISIBLE_DEVICES (runtime level) and then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (HIP level, indices into
    what ROCR left visible): the composition of the two lists is what a LOCAL_RANK indexes into."""
    env = os.environ if env is None else env

    def ids(name):
        v = env.get(name)
        if v is None or v.strip() == "":
            return None
        out = []
        for t in v.split(","):
            t = t.strip()
            if not t.lstrip("-").isdigit() or int(t) < 0:  # HIP stops at the first entry it cannot use (UUIDs are not handled here)
                break
            out.append(int(t))
        return out

    rocr = ids("ROCR_VISIBLE_DEVICES")
    hip = ids("HIP_VISIBLE_DEVICES")
    if hip is None:
        hip = ids("CUDA_VISIBLE_DEVICES")
    if rocr is None:
        return hip
    if hip is None:
        return rocr
    return [rocr[i] for i in hip if i < len(rocr)]


def device_for_local_rank(local_rank, n_visible, env=None):
    """(ordinal, physical id) of the device rank `local_rank` of this node uses: ordinal = LOCAL_RANK (one process per GPU, the
    launcher's numbering), physical = what that ordinal means under the *_VISIBLE_DEVICES permutation in force -- the id the
    reference's --gpu_id / physical device lists speak (clair3/CallVariantsFromCffiGPU.py:45-73).  Raises ValueError with one clear
    sentence when the node does not show that many devices."""
    local_rank, n_visible = int(local_rank), int(n_visible)
    if not 0 <= local_rank < n_visible:
        raise ValueError(f"LOCAL_RANK {local_rank} needs device ordinal {local_rank}, but this process sees {n_visible} HIP device(s)"
                         " (check --nproc-per-node against the node and *_VISIBLE_DEVICES)")
    vis = visible_device_ids(env)
    if vis is None:
        return local_rank, local_rank
    if local_rank >= len(vis):
        raise ValueError(f"LOCAL_RANK {local_rank} is beyond the {len(vis)} device(s) *_VISIBLE_DEVICES names")
    return local_rank, vis[local_rank]


def preflight(n_ranks_on_node, n_visible, local_rank=0):
    """None if `n_ranks_on_node` ranks fit the devices this process sees, else ONE sentence for stderr (bench.py --gpus N exits 2
    with it before any rendezvous; only local rank 0 prints)."""
    if n_visible >= n_ranks_on_node:
        return None
    return (f"{n_ranks_on_node} ranks on this node but only {n_visible} HIP device(s) visible: one process per GPU is the contract "
            f"(local rank {local_rank} stops here)")


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def numa_cpus_of_pci(pci_bus_id, sysfs="/sys"):
    """(node, cpus) of the NUMA node a PCI device hangs off: /sys/bus/pci/devices/<id>/numa_node and
    /sys/devices/system/node/node<N>/cpulist; (-1, empty set) when the platform does not say (one-socket boxes, VMs)."""
    try:
        node = int(open(os.path.join(sysfs, "bus/pci/devices", pci_bus_id, "numa_node")).read().strip())
    except (OSError, ValueError):
        return -1, set()
    if node < 0:
        return -1, set()
    try:
        cpus = _parse_cpulist(open(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")).read())
    except (OSError, ValueError):
        return node, set()
    return node, cpus


def pci_bus_id_from_kfd(physical_id, sysfs="/sys"):
    """PCI address of the physical_id-th GPU in the order the runtime enumerates them (the kfd topology's GPU nodes), read from
    sysfs alone: NO HIP call, so a worker can place itself before it forks its decode pool and before the runtime starts
    (clair3_amd/callvar.py: a fork behind HIP's initialisation stalls the device, profiles/r05_l_fork_stall.txt).  None if the
    topology is not there."""
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return None
    gpus = []
    for n in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(base, str(n), "properties")).read().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) <= 0:
                continue  # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
        except (OSError, KeyError, ValueError):
            continue
        gpus.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}")
    return gpus[physical_id] if 0 <= physical_id < len(gpus) else None


def pin_to_device_numa(device, sysfs="/sys", pci_bus_id=None, setaffinity=None, getaffinity=None, use_hip=True):
    """Restrict THIS process (and everything it starts afterwards: the library's staging threads are created on the first staged
    copy, the decode pool is forked by the loop) to the CPUs of the NUMA node of HIP device ordinal `device`.  Never widens the
    set the process already has (cgroup / taskset), never leaves it empty, C3HIP_NUMA_PIN=0 switches it off.  Returns a dict that
    says what happened (job.py / bench.py put it in their reports)."""
    info = {"device": int(device), "pinned": False}
    if os.environ.get("C3HIP_NUMA_PIN", "1").strip().lower() in ("0", "false", "no", "off"):
        info["why"] = "C3HIP_NUMA_PIN=0"
        return info
    setaffinity = setaffinity or getattr(os, "sched_setaffinity", None)
    getaffinity = getaffinity or getattr(os, "sched_getaffinity", None)
    if setaffinity is None or getaffinity is None:
        info["why"] = "no sched_setaffinity on this platform"
        return info
    why = None
    if pci_bus_id is None and use_hip:  # the runtime's own answer where the runtime may be asked (a rank of a job, bench.py) ...
        try:
            from . import _lib
            pci_bus_id = _lib.pci_bus_id(device)
            info["pci_from"] = "c3_device_pci_bus_id"
        except Exception as e:  # no device, no library: placement is an optimisation, never an error
            why = f"{e}"
    if pci_bus_id is None:  # ... the kfd topology in sysfs otherwise (no HIP call: the stage-B worker before it forks its pool)
        vis = visible_device_ids()
        physical = vis[device] if vis is not None and 0 <= int(device) < len(vis) else int(device)
        pci_bus_id = pci_bus_id_from_kfd(physical, sysfs)
        info["pci_from"] = "kfd topology"
    if pci_bus_id is None:
        info.pop("pci_from", None)
        info["why"] = "no PCI address (no kfd topology in sysfs" + (f"; the runtime: {why})" if why else ")")
        return info
    node, cpus = numa_cpus_of_pci(pci_bus_id, sysfs)
    info.update(pci=pci_bus_id, numa_node=node)
    have = set(getaffinity(0))
    want = cpus & have
    if node < 0 or not cpus:
        info["why"] = "the platform names no NUMA node for the device"
    elif not want:
        info["why"] = "none of the node's CPUs is in this process's allowed set"
    elif want == have:
        info["why"] = "already inside the node"
        info["cpus"] = len(have)
    else:
        setaffinity(0, want)
        info.update(pinned=True, cpus=len(want), of=len(have))
    return info


def init_from_env(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, world_size, local_rank); a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            if backend is None:
                # C3_DIST_BACKEND: the control plane of a job whose ranks cannot form an RCCL group of their own -- the two-ranks-on-one-GPU
                # arrangement of tests/test_comm_gpu.py (real RCCL refuses two ranks on one device); the rows then still travel on
                # c3_gather_rows (RowExchange._cuda_job reads the same variable; job.run_job leaves the rows on the device for it),
                # everything else on gloo
                backend = os.environ.get("C3_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend == "nccl":
                torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_rows(y_local, n_total, dst=0):
    """Gather the per-rank probability rows (torch tensor (n_r, W) float32, on the GPU for nccl / host for
    gloo) to rank ``dst`` in rank order.  Every rank's n_r must equal shard_range(n_total, r, world).
    Returns the (n_total, W) tensor on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return y_local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    if y_local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {y_local.shape[0]} rows, its shard has {counts[rank]}")
    width = y_local.shape[1]
    pad = max(counts)
    send = y_local
    if y_local.shape[0] != pad:  # pad to a common size: gather needs equal shapes
        send = torch.zeros((pad, width), dtype=y_local.dtype, device=y_local.device)
        send[: y_local.shape[0]] = y_local
    send = send.contiguous()
    if rank == dst:
        parts = [torch.empty((pad, width), dtype=y_local.dtype, device=y_local.device) for _ in range(world)]
        dist.gather(send, parts, dst=dst)
        return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    dist.gather(send, None, dst=dst)
    return None


class RowGatherer:
    """Collect the per-step rows of this rank and send them to ``dst`` ``every`` steps at a time -- one collective per
    super-batch instead of one per step.  ``add(y)`` returns what ``gather_rows`` returns when a group is
    complete (the rows of the group's steps, rank-major: all steps of rank 0, then all steps of rank 1, ...; None on the
    other ranks) and None otherwise; ``flush()`` sends an incomplete group.  Every step must bring the same number of
    rows per rank (``rows_per_step_total`` = that number x world size)."""

    def __init__(self, rows_per_step_total, every=8, dst=0, exchange=None):
        self.n_total, self.every, self.dst, self.pending = int(rows_per_step_total), int(every), dst, []
        self.exchange = exchange  # a RowExchange: RCCL directly, torch.distributed if that does not come up; None = gather_rows

    def add(self, y):
        self.pending.append(y)
        return self.flush() if len(self.pending) >= self.every else None

    def flush(self):
        if not self.pending:
            return None
        import torch
        rows = torch.cat(self.pending) if len(self.pending) > 1 else self.pending[0]
        k = len(self.pending)
        self.pending = []
        if self.exchange is not None and self.exchange.world > 1:
            counts = [shard_range(self.n_total * k, r, self.exchange.world) for r in range(self.exchange.world)]
            return self.exchange.gather(rows, [b - a for a, b in counts], dst=self.dst)
        return gather_rows(rows, self.n_total * k, dst=self.dst)


def predict_sharded(model, x_all, dst=0):
    """Whole-job helper: every rank passes the same host window array (or its own memmap of it), computes its
    contiguous shard on its GPU and rank ``dst`` receives all rows in window order (numpy), others None."""
    import numpy as np
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    n = len(x_all)
    lo, hi = shard_range(n, rank, world)
    x = np.ascontiguousarray(x_all[lo:hi])
    if world == 1:
        return model.predict_numpy(x)
    xd = torch.from_numpy(x).cuda()
    yd = model.forward(xd, checked=True)  # the range guard of the host path, on the device-resident entry
    out = gather_rows(yd, n, dst)
    return out.cpu().numpy() if out is not None else None


def gather_counts(y_local, counts, dst=0):
    """gather_rows for arbitrary per-rank row counts (a file-sharded job: ranks own whole files, so their window counts
    are whatever the files hold).  counts[r] = rows of rank r, identical on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return y_local
    world, rank = dist.get_world_size(), dist.get_rank()
    if y_local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {y_local.shape[0]} rows, expected {counts[rank]}")
    home = y_local.device
    if y_local.is_cuda and dist.get_backend() != "nccl":  # device rows under a host control plane (gloo): the fallback goes through the host
        y_local = y_local.cpu()
        out = gather_counts(y_local, counts, dst=dst)
        return out.to(home) if out is not None else None
    width, pad = y_local.shape[1], max(max(counts), 1)
    send = torch.zeros((pad, width), dtype=y_local.dtype, device=y_local.device)
    send[: y_local.shape[0]] = y_local
    if rank == dst:
        parts = [torch.empty((pad, width), dtype=y_local.dtype, device=y_local.device) for _ in range(world)]
        dist.gather(send, parts, dst=dst)
        return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    dist.gather(send, None, dst=dst)
    return None


class RcclComm:
    """The gather on RCCL itself (c3_gather_rows: grouped ncclSend / ncclRecv on the caller's HIP stream) -- the data
    path of a sharded job touches no framework.  torch.distributed (any backend, gloo is enough) is used ONCE, as the
    control plane that carries rank 0's 128-byte unique id to the other ranks; a launcher with another store can pass
    ``unique_id`` itself.  world == 1 needs neither RCCL nor a rendezvous.

    ``create_timeout_s``: ncclCommInitRank blocks until every rank has joined; it runs on a helper thread and a rendezvous
    that does not complete in time raises ``TimeoutError`` instead of hanging the job (RowExchange then gathers through
    torch.distributed)."""

    def __init__(self, rank=0, world=1, device=0, unique_id=None, create_timeout_s=None):
        import ctypes as C
        from . import _lib
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self._h = None
        idbuf = (C.c_char * 128)()
        if self.world > 1:
            if unique_id is None:
                import torch
                import torch.distributed as dist
                # rank 0 ALWAYS takes part in the broadcast: if it cannot make an id (no librccl, ncclGetUniqueId failed) it sends
                # 128 zero bytes, which every rank -- itself included -- reads as "no direct path"; raising before the broadcast
                # would leave the other ranks inside it while this one moves on to the next collective
                t = torch.zeros(128, dtype=torch.uint8)
                id_error = None
                if self.rank == 0:
                    try:
                        _lib.check(_lib.lib().c3_comm_unique_id(idbuf), "c3_comm_unique_id")
                        t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).clone()
                    except Exception as e:
                        id_error = e
                if dist.get_backend() == "nccl":
                    t = t.cuda(self.device)
                dist.broadcast(t, src=0)
                unique_id = bytes(t.cpu().numpy().tobytes())
                if not any(unique_id):
                    raise _lib.C3Error(f"rank 0 could not create an RCCL unique id{': ' + repr(id_error) if id_error else ''}")
            idbuf.raw = unique_id
        elif unique_id is not None:  # a world of one normally needs no id; under C3HIP_FORCE_RCCL=1 c3_comm_create uses the caller's if given
            idbuf.raw = unique_id
        import threading
        box, lock = {}, threading.Lock()

        def create():
            h = _lib.lib().c3_comm_create(idbuf, self.rank, self.world, self.device)
            err = _lib.last_error() if not h else ""  # thread-local: read it on the thread that failed
            with lock:
                if box.get("abandoned"):  # the caller gave up waiting: nobody will ever use or close this communicator
                    if h:
                        _lib.lib().c3_comm_destroy(C.c_void_p(h))
                    return
                box["h"], box["err"] = h, err

        if create_timeout_s is None or self.world == 1:
            create()
        else:
            th = threading.Thread(target=create, daemon=True)
            th.start()
            th.join(create_timeout_s)
            with lock:
                if "h" not in box:
                    box["abandoned"] = True
            if box.get("abandoned"):
                raise TimeoutError(f"ncclCommInitRank did not return within {create_timeout_s} s")
        if not box.get("h"):
            raise _lib.C3Error(f"c3_comm_create: {box.get('err')}")
        self._h = C.c_void_p(box["h"])

    def ranks_seen(self):
        """(ranks, this rank) as RCCL reports them (ncclCommCount / ncclCommUserRank)"""
        import ctypes as C
        from . import _lib
        n, r = C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().c3_comm_count(self._h, C.byref(n), C.byref(r)), "c3_comm_count")
        return n.value, r.value

    def gather(self, y_dev, counts, dst=0, stream=None, timeout_s=None):
        """y_dev: this rank's (counts[rank], W) float32 CUDA tensor.  Returns the (sum(counts), W) tensor on ``dst`` (rows in
        rank order), None elsewhere.  Asynchronous on the current stream, like any other kernel -- unless ``timeout_s`` is
        given: then the call waits for the stream (c3_stream_wait, polling hipStreamQuery) and raises ``TimeoutError`` after
        aborting the communicator (ncclCommAbort) if the collective has not finished by then."""
        import ctypes as C
        import torch
        from . import _lib
        if y_dev.shape[0] != counts[self.rank] or y_dev.dtype != torch.float32 or not y_dev.is_cuda:
            raise ValueError("rows must be a float32 CUDA tensor holding this rank's count")
        y_dev = y_dev.contiguous()
        width = int(y_dev.shape[1])
        out = torch.empty((int(sum(counts)), width), dtype=torch.float32, device=y_dev.device) if self.rank == dst else None
        cnt = (C.c_int64 * self.world)(*[int(c) for c in counts])
        s = torch.cuda.current_stream(y_dev.device).cuda_stream if stream is None else stream
        _lib.check(_lib.lib().c3_gather_rows(self._h, y_dev.data_ptr(), width, cnt, out.data_ptr() if out is not None else None,
                                             dst, C.c_void_p(s)), "c3_gather_rows")
        if timeout_s is not None:
            rc = _lib.lib().c3_stream_wait(C.c_void_p(s), self.device, int(timeout_s * 1000))
            if rc == 1:
                _lib.lib().c3_comm_abort(self._h)
                raise TimeoutError(f"c3_gather_rows did not finish within {timeout_s} s (communicator aborted)")
            _lib.check(rc, "c3_stream_wait")
        return out

    def close(self):
        from . import _lib
        if getattr(self, "_h", None):
            _lib.lib().c3_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RowExchange:
    """The gather of a sharded job with a way out: rows travel on RCCL directly (RcclComm) as long as that works, and through
    torch.distributed (``gather_counts``: backend "nccl" = RCCL under PyTorch on GPUs, "gloo" on CPU) from the moment it
    does not -- a rendezvous or a first collective that does not complete within ``timeout_s`` must cost a job one timeout,
    not its life.  The FIRST gather is the guarded one (waits for the stream, then all ranks agree through one small
    all-reduce whether everybody got through); later gathers stay asynchronous.  ``mode`` says which path carries the rows:
    "rccl_direct", "torch_fallback" or "single" (world of one)."""

    def __init__(self, rank, world, device=0, timeout_s=20.0, direct=True):
        self.rank, self.world, self.device, self.timeout_s = int(rank), int(world), int(device), float(timeout_s)
        self.mode, self.comm, self.ranks_seen, self.fallback_reason, self._proven = "single", None, 1, None, False
        if self.world == 1:
            return
        self.mode = "torch_fallback"
        if direct and self._cuda_job():
            ok, why = True, None
            try:
                self.comm = RcclComm(rank, world, device, create_timeout_s=self.timeout_s)
                self.ranks_seen = self.comm.ranks_seen()[0]
                ok = self.ranks_seen == self.world
                why = None if ok else f"RCCL sees {self.ranks_seen} ranks, the job has {self.world}"
            except Exception as e:  # no librccl, rendezvous timed out, ...
                ok, why = False, repr(e)
            if self._all_ok(ok):
                self.mode = "rccl_direct"
            else:
                self.fallback_reason = why or "another rank could not create its communicator"
                self.comm = None
        elif direct:
            self.fallback_reason = "rows are not on GPUs"

    def _cuda_job(self):
        """rows live on GPUs (the direct path moves device memory): a GPU job under torch.distributed whose control plane is
        nccl -- or any other backend the launcher asked for BY NAME (C3_DIST_BACKEND: gloo as the control plane of a GPU job;
        the id broadcast and the agreement all-reduce then use host tensors).  An ordinary gloo job on a host that happens to
        have GPUs (the CPU tests) is not one: it never touches RCCL."""
        import torch
        import torch.distributed as dist
        if not (torch.cuda.is_available() and dist.is_initialized()):
            return False
        return dist.get_backend() == "nccl" or bool(os.environ.get("C3_DIST_BACKEND"))

    def _all_ok(self, ok):
        """every rank learns whether EVERY rank succeeded (one 1-element all-reduce on the control plane)"""
        import torch
        import torch.distributed as dist
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if dist.get_backend() == "nccl":
            t = t.cuda(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def gather(self, y, counts, dst=0):
        """y: this rank's (counts[rank], W) float32 rows (CUDA tensor on a GPU job, host tensor under gloo).  Returns all rows
        in rank order on ``dst``, None elsewhere."""
        if self.world == 1:
            return y
        if self.mode == "rccl_direct":
            guarded = not self._proven
            ok, out, why = True, None, None
            try:
                out = self.comm.gather(y, counts, dst=dst, timeout_s=self.timeout_s if guarded else None)
            except Exception as e:
                ok, why = False, repr(e)
            if guarded:
                if self._all_ok(ok):
                    self._proven = True
                    return out
                self.mode, self.fallback_reason, self.comm = "torch_fallback", why or "another rank's first gather failed", None
            elif ok:
                return out
            else:
                raise RuntimeError(f"RCCL gather failed after it had worked: {why}")
        return gather_counts(y, counts, dst=dst)

    def report(self):
        return {"gather": self.mode, "rccl_ranks_seen": self.ranks_seen, "fallback_reason": self.fallback_reason}
