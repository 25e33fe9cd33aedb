"""c3_gather_rows at world = 2 (grouped ncclSend / ncclRecv of clair3_amd/csrc/c3_comm.h) before it ever meets 8 GPUs: two
processes on the ONE MI355X of a lease, the library bound to tests/stubs/fake_rccl.cpp through C3HIP_RCCL_LIB (real RCCL refuses
two ranks on one device).  Everything above the wire is the product's: RcclComm (id broadcast, communicator creation on a helper
thread, ncclCommCount), job.run_job with the rows left on the device until the gather, RowExchange's guarded first gather."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from clair3_amd import job, synthetic as syn
from tests.util import ROOT

pytestmark = pytest.mark.gpu

STUB_SRC = os.path.join(ROOT, "tests", "stubs", "fake_rccl.cpp")
STUB = os.path.join(ROOT, "tests", "stubs", "libfake_rccl.so")

RANK = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    sys.path.insert(0, {root!r})
    from clair3_amd import dist as c3dist, job, synthetic as syn
    from clair3_amd.model import Clair3_F
    rank, world, _ = c3dist.init_from_env(backend="gloo")  # control plane only; both ranks drive device 0
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=4)
    m = Clair3_F(add_indel_length=True, predict=True, input_channels=8).to("cuda:0")
    m.load_state_dict(sd)
    comm = c3dist.RcclComm(rank, world, 0, create_timeout_s=60)
    assert comm.ranks_seen() == (world, rank)
    res = job.run_job(m, {lst!r}, rank=rank, world=world, batch_size=100, comm=comm)
    assert res["rows_path"] == "device" and res["gather"] == "rccl_direct"
    # the guarded exchange on the same wire: first gather waited for and agreed on, later ones asynchronous
    c3dist.RowExchange._cuda_job = lambda self: True
    ex = c3dist.RowExchange(rank, world, device=0, timeout_s=60.0)
    counts = [5, 3]
    for it in range(3):
        y = torch.full((counts[rank], 90), 10.0 * it + rank, device="cuda:0")
        out = ex.gather(y, counts, dst=0)
        torch.cuda.synchronize()
        if rank == 0:
            assert out.shape == (8, 90) and bool((out[:5] == 10.0 * it).all()) and bool((out[5:] == 10.0 * it + 1).all())
        else:
            assert out is None
    rep = ex.report()
    assert rep["gather"] == "rccl_direct" and rep["rccl_ranks_seen"] == world, rep
    # gather to a rank other than 0, and an empty contribution
    y = torch.full((counts[rank] if rank else 0, 24), float(rank), device="cuda:0")
    out = comm.gather(y, [0, 3], dst=1, timeout_s=60.0)
    if rank == 1:
        assert out.shape == (3, 24) and bool((out == 1.0).all())
    if rank == 0:
        np.save({out!r}, res["rows"])
        open({out!r} + ".txt", "w").write(repr((res["per_rank"], res["segments_local"])))
    comm.close()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stub():
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(STUB_SRC):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", STUB_SRC, "-o", STUB])
    return STUB


def test_two_ranks_gather_rows_over_the_send_recv_path(tmp_path):
    from clair3_amd.model import Clair3_F
    n = 531
    lst, counts = job.write_synthetic_job(str(tmp_path / "job"), syn.FULL_ALIGNMENT, n, channels=8, per_file=300, unique=n, seed=12)
    out = str(tmp_path / "rows.npy")
    script = tmp_path / "rank.py"
    script.write_text(RANK.format(root=ROOT, lst=lst, out=out))
    env = dict(os.environ, C3HIP_RCCL_LIB=_stub(), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    y = np.load(out)
    per_rank, _ = eval(open(out + ".txt").read())
    assert per_rank == [266, 265]  # two files on two ranks: contiguous window ranges
    # the same job in this process on one rank: sharding and the wire change no bit and no order
    sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=4)
    m = Clair3_F(add_indel_length=True, predict=True, input_channels=8).to("cuda:0")
    m.load_state_dict(sd)
    single = job.run_job(m, lst, batch_size=100)
    assert y.shape == (n, 90) and np.array_equal(y, single["rows"])


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """`python bench.py --gpus 2 ...` typed PLAINLY (no torchrun around it: bench.py starts its own two ranks)
    executes end to end with two ranks on the lease's ONE device: gloo as the control plane (C3_DIST_BACKEND), both ranks on device 0 (C3_BENCH_DEVICE), the
    rows of every 8 steps gathered to rank 0 on c3_gather_rows through the stand-in for librccl.  Not a measurement (two ranks
    share one GPU): what is checked is that the N > 1 path runs -- barriers, the MAX over ranks, the guarded first gather, the
    per-rank rates -- and prints the one parseable line with the gather on the direct path."""
    import json
    env = dict(os.environ, C3HIP_RCCL_LIB=_stub(), C3_DIST_BACKEND="gloo", C3_BENCH_DEVICE="0", OMP_NUM_THREADS="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", C3_BENCH_FULL=str(tmp_path / "full.json"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "2", "--repeats", "2",
           "--workload", "full_alignment"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert len(lines[0]) <= 4096
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 9
    assert line["config"]["windows_per_step"] == 512 and line["value"] > 0
    assert abs(line["value"] - 512 / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3  # whole-job windows / MAX-over-ranks time
    mg = line["multi_gpu"]
    assert mg["gather"] == "rccl_direct" and mg["rccl_ranks_seen"] == 2 and mg["ranks"] == 2, mg
    assert len(mg["per_rank_windows_per_s"]) == 2 and all(v > 0 for v in mg["per_rank_windows_per_s"])
    assert "roofline" in line and "cpu_baseline" not in line  # the CPU baseline is an N = 1 leg


REAL_RCCL = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    WITH_TORCH = {with_torch!r}
    if WITH_TORCH:
        import torch
    from clair3_amd import _lib, dist as c3dist
    L = _lib.lib()
    assert os.environ.get("C3HIP_FORCE_RCCL") == "1" and not os.environ.get("C3HIP_RCCL_LIB")
    # 1. the 128-byte id of the REAL library
    idbuf = (C.c_char * 128)()
    _lib.check(L.c3_comm_unique_id(idbuf), "c3_comm_unique_id")
    assert any(idbuf.raw)
    mapped = sorted({{ln.split()[-1] for ln in open("/proc/self/maps") if "rccl" in ln}})
    assert mapped and not any("fake_rccl" in m for m in mapped), mapped
    # 2. a one-rank communicator on it: ncclCommInitRank(nranks = 1), ncclCommCount / ncclCommUserRank
    comm = c3dist.RcclComm(0, 1, 0, unique_id=idbuf.raw)
    assert comm.ranks_seen() == (1, 0)
    rows = np.random.default_rng(5).random((777, 90), dtype=np.float32)
    if WITH_TORCH:
        # 3. the gather = a grouped self ncclSend / ncclRecv on torch's current stream
        y = torch.from_numpy(rows).cuda(0)
        out = comm.gather(y, [777], dst=0, timeout_s=120.0)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert out.data_ptr() != y.data_ptr()
    else:
        hip = C.CDLL("libamdhip64.so.7")  # already in the process (libc3hip brought it): device buffers without PyTorch
        hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        a, b = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(a), rows.nbytes) == 0 and hip.hipMalloc(C.byref(b), rows.nbytes) == 0
        assert hip.hipMemcpy(a, rows.ctypes.data, rows.nbytes, 1) == 0
        cnt = (C.c_int64 * 1)(777)
        _lib.check(L.c3_gather_rows(comm._h, a, 90, cnt, b, 0, None), "c3_gather_rows")
        rc = L.c3_stream_wait(None, 0, 120000)
        assert rc == 0, _lib.last_error()
        got = np.empty_like(rows)
        assert hip.hipMemcpy(got.ctypes.data, b, rows.nbytes, 2) == 0
    assert np.array_equal(got, rows)  # bit for bit
    # 4. an id made by c3_comm_create itself (no id from the caller), then destroy both
    comm2 = c3dist.RcclComm(0, 1, 0)
    assert comm2.ranks_seen() == (1, 0)
    comm2.close()
    comm.close()
    print("REAL_RCCL_OK " + ";".join(mapped))
""")


def _real_rccl(tmp_path, with_torch):
    script = tmp_path / ("real_rccl_%d.py" % with_torch)
    script.write_text(REAL_RCCL.format(root=ROOT, with_torch=with_torch))
    env = dict(os.environ, C3HIP_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    env.pop("C3HIP_RCCL_LIB", None)
    return subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)


def test_one_rank_on_the_real_librccl(tmp_path):
    """Every RCCL call c3_comm.h makes, executed ONCE against the REAL library on the lease's one GPU: dlopen +
    symbol binding, RTLD_NOLOAD sharing of the copy PyTorch ships, ncclGetUniqueId, ncclCommInitRank(nranks = 1), ncclCommCount,
    ncclCommUserRank, a grouped self ncclSend / ncclRecv in c3_gather_rows (C3HIP_FORCE_RCCL=1 makes a world of one take the
    wire path), ncclCommDestroy.  The gathered rows are compared bit for bit."""
    r = _real_rccl(tmp_path, True)
    assert r.returncode == 0 and "REAL_RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
    libs = r.stdout.strip().splitlines()[-1].split(" ", 1)[1]
    assert "torch" in libs, libs  # a process with PyTorch shares PyTorch's copy instead of loading a second librccl


def test_one_rank_on_the_system_librccl_without_pytorch(tmp_path):
    """The same sequence in a process that never imports torch: libc3hip then dlopens the SYSTEM librccl (/opt/rocm/lib)."""
    r = _real_rccl(tmp_path, False)
    assert r.returncode == 0 and "REAL_RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-6000:]
