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
    """`bench.py --gpus 2` -- the command the driver's scaling run uses, launched the way it launches it -- executes end to end with
    two ranks on the lease's ONE device: gloo as the control plane (C3_DIST_BACKEND), both ranks on device 0 (C3_BENCH_DEVICE), the
    rows of every 8 steps gathered to rank 0 on c3_gather_rows through the stand-in for librccl.  Not a measurement (two ranks
    share one GPU): what is checked is that the N > 1 path runs -- barriers, the MAX over ranks, the guarded first gather, the
    per-rank rates -- and prints the one parseable line with the gather on the direct path."""
    import json
    env = dict(os.environ, C3HIP_RCCL_LIB=_stub(), C3_DIST_BACKEND="gloo", C3_BENCH_DEVICE="0", OMP_NUM_THREADS="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", C3_BENCH_FULL=str(tmp_path / "full.json"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "2",
           "--repeats", "2", "--workload", "full_alignment"]
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
