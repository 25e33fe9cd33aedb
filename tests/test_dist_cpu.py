"""world_size-2 gloo test of the N>1 path (sharding + gather to rank 0) on CPU; the per-rank "forward" is
the CPU oracle so the check is end-to-end on real rows without a GPU."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from tests.util import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    sys.path.insert(0, {root!r})
    from clair3_amd import dist as c3dist, synthetic as syn
    from oracle import oracle
    rank, world, _ = c3dist.init_from_env(backend="gloo")
    n = {n}
    sd = syn.make_state_dict(syn.PILEUP, seed=11)
    x = syn.make_pileup_windows(n, seed=12)
    lo, hi = c3dist.shard_range(n, rank, world)
    y_local = torch.from_numpy(oracle.pileup_forward(sd, x[lo:hi], n_threads=1))
    y_all = c3dist.gather_rows(y_local, n, dst=0)
    if rank == 0:
        np.save({out!r}, y_all.numpy())
    else:
        assert y_all is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gather_matches_single_process(tmp_path):
    from clair3_amd import synthetic as syn
    from oracle import oracle
    n = 21  # odd: ranks get 11 and 10 rows, exercising the padded gather
    out = str(tmp_path / "y.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, n=n, out=out))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    y = np.load(out)
    sd = syn.make_state_dict(syn.PILEUP, seed=11)
    x = syn.make_pileup_windows(n, seed=12)
    y_single = oracle.pileup_forward(sd, x, n_threads=1)
    assert y.shape == (n, 24)
    assert np.array_equal(y, y_single)  # sharding must not change a single bit, nor the row order


GATHERER = textwrap.dedent("""
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, {root!r})
    from clair3_amd import dist as c3dist
    rank, world, _ = c3dist.init_from_env(backend="gloo")
    B, W, steps, every = 5, 7, 11, 4
    g = c3dist.RowGatherer(B * world, every=every, dst=0)
    got = []
    for step in range(steps):  # row value encodes (step, rank, row)
        y = torch.tensor([[1000.0 * step + 100.0 * rank + r] * W for r in range(B)])
        out = g.add(y)
        if out is not None:
            got.append(out)
    out = g.flush()
    if out is not None:
        got.append(out)
    assert g.flush() is None
    if rank == 0:
        np.save({out!r}, torch.cat(got).numpy())
    else:
        assert not got
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
""")


def test_row_gatherer_groups_steps_and_loses_nothing(tmp_path):
    """bench.py --gpus N sends the rows of 8 steps per collective: groups of `every` steps plus the flushed remainder,
    rank-major inside a group, every (step, rank, row) exactly once"""
    out = str(tmp_path / "g.npy")
    script = tmp_path / "gatherer.py"
    script.write_text(GATHERER.format(root=ROOT, out=out))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    y = np.load(out)
    B, W, steps, every, world = 5, 7, 11, 4, 2
    assert y.shape == (steps * world * B, W)
    want = []
    for g0 in range(0, steps, every):
        group = range(g0, min(g0 + every, steps))
        for rank in range(world):
            for step in group:
                want += [1000.0 * step + 100.0 * rank + r for r in range(B)]
    assert np.array_equal(y[:, 0], np.array(want, dtype=np.float32))


EXCHANGE = textwrap.dedent("""
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, {root!r})
    from clair3_amd import dist as c3dist
    rank, world, _ = c3dist.init_from_env(backend="gloo")
    scenario = {scenario!r}

    class FakeComm:  # what RowExchange needs of RcclComm, with the failure under test
        def __init__(self, rank, world, device, create_timeout_s=None):
            self.rank, self.world, self.device = rank, world, device
            if scenario == "create_fails_on_rank_1" and rank == 1:
                raise TimeoutError("ncclCommInitRank did not return")
        def ranks_seen(self):
            return (1 if scenario == "rccl_sees_one_rank" and self.rank == 0 else self.world), self.rank
        def gather(self, y, counts, dst=0, timeout_s=None):
            if scenario == "first_gather_times_out_on_rank_0":
                assert timeout_s is not None  # the first gather is the guarded one
                if self.rank == 0:
                    raise TimeoutError("c3_gather_rows did not finish")
                return None  # the sender's ncclSend was accepted: rank 1 believes it got through
            return c3dist.gather_counts(y, counts, dst=dst)  # a working direct path
        def close(self):
            pass

    if scenario == "unique_id_fails_on_rank_0":
        # the REAL RcclComm: rank 0 cannot make an id (no librccl / ncclGetUniqueId failed).  It must still take part in the
        # broadcast -- the other ranks are inside it -- and everybody must then agree on the fallback (ADVICE r3: this used
        # to leave ranks 1.. in dist.broadcast while rank 0 went on to the all-reduce)
        from clair3_amd import _lib
        L = _lib.lib()
        real = L.c3_comm_unique_id
        L.c3_comm_unique_id = (lambda buf: 1) if rank == 0 else real
        L.c3_comm_create = lambda *a: (_ for _ in ()).throw(AssertionError("no rank may try to create a communicator"))
        c3dist.RowExchange._cuda_job = lambda self: True
    elif scenario != "gloo_job":
        c3dist.RcclComm = FakeComm
        c3dist.RowExchange._cuda_job = lambda self: True
    ex = c3dist.RowExchange(rank, world, device=0, timeout_s=1.0)
    counts = [3, 5]
    outs = []
    for it in range(3):
        y = torch.full((counts[rank], 4), 10.0 * it + rank)
        outs.append(ex.gather(y, counts, dst=0))
    rep = ex.report()
    if rank == 0:
        for it, o in enumerate(outs):
            assert o.shape == (8, 4) and bool((o[:3] == 10.0 * it).all()) and bool((o[3:] == 10.0 * it + 1).all())
        open({out!r}, "w").write(repr((rep["gather"], rep["rccl_ranks_seen"], bool(rep["fallback_reason"]))))
    else:
        assert all(o is None for o in outs)
        open({out!r} + ".1", "w").write(rep["gather"])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
""")


def _run_exchange(tmp_path, scenario):
    out = str(tmp_path / f"{scenario}.txt")
    script = tmp_path / f"{scenario}.py"
    script.write_text(EXCHANGE.format(root=ROOT, out=out, scenario=scenario))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return eval(open(out).read()), open(out + ".1").read()


def test_row_exchange_falls_back_together(tmp_path):
    """dist.RowExchange (what bench.py --gpus N and job.run_job gather with): RCCL directly while it works, torch.distributed
    for EVERY rank from the moment one rank's rendezvous or first collective fails -- and the rows arrive either way"""
    assert _run_exchange(tmp_path, "gloo_job") == (("torch_fallback", 1, True), "torch_fallback")  # host rows: nothing for RCCL
    assert _run_exchange(tmp_path, "direct_works") == (("rccl_direct", 2, False), "rccl_direct")
    assert _run_exchange(tmp_path, "create_fails_on_rank_1") == (("torch_fallback", 2, True), "torch_fallback")
    assert _run_exchange(tmp_path, "rccl_sees_one_rank") == (("torch_fallback", 1, True), "torch_fallback")
    assert _run_exchange(tmp_path, "first_gather_times_out_on_rank_0") == (("torch_fallback", 2, True), "torch_fallback")
    assert _run_exchange(tmp_path, "unique_id_fails_on_rank_0") == (("torch_fallback", 1, True), "torch_fallback")
