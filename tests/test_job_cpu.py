"""The sharded whole-job path (clair3_amd/job.py; BASELINE.json configs[3]) on CPU: files of <= N windows, the file list
split into contiguous per-rank runs, every rank through the worker pipeline, one gather to rank 0.  The per-rank "model" is
the CPU oracle behind the submit/wait contract, so rows and order are checked end to end without a GPU (world_size 2, gloo)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from clair3_amd import job, synthetic as syn
from tests.util import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch
    sys.path.insert(0, {root!r})
    from clair3_amd import dist as c3dist, job, synthetic as syn
    from tests.test_worker import OracleModel
    rank, world, _ = c3dist.init_from_env(backend="gloo")
    sd = syn.make_state_dict(syn.PILEUP, seed=11)
    res = job.run_job(OracleModel(sd), {lst!r}, rank=rank, world=world, batch_size=7)
    if rank == 0:
        np.save({out!r}, res["rows"])
        open({out!r} + ".pos", "w").write(str(res["per_rank"]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_files_is_contiguous_balanced_and_complete():
    for counts, world in (([10, 10, 10, 5], 2), ([10] * 7 + [3], 8), ([5], 3), ([4] * 8 + [1], 4), ([], 2), ([1, 100, 1], 2)):
        cuts = job.shard_files(counts, world)
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == len(counts)
        assert all(a <= b for a, b in zip(cuts, cuts[1:]))
    cuts = job.shard_files([10000] * 150, 8)
    sizes = [cuts[r + 1] - cuts[r] for r in range(8)]
    assert max(sizes) - min(sizes) <= 1  # 150 equal files over 8 ranks: 19 or 18 each


def test_synthetic_job_files_have_the_reference_format(tmp_path):
    from clair3_amd import worker
    lst, counts = job.write_synthetic_job(str(tmp_path), syn.PILEUP, 25, per_file=10, unique=8)
    assert counts == [10, 10, 5]
    names, c2 = job.file_window_counts(lst)
    assert c2 == counts and names[0] == "tensor_00000"
    batches = list(worker.iter_batches(lst, 1000))
    assert [len(b[0]) for b in batches] == [10, 10, 5]
    assert batches[2][1][4].startswith("chrS:25:") and batches[0][0].dtype == np.int8


def test_single_rank_job_rows_in_window_order(tmp_path):
    from oracle import oracle
    from tests.test_worker import OracleModel
    lst, counts = job.write_synthetic_job(str(tmp_path), syn.PILEUP, 23, per_file=6, unique=23, seed=3)
    sd = syn.make_state_dict(syn.PILEUP, seed=11)
    res = job.run_job(OracleModel(sd), lst, batch_size=4)
    x = syn.make_windows(syn.PILEUP, 23, seed=3, channels=18)
    assert np.array_equal(res["rows"], oracle.pileup_forward(sd, x, n_threads=1))
    assert [int(p.split(":")[1]) for p in res["positions"]] == list(range(1, 24))


def test_two_rank_job_matches_single_process(tmp_path):
    from oracle import oracle
    d = tmp_path / "job"
    lst, counts = job.write_synthetic_job(str(d), syn.PILEUP, 37, per_file=8, unique=37, seed=5)  # files 8,8,8,8,5
    out = str(tmp_path / "y.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, lst=lst, out=out))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    y = np.load(out)
    sd = syn.make_state_dict(syn.PILEUP, seed=11)
    x = syn.make_windows(syn.PILEUP, 37, seed=5, channels=18)
    assert np.array_equal(y, oracle.pileup_forward(sd, x, n_threads=1)), "rows of the two-rank job differ or are out of order"
    assert open(out + ".pos").read() == "[19, 18]"  # five files on two ranks: contiguous WINDOW ranges (file 2 is cut at window 3)


def test_shard_segments_by_files_or_by_window_ranges():
    # plenty of files: whole files per rank, exactly shard_files' cuts
    counts = [10] * 16
    segs = job.shard_segments(counts, 4)
    assert [[f for f, _, _ in s] for s in segs] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15]]
    assert all(lo == 0 and hi == 10 for s in segs for _, lo, hi in s)
    # few files: contiguous window ranges, files cut where the range ends
    for counts, world in (([10000, 10000], 8), ([7, 0, 5, 1], 3), ([3], 4), ([], 2), ([8, 8, 8, 8, 5], 2)):
        segs = job.shard_segments(counts, world)
        assert len(segs) == world
        flat = [(f, w) for s in segs for f, lo, hi in s for w in range(lo, hi)]
        assert flat == [(f, w) for f, c in enumerate(counts) for w in range(c)]  # complete, in order, nothing twice
        sizes = [sum(hi - lo for _, lo, hi in s) for s in segs]
        assert max(sizes) - min(sizes) <= 1
    assert job.shard_segments([10000, 10000], 8)[3] == [(0, 7500, 10000)]
