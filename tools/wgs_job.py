#!/usr/bin/env python3
"""BASELINE.json configs[3] as a runnable, timed job: a synthetic stand-in for the candidate windows of a 50x ONT genome
(~8 M pileup + ~1.5 M full-alignment windows; no BAM exists offline), cut into tensor files of <= 10 000 windows
like preprocess/SelectCandidates.py:379 does, the file list sharded contiguously over the ranks, every rank through the
worker pipeline of its GPU, rows gathered to rank 0 on RCCL (clair3_amd/job.py).

    python tools/wgs_job.py --scale 0.02                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29531 \\
        tools/wgs_job.py --scale 1.0                           # the whole stand-in on an 8-GPU node

Prints one JSON line per model kind on rank 0: windows, files, per-rank split, windows/s (files -> rows on rank 0), and an
order check (the k-th row belongs to the k-th window of the list: rows of a sample of windows are recomputed alone).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FULL_JOB = {"pileup": 8_000_000, "full_alignment": 1_500_000}  # SURVEY 8d config 4 (order of magnitude, synthetic)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.01, help="fraction of the 8 M / 1.5 M window stand-in")
    ap.add_argument("--kinds", default="pileup,full_alignment")
    ap.add_argument("--dir", default=None, help="where to write the tensor files (default: a temp dir, removed afterwards)")
    ap.add_argument("--batch", type=int, default=1000, help="the reference's GPU batch (CallVariantsFromCffi.py:265-269)")
    ap.add_argument("--handles", type=int, default=1)
    args = ap.parse_args()

    import torch
    from clair3_amd import dist as c3dist, job, synthetic as syn
    from clair3_amd.model import Clair3_F, Clair3_P
    rank, world, local = c3dist.init_from_env(backend="gloo" if os.environ.get("C3_JOB_GLOO") else None)
    n_vis = torch.cuda.device_count()
    err = c3dist.preflight(int(os.environ.get("LOCAL_WORLD_SIZE", world)), n_vis, local)
    if err:
        if local == 0:
            print(f"[wgs_job] {err}", file=sys.stderr)
        sys.exit(2)
    c3dist.pin_to_device_numa(local)  # this rank's staging threads on its GPU's NUMA node (before the first staged copy)
    torch.cuda.set_device(local)
    exchange = c3dist.RowExchange(rank, world, device=local)  # RCCL directly; torch.distributed if that does not come up
    base = args.dir or tempfile.mkdtemp(prefix="c3_wgs_job_")
    for kind in args.kinds.split(","):
        n = max(1, int(FULL_JOB[kind] * args.scale))
        d = os.path.join(base, kind)
        ch, indel, cls = (18, False, Clair3_P) if kind == syn.PILEUP else (8, True, Clair3_F)
        if rank == 0:
            t0 = time.perf_counter()
            list_fn, counts = job.write_synthetic_job(d, kind, n, channels=ch)
            t_write = time.perf_counter() - t0
        if world > 1:
            torch.distributed.barrier()
        list_fn = os.path.join(d, "tensor_can_fn_list")
        sd = syn.make_state_dict(kind, ch, indel, seed=0)
        models = []
        for _ in range(args.handles):
            m = cls(add_indel_length=indel, predict=True, input_channels=ch).to(local)
            m.load_state_dict(sd)
            models.append(m)
        model = models if len(models) > 1 else models[0]
        job.run_job(model, list_fn, rank, world, batch_size=args.batch, exchange=exchange)  # warm-up pass: workspaces, page cache
        if world > 1:
            torch.distributed.barrier()
        res = job.run_job(model, list_fn, rank, world, batch_size=args.batch, exchange=exchange)
        if rank == 0:
            y = res["rows"]
            assert y.shape == (n, 90 if indel else 24) and np.isfinite(y).all()
            # order: window g of the list is tile g % unique of the seeded windows; recompute a sample of them alone
            unique = min(2048, n)
            basew = syn.make_windows(kind, unique, seed=0, channels=ch)
            pick = np.unique(np.concatenate([[0, n - 1], np.random.default_rng(0).integers(0, n, 62)]))
            y_alone = models[0].predict_numpy(basew[pick % unique])
            order_ok = bool(np.array_equal(y[pick], y_alone))
            print(json.dumps({"job": "configs[3] stand-in", "kind": kind, "windows": n, "files": len(counts), "n_gpus": world,
                              "windows_per_rank": res["per_rank"], "batch": args.batch, "handles": args.handles,
                              "candidate_windows_per_s": n / res["total_s"], "total_s": res["total_s"],
                              "rank0_compute_s": res["compute_s"], "gather_s": res["gather_s"], "write_files_s": t_write,
                              "rows_in_window_order": order_ok, **exchange.report()}),
                  flush=True)
            assert order_ok
        if world > 1:
            torch.distributed.barrier()
        del models, model
    if exchange.comm is not None:
        exchange.comm.close()
    if args.dir is None and rank == 0:
        shutil.rmtree(base, ignore_errors=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
