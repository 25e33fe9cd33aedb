#!/usr/bin/env python3
"""Throughput benchmark of the MI355X-native Clair3 inference path (BASELINE.json metric:
candidate-windows/sec, pileup + full-alignment; GT-call concordance vs the reference arithmetic).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a torchrun environment: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic candidate windows.  One process per GPU, windows
sharded with no data-path collective (weak scaling: every GPU gets the configured batch); for N > 1 the (B, 24|90)
probability rows travel to rank 0 in one RCCL gather per GATHER_EVERY steps.  Rank 0 prints ONE JSON line:

  value           whole-job candidate-windows/s of the headline workload = BASELINE.json configs[2] "ONT r10.4.1
                  full-alignment model, synthetic (B=256, 89, 33, 8)", windows RESIDENT IN HBM when the timed region
                  starts (the bench contract: c3_predict_device, every kernel of the forward pass).  The better of
                  "one_batch_in_flight" (K steps strictly one after the other on one handle) and three handles on three
                  HIP streams; both are reported;
  host_inclusive  the SURVEY 8d metric: the same K steps through c3_predict_submit / c3_predict_wait from PAGEABLE numpy
                  windows to probability rows landed in host memory (staging copy + H2D + kernels + D2H, one handle, a
                  ring of three slots), at the configured batch and at the reference's GPU batch of 1000
                  (clair3/CallVariantsFromCffi.py:265-269).  N = 1 only;
  roofline        dominant kernel family.  Its TIME is a share of the un-instrumented one-batch-in-flight step: a profiled
                  pass over the same steps brackets every kernel launch with HIP events on the launch stream and gives the
                  family's share of the sum of all kernels; kernel_us_per_step = share x ms_per_step, so it is <= the step
                  by construction (bracketed launches are a few per cent longer than free-running ones; the raw event
                  sums are in `events`).  achieved = ALGORITHMIC FLOP of the family (2*MACs of the reference layer
                  shapes) / that time; peak = the dense peak of the matrix instruction the family issues
                  (2500 TFLOP/s for the 16-bit forms every contraction uses, 157.3 for the fp32 fallbacks); frac =
                  achieved / peak; whole_network_frac = FLOP of the WHOLE forward pass / ms_per_step / peak; mfma_util =
                  FLOP the matrix instructions EXECUTE (three fp16 piece products per fp32 product, tile padding) / time /
                  peak; traffic / fabric_frac_of_hbm_peak from the committed rocprofv3 PMC passes
                  (profiles/pmc_traffic*.json): FETCH_SIZE / WRITE_SIZE count requests on the memory side of the L2 --
                  Infinity-Cache hits included -- so they are FABRIC bytes, an upper bound of what reaches HBM;
  cpu_baseline    the reference CPU path on this node's host cores, rank 0, N = 1, bounded sample: kind "reference" = the
                  reference's own clair3/model.py modules called as its _torch_predict does (staged into the git-ignored
                  oracle/_ref by oracle/stage_reference.py; "port" = oracle/torch_port.py, the same ATen operators, when
                  no staged copy is there): (i) in-process at its best thread count, (ii) one thread (how the
                  reference pipeline runs its workers) x usable cores;
  gt_concordance  arg-max of the gt21 and zygosity heads of the GPU rows vs the CPU baseline's rows on the same batch.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)

WORKLOADS = {
    # name: (kind, batch, channels, add_indel_length, FLOP/window, algorithmic bytes/window, BASELINE.json config)
    "full_alignment": ("full_alignment", 256, 8, True, 451538432, 23856,
                       "configs[2]: ONT r10.4.1 full-alignment model, synthetic (B=256, 89, 33, 8) int8, 1xMI355X"),
    "pileup": ("pileup", 1024, 18, False, 47785984, 690,
               "configs[1]: ONT r10.4.1 pileup model, synthetic (B=1024, 33, 18) [= (1024, 33, 594/33)] int8, 1xMI355X"),
    "full_alignment_dwell": ("full_alignment", 256, 9, True, 452419712, 26793,
                             "configs[4]: ONT --enable_dwell_time full-alignment, synthetic (B=256, 89, 33, 9) int8"),
}
TRAFFIC_FILE = {"full_alignment": "pmc_traffic.json", "pileup": "pmc_traffic_pileup.json"}
HOST_SLOTS = 3
GATHER_EVERY = 8  # steps per gather of probability rows to rank 0 (N > 1)


def build_model(kind, channels, indel, device):
    from clair3_amd import synthetic as syn
    from clair3_amd.model import Clair3_F, Clair3_P
    cls = Clair3_P if kind == syn.PILEUP else Clair3_F
    m = cls(add_indel_length=indel, predict=True, input_channels=channels).to(device)
    m.eval()
    sd = syn.make_state_dict(kind, channels, indel, seed=0)
    m.load_state_dict(sd)
    return m, sd


def host_leg(model, x_host, steps, warmup, slots=None):
    """K steps host to host: pageable numpy windows -> rows in host memory, ring of HOST_SLOTS submits in flight on ONE
    handle (c3_predict_submit / c3_predict_wait) -- or, with a list of handles, batch i on handle i % len with `slots` submits
    in flight on each (what clair3_amd.worker.predict_batches does with a list).  Returns (seconds, last rows)."""
    models = list(model) if isinstance(model, (list, tuple)) else [model]
    slots = slots or HOST_SLOTS
    tickets = []
    y = None

    def run(k):
        nonlocal y
        for i in range(k):
            if len(tickets) == slots * len(models):
                mi, t = tickets.pop(0)
                y = mi.wait(t)
            mi = models[i % len(models)]
            tickets.append((mi, mi.submit(x_host, slot=(i // len(models)) % slots)))
        while tickets:
            mi, t = tickets.pop(0)
            y = mi.wait(t)

    run(warmup)
    t0 = time.perf_counter()
    run(steps)
    return time.perf_counter() - t0, y


def host_leg_median(model, x_host, steps, warmup, slots=None, passes=3):
    """host_leg, `passes` times back to back (the first with the warm-up): (median seconds, last rows, windows-per-second-free list of the
    passes' seconds).  A host leg of 20 steps is 7 ms of wall time: one scheduler hiccup of the box's host (seen: 431 k against 786 k
    windows/s on the same binary a minute apart) must not be the figure."""
    els, y = [], None
    for i in range(max(1, passes)):
        el, y = host_leg(model, x_host, steps, warmup if i == 0 else 0, slots)
        els.append(el)
    return sorted(els)[len(els) // 2], y, els


def run_workload(name, args, rank, world, local):
    import torch
    import torch.distributed as dist
    from clair3_amd import dist as c3dist, synthetic as syn
    kind, batch, channels, indel, flop_w, bytes_w, cfg = WORKLOADS[name]
    if args.batch:
        batch = args.batch
    if os.environ.get("C3_BENCH_DEVICE"):  # every rank on ONE device: the N > 1 code path on a one-GPU box (tests/test_comm_gpu.py), never a measurement
        local = int(os.environ["C3_BENCH_DEVICE"])
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    ctl = dev if (world == 1 or dist.get_backend() == "nccl") else torch.device("cpu")  # where the tensors of control-plane collectives live
    model, sd = build_model(kind, channels, indel, local)
    x_host = syn.make_windows(kind, batch, seed=1000 + rank, channels=channels)
    x = torch.from_numpy(x_host).to(dev)  # inputs resident in HBM before the timed region
    n_total = batch * world
    # --streams S > 1: S model handles (own workspace each) on S HIP streams, steps issued round-robin -- a worker
    # that keeps S batches in flight, which is how the reference itself drives a GPU (free_MB // 8000 concurrent
    # workers per device, clair3/CallVariantsFromCffiGPU.py:55-56).  Every step is still one complete forward pass.
    S = args.streams if args.streams > 0 else 3
    models = [model] + [build_model(kind, channels, indel, local)[0] for _ in range(S - 1)]
    streams = [torch.cuda.Stream(device=dev) for _ in models]
    # part of model set-up, like the weight upload: every handle sizes its device workspace on its first batch
    model(x)
    for mi, st in zip(models, streams):
        with torch.cuda.stream(st):
            mi(x)
    torch.cuda.synchronize()
    # an idle MI355X needs ~20 ms of load before its clocks settle: 60 ms of the same forward passes, untimed
    t_ramp = time.perf_counter() + 0.06
    while time.perf_counter() < t_ramp:
        for mi, st in zip(models, streams):
            with torch.cuda.stream(st):
                mi(x)
        torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    last_rows = [None]
    # N > 1: the rows of GATHER_EVERY steps travel to rank 0 in one gather -- on RCCL directly (c3_gather_rows), watched: a
    # rendezvous or first collective that does not complete in time sends every rank to torch.distributed's gather instead
    exchange = c3dist.RowExchange(rank, world, device=local, timeout_s=args.gather_timeout, direct=not args.torch_gather)

    def timed(n_streams):
        counter = [0]

        def step():
            i = counter[0] % n_streams
            counter[0] += 1
            if n_streams == 1:
                y = model(x)  # c3_predict_device on torch's current stream
            else:
                with torch.cuda.stream(streams[i]):
                    y = models[i](x)
            last_rows[0] = y
            if world > 1:
                if n_streams > 1:
                    torch.cuda.current_stream(dev).wait_stream(streams[i])
                    y.record_stream(torch.cuda.current_stream(dev))  # the gather below reads y on the default stream
                return gatherer.add(y)
            return y

        gatherer = c3dist.RowGatherer(n_total, every=GATHER_EVERY, dst=0, exchange=exchange)
        for _ in range(args.warmup):
            step()
        gatherer.flush()
        blocks, own, outs = [], [], []
        for rep in range(max(1, args.repeats)):  # every block: EXACTLY K steps between two fences, MAX over ranks
            if rep:  # the W untimed steps again in front of every later block: the fences and the bookkeeping between two blocks are an
                     # idle gap after which a 7 ms block (the driver's 20 steps) would start on clocks that have begun to fall
                for _ in range(args.warmup):
                    step()
                gatherer.flush()
            fence()
            t0 = time.perf_counter()
            out = None
            for _ in range(args.steps):
                got = step()
                out = got if got is not None else out
            got = gatherer.flush()
            out = got if got is not None else out
            torch.cuda.synchronize()
            own.append(time.perf_counter() - t0)
            fence()
            elapsed = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            blocks.append(elapsed)
            outs.append(out)
        if rank == 0:  # checked behind the last block: nothing but the fences stands between two timed blocks
            for out in outs:
                assert out is not None and out.shape[1] == (90 if indel else 24) and out.shape[0] % n_total == 0 and out.shape[0] > 0
                assert bool(torch.isfinite(out).all())
        return blocks, own

    def stats(blocks, n=n_total):
        r = sorted(n * args.steps / b for b in blocks)
        return {"median": r[len(r) // 2] if len(r) % 2 else 0.5 * (r[len(r) // 2 - 1] + r[len(r) // 2]), "min": r[0], "max": r[-1],
                "repeats": len(r), "steps_per_repeat": args.steps}

    describe = lambda mi: mi.describe()
    single_blocks, single_own = timed(1)
    variants = {"one_batch_in_flight": describe(model)}
    if S > 1:
        for mi in models:
            mi.sharing(S)  # the caller's hint: S handles feed this GPU side by side (tile shapes of the pileup kernels follow it)
    multi_blocks, multi_own = timed(S) if S > 1 else (single_blocks, single_own)
    if S > 1:
        variants[f"{S}_batches_in_flight"] = describe(models[-1])
        for mi in models:
            mi.sharing(1)
    med = lambda b: float(np.median(b))
    single, multi = med(single_blocks), med(multi_blocks)
    elapsed, in_flight, own = (multi, S, multi_own) if multi <= single else (single, 1, single_own)
    per_rank = None
    if world > 1:  # every rank's own K-step time (before the MAX), so the driver can see N ranks at work
        t = torch.tensor([med(own)], dtype=torch.float64, device=ctl)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        per_rank = [batch * args.steps / float(p.item()) for p in parts]
    # the device-resident entry is unchecked (asynchronous): the handles report afterwards whether any batch came near
    # the range of the fp16x3 kernels
    flags = [m.range_status() for m in models]
    res = {
        "workload": cfg, "batch_per_gpu": batch, "windows_per_step": n_total, "batches_in_flight": in_flight,
        "value": n_total * args.steps / elapsed, "ms_per_step": 1e3 * elapsed / args.steps,
        "value_stats": stats(multi_blocks if in_flight > 1 else single_blocks),
        "one_batch_in_flight": {"value": n_total * args.steps / single, "ms_per_step": 1e3 * single / args.steps, **stats(single_blocks)},
        f"{S}_batches_in_flight": {"value": n_total * args.steps / multi, "ms_per_step": 1e3 * multi / args.steps, **stats(multi_blocks)},
        "kernel_variants": variants,
        "flop_per_window": flop_w, "bytes_per_window": bytes_w,
        "range_flag_raised": any(f for f, _ in flags), "on_fp32_fallback": any(o for _, o in flags),
        "rows_rank0": last_rows[0].cpu().numpy() if rank == 0 else None,
    }
    if world > 1:
        res["multi_gpu"] = {**exchange.report(), "per_rank_windows_per_s": per_rank, "ranks": world,
                            "gather_every_steps": GATHER_EVERY}

    # ---- host-inclusive leg (H2D + kernels + D2H), N = 1 ----
    if world == 1 and not args.no_host_leg:
        hl = {"slots_in_flight": HOST_SLOTS, "handles": 1,
              "path": "pageable numpy -> staging pool memcpy -> pinned -> H2D -> kernels -> D2H -> numpy (c3_predict_submit / _wait)"}
        # the supplementary host legs run >= 100 (B = 256 / 1024) and >= 25 (B = 1000) steps whatever --steps says: a ring of three
        # batches needs more than the driver's 20 steps before filling and draining it stop showing (0.875 against 0.955)
        hsteps = max(args.steps, 100)
        el, y, els = host_leg_median(model, x_host, hsteps, args.warmup)
        assert y.shape == (batch, 90 if indel else 24) and np.isfinite(y).all()
        hl.update({"value": batch * hsteps / el, "ms_per_step": 1e3 * el / hsteps, "batch": batch, "steps": hsteps,
                   "passes": [round(batch * hsteps / e) for e in els],  # value = the median pass
                   "frac_of_device_resident_one_in_flight": (batch * hsteps / el) / res["one_batch_in_flight"]["value"]})
        # the same ring over EXACTLY the driver's --steps / --warmup (the figure above runs >= 100 steps, outside the
        # driver's consistency check): filling and draining three slots is part of these few steps
        el_d, _, els_d = host_leg_median(model, x_host, args.steps, args.warmup)
        hl["at_driver_steps"] = {"value": batch * args.steps / el_d, "ms_per_step": 1e3 * el_d / max(args.steps, 1), "steps": args.steps,
                                 "warmup": args.warmup, "passes": [round(batch * args.steps / e) for e in els_d]}
        if not args.batch:
            bref = 1000  # the reference's GPU batch (CallVariantsFromCffi.py:265-269: predictBatchSize * 5)
            xb = syn.make_windows(kind, bref, seed=2000, channels=channels)
            k = max(25, args.steps * batch // bref)
            el, y, els_b = host_leg_median(model, xb, k, 3)
            xd = torch.from_numpy(xb).to(dev)
            for _ in range(3):
                model(xd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                model(xd)
            torch.cuda.synchronize()
            el_dev = time.perf_counter() - t0
            hl["batch_1000"] = {"value": bref * k / el, "ms_per_step": 1e3 * el / k, "steps": k, "passes": [round(bref * k / e) for e in els_b],
                                "device_resident_one_in_flight": bref * k / el_dev,
                                "frac_of_device_resident": el_dev / el}
            # the reference loop's own call: ONE blocking _torch_predict per batch (predict._hip_predict -> c3_predict, which cuts
            # the batch into chunks that travel through the ring), every piece through the staging buffer: the product's default.
            # (until round 5 a --page-locked-legs switch timed the same call on hipHostRegister'ed windows; the entries are gone from the ABI,
            # include/c3hip.h says why)
            xs = np.array(xb, copy=True)
            for _ in range(3):
                y = model.predict_numpy(xs)
            t0 = time.perf_counter()
            for _ in range(k):
                y = model.predict_numpy(xs)
            el_sync = time.perf_counter() - t0
            hl["batch_1000"]["sync_call"] = {"value": bref * k / el_sync, "ms_per_call": 1e3 * el_sync / k,
                                             "frac_of_device_resident": el_dev / el_sync,
                                             "path": "_hip_predict on the worker's model = c3_predict, every piece through the staging buffer (the default): one blocking call per batch"}
            del xs
            # what the UNMODIFIED reference loop gets after callvar.install(): its batch generator rebound to
            # worker.lookahead_batches over the tensor FILES of stage A (memory-mapped .npy + .info, batches of 1000 that never
            # span files; consecutive batches of a file travel in one forward pass, two groups in flight beyond the one being
            # read) and one blocking _torch_predict (= predict._hip_predict) per batch
            from clair3_amd import predict as c3predict, worker as c3worker
            import shutil
            per_file = 10000 if kind == syn.PILEUP else 4000  # <= 10 000 windows per file (preprocess/SelectCandidates.py:379)
            n_files = 40 if kind == syn.PILEUP else 6  # a fixed job (400 000 / 24 000 windows): long enough that pipeline fill and drain do not dominate
            tdir = tempfile.mkdtemp(prefix="c3_bench_files_")
            try:
                names = []
                tiled = np.concatenate([xb] * (per_file // bref))
                info = "\n".join("chrS:%d:%s\t30-RA 30 " % (i + 1, "ACGT" * 8 + "A") for i in range(per_file)) + "\n"
                for fi in range(n_files):
                    np.save(os.path.join(tdir, f"t{fi}.npy"), tiled)
                    with open(os.path.join(tdir, f"t{fi}.info"), "w") as fh:
                        fh.write(info)
                    names.append(f"t{fi}")
                list_fn = os.path.join(tdir, "tensor_list")
                with open(list_fn, "w") as fh:
                    fh.write("\n".join(names) + "\n")
                del tiled

                def loop(group_windows):
                    gen = c3worker.lookahead_batches(model, c3worker.iter_tensor_files(list_fn), bref, c3predict._PENDING, depth=2,
                                                     group_windows=group_windows)
                    n = 0
                    for X, _, _ in gen:  # clair3/CallVariantsFromCffi.py:308-317
                        yy = c3predict._hip_predict(model, None, X)
                        n += len(yy)
                    assert n == n_files * per_file and np.isfinite(yy).all()
                    return n
                gw = c3worker.group_windows_for(model)
                loop(gw)
                el_loops = []
                for _ in range(3):  # a 35 - 45 ms job: the best of three passes (the spread goes into the line)
                    t0 = time.perf_counter()
                    n_loop = loop(gw)
                    el_loops.append(time.perf_counter() - t0)
                el_loop = min(el_loops)
                t0 = time.perf_counter()
                loop(0)
                el_loop1 = time.perf_counter() - t0
            finally:
                shutil.rmtree(tdir, ignore_errors=True)
            hl["batch_1000"]["dropin_loop"] = {"value": n_loop / el_loop, "ms_per_call": 1e3 * el_loop / (n_loop / bref),
                                               "frac_of_device_resident": (n_loop / el_loop) / (bref * k / el_dev),
                                               "passes": [round(n_loop / e) for e in el_loops],
                                               "windows_per_forward_pass": gw, "tensor_files": n_files, "windows_per_file": per_file,
                                               "one_forward_pass_per_batch": {"value": n_loop / el_loop1,
                                                                              "frac_of_device_resident": (n_loop / el_loop1) / (bref * k / el_dev)},
                                               "path": "callvar.install(): tensor_generator_for_chunk -> worker.lookahead_batches over memory-mapped tensor "
                                                       "files (groups of consecutive batches per forward pass, 2 groups ahead), one blocking _torch_predict "
                                                       "per batch of 1000 = wait for rows already in flight; frac is against the device-resident rate at B=1000"}
            # every handle of the device-resident headline fed from the host: batch i on handle i % S, two submits in flight each
            if len(models) > 1:
                el3, y = host_leg(models, xb, k, 3, slots=2)
                hl["batch_1000"]["all_handles"] = {"value": bref * k / el3, "ms_per_step": 1e3 * el3 / k, "handles": len(models),
                                                   "slots_in_flight_per_handle": 2}
        res["host_inclusive"] = hl

    if args.no_profiled_pass:
        return res
    # ---- profiled pass: HIP events around every kernel launch on the launch stream ----
    model.profile(True)
    model.profile_reset()
    for _ in range(args.steps):
        model(x)
    torch.cuda.synchronize()
    stats = model.profile_read()
    model.profile(False)
    res["kernels"] = {r["name"]: {"launches": r["launches"], "avg_us": 1e3 * r["total_ms"] / r["launches"],
                                  "tflops": r["flops"] / r["total_ms"] / 1e9 if r["total_ms"] > 0 else None,
                                  "mfma_util": (r["mfma_flops"] / r["total_ms"] / 1e9 / r["mfma_peak_tflops"])
                                  if r["total_ms"] > 0 and r["mfma_peak_tflops"] > 0 else None}
                      for r in stats}
    if kind == syn.FULL_ALIGNMENT:
        dom = [r for r in stats if r["name"].startswith(("fa.conv", "fa.res", "fa.stage"))]
        dom_name = ("3x3 convolution family of Clair3_F (conv1 -- computed inside fa.res1a / fa.res1b, its algorithmic "
                    "FLOP counted once, in fa.res1a --, the two other stride-2 convs, six residual-block convs): the launches named "
                    "fa.conv* / fa.res* in `kernels`")
    else:
        dom = [r for r in stats if r["name"].startswith(("p.lstm", "p.proj"))]
        dom_name = "the two BiLSTM layers of Clair3_P (input projections + recurrences): the launches named p.lstm* / p.proj* in `kernels`"
    ms = sum(r["total_ms"] for r in dom)
    fl = sum(r["flops"] for r in dom)
    mf = sum(r["mfma_flops"] for r in dom)
    launches = sum(r["launches"] for r in dom)
    peak = max([r["mfma_peak_tflops"] for r in dom] + [0.0]) or 2500.0
    ms_all = max(sum(r["total_ms"] for r in stats), 1e-9)
    step_ms_profiled = ms_all / max(args.steps, 1)
    # The family's time is its SHARE (from the event-bracketed pass) of the un-instrumented step: bracketing every launch
    # with events makes each a little longer (the eight convolution launches alone read longer than the whole free-running
    # step), so the raw event sum is not a duration of the step the line reports -- the share of it is.
    step_us = 1e3 * res["one_batch_in_flight"]["ms_per_step"]
    share, fam_us = family_time(ms, ms_all, step_us)
    fl_step, mf_step = fl / max(args.steps, 1), mf / max(args.steps, 1)
    achieved = fl_step / fam_us / 1e6 if fam_us > 0 else 0.0  # FLOP / us / 1e6 = TFLOP/s
    traffic, traffic_note, fabric_frac = None, None, None
    tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE.get(name, ""))
    if name in TRAFFIC_FILE and os.path.exists(tpath) and not args.batch:
        # bytes on the memory side of the L2 from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, collected in their
        # own runs by tools/gpu_round.sh pmc_fa / pmc_p; bench.py cannot run rocprof on itself)
        with open(tpath) as fh:
            tj = json.load(fh)
        traffic = tj.get("fabric_bytes_per_launch", tj.get("hbm_bytes_per_launch"))
        step_bytes = tj.get("fabric_bytes_per_step", tj.get("hbm_bytes_per_step"))
        traffic_note = {"source": "profiles/%s (%s): FETCH_SIZE x2 + WRITE_SIZE = requests on the memory side of the L2, Infinity-Cache hits "
                                  "included (MI355X_MICROARCH.md): fabric bytes, an upper bound of HBM bytes" % (TRAFFIC_FILE[name], tj.get("tag", "")),
                        "unit": "bytes per launch of the dominant family (PMC)",
                        "fabric_bytes_per_step_all_kernels": step_bytes,
                        "algorithmic_bytes_per_step": bytes_w * batch,
                        "ratio_to_algorithmic": (step_bytes / (bytes_w * batch)) if step_bytes else None,
                        "l2_hit_rate": tj.get("l2_hit_rate")}
        if step_bytes:
            fabric_frac = step_bytes / (res["one_batch_in_flight"]["ms_per_step"] * 1e-3) / (HBM_PEAK_GBS * 1e9)
    whole_one = res["one_batch_in_flight"]["value"] / world * flop_w / (peak * 1e12)
    res["roofline"] = {
        "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
        "mfma_util": (mf_step / fam_us / 1e6 / peak) if fam_us > 0 else None,
        "traffic": traffic, "traffic_note": traffic_note, "fabric_frac_of_hbm_peak": fabric_frac, "kernel": dom_name,
        "avg_launch_us": fam_us * max(args.steps, 1) / max(launches, 1), "launches": launches,
        "share_of_step_time": share,
        "kernel_us_per_step": fam_us,  # the dominant family's share of the free-running step: <= ms_per_step of one batch in flight
        "step_us_one_batch_in_flight": step_us,
        # the RAW event-based figures next to the share-based ones (a share of the free-running step also carries that step's
        # launch gaps): the family's bracketed launch durations as measured, and the fraction they give -- comparable with round 4's lines
        "events": {"family_us_per_step": 1e3 * ms / max(args.steps, 1), "all_kernels_us_per_step": 1e3 * step_ms_profiled,
                   "achieved": (fl_step / (1e3 * ms / max(args.steps, 1)) / 1e6) if ms > 0 else None,
                   "frac": (fl_step / (1e3 * ms / max(args.steps, 1)) / 1e6 / peak) if ms > 0 else None,
                   "avg_launch_us": 1e3 * ms / max(launches, 1),
                   "note": "raw HIP-event sums of the profiled pass (every launch bracketed): longer than the free-running step"},
        "kernel_variants": describe(model),
        "whole_network_frac": whole_one, "whole_network_achieved": whole_one * peak,
        "whole_forward_frac": res["value"] / world * flop_w / (peak * 1e12),
        "whole_forward_frac_one_in_flight": whole_one,
        "algorithmic_bytes_per_window": bytes_w,
        "hbm_algorithmic_gbs": res["value"] / world * bytes_w / 1e9, "hbm_peak_gbs": HBM_PEAK_GBS,
        "note": "achieved = ALGORITHMIC FLOP (direct 3x3 convolution / LSTM shapes of the reference, SURVEY 8d) of the family / its share of "
                "the free-running step; peak = dense peak of the matrix instruction in use (fp16 inputs, fp32 accumulate); every fp32 product "
                "is formed from three fp16 piece products (fp16x3, DESIGN.md 1), so mfma_util ~ 3 x frac is what the matrix pipe actually executes",
    }
    return res


def family_time(family_event_ms, all_kernels_event_ms, step_us):
    """(share, microseconds per step) of a kernel family: its share of the event-bracketed kernel time of the profiled pass,
    applied to the FREE-RUNNING step -- never longer than the step it is a part of (tests/test_bench_line.py)."""
    # the family's launches are a subset of the launches of the pass: a share above one means the profile is inconsistent (a family
    # counted twice, a stale record) and is an error, not something to clamp
    if family_event_ms > all_kernels_event_ms * (1.0 + 1e-9):
        raise ValueError(f"inconsistent profile: the family's bracketed time {family_event_ms} exceeds that of all kernels {all_kernels_event_ms}")
    share = family_event_ms / max(all_kernels_event_ms, 1e-12)
    return share, share * step_us


def staged_reference():
    """The reference's own modules for the CPU baseline: oracle/_ref (staged by oracle/stage_reference.py, travels with the
    snapshot) or $CLAIR3_REFERENCE.  /root/reference itself is never read by the bench."""
    for cand in (os.environ.get("CLAIR3_REFERENCE"), os.path.join(ROOT, "oracle", "_ref")):
        if cand and os.path.isfile(os.path.join(cand, "clair3", "model.py")):
            return cand
    return None


def cpu_worker(name, threads, budget_s, batch, rows_path):
    """One clean process per thread count (OMP_NUM_THREADS is set by the parent): times the reference CPU path on the same
    synthetic batch and prints one JSON line.  kind "reference": the reference's own modules and model call
    (clair3/model.py Clair3_P / Clair3_F in eval(), clair3/CallVariantsFromCffi.py:48-52 _torch_predict) from the staged
    copy; kind "port" (no staged copy): oracle/torch_port.py, the same ATen operators restated."""
    import torch
    from clair3_amd import synthetic as syn
    kind, b, channels, indel, _, _, _ = WORKLOADS[name]
    b = batch or b
    torch.set_num_threads(threads)
    sd_np = syn.make_state_dict(kind, channels, indel, seed=0)
    x = syn.make_windows(kind, b, seed=1000, channels=channels)
    if threads == 1:
        x = x[: max(8, b // 8)]  # one thread: a bounded slice of the batch (rate per window is what is reported)
    ref = staged_reference() if not os.environ.get("C3_BENCH_CPU_PORT") else None
    how = "port"
    if ref:
        try:
            sys.path.insert(0, ref)
            from clair3.model import Clair3_F, Clair3_P
            m = (Clair3_P if kind == "pileup" else Clair3_F)(add_indel_length=indel, predict=True, input_channels=channels)
            m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()})  # strict, as :28
            m.eval()
            device = torch.device("cpu")

            def _torch_predict(model, device, X):  # clair3/CallVariantsFromCffi.py:48-52, the call being replaced
                with torch.inference_mode():
                    X_tensor = torch.from_numpy(X).to(device)
                    Y = model(X_tensor)
                return Y.detach().cpu().numpy()
            run = lambda xs: _torch_predict(m, device, xs)
            how = "reference"
        except Exception as e:
            print(f"[bench] staged reference unusable ({e!r}); timing the port", file=sys.stderr)
    if how == "port":
        from oracle import torch_port
        sd = torch_port.to_torch(sd_np)
        kw = {"lstms": torch_port.make_lstms(sd)} if kind == "pileup" else {}
        run = lambda xs: torch_port.forward(kind, sd, torch.from_numpy(xs), indel, **kw)
    run(np.ascontiguousarray(x[: max(1, len(x) // 4)]))  # warm-up
    times = []
    y = None
    t_start = time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_start < budget_s or not times):
        t0 = time.perf_counter()
        y = run(x)
        times.append(time.perf_counter() - t0)
    if rows_path and rows_path != "-":
        np.save(rows_path, np.asarray(y, dtype=np.float32))
    print(json.dumps({"threads": threads, "batch": int(len(x)), "reps": len(times), "median_s": float(np.median(times)),
                      "torch": torch.__version__, "kind": how}), flush=True)


def ref_gpu_worker(name, budget_s, batch):
    """The reference AS IT IS on this GPU: its own modules (staged copy) moved to the device by PyTorch-ROCm and called the way
    its worker does (clair3/CallVariantsFromCffi.py:48-52: H2D, forward through ATen / MIOpen / rocBLAS, D2H per batch) -- what
    `--use_gpu` gives a user of the reference on an MI355X today.  Prints one JSON line."""
    import torch
    from clair3_amd import synthetic as syn
    kind, b, channels, indel, _, _, _ = WORKLOADS[name]
    ref = staged_reference()
    if not ref or not torch.cuda.is_available():
        print(json.dumps({"error": "no staged reference or no GPU"}))
        return
    sys.path.insert(0, ref)
    from clair3.model import Clair3_F, Clair3_P
    device = torch.device("cuda")
    m = (Clair3_P if kind == "pileup" else Clair3_F)(add_indel_length=indel, predict=True, input_channels=channels)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in syn.make_state_dict(kind, channels, indel, seed=0).items()})
    m.to(device)
    m.eval()

    def _torch_predict(model, device, X):  # clair3/CallVariantsFromCffi.py:48-52
        with torch.inference_mode():
            X_tensor = torch.from_numpy(X).to(device)
            Y = model(X_tensor)
        return Y.detach().cpu().numpy()
    out = {"torch": torch.__version__}
    for bb in sorted({batch or b, 1000}):
        x = syn.make_windows(kind, bb, seed=1000, channels=channels)
        t0 = time.perf_counter()
        y = _torch_predict(m, device, x)  # first call: MIOpen / rocBLAS kernel selection
        first = time.perf_counter() - t0
        _torch_predict(m, device, x)
        times = []
        t_start = time.perf_counter()
        while len(times) < 20 and (time.perf_counter() - t_start < budget_s / 2 or len(times) < 3):
            t0 = time.perf_counter()
            y = _torch_predict(m, device, x)
            times.append(time.perf_counter() - t0)
        out[f"batch_{bb}"] = {"value": bb / float(np.median(times)), "ms_per_call": 1e3 * float(np.median(times)), "calls": len(times),
                              "first_call_s": first, "finite": bool(np.isfinite(y).all())}
    print(json.dumps(out), flush=True)


def reference_gpu(name, budget_s, r):
    """rank 0, N = 1: run ref_gpu_worker in its own process (bounded; a failure only costs this entry)"""
    import subprocess
    if not staged_reference():
        return None
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--ref-gpu-worker", name, str(budget_s), "0"],
                           capture_output=True, text=True, timeout=budget_s * 4 + 240)
        got = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}
    if "error" in got:
        return got
    got["what"] = ("the reference's own clair3/model.py modules on this MI355X through PyTorch-ROCm (ATen / MIOpen / rocBLAS), one blocking "
                   "_torch_predict per batch as its GPU loop does (clair3/CallVariantsFromCffi.py:48-52,317) -- the path libc3hip replaces")
    mine = r.get("host_inclusive", {}).get("batch_1000", {})
    if "batch_1000" in got and mine.get("sync_call"):
        got["libc3hip_blocking_call_speedup_at_batch_1000"] = mine["sync_call"]["value"] / got["batch_1000"]["value"]
        if mine.get("dropin_loop"):
            got["libc3hip_dropin_loop_speedup_at_batch_1000"] = mine["dropin_loop"]["value"] / got["batch_1000"]["value"]
    return got


def cpu_baseline(name, budget_s, batch, gpu_rows):
    """Reference CPU arithmetic on this node's host cores (rank 0, N=1): bounded sample of the same workload, every
    candidate thread count in its own process.  Reports (i) the in-process best and (ii) one thread x cores (how the
    reference pipeline runs: `parallel -j` single-threaded workers, scripts/clair3_c_impl_pipeline.py:229-266)."""
    import subprocess
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    # a container may see every core of the host but own only a CPU-time quota (cgroup v2 cpu.max: "quota period")
    quota = None
    try:
        q, per_us = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(int(q) / int(per_us))))
    except Exception:
        pass
    usable = min(cores, quota) if quota else cores
    cands = [1] + sorted({t for t in (16, 32, 64, 128) if t <= cores} | ({cores} if cores < 16 else set()))
    per = max(2.0, budget_s / len(cands))
    runs = []
    rows_path = os.path.join(tempfile.gettempdir(), f"c3_bench_cpu_rows_{os.getpid()}_{name}.npy")
    for th in cands:
        env = dict(os.environ, OMP_NUM_THREADS=str(th), MKL_NUM_THREADS=str(th))
        save = rows_path if th == cands[1 if len(cands) > 1 else 0] else "-"
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", name, str(th), str(per),
                                str(batch), save], env=env, capture_output=True, text=True, timeout=per * 6 + 180)
            runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
        except Exception as e:  # a failed candidate must not kill the benchmark line
            print(f"[bench] cpu worker threads={th} failed: {e!r}", file=sys.stderr)
        multi = [r for r in runs if r["threads"] > 1]
        if len(multi) >= 2 and multi[-1]["median_s"] > 1.5 * min(r["median_s"] for r in multi):
            break  # past the sweet spot: larger thread counts only get worse
    if not runs:
        return None, None
    rate = lambda r: r["batch"] / r["median_s"]
    best = max(runs, key=rate)
    one = next((r for r in runs if r["threads"] == 1), None)
    kind = "reference" if all(r.get("kind") == "reference" for r in runs) else "port"
    what = ("the reference's own clair3/model.py module in eval() called as clair3/CallVariantsFromCffi.py:48-52 _torch_predict does "
            "(staged copy oracle/_ref)") if kind == "reference" else "oracle/torch_port.py = the ATen/oneDNN operators the reference modules call"
    out = {"value": rate(best), "unit": "candidate-windows/s", "cores": best["threads"], "kind": kind,
           "sample": f"{best['reps']} x one batch of {best['batch']} windows, median, own process; {what}; in-process rates by thread count "
                     f"{ {r['threads']: round(rate(r)) for r in runs} } windows/s on {cores} visible cores; torch {best['torch']}",
           "ms_per_batch": 1e3 * best["median_s"], "host_cores_visible": cores, "cpu_quota_cores": quota}
    if one:
        out["per_core"] = {"value": rate(one), "unit": "candidate-windows/s/core", "threads": 1,
                           "sample": f"{one['reps']} x {one['batch']} windows, one thread"}
        out["per_core_x_cores"] = {"value": rate(one) * usable, "cores": usable, "cpu_quota_cores": quota,
                                   "note": "extrapolation: one-thread rate x usable cores = min(visible cores, cgroup cpu.max quota) -- "
                                           "the pipeline-like mode (single-threaded workers side by side); an upper bound, memory "
                                           "bandwidth and turbo make the real figure lower"}
    conc = None
    if gpu_rows is not None and os.path.exists(rows_path):
        y_cpu = np.load(rows_path)
        os.unlink(rows_path)
        if y_cpu.shape == gpu_rows.shape:
            heads = {"gt21": (0, 21), "zygosity": (21, 24)}
            conc = {"windows": int(len(y_cpu)), "max_abs_dy": float(np.abs(y_cpu.astype(np.float64) - gpu_rows).max()),
                    "vs": "cpu_baseline rows on the same batch (" + ("the reference's own modules" if kind == "reference" else
                                                                      "oracle/torch_port.py, pinned to the reference's goldens at 2e-6") + ")"}
            for k, (lo, hi) in heads.items():
                a, b = gpu_rows[:, lo:hi].argmax(1), y_cpu[:, lo:hi].argmax(1)
                top2 = np.sort(y_cpu[:, lo:hi], axis=1)[:, -2:]
                diff = a != b
                conc[k] = {"identical": int((~diff).sum()), "differ": int(diff.sum()),
                           "differ_outside_near_ties_1e-6": int((diff & ((top2[:, 1] - top2[:, 0]) > 1e-6)).sum())}
    return out, conc


def _r(v, nd=4):
    """numbers of the short line: enough digits to recompute, not fifteen"""
    if isinstance(v, float):
        return float(f"{v:.{nd + 2}g}")
    return v


def short_line(full, names, full_path):
    """The ONE line on stdout (<= 4 KB).  value, ms_per_step and roofline all come from the SAME leg: one batch in flight on one
    handle, windows resident in HBM (the bench contract); the SURVEY 8d host-to-host rate stands next to it as host_inclusive.
    Everything else measured is in `full` (gpurun_out/bench_full.json)."""
    def roof(r):
        if not r:
            return None
        out = {k: _r(r.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "mfma_util", "traffic",
                                         "avg_launch_us", "launches", "kernel_us_per_step", "step_us_one_batch_in_flight")}
        ev = r.get("events") or {}
        if ev.get("frac") is not None:  # the raw event-bracketed launch durations and the fraction they give (the share-based one is `frac`)
            out["events"] = {"family_us_per_step": _r(ev.get("family_us_per_step")), "frac": _r(ev.get("frac")), "avg_launch_us": _r(ev.get("avg_launch_us"))}
        out["whole_network_frac"] = _r(r.get("whole_network_frac", r.get("whole_forward_frac_one_in_flight")))
        out["fabric_frac_of_hbm_peak"] = _r(r.get("fabric_frac_of_hbm_peak", r.get("hbm_frac")))
        out["kernel"] = "Clair3_F 3x3 convolution launches (fa.conv* / fa.res*)" if "convolution" in r.get("kernel", "") else \
                        "Clair3_P BiLSTM launches (p.lstm* / p.proj*)"
        tn = r.get("traffic_note") or {}
        out["fabric_bytes_per_step"] = tn.get("fabric_bytes_per_step_all_kernels", tn.get("hbm_bytes_per_step_all_kernels"))
        out["algorithmic_bytes_per_step"] = tn.get("algorithmic_bytes_per_step")
        out["l2_hit_rate"] = _r(tn.get("l2_hit_rate"))
        src = tn.get("source") or ""
        out["traffic_source"] = (src.split(":")[0] + ": L2 memory-side requests (FETCH_SIZE x2 + WRITE_SIZE), Infinity-Cache hits included") if src else None
        return out

    def cpu(c):
        if not c:
            return None
        out = {"value": _r(c["value"]), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
               "sample": c["sample"].split(";")[0] + "; the reference's modules called as _torch_predict does" if c["kind"] == "reference" else c["sample"].split(";")[0]}
        if "per_core" in c:
            out["one_thread"] = _r(c["per_core"]["value"])
        if "per_core_x_cores" in c:
            out["one_thread_x_usable_cores"] = {"value": _r(c["per_core_x_cores"]["value"]), "cores": c["per_core_x_cores"]["cores"]}
        return out

    def host(hl):
        if not hl:
            return None
        out = {"value": _r(hl["value"]), "unit": "candidate-windows/s", "batch": hl["batch"], "slots_in_flight": hl["slots_in_flight"],
               "frac_of_device_resident": _r(hl["frac_of_device_resident_one_in_flight"]), "steps": hl.get("steps")}
        if hl.get("passes"):
            out["median_of_passes"] = len(hl["passes"])  # every host leg: the median of that many back-to-back passes (all of them in the full record)
        if hl.get("at_driver_steps"):
            out["at_driver_steps"] = {"value": _r(hl["at_driver_steps"]["value"]), "steps": hl["at_driver_steps"]["steps"]}
        b = hl.get("batch_1000")
        if b:
            out["batch_1000"] = {"ring": _r(b["value"]), "device_resident": _r(b["device_resident_one_in_flight"]),
                                 "blocking_call": _r(b.get("sync_call", {}).get("value")),
                                 "dropin_loop": _r(b.get("dropin_loop", {}).get("value"))}
        return out

    def conc(g):
        if not g:
            return None
        return {"windows": g["windows"], "max_abs_dy": _r(g["max_abs_dy"]), "gt21_differ": g["gt21"]["differ"],
                "zygosity_differ": g["zygosity"]["differ"],
                "differ_outside_near_ties": sum(g[h].get("differ_outside_near_ties_1e-6", g[h].get("differ_outside_near_ties_1e-5", 0)) for h in ("gt21", "zygosity"))}

    def sub(r, S):
        o = {"value": _r(r["one_batch_in_flight"]["value"]), "ms_per_step": _r(r["one_batch_in_flight"]["ms_per_step"]),
             "batch_per_gpu": r["config"]["batch_per_gpu"], "batches_in_flight": 1,
             "value_stats": {k: _r(v) for k, v in r["one_batch_in_flight"].items() if k in ("median", "min", "max", "repeats")}}
        k = next((k for k in r if k.endswith("_batches_in_flight") and k != "one_batch_in_flight"), None)
        if k:
            o[k] = _r(r[k]["value"])
        if r.get("roofline"):
            o["roofline"] = roof(r["roofline"])
        if r.get("host_inclusive"):
            o["host_inclusive"] = host(r["host_inclusive"])
        if r.get("cpu_baseline"):
            o["cpu_baseline"] = cpu(r["cpu_baseline"])
        if r.get("gt_concordance"):
            o["gt_concordance"] = conc(r["gt_concordance"])
        rg = (r.get("reference_on_this_gpu") or {}).get("batch_1000")
        if rg:
            o["reference_modules_on_this_gpu_batch_1000"] = _r(rg["value"])
        if r.get("multi_gpu"):
            mg = r["multi_gpu"]
            o["multi_gpu"] = {k: mg.get(k) for k in ("gather", "rccl_ranks_seen", "fallback_reason", "ranks")}
            o["multi_gpu"]["per_rank_windows_per_s"] = [_r(v) for v in (mg.get("per_rank_windows_per_s") or [])]
        o["range_flag_raised"] = r["range_flag_raised"]
        o["on_fp32_fallback"] = r["on_fp32_fallback"]
        return o

    head = sub(full, 0)
    line = {"metric": full["metric"], "value": head.pop("value"), "unit": full["unit"], "n_gpus": full["n_gpus"], "steps": full["steps"],
            "warmup": full["warmup"], "ms_per_step": head.pop("ms_per_step"), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": full["dtype"], "data": "synthetic",
            "config": {"workload": full["config"]["workload"], "batch_per_gpu": full["config"]["batch_per_gpu"],
                       "windows_per_step": full["config"]["windows_per_step"], "batches_in_flight": 1,
                       "inputs": "resident in HBM; host_inclusive = pageable host windows in, rows in host memory out"}}
    head.pop("batch_per_gpu"), head.pop("batches_in_flight")
    line.update(head)
    for n in names[1:]:  # the other workloads: the same leg, their numbers only
        if n in full:
            o = sub(full[n], 0)
            c = {"value": o["value"], "ms_per_step": o["ms_per_step"], "batch_per_gpu": o["batch_per_gpu"], "batches_in_flight": 1}
            if o.get("roofline"):
                c["roofline"] = {k: o["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "mfma_util", "kernel_us_per_step",
                                                               "fabric_bytes_per_step")}
            if o.get("host_inclusive"):
                c["host_inclusive"] = o["host_inclusive"]["value"]
            if o.get("cpu_baseline"):
                c["cpu_baseline"] = {k: o["cpu_baseline"][k] for k in ("value", "cores", "kind")}
            if o.get("gt_concordance"):
                c["gt_concordance"] = o["gt_concordance"]
            line[n] = c
    line["full_record"] = os.path.relpath(full_path, ROOT) if full_path else None
    return line


PLACEMENT = {}  # dist.pin_to_device_numa's report of this rank (N > 1), printed in the full record


def self_launch(n):
    """`python bench.py --gpus N` typed plainly, N > 1: no torchrun environment around this process, so it starts its own N
    ranks -- exactly the command the module docstring names (one process per GPU, rendezvous on 127.0.0.1 at a free port) with
    this process's own arguments -- and returns their exit code.  Only rank 0 writes to stdout (its ONE JSON line), which the
    children inherit; torchrun's chatter goes to stderr."""
    import socket
    import subprocess
    if not os.environ.get("C3_BENCH_DEVICE"):  # (C3_BENCH_DEVICE: every rank on one device -- the one-GPU code-path test)
        try:
            from clair3_amd import _lib
            have = _lib.device_count()
        except Exception as e:
            print(f"[bench] --gpus {n}: cannot count HIP devices ({e!r})", file=sys.stderr)
            return 2
        if have < n:
            print(f"[bench] --gpus {n} but this node shows {have} HIP device(s)", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="all", choices=["all"] + list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch (parity/experiments only)")
    ap.add_argument("--streams", type=int, default=0,
                    help="batches kept in flight per GPU (model handles x HIP streams); 0 = 3")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-leg", action="store_true")
    ap.add_argument("--no-profiled-pass", action="store_true", help="skip the HIP-event pass (for rocprofv3 runs of one leg)")
    ap.add_argument("--no-reference-gpu", action="store_true", help="skip the leg that times the reference's own modules on the GPU through PyTorch-ROCm")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of K steps each; value = the median block")
    ap.add_argument("--gather-timeout", type=float, default=20.0, help="seconds RCCL's rendezvous / first gather may take before the rows go through torch.distributed (N > 1)")
    ap.add_argument("--torch-gather", action="store_true", help="N > 1: gather through torch.distributed from the start")
    ap.add_argument("--ref-gpu-worker", nargs=3, metavar=("WORKLOAD", "BUDGET", "BATCH"), help=argparse.SUPPRESS)
    ap.add_argument("--cpu-worker", nargs=5, metavar=("WORKLOAD", "THREADS", "BUDGET", "BATCH", "ROWS"), help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_worker:
        cpu_worker(args.cpu_worker[0], int(args.cpu_worker[1]), float(args.cpu_worker[2]), int(args.cpu_worker[3]),
                   args.cpu_worker[4])
        return
    if args.ref_gpu_worker:
        ref_gpu_worker(args.ref_gpu_worker[0], float(args.ref_gpu_worker[1]), int(args.ref_gpu_worker[2]))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))

    from clair3_amd import dist as c3dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not os.environ.get("C3_BENCH_DEVICE"):
        # preflight of a rank (the driver's torch.distributed.run launch; self_launch checks before it starts anything): one process per
        # GPU needs that many devices -- say so in one sentence and stop with rc 2 instead of a traceback out of set_device per rank
        from clair3_amd import _lib
        lrank = int(os.environ.get("LOCAL_RANK", "0"))
        err = c3dist.preflight(int(os.environ.get("LOCAL_WORLD_SIZE", os.environ["WORLD_SIZE"])), _lib.device_count(), lrank)
        if err:
            if lrank == 0:
                print(f"[bench] {err}", file=sys.stderr)
            sys.exit(2)
        # every rank's host side on the NUMA node of its GPU, before the runtime and the library start their threads (N > 1 only: the
        # one-GPU line keeps the whole host for its cpu_baseline leg)
        PLACEMENT.update(c3dist.pin_to_device_numa(lrank))
    rank, world, local = c3dist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}: launch with torch.distributed.run "
                  f"--nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)

    names = ["full_alignment", "pileup", "full_alignment_dwell"] if args.workload == "all" else [args.workload]
    results = {n: run_workload(n, args, rank, world, local) for n in names}
    head = results[names[0]]

    if rank == 0:
        def sub_line(n, r, budget):
            rows = r.pop("rows_rank0", None)
            sub = {"value": r["value"], "unit": "candidate-windows/s", "ms_per_step": r["ms_per_step"],
                   "batches_in_flight": r["batches_in_flight"], "one_batch_in_flight": r["one_batch_in_flight"],
                   **{k: v for k, v in r.items() if k.endswith("_batches_in_flight") and k != "one_batch_in_flight"},
                   "config": {"workload": r["workload"], "batch_per_gpu": r["batch_per_gpu"]},
                   "value_stats": r["value_stats"], "kernel_variants": r["kernel_variants"],
                   "range_flag_raised": r["range_flag_raised"], "on_fp32_fallback": r["on_fp32_fallback"]}
            for key in ("roofline", "kernels", "multi_gpu"):
                if key in r:
                    sub[key] = r[key]
            if "host_inclusive" in r:
                sub["host_inclusive"] = r["host_inclusive"]
            if world == 1 and not args.no_reference_gpu and budget > 0:
                sub["reference_on_this_gpu"] = reference_gpu(n, min(budget, 15.0), r)
            if world == 1 and not args.no_cpu_baseline and budget > 0:
                sub["cpu_baseline"], sub["gt_concordance"] = cpu_baseline(n, budget, args.batch, rows)
                if sub["cpu_baseline"]:
                    sub["speedup_vs_cpu_baseline"] = {
                        "device_resident_vs_in_process_best": r["value"] / sub["cpu_baseline"]["value"],
                        "host_inclusive_vs_in_process_best": (r["host_inclusive"]["value"] / sub["cpu_baseline"]["value"]) if "host_inclusive" in r else None,
                        "device_resident_vs_per_core_x_cores": (r["value"] / sub["cpu_baseline"]["per_core_x_cores"]["value"]) if "per_core_x_cores" in sub["cpu_baseline"] else None,
                    }
            return sub

        h = sub_line(names[0], head, args.cpu_budget)
        full = {
            "metric": "candidate-windows/sec", "value": h["one_batch_in_flight"]["value"], "unit": "candidate-windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "repeats": args.repeats,
            "ms_per_step": h["one_batch_in_flight"]["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32(fp16x3)", "data": "synthetic",
            "dtype_note": "fp32 storage, accumulation and results; every fp32 product formed from two fp16 pieces per operand (three 16-bit MFMA products, DESIGN.md 1)",
            "config": {"workload": head["workload"], "batch_per_gpu": head["batch_per_gpu"],
                       "windows_per_step": head["windows_per_step"], "weights": "seeded random (no checkpoints offline)",
                       "sharding": f"windows x{world}, rows gathered to rank 0 every {GATHER_EVERY} steps (one gather: {head.get('multi_gpu', {}).get('gather')})" if world > 1 else "single GPU",
                       "inputs": "resident in HBM before the timed region (host-to-host rate: host_inclusive)", "batches_in_flight": 1},
            **{k: v for k, v in h.items() if k not in ("value", "unit", "ms_per_step", "config", "batches_in_flight")},
        }
        for n in names[1:]:
            full[n] = sub_line(n, results[n], args.cpu_budget / 2 if n == "pileup" else 0)
        if world > 1:
            full["placement_rank0"] = PLACEMENT  # dist.pin_to_device_numa: which NUMA node / how many CPUs rank 0's host side got
        # everything measured goes to a file; stdout carries ONE short line the driver can parse (round 3's 21 KB line could not be)
        full_path = os.environ.get("C3_BENCH_FULL") or os.path.join(ROOT, "gpurun_out", "bench_full.json")
        try:
            os.makedirs(os.path.dirname(full_path), exist_ok=True)
            with open(full_path, "w") as fh:
                json.dump(full, fh, indent=1)
        except OSError as e:
            print(f"[bench] could not write {full_path}: {e!r}", file=sys.stderr)
            full_path = None
        print(json.dumps(short_line(full, names, full_path)), flush=True)

    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
