"""The reference's UNMODIFIED GPU worker -- stage B of clair3/CallVariantsFromCffiGPU.py:163-199,289-318, i.e.
``python clair3.py CallVariantsFromCffi --use_gpu True --gpu_id G --cpu_threads N --output_tensor_can_fn_list LIST`` with its
loop at clair3/CallVariantsFromCffi.py:196-353 (model created before the ProcessPoolExecutor, one blocking _torch_predict
per batch of 1000, forked decode workers, shared-memory hand-off of the rows, VCF text) -- against the REAL libc3hip on an
MI355X, in its own process, next to the same command on the reference's own modules on the CPU.

The reference modules come from the git-ignored oracle/_ref/ (staged by oracle/stage_reference.py from __graft_entry__.build();
the GPU box has no /root/reference).  This is the test of everything no stand-in can show: HIP initialised before the fork of
the decode workers, page-locking of numpy pages, the staging threads, CUDA_VISIBLE_DEVICES set by --gpu_id after the library
is loaded, the rebound batch generator running two batches ahead of the loop, decoder columns through shared memory."""
import json
import os
import re

import pytest

from clair3_amd import synthetic as syn
from tests import refloop

pytestmark = pytest.mark.gpu

CASES = {
    # name: (kind, channels, indel heads, pileup flag, dwell flag, windows per tensor file)
    "full_alignment": (syn.FULL_ALIGNMENT, 8, True, False, False, [2300, 1000, 37]),
    "pileup": (syn.PILEUP, 18, False, True, False, [5000, 1200, 1]),
    "dwell": (syn.FULL_ALIGNMENT, 9, True, False, True, [1100]),
}


@pytest.fixture(scope="module")
def ref():
    root = refloop.reference_root()
    if root is None:
        pytest.skip("no reference modules: run tools/stage_reference.sh in the build container (oracle/_ref travels with the snapshot)")
    return root


def file_window(directory, sizes):
    """job-wide window index -> (the loop's batch that holds the window, its index in it, reference base): refloop.write_job numbers positions over the whole job and
    tests/test_decode_dropin.alt_infos puts "ACGT"[j % 4] (j = index within the file) at the window's centre"""
    import numpy as np

    def get(g, batch=1000):
        for i, n in enumerate(sizes):
            if g < n:
                lo = g // batch * batch  # the loop's batches of 1000 never span files (clair3/CallVariantsFromCffi.py:106-148)
                xb = np.array(np.load(os.path.join(directory, f"t{i}.npy"), mmap_mode="r")[lo:lo + batch])
                return xb, g - lo, "ACGT"[g % 4]
            g -= n
        raise IndexError(g)
    return get


def elapsed(out):
    m = re.search(r"Total time elapsed: ([0-9.]+) s", out)
    return float(m.group(1)) if m else None


@pytest.fixture(scope="module")
def jobs(tmp_path_factory, ref):
    """tensor files, checkpoint and the reference's own CPU run of every case (made once)"""
    made = {}

    def get(name):
        if name in made:
            return made[name]
        kind, channels, indel, pileup, dwell, sizes = CASES[name]
        d = str(tmp_path_factory.mktemp(name))
        lst = refloop.write_job(d, kind, sizes, channels=channels)
        ck = os.path.join(d, "model")  # the loader appends .pt (clair3/CallVariantsFromCffi.py:21-22)
        refloop.write_checkpoint(ck + ".pt", kind, channels, indel)
        want = os.path.join(d, "reference_cpu.vcf")
        rc, out = refloop.run_worker(ref, lst, ck, want, pileup, indel, dwell=dwell, hip=False)
        assert rc == 0, out[-3000:]
        assert f"Total processed positions : {sum(sizes)}" in out, out[-3000:]
        made[name] = dict(dir=d, lst=lst, ck=ck, want=want, n=sum(sizes), ref_s=elapsed(out), kind=kind, channels=channels,
                          indel=indel, window=file_window(d, sizes))
        return made[name]
    return get


def explain_call_differences(job, diffs, ref_root):
    """a differing call is accepted only with proof that it is a near-tie.  For every differing record: find
    its window in the job's tensors, compute the reference's fp32 row (its own module on the CPU) and the library's row on the
    GPU, enumerate the joint outcome probabilities the reference decoder ranks (clair3/CallVariants.py:510-660) for both rows
    and require that (i) the rows are within the 1e-4 gate, (ii) the two rows rank at least one pair of outcomes differently --
    otherwise the rows cannot explain the difference -- and (iii) every pair they rank differently is closer than 1e-6 in the
    REFERENCE's own row (:722-751 picks the largest).  Returns the explanations (also printed)."""
    import numpy as np
    import torch
    from tests import refmodels
    from tests.test_parity_gpu import make_model
    kind, channels, indel = job["kind"], job["channels"], job["indel"]
    sd = {k: v.numpy() for k, v in torch.load(job["ck"] + ".pt", map_location="cpu").items()}
    m_ref = refmodels.reference_model(ref_root, kind, sd, indel, channels)
    m_hip = make_model(kind, channels, indel, sd)
    notes = []
    for (chrom, pos), a, b in diffs:
        g = (int(pos) - 5000) // 41
        assert 5000 + 41 * g == int(pos) and chrom == f"chr{1 + g % 3}", (chrom, pos)
        xb, j, base = job["window"](g)  # the same batch the worker's loop formed: the CPU kernels ATen picks depend on its size
        y_ref = refmodels.reference_rows(m_ref, xb)[j]
        y_hip = m_hip.predict_numpy(np.ascontiguousarray(xb))[j]
        p_ref = refmodels.outcome_probabilities(ref_root, y_ref, base, indel)
        p_hip = refmodels.outcome_probabilities(ref_root, y_hip, base, indel)
        inv = refmodels.order_inversions(p_ref, p_hip)
        top2 = np.sort(p_ref)[-2:]
        note = {"record": f"{chrom}:{pos}", "window": g, "libc3hip": a, "reference_cpu": b,  # compare_vcfs(got, want): a = the library's run
                "max_abs_dy": float(np.abs(y_ref.astype(np.float64) - y_hip).max()),
                "reference_top2_joint": [float(top2[1]), float(top2[0])], "reference_top2_gap": float(top2[1] - top2[0]),
                "pairs_ranked_differently": len(inv), "largest_reference_gap_among_them": max([g_ for _, _, g_ in inv], default=None)}
        print("CALL DIFFERS:", json.dumps(note, default=str))
        notes.append(note)
        assert note["max_abs_dy"] <= 1e-4, note
        # (the worker's CPU run may have used another thread count than this re-evaluation: a tie of the two LARGEST outcomes
        # within 1e-6 explains the record as well)
        assert inv or note["reference_top2_gap"] <= 1e-6, f"rows rank every outcome alike yet the call differs -- not a near-tie: {note}"
        assert (note["largest_reference_gap_among_them"] or 0.0) <= 1e-6, f"outcomes further apart than 1e-6 ranked differently: {note}"
    return notes


def check(name, job, got_vcf, out, tag, min_share=0.5):
    s = refloop.compare_vcfs(got_vcf, job["want"])
    s.update(case=name, run=tag, windows=job["n"], loop_seconds=elapsed(out), reference_cpu_loop_seconds=job["ref_s"])
    assert s["records_a"] == s["records_b"] and s["records_a"] >= int(job["n"] * min_share), s
    assert not s["only_a"] and not s["only_b"], s
    # the same calls.  QUAL is printed with two decimals from a log of probabilities that agree to ~1e-6, so a handful of rows may
    # differ in the last digit (counted in qual_only).  A different CALL must be proven a near-tie of the reference's own joint
    # outcome probabilities, record by record -- no allowance by count
    s["near_ties"] = explain_call_differences(job, s["call_differs"], refloop.reference_root()) if s["call_differs"] else []
    os.makedirs(os.path.join(refloop.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(refloop.ROOT, "gpurun_out", f"ref_loop_{name}_{tag}.json"), "w") as f:
        json.dump(s, f, default=str)
    print(json.dumps(s, default=str))
    assert len(s["near_ties"]) == len(s["call_differs"])
    assert s["identical_text"] + s["qual_only"] >= s["records_a"] - len(s["call_differs"])
    assert s["identical_text"] >= 0.98 * s["records_a"], s
    return s


@pytest.mark.parametrize("name", ["full_alignment", "pileup", "dwell"])
def test_unmodified_worker_command_on_libc3hip(name, ref, jobs):
    kind, channels, indel, pileup, dwell, sizes = CASES[name]
    job = jobs(name)
    got = os.path.join(job["dir"], "hip.vcf")
    report = os.path.join(job["dir"], "torch_report.json")
    rc, out = refloop.run_worker(ref, job["lst"], job["ck"], got, pileup, indel, dwell=dwell, hip=True, extra_env={"C3HIP_LAZY_TORCH_REPORT": report})
    assert rc == 0, out[-3000:]
    assert "tensor_generator_for_chunk" in out  # install() ran in that process
    assert f"Total processed positions : {sum(sizes)}" in out, out[-3000:]
    check(name, job, got, out, "hip")
    # round 6: the whole worker process -- argument parsing, the loader (clair3_amd/ptfile.py), the loop, its decode pool -- ran without
    # ever importing torch (clair3_amd/lazy_torch.py: a stand-in answered Run()'s thread counts and torch.device("cpu"))
    assert "torch is imported on first use" in out
    assert json.load(open(report)) == {"installed": True, "real_loaded": False, "first_touch": None}


def test_the_worker_with_torch_imported_up_front_prints_the_same_vcf(ref, jobs):
    """C3HIP_LAZY_TORCH=0 C3HIP_PTFILE=0 (the process as it was before round 6: torch imported by the reference's modules, the checkpoint
    read by torch.load): character for character the VCF of the default run"""
    job = jobs("full_alignment")
    got, report = os.path.join(job["dir"], "hip_eager.vcf"), os.path.join(job["dir"], "torch_report_eager.json")
    rc, out = refloop.run_worker(ref, job["lst"], job["ck"], got, False, True, hip=True,
                                 extra_env={"C3HIP_LAZY_TORCH": "0", "C3HIP_PTFILE": "0", "C3HIP_LAZY_TORCH_REPORT": report})
    assert rc == 0, out[-3000:]
    assert "torch is imported on first use" not in out and not os.path.exists(report)  # no stand-in: nothing to report
    check("full_alignment", job, got, out, "hip_eager")
    plain = os.path.join(job["dir"], "hip.vcf")
    if os.path.exists(plain):
        t = refloop.compare_vcfs(got, plain)
        assert t["identical_text"] == t["records_a"] == t["records_b"], t


def test_blocking_calls_without_the_lookahead_generator(ref, jobs):
    """C3HIP_PREFETCH_DEPTH=0: the reference's own generator and one blocking c3_predict per batch (chunks through the ring, every
    piece through the staging buffer) -- same VCF"""
    job = jobs("full_alignment")
    got = os.path.join(job["dir"], "hip_blocking.vcf")
    rc, out = refloop.run_worker(ref, job["lst"], job["ck"], got, False, True, hip=True, extra_env={"C3HIP_PREFETCH_DEPTH": "0"})
    assert rc == 0, out[-3000:]
    check("full_alignment", job, got, out, "hip_blocking")


@pytest.mark.parametrize("name", ["full_alignment", "pileup"])
def test_decoder_columns_through_the_reference_loop(name, ref, jobs):
    """install(decoder=True): 121-column (full alignment) / 55-column (pileup, 24 probabilities) rows through the loop's shared memory
    into its forked decode workers, the rebound batch_output reading the device's decoder columns there -- same VCF text as the
    reference's enumeration"""
    kind, channels, indel, pileup, dwell, sizes = CASES[name]
    job = jobs(name)
    got = os.path.join(job["dir"], "hip_decoder.vcf")
    rc, out = refloop.run_worker(ref, job["lst"], job["ck"], got, pileup, indel, hip=True, decoder=True)
    assert rc == 0, out[-3000:]
    s = check(name, job, got, out, "hip_decoder")
    # and against the run without the columns on the same device rows: character for character
    plain = os.path.join(job["dir"], "hip.vcf")
    if os.path.exists(plain):
        t = refloop.compare_vcfs(got, plain)
        assert t["identical_text"] == t["records_a"] == t["records_b"], t
    assert s["records_a"] > 0


@pytest.mark.parametrize("name,flags", [("full_alignment", ["--gvcf", "True"]), ("pileup", ["--gvcf", "True"]),
                                        ("pileup", ["--haploid_sensitive", "--gvcf", "True", "--qual", "8"]), ("full_alignment", ["--haploid_precise"]),
                                        ("full_alignment", ["--enable_long_indel", "True"])])
def test_gvcf_and_haploid_rows_through_the_worker_command(name, flags, ref, jobs):
    """--gvcf True (rows carry the PL field, clair3/CallVariants.py:1360-1378), the haploid modes (:1191-1199, :1327-1329) and --enable_long_indel
    (lookups up to 100 000 bases, get_long_indel_read_count :383-402): the worker
    command on libc3hip with the decoder columns -- c3_vcf_rows restates compute_PL (:1397-1454) from the row's own probabilities and the
    haploid rules -- against the same command on the reference's modules"""
    kind, channels, indel, pileup, dwell, sizes = CASES[name]
    job = dict(jobs(name))
    tag = "hip" + "".join(f.strip("-").replace("haploid_", "_h").replace("enable_long_indel", "_longindel").replace("True", "") for f in flags if not f.isdigit())
    want, got = os.path.join(job["dir"], f"reference_cpu_{tag}.vcf"), os.path.join(job["dir"], f"{tag}.vcf")
    rc, out = refloop.run_worker(ref, job["lst"], job["ck"], want, pileup, indel, hip=False, extra_args=flags)
    assert rc == 0 and f"Total processed positions : {sum(sizes)}" in out, out[-3000:]
    job["want"] = want
    rc, out = refloop.run_worker(ref, job["lst"], job["ck"], got, pileup, indel, hip=True, decoder=True, extra_args=flags)
    assert rc == 0, out[-3000:]
    s = check(name, job, got, out, tag, min_share=0.02 if any("haploid" in f for f in flags) else 0.5)  # (a haploid run prints no heterozygous call)
    rows = [r for r in open(got) if r.strip() and not r.startswith("#")]
    if "--gvcf" in flags:
        assert rows and all(r.split("\t")[8] == "GT:GQ:DP:AD:AF:PL" and r.rstrip("\n").split(":")[-1].replace(",", "").isdigit() for r in rows)
    if any("haploid" in f for f in flags):
        assert rows and all(r.split("\t")[9].split(":")[0] in ("0", "1") for r in rows)
    # the PL field is part of the compared text: identical rows are identical in it (qual_only rows may differ in a PL by one as well)
    assert s["identical_text"] >= 0.98 * s["records_a"], s


def test_gpu_wrapper_slot_probe_on_the_device(ref):
    """CallVariantsFromCffiGPU.check_gpu_memory after install(): hipMemGetInfo instead of nvidia-smi, one slot per MI355X"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, sys.argv[1]); from clair3_amd import callvar; callvar.install(worker=False);"
            "import clair3.CallVariantsFromCffiGPU as g; print('SLOTS', g.check_gpu_memory(8000, None), g.get_gpu_memory(0))")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([refloop.ROOT, refloop.STUBS]))
    r = subprocess.run([sys.executable, "-c", code, ref], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"SLOTS \[([0-9, ]*)\] \[(\d+)\]", r.stdout)
    assert m, r.stdout
    slots = [int(v) for v in m.group(1).split(",") if v.strip()]
    assert slots == sorted(set(slots)) and len(slots) >= 1  # one slot per visible device
    assert int(m.group(2)) > 100_000  # MB free on an MI355X
    assert "would start" in r.stdout


@pytest.mark.parametrize("name,decoder", [("pileup", False), ("pileup", True), ("full_alignment", False), ("full_alignment", True)])
def test_legacy_stdin_worker_on_libc3hip(name, decoder, ref, tmp_path):
    """the reference's other worker, ``clair3.py CallVariants --tensor_fn PIPE`` (clair3/CallVariants.py:1456-1621: text
    tensors on stdin, int32 pileup windows, batches of predictBatchSize, decode on a thread beside the next load): install()
    rebinds its twins of the four model-call functions as well -- same VCF as the same command on the reference's CPU modules"""
    kind, channels, indel, pileup, _, _ = CASES[name]
    n = 900 if pileup else 260
    d = str(tmp_path)
    txt = os.path.join(d, "tensors.txt")
    refloop.write_pipe_tensors(txt, kind, n, channels=channels)
    ck = os.path.join(d, "model")
    refloop.write_checkpoint(ck + ".pt", kind, channels, indel)
    want, got = os.path.join(d, "reference_cpu.vcf"), os.path.join(d, "hip.vcf")
    rc, out = refloop.run_legacy_worker(ref, txt, ck, want, pileup, indel, hip=False)
    assert rc == 0 and f"Total processed positions in None : {n}" in out, out[-3000:]
    report = os.path.join(d, "torch_report.json")
    rc, out = refloop.run_legacy_worker(ref, txt, ck, got, pileup, indel, hip=True, decoder=decoder, extra_env={"C3HIP_LAZY_TORCH_REPORT": report})
    assert rc == 0 and f"Total processed positions in None : {n}" in out, out[-3000:]
    assert json.load(open(report)) == {"installed": True, "real_loaded": False, "first_touch": None}  # this worker too never imports torch
    assert "clair3.CallVariants._torch_predict" in out  # run_reference lists what install() rebound
    x_all = syn.make_windows(kind, n, seed=40, channels=channels)  # what refloop.write_pipe_tensors wrote
    job = dict(want=want, n=n, ref_s=None, ck=ck, kind=kind, channels=channels, indel=indel,
               window=lambda g: (x_all[g:g + 1], 0, "ACGT"[g % 4]))
    check(name, job, got, out, "legacy_decoder" if decoder else "legacy")
