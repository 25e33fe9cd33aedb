"""End to end through the reference's own stage-B worker command (clair3.py CallVariantsFromCffi --use_gpu True ...: tensor files ->
model -> decode processes -> VCF) on libc3hip, with and without the decoder columns, next to the same command on the reference's
modules through PyTorch (GPU if torch sees one, as --use_gpu does there).  Needs an MI355X and oracle/_ref.
python tests/diag/worker_throughput.py [windows per file] [files] [cpu_threads]"""
import atexit
import json
import os
import re
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clair3_amd import synthetic as syn  # noqa: E402
from tests import refloop  # noqa: E402

per_file = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
files = int(sys.argv[2]) if len(sys.argv) > 2 else 6
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ref = refloop.reference_root()
out = {}
only = os.environ.get("C3_WT_ONLY")  # "pileup" / "full_alignment": one of the two jobs
extra_args = os.environ.get("C3_WT_EXTRA_ARGS", "").split()  # e.g. "--gvcf True": appended to every worker command
legs = os.environ.get("C3_WT_LEGS")  # e.g. "libc3hip_decoder_columns": these legs alone, timing only (no VCF comparison)
for name, kind, channels, indel, pileup in (("full_alignment", syn.FULL_ALIGNMENT, 8, True, False), ("pileup", syn.PILEUP, 18, False, True)):
    if only and only != name:
        continue
    d = tempfile.mkdtemp(prefix="c3_worker_")
    atexit.register(shutil.rmtree, d, ignore_errors=True)  # 5.6 GB of tensor files per 240 k-window job: a sweep that leaves them fills /tmp
    n = per_file * files * (3 if pileup else 1)
    lst = refloop.write_job(d, kind, [per_file * (3 if pileup else 1)] * files, channels=channels)
    ck = os.path.join(d, "model")
    refloop.write_checkpoint(ck + ".pt", kind, channels, indel)
    res = {}
    # (round 6: a worker process on libc3hip no longer imports torch -- clair3_amd/lazy_torch.py, ptfile.py; the *_torch_imported leg is
    # the same command with C3HIP_LAZY_TORCH=0 C3HIP_PTFILE=0, i.e. the process as it was)
    for tag, kw in (("libc3hip", dict(hip=True)), ("libc3hip_decoder_columns", dict(hip=True, decoder=True)),
                    ("libc3hip_decoder_columns_torch_imported", dict(hip=True, decoder=True, extra_env={"C3HIP_LAZY_TORCH": "0", "C3HIP_PTFILE": "0"})),
                    ("reference_modules_pytorch", dict(hip=False))):
        if legs and tag not in legs.split(","):
            continue
        vcf = os.path.join(d, tag + ".vcf")
        report = os.path.join(d, tag + ".torch.json")
        kw = dict(kw, extra_env=dict(kw.get("extra_env") or {}, C3HIP_LAZY_TORCH_REPORT=report))
        t0 = time.perf_counter()
        rc, log = refloop.run_worker(ref, lst, ck, vcf, pileup, indel, cpu_threads=threads, extra_args=extra_args, **kw)
        wall = time.perf_counter() - t0
        m = re.search(r"Total time elapsed: ([0-9.]+) s", log)
        assert rc == 0 and f"Total processed positions : {n}" in log, log[-2000:]
        res[tag] = {"loop_seconds": float(m.group(1)), "process_wall_seconds": round(wall, 2), "windows_per_s_in_the_loop": round(n / float(m.group(1))),
                    "windows_per_s_of_the_process": round(n / wall)}
        if os.path.exists(report):
            res[tag]["torch_imported"] = json.load(open(report))["real_loaded"]
    if legs:
        out[name] = {"windows": n, "tensor_files": files, "cpu_threads": threads, **res}
        print(name, json.dumps(out[name]), flush=True)
        continue
    # libc3hip's VCF against the reference modules' (PyTorch): identical text, rows whose QUAL / GQ differ in the last digit
    # (probabilities that agree to ~1e-6 on either side of a rounding boundary) and rows whose CALL differs -- each of those
    # listed, so that a near-tie can be told from a defect (tests/test_reference_loop_gpu.py proves near-ties record by record)
    cmp_ = refloop.compare_vcfs(os.path.join(d, "libc3hip.vcf"), os.path.join(d, "reference_modules_pytorch.vcf"))
    res["vcf_identical"] = cmp_["identical_text"]
    res["vcf_records"] = [cmp_["records_a"], cmp_["records_b"]]
    res["vcf_qual_last_digit_only"] = cmp_["qual_only"]
    res["vcf_max_qual_diff"] = cmp_["max_qual_diff"]
    res["vcf_call_differs"] = [[list(k), a, b] for k, a, b in cmp_["call_differs"][:20]]
    res["vcf_only_in_one"] = [cmp_["only_a"], cmp_["only_b"]]
    if cmp_["call_differs"]:
        # every differing call must be a near-tie of the reference's own joint outcome probabilities -- the proof of
        # tests/test_reference_loop_gpu.py (reference module on the CPU and the library on the GPU on the loop's own batch)
        from tests.test_reference_loop_gpu import explain_call_differences, file_window
        job = dict(ck=ck, kind=kind, channels=channels, indel=indel, window=file_window(d, [per_file * (3 if pileup else 1)] * files))
        res["vcf_call_differs_explained"] = explain_call_differences(job, cmp_["call_differs"], ref)
    if "libc3hip_decoder_columns_torch_imported" in res:
        cmp3 = refloop.compare_vcfs(os.path.join(d, "libc3hip_decoder_columns_torch_imported.vcf"), os.path.join(d, "libc3hip_decoder_columns.vcf"))
        res["vcf_torch_imported_vs_not_identical"] = [cmp3["identical_text"], cmp3["records_a"], cmp3["records_b"]]
    if "libc3hip_decoder_columns" in res:
        cmp2 = refloop.compare_vcfs(os.path.join(d, "libc3hip_decoder_columns.vcf"), os.path.join(d, "libc3hip.vcf"))
        res["vcf_decoder_columns_vs_plain_identical"] = [cmp2["identical_text"], cmp2["records_a"], cmp2["records_b"]]
    out[name] = {"windows": n, "tensor_files": files, "cpu_threads": threads, **res}
    print(name, json.dumps(out[name]), flush=True)
print(json.dumps(out))
