"""End to end through the reference's own stage-B worker command (clair3.py CallVariantsFromCffi --use_gpu True ...: tensor files ->
model -> decode processes -> VCF) on libc3hip, with and without the decoder columns, next to the same command on the reference's
modules through PyTorch (GPU if torch sees one, as --use_gpu does there).  Needs an MI355X and oracle/_ref.
python tests/diag/worker_throughput.py [windows per file] [files] [cpu_threads]"""
import json
import os
import re
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
for name, kind, channels, indel, pileup in (("full_alignment", syn.FULL_ALIGNMENT, 8, True, False), ("pileup", syn.PILEUP, 18, False, True)):
    d = tempfile.mkdtemp(prefix="c3_worker_")
    n = per_file * files * (3 if pileup else 1)
    lst = refloop.write_job(d, kind, [per_file * (3 if pileup else 1)] * files, channels=channels)
    ck = os.path.join(d, "model")
    refloop.write_checkpoint(ck + ".pt", kind, channels, indel)
    res = {}
    for tag, kw in (("libc3hip", dict(hip=True)), ("libc3hip_decoder_columns", dict(hip=True, decoder=True)), ("reference_modules_pytorch", dict(hip=False))):
        if tag == "libc3hip_decoder_columns" and not indel:
            continue
        vcf = os.path.join(d, tag + ".vcf")
        t0 = time.perf_counter()
        rc, log = refloop.run_worker(ref, lst, ck, vcf, pileup, indel, cpu_threads=threads, **kw)
        wall = time.perf_counter() - t0
        m = re.search(r"Total time elapsed: ([0-9.]+) s", log)
        assert rc == 0 and f"Total processed positions : {n}" in log, log[-2000:]
        res[tag] = {"loop_seconds": float(m.group(1)), "process_wall_seconds": round(wall, 2), "windows_per_s_in_the_loop": round(n / float(m.group(1)))}
    res["vcf_identical"] = refloop.compare_vcfs(os.path.join(d, "libc3hip.vcf"), os.path.join(d, "reference_modules_pytorch.vcf"))["identical_text"]
    out[name] = {"windows": n, "tensor_files": files, "cpu_threads": threads, **res}
    print(name, json.dumps(out[name]), flush=True)
print(json.dumps(out))
