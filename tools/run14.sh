#!/bin/bash
# round 6, run 14: where the remaining 0.6 s of a worker process go
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY' 2>&1 | tee gpurun_out/worker_process_startup.txt
import os, subprocess, sys, tempfile, time, re
sys.path.insert(0, os.getcwd())
t0 = time.perf_counter()
import numpy as np
t1 = time.perf_counter()
from clair3_amd import _lib, synthetic as syn
from clair3_amd.model import Clair3_F
t2 = time.perf_counter()
L = _lib.lib()
t3 = time.perf_counter()
n = _lib.device_count()
t4 = time.perf_counter()
m = Clair3_F(add_indel_length=True, predict=True)
m.to("cuda:0")
t5 = time.perf_counter()
sd = syn.make_state_dict(syn.FULL_ALIGNMENT, 8, True, seed=2)
t6 = time.perf_counter()
m.load_state_dict(sd)
t7 = time.perf_counter()
x = syn.make_fa_windows(256, seed=1)
t8 = time.perf_counter()
y = m.predict_numpy(x)
t9 = time.perf_counter()
y = m.predict_numpy(x)
t10 = time.perf_counter()
print(f"in one process: numpy {t1-t0:.3f}  clair3_amd {t2-t1:.3f}  dlopen libc3hip (+ libamdhip64) {t3-t2:.3f}  device count (hipInit) {t4-t3:.3f}  "
      f"model create {t5-t4:.3f}  load_state_dict (pack + upload) {t7-t6:.3f}  first predict of 256 {t9-t8:.3f}  second {t10-t9:.4f}")
from tests import refloop
d = tempfile.mkdtemp(prefix="c3_imp_")
lst = refloop.write_job(d, syn.FULL_ALIGNMENT, [2000] * 2, channels=8)
ck = os.path.join(d, "model"); refloop.write_checkpoint(ck + ".pt", syn.FULL_ALIGNMENT, 8, True)
ref = refloop.reference_root()
env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.getcwd(), os.path.join(os.getcwd(), "tests", "stubs")]))
cmd = [sys.executable, "-X", "importtime", "-m", "clair3_amd.run_reference", "--ref", ref, "--decoder", "CallVariantsFromCffi", "--chkpnt_fn", ck, "--bam_fn", "u.bam",
       "--call_fn", os.path.join(d, "o.vcf"), "--sampleName", "S", "--platform", "ont", "--use_gpu", "True", "--cpu_threads", "8", "--threads", "4",
       "--output_tensor_can_fn_list", lst, "--gpu_id", "0", "--add_indel_length"]
t = time.perf_counter()
r = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=d)
wall = time.perf_counter() - t
assert r.returncode == 0, r.stderr[-2000:]
rows = []
for line in r.stderr.splitlines():
    mm = re.match(r"import time:\s+(\d+) \|\s+(\d+) \| (\s*)(\S+)", line)
    if mm:
        rows.append((int(mm.group(2)), len(mm.group(3)) // 2, mm.group(4)))
top = sorted([r_ for r_ in rows if r_[1] == 0], reverse=True)[:14]
print(f"worker command under -X importtime: wall {wall:.2f} s; cumulative import time of top-level imports {sum(r_[0] for r_ in rows if r_[1] == 0) / 1e6:.2f} s")
for us, _, name in top:
    print(f"   {us / 1e3:8.1f} ms  {name}")
PY
