#!/bin/bash
# round 6, run 13: a worker process that never imports torch (clair3_amd/lazy_torch.py, ptfile.py): the worker-command tests, then the
# stage-B command's process wall time with and without it on the 120 000- and 240 000-window jobs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_reference_loop_gpu.py tests/test_ptfile.py tests/test_parity_gpu.py -m gpu -q -x 2>&1 | tail -5
for rep in 1 2; do
  C3_WT_LEGS=libc3hip_decoder_columns,libc3hip_decoder_columns_torch_imported timeout 900 python tests/diag/worker_throughput.py 4000 30 8 2>&1 | tail -1 > gpurun_out/worker_lazy_torch_120k_$rep.json
  cat gpurun_out/worker_lazy_torch_120k_$rep.json | cut -c1-1500
done
C3_WT_ONLY=full_alignment C3_WT_LEGS=libc3hip_decoder_columns,libc3hip_decoder_columns_torch_imported timeout 900 python tests/diag/worker_throughput.py 8000 30 8 2>&1 | tail -1 > gpurun_out/worker_lazy_torch_240k.json
cat gpurun_out/worker_lazy_torch_240k.json | cut -c1-1500
# where the rest of a process goes: the interpreter's own import times of the worker command on a tiny job
python - <<'PY'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
from clair3_amd import synthetic as syn
from tests import refloop
d = tempfile.mkdtemp(prefix="c3_imp_")
lst = refloop.write_job(d, syn.FULL_ALIGNMENT, [2000] * 2, channels=8)
ck = os.path.join(d, "model"); refloop.write_checkpoint(ck + ".pt", syn.FULL_ALIGNMENT, 8, True)
ref = refloop.reference_root()
for tag, env in (("default", {}), ("torch imported", {"C3HIP_LAZY_TORCH": "0", "C3HIP_PTFILE": "0"})):
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        rc, log = refloop.run_worker(ref, lst, ck, os.path.join(d, "o.vcf"), False, True, decoder=True, cpu_threads=8, extra_env=env)
        ts.append(time.perf_counter() - t0)
        assert rc == 0, log[-1500:]
    print(f"4000-window job, {tag}: process wall {min(ts):.2f} s (best of 3: {[round(t, 2) for t in ts]})")
PY
