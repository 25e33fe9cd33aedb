#!/bin/bash
# round 6, run 3: the narrower-gx2 upper bound (tools/dense_probe.hip gx2), the loop timeline with the C row printer, the pileup worker sweep, the driver's bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc -I tools tools/dense_probe.hip -o /tmp/dense_probe && timeout 300 /tmp/dense_probe 1024 gx2 > gpurun_out/dense_probe_gx2.txt 2>&1; echo "gx2 probe rc=$?"; cat gpurun_out/dense_probe_gx2.txt
for t in 2 8; do timeout 600 python tests/diag/loop_timeline.py 4000 60 $t full_alignment > gpurun_out/loop_timeline_fa_$t.txt 2>&1; echo "timeline fa $t rc=$?"; tail -3 gpurun_out/loop_timeline_fa_$t.txt | cut -c1-1800; done
: > gpurun_out/worker_sweep_pileup.txt
for t in 2 4 8 16; do
  echo "== pileup C3HIP_ROWS_C=1 cpu_threads=$t" >> gpurun_out/worker_sweep_pileup.txt
  C3_WT_ONLY=pileup C3_WT_LEGS=libc3hip_decoder_columns timeout 600 python tests/diag/worker_throughput.py 4000 30 $t 2>&1 | tail -1 >> gpurun_out/worker_sweep_pileup.txt
done
echo "== pileup C3HIP_ROWS_C=0 cpu_threads=8" >> gpurun_out/worker_sweep_pileup.txt
C3HIP_ROWS_C=0 C3_WT_ONLY=pileup C3_WT_LEGS=libc3hip_decoder_columns timeout 600 python tests/diag/worker_throughput.py 4000 30 8 2>&1 | tail -1 >> gpurun_out/worker_sweep_pileup.txt
cat gpurun_out/worker_sweep_pileup.txt | cut -c1-300
df -h /tmp | tail -1
tools/gpu_round.sh "bench20"
