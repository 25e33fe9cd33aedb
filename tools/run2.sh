#!/bin/bash
# round 6, run 2: the GPU suite, the stage-B worker command's --cpu_threads sweep with and without the one-pass C row printer, the bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tools/gpu_round.sh "test"
# VCF identity of the three legs once (4000 x 30 files, 8 decode processes), then the timing-only sweeps on the 240 k-window job
timeout 900 python tests/diag/worker_throughput.py 4000 30 8 > gpurun_out/worker_throughput_30.txt 2>&1; echo "wt30 rc=$?"; tail -2 gpurun_out/worker_throughput_30.txt | cut -c1-1500
: > gpurun_out/worker_sweep.txt
for rows_c in 1 0; do
  for t in 2 4 8 16 32; do
    [ $rows_c = 0 ] && [ $t != 8 ] && [ $t != 16 ] && continue
    for rep in 1 2; do
      echo "== C3HIP_ROWS_C=$rows_c cpu_threads=$t rep=$rep" >> gpurun_out/worker_sweep.txt
      C3HIP_ROWS_C=$rows_c C3_WT_ONLY=full_alignment C3_WT_LEGS=libc3hip_decoder_columns timeout 600 python tests/diag/worker_throughput.py 4000 60 $t 2>&1 | tail -1 >> gpurun_out/worker_sweep.txt
    done
  done
done
for t in 4 8 16; do
  echo "== pileup C3HIP_ROWS_C=1 cpu_threads=$t" >> gpurun_out/worker_sweep.txt
  C3_WT_ONLY=pileup C3_WT_LEGS=libc3hip_decoder_columns timeout 600 python tests/diag/worker_throughput.py 4000 30 $t 2>&1 | tail -1 >> gpurun_out/worker_sweep.txt
done
cat gpurun_out/worker_sweep.txt | cut -c1-400
tools/gpu_round.sh "bench20"
