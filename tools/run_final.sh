#!/bin/bash
# the round's ONE final collection: everything tools/store_final.sh copies into profiles/<tag>_*
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tools/gpu_round.sh "info test smoke bench bench20 prof_fa prof_p pmc_fa pmc_p l2_fa l2_p sq_fa sq_p"
timeout 900 python tests/diag/worker_throughput.py 4000 30 8 > gpurun_out/worker_throughput_30.txt 2>&1; echo "wt30 rc=$?"
C3_WT_ONLY=full_alignment C3_WT_LEGS=libc3hip_decoder_columns timeout 600 python tests/diag/worker_throughput.py 4000 60 8 > gpurun_out/worker_throughput_60_decoder.txt 2>&1; echo "wt60 rc=$?"; tail -1 gpurun_out/worker_throughput_60_decoder.txt | cut -c1-300
tools/batch_sweep.sh > gpurun_out/batch_sweep.txt 2>&1; echo "sweep rc=$?"; tail -12 gpurun_out/batch_sweep.txt | cut -c1-200
df -h /tmp | tail -1
