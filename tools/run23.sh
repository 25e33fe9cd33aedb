#!/bin/bash
# round 6, run 42: the full-alignment drop-in loop's group size end to end (worker command, 240 000 windows, decoder columns): one forward pass per
# batch of 1000 against groups of 2000 (default), alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/ab_fa_group.txt
for rep in 1 2 3; do for g in 2000 1000; do
  C3HIP_PREFETCH_GROUP=$g C3_WT_ONLY=full_alignment C3_WT_LEGS=libc3hip_decoder_columns timeout 600 python tests/diag/worker_throughput.py 4000 60 8 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['full_alignment']; r=d['libc3hip_decoder_columns']
print('worker command, %d full-alignment windows, C3HIP_PREFETCH_GROUP=$g: loop %.2f s = %d windows/s, process %.2f s' % (d['windows'], r['loop_seconds'], r['windows_per_s_in_the_loop'], r['process_wall_seconds']))" >> gpurun_out/ab_fa_group.txt
done; done
cat gpurun_out/ab_fa_group.txt
