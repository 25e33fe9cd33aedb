#!/bin/bash
# round 6, run 26: the pileup drop-in loop's group size now that 1000-window batches overlap in the ring's lanes: groups of 4000 windows per forward
# pass (default) against one forward pass per batch, in the bench's loop leg and end to end through the worker command
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/ab_pileup_group.txt
for rep in 1 2 3; do
  C3_BENCH_FULL=/tmp/ab_full.json timeout 600 python bench.py --gpus 1 --workload pileup --streams 1 --no-cpu-baseline --no-reference-gpu --no-profiled-pass --steps 100 --warmup 5 --repeats 3 > /dev/null 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python - >> gpurun_out/ab_pileup_group.txt <<PY
import json
d=json.load(open('/tmp/ab_full.json')); b=d['host_inclusive']['batch_1000']; l=b['dropin_loop']
print("bench loop leg rep $rep: groups of %d windows %.0f %s | one forward pass per batch %.0f | ring B=1000 %.0f | device-resident %.0f" % (l['windows_per_forward_pass'], l['value'], l['passes'], l['one_forward_pass_per_batch']['value'], b['value'], b['device_resident_one_in_flight']))
PY
done
for g in 4000 1000 2000 4000 1000 2000; do
  C3HIP_PREFETCH_GROUP=$g C3_WT_ONLY=pileup C3_WT_LEGS=libc3hip_decoder_columns timeout 600 python tests/diag/worker_throughput.py 8000 30 8 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['pileup']; r=d['libc3hip_decoder_columns']
print('worker command, %d pileup windows, C3HIP_PREFETCH_GROUP=$g: loop %.2f s = %d windows/s, process %.2f s' % (d['windows'], r['loop_seconds'], r['windows_per_s_in_the_loop'], r['process_wall_seconds']))" >> gpurun_out/ab_pileup_group.txt
done
cat gpurun_out/ab_pileup_group.txt
