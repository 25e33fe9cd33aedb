#!/bin/bash
# round 6, run 11: two against three lanes of the ring (same box, alternating) + the ring tests
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_worker.py tests/test_job_gpu.py -m gpu -q -x 2>&1 | tail -3
: > gpurun_out/ab_ring_lanes3.txt
for rep in 1 2; do for lanes in 1 2 3; do for wl in full_alignment pileup; do
  C3HIP_RING_LANES=$lanes C3_BENCH_FULL=/tmp/ab_full.json timeout 600 python bench.py --gpus 1 --workload $wl --streams 1 --no-cpu-baseline --no-reference-gpu --no-profiled-pass --steps 100 --warmup 5 --repeats 3 > /dev/null 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python - >> gpurun_out/ab_ring_lanes3.txt <<PY
import json
d=json.load(open('/tmp/ab_full.json')); h=d['host_inclusive']; b=h.get('batch_1000',{})
print("C3HIP_RING_LANES=$lanes rep $rep $wl: one in flight %.0f | three in flight %.0f | ring B=%d %.0f (at driver steps %.0f) | B=1000: ring %.0f  blocking call %.0f  drop-in loop %.0f" % (d['one_batch_in_flight']['value'], d.get('3_batches_in_flight',{}).get('value',0), h['batch'], h['value'], h['at_driver_steps']['value'], b.get('value',0), b.get('sync_call',{}).get('value',0), b.get('dropin_loop',{}).get('value',0)))
PY
done; done; done
cat gpurun_out/ab_ring_lanes3.txt
