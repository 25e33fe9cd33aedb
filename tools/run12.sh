#!/bin/bash
# round 6, run 12: the blocking call in equal lane-sized pieces (default) against round 5's ring, same box, alternating; then the GPU suite and the soak
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/ab_blocking_call.txt
for rep in 1 2; do for cfg in "old 1 0" "new - -"; do set -- $cfg; for wl in full_alignment; do
  if [ $1 = old ]; then export C3HIP_RING_LANES=$2 C3HIP_TAIL_STREAM=$3; else unset C3HIP_RING_LANES C3HIP_TAIL_STREAM; fi
  C3_BENCH_FULL=/tmp/ab_full.json timeout 600 python bench.py --gpus 1 --workload $wl --streams 1 --no-cpu-baseline --no-reference-gpu --no-profiled-pass --steps 100 --warmup 5 --repeats 3 > /dev/null 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python - >> gpurun_out/ab_blocking_call.txt <<PY
import json
d=json.load(open('/tmp/ab_full.json')); h=d['host_inclusive']; b=h.get('batch_1000',{})
print("$1 rep $rep $wl: one in flight %.0f | ring B=%d %.0f (at driver steps %.0f) | B=1000: ring %.0f  blocking call %.0f  drop-in loop %.0f  device-resident %.0f" % (d['one_batch_in_flight']['value'], h['batch'], h['value'], h['at_driver_steps']['value'], b.get('value',0), b.get('sync_call',{}).get('value',0), b.get('dropin_loop',{}).get('value',0), b.get('device_resident_one_in_flight',0)))
PY
done; done; done
unset C3HIP_RING_LANES C3HIP_TAIL_STREAM
cat gpurun_out/ab_blocking_call.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 600 python tests/diag/ring_soak.py 2>&1 | tail -1 | cut -c1-300
