#!/bin/bash
# round 6, run 28: the full-alignment ring at the driver's 20 steps: lanes dealt in submit order (default) against slot % lanes, alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/ab_lane_order_fa20.txt
for rep in 1 2 3 4; do for cfg in "slot" "rr"; do
  export C3HIP_LANE_ORDER=$cfg
  C3_BENCH_FULL=/tmp/ab_full.json timeout 600 python bench.py --gpus 1 --workload full_alignment --streams 1 --no-cpu-baseline --no-reference-gpu --no-profiled-pass --steps 20 --warmup 5 > /dev/null 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python - >> gpurun_out/ab_lane_order_fa20.txt <<PY
import json
d=json.load(open('/tmp/ab_full.json')); h=d['host_inclusive']; b=h.get('batch_1000',{})
print("order %-4s rep $rep: one in flight %.0f | ring B=%d %.0f over %d steps, %.0f over the driver's %d | B=1000: ring %.0f  blocking call %.0f  drop-in loop %.0f" % ("$cfg", d['one_batch_in_flight']['value'], h['batch'], h['value'], h['steps'], h['at_driver_steps']['value'], h['at_driver_steps']['steps'], b.get('value',0), b.get('sync_call',{}).get('value',0), b.get('dropin_loop',{}).get('value',0)))
PY
done; done
unset C3HIP_LANE_ORDER
cat gpurun_out/ab_lane_order_fa20.txt
