#!/bin/bash
# round 6, run 25: the ring's lanes dealt in submit order and ring batches beside another lane's batch on the kernel forms for a shared chip
# (the default now) against round 6's first form (C3HIP_LANE_ORDER=slot C3HIP_LANE_SHARING=0), pileup network, same box, alternating; GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/ab_lane_sharing.txt
for rep in 1 2 3; do for cfg in "r6a slot 0" "new - -"; do set -- $cfg
  if [ $1 = r6a ]; then export C3HIP_LANE_ORDER=$2 C3HIP_LANE_SHARING=$3; else unset C3HIP_LANE_ORDER C3HIP_LANE_SHARING; fi
  C3_BENCH_FULL=/tmp/ab_full.json timeout 600 python bench.py --gpus 1 --workload pileup --streams 1 --no-cpu-baseline --no-reference-gpu --no-profiled-pass --steps 100 --warmup 5 --repeats 3 > /dev/null 2>/tmp/ab.err || tail -3 /tmp/ab.err
  python - >> gpurun_out/ab_lane_sharing.txt <<PY
import json
d=json.load(open('/tmp/ab_full.json')); h=d['host_inclusive']; b=h.get('batch_1000',{})
print("%-4s rep $rep: one in flight %.0f | ring B=%d %.0f (at driver steps %.0f) | B=1000: ring %.0f  blocking call %.0f  drop-in loop %.0f %s  device-resident %.0f" % ("$1", d['one_batch_in_flight']['value'], h['batch'], h['value'], h['at_driver_steps']['value'], b.get('value',0), b.get('sync_call',{}).get('value',0), b.get('dropin_loop',{}).get('value',0), b.get('dropin_loop',{}).get('passes'), b.get('device_resident_one_in_flight',0)))
PY
done; done
unset C3HIP_LANE_SHARING C3HIP_LANE_ORDER
cat gpurun_out/ab_lane_sharing.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 | cut -c1-400
timeout 600 python tests/diag/ring_soak.py 2>&1 | tail -1 | cut -c1-300
