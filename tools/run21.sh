#!/bin/bash
# round 6, run 38: two or three lanes for the pileup network, now that a batch beside others runs on the shared-chip forms and a lane is one stream
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for rep in 1 2; do for l in 2 3; do export C3HIP_RING_LANES=$l
  python tools/ring_fresh.py pileup 1024 2>&1 | grep -v amdgpu.ids
  python tools/ring_fresh.py pileup 1000 2>&1 | grep -v amdgpu.ids
  for st in 1 3; do C3_BENCH_FULL=/tmp/f.json python bench.py --gpus 1 --workload pileup --streams $st --no-cpu-baseline --no-reference-gpu --no-profiled-pass --steps 100 --warmup 5 > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/f.json')); h=d['host_inclusive']; b=h['batch_1000']; print('   lanes $l bench.py pileup --streams $st: one in flight %.0f | ring %.0f %s at driver steps %.0f | B=1000 ring %.0f blocking %.0f loop %.0f' % (d['one_batch_in_flight']['value'], h['value'], h['passes'], h['at_driver_steps']['value'], b['value'], b['sync_call']['value'], b['dropin_loop']['value']))"; done
done; done; } | tee gpurun_out/ring_lanes_pileup.txt
