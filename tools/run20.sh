#!/bin/bash
# round 6, run 36: the new defaults (no tail stream, lane batches on their lane's stream alone, lazy transfer stream) in every kind of process
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for rep in 1 2; do python tools/ring_bisect.py 2>&1 | grep -v amdgpu.ids | sed -n '1p;4p;5p' | cut -c1-100; done
  for wl in full_alignment pileup; do for st in 1 3; do C3_BENCH_FULL=/tmp/f.json python bench.py --gpus 1 --workload $wl --streams $st --no-cpu-baseline --no-reference-gpu --no-profiled-pass --steps 100 --warmup 5 > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/f.json')); h=d['host_inclusive']; b=h['batch_1000']; print('   bench.py $wl --streams $st: one in flight %.0f | ring %.0f %s at driver steps %.0f | B=1000 ring %.0f blocking %.0f loop %.0f' % (d['one_batch_in_flight']['value'], h['value'], h['passes'], h['at_driver_steps']['value'], b['value'], b['sync_call']['value'], b['dropin_loop']['value']))"; done; done
  C3_BENCH_FULL=/tmp/f.json python bench.py --gpus 1 --no-cpu-baseline --no-reference-gpu --no-profiled-pass --steps 20 --warmup 5 > /dev/null 2>&1; python -c "
import json; d=json.load(open('/tmp/f.json'))
for n,x in (('FA',d),('pileup',d['pileup']),('dwell',d['full_alignment_dwell'])):
    h=x['host_inclusive']; print('   the driver\'s command (all workloads, 20 steps):', n, 'one in flight %.0f | ring %.0f %s at driver steps %.0f %s' % (x['one_batch_in_flight']['value'], h['value'], h['passes'], h['at_driver_steps']['value'], h['at_driver_steps']['passes']))"
} | tee gpurun_out/ring_streams_defaults.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 | cut -c1-300
timeout 600 python tests/diag/ring_soak.py 2>&1 | tail -1 | cut -c1-300
