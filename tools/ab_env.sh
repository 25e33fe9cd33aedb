#!/bin/bash
# A/B of environment switches on one box: tools/ab_env.sh <workload> "<VAR=a>" "<VAR=b>" ...  (each setting twice, interleaved)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
wl=$1; shift
for rep in 1 2; do for setting in "$@"; do
  echo "== $wl $setting (run $rep)"
  env $setting C3_BENCH_FULL=/tmp/ab_full.json timeout 300 python bench.py --gpus 1 --workload $wl --no-cpu-baseline --no-reference-gpu ${AB_FLAGS:-} --steps ${STEPS:-200} 2>> gpurun_out/ab_env.err > /dev/null; python -c "
import sys,json
d=json.load(open('/tmp/ab_full.json'))
h=d.get('host_inclusive',{}); b=h.get('batch_1000',{})
print('  device: one %.0f  best-of-legs %.0f | host ring B=%d %.0f | B=1000: dev %.0f ring %.0f (%.3f) sync %.0f (%.3f) dropin %.0f (%.3f)' % (d['one_batch_in_flight']['value'], d['value'], h.get('batch',0), h.get('value',0), b.get('device_resident_one_in_flight',0), b.get('value',0), b.get('frac_of_device_resident',0), b.get('sync_call',{}).get('value',0), b.get('sync_call',{}).get('frac_of_device_resident',0), b.get('dropin_loop',{}).get('value',0), b.get('dropin_loop',{}).get('frac_of_device_resident',0)))
print('  kernels:', ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d.get('kernels',{}).items()))
"
done; done
