#!/bin/bash
# device-resident rate (one batch in flight) and per-kernel HIP-event averages over batch sizes: bash tools/batch_sweep.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for cfg in "pileup 1000" "pileup 2000" "pileup 4000" "pileup 8000" "pileup 16000" "full_alignment 256" "full_alignment 512" "full_alignment 1000" "full_alignment 2000"; do set -- $cfg
  C3_BENCH_FULL=/tmp/sweep_full.json timeout 300 python bench.py --gpus 1 --workload $1 --batch $2 --streams 1 --no-host-leg --no-cpu-baseline --no-reference-gpu --steps 30 --warmup 3 --repeats 3 > /dev/null 2>&1
  python -c "
import json
d=json.load(open('/tmp/sweep_full.json'))
print('$1 B=$2: %.0f windows/s one in flight (%.3f ms/step)' % (d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()), '|', d.get('kernel_variants', {}).get('one_batch_in_flight', '') if isinstance(d.get('kernel_variants'), dict) else d.get('kernel_variants', ''))
"
done
