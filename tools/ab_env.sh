# A/B of one environment switch on one box: bash tools/ab_env.sh VAR "v1 v2 ..." workload [repeats]
pyb() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  one in flight %.0f (%.4f ms)  three %.0f |' % (d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step'], d['value']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; }
for rep in $(seq 1 ${4:-2}); do for v in $2; do echo "== $3 $1=$v"; env $1=$v timeout 600 python bench.py --gpus 1 --workload $3 --no-cpu-baseline --no-host-leg --steps 200 --warmup 10 2> gpurun_out/ab_env.err | pyb; done; done
