for st in 0 2; do
echo "== stagger $st"
C3HIP_PLANES_STAGGER=$st python bench.py --gpus 1 --no-cpu-baseline --no-host-leg --steps 100 --workload full_alignment 2> gpurun_out/bench_d.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['one_batch_in_flight'], d['roofline']['frac'], d['roofline']['mfma_util'])
print({k:(round(v['avg_us'],1), round(v['mfma_util'] or 0,3)) for k,v in d['kernels'].items()})
"
done
