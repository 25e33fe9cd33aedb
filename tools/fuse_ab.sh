pyb() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.0f windows/s  %.4f ms/step' % (d['value'], d['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; }
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -8
for f in 0 1 0 1; do echo "== full_alignment C3HIP_CONV1_FUSED=$f"; C3HIP_CONV1_FUSED=$f timeout 600 python bench.py --gpus 1 --workload full_alignment --no-cpu-baseline --no-host-leg --streams 1 --steps 100 --warmup 5 2> gpurun_out/fuse.err | pyb; done
