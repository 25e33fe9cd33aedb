# same-box comparison of two builds of libc3hip.so (C3HIP_LIB): bench.py device-resident rates, one batch in flight and three
pyb() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  one in flight %.0f (%.4f ms)  three %.0f | sum of kernels %.1f us |' % (d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step'], d['value'], d['roofline']['step_us_sum_of_kernels']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; }
for rep in 1 2; do
for lib in "$@"; do
  for wl in ${WLS:-full_alignment pileup}; do
    echo "== $wl $lib"
    C3HIP_LIB=$PWD/$lib timeout 600 python bench.py --gpus 1 --workload $wl --no-cpu-baseline --no-host-leg --steps 200 --warmup 10 2> gpurun_out/ab.err | pyb
  done
done
done
