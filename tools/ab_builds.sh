# same-box comparison of two builds of libc3hip.so (C3HIP_LIB): bench.py device-resident rates, one batch in flight and three
pyb() { python -c "
import sys,json
d=json.load(open('/tmp/ab_full.json'))
three=[v['value'] for k,v in d.items() if k.endswith('_batches_in_flight') and k!='one_batch_in_flight']
print('  one in flight %.0f (%.4f ms)  three %.0f | sum of kernels %.1f us |' % (d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step'], three[0] if three else 0, d['roofline']['step_us_sum_of_kernels']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; }
for rep in 1 2; do
for lib in "$@"; do
  for wl in ${WLS:-full_alignment pileup}; do
    echo "== $wl $lib"
    C3_BENCH_FULL=/tmp/ab_full.json C3HIP_LIB=$PWD/$lib timeout 600 python bench.py --gpus 1 --workload $wl --no-cpu-baseline --no-host-leg --no-reference-gpu --steps ${STEPS:-200} --warmup 10 2> gpurun_out/ab.err > /dev/null; pyb < /dev/null
  done
done
done
