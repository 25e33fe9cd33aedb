# A/B of one environment switch on the bench's device-resident legs: bash tools/benchfa.sh VAR "v1 v2 ..." [workloads]
VAR=${1:-C3HIP_LANES}; VALS=${2:-"1 2"}; WL=${3:-"full_alignment pileup"}
for wl in $WL; do for v in $VALS; do
echo "== $wl $VAR=$v"
env $VAR=$v python bench.py --gpus 1 --no-cpu-baseline --no-host-leg --steps 100 --workload $wl 2> gpurun_out/bench_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  3-in-flight %.0f  one-in-flight %.0f  (%.4f ms/step)  frac %.3f mfma_util %.3f' % (d['value'], d['one_batch_in_flight']['value'], d['one_batch_in_flight']['ms_per_step'], d['roofline']['frac'], d['roofline']['mfma_util']))
print('  ', {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
"
done; done
