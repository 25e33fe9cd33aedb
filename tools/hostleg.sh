# sync-call / ring / multi-handle host-inclusive rates (bench.py host_inclusive) under the switches of c3_predict
for cfg in "0 1" "-1 0" "-1 1"; do
  set -- $cfg
  for wl in full_alignment pileup; do
    echo "== $wl C3HIP_PREDICT_CHUNK=$1 C3HIP_PREDICT_REGISTER=$2"
    if [ $1 = -1 ]; then unset C3HIP_PREDICT_CHUNK; else export C3HIP_PREDICT_CHUNK=$1; fi
    C3HIP_PREDICT_REGISTER=$2 timeout 600 python bench.py --gpus 1 --workload $wl --no-cpu-baseline --steps 100 --warmup 5 2> gpurun_out/hl.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
h=d['host_inclusive']; b=h['batch_1000']
print('  device one-in-flight %.0f (3 in flight %.0f) | host ring B=%d %.0f | B=1000: ring %.0f  sync %.0f  all handles %.0f  device %.0f' % (d['one_batch_in_flight']['value'], d['value'], h['batch'], h['value'], b['value'], b['sync_call']['value'], b.get('all_handles',{}).get('value',0), b['device_resident_one_in_flight']))
"
  done
done
