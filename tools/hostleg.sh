# sync-call / ring / multi-handle host-inclusive rates (bench.py host_inclusive), with and without page-locking the caller's pages
for reg in ${REGS:-0 1 0 1}; do
  for wl in full_alignment pileup; do
    echo "== $wl C3HIP_PREDICT_REGISTER=$reg"
    C3HIP_PREDICT_REGISTER=$reg timeout 600 python bench.py --gpus 1 --workload $wl --no-cpu-baseline --steps 100 --warmup 5 2> gpurun_out/hl.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
h=d['host_inclusive']; b=h['batch_1000']
print('  device one-in-flight %.0f (3 in flight %.0f) | host ring B=%d %.0f | B=1000: ring %.0f  sync %.0f  all handles %.0f  registered %.0f  device %.0f' % (d['one_batch_in_flight']['value'], d['value'], h['batch'], h['value'], b['value'], b['sync_call']['value'], b.get('all_handles',{}).get('value',0), h.get('batch_1000_registered_source',{}).get('value',0), b['device_resident_one_in_flight']))
"
  done
done
