#!/bin/bash
# tools/store_final.sh <tag>: copy what one `tools/gpu_round.sh "info test smoke bench bench20 prof_fa prof_p pmc_fa pmc_p sq_fa sq_p"` call (plus
# the two worker_throughput runs) left in gpurun_out/ into profiles/<tag>_*, with the PMC passes reduced to the JSON files bench.py reads.
set -e
cd "$(dirname "$0")/.."
tag=${1:-r05_final}
python tools/pmc_traffic.py $tag > /dev/null; python tools/pmc_traffic.py pileup $tag > /dev/null
python tools/pmc_sq_summary.py fa > profiles/${tag}_pmc_sq_fa.md; python tools/pmc_sq_summary.py p > profiles/${tag}_pmc_sq_pileup.md
cp profiles/pmc_traffic.json profiles/${tag}_pmc_traffic.json; cp profiles/pmc_traffic_pileup.json profiles/${tag}_pmc_traffic_pileup.json
cp gpurun_out/pytest_gpu.txt profiles/${tag}_pytest_gpu.txt; cp gpurun_out/smoke.txt profiles/${tag}_smoke.txt
cp gpurun_out/bench.json profiles/${tag}_bench.json; cp gpurun_out/bench_full.json profiles/${tag}_bench_full.json
cp gpurun_out/bench20.json profiles/${tag}_bench20.json; cp gpurun_out/bench20_full.json profiles/${tag}_bench20_full.json
cp gpurun_out/prof_fa/c3_kernel_stats.csv profiles/${tag}_kernel_stats_fa_one_in_flight.csv; cp gpurun_out/prof_p/c3_kernel_stats.csv profiles/${tag}_kernel_stats_pileup_one_in_flight.csv
cp gpurun_out/info.txt profiles/${tag}_box_info.txt
for f in ref_loop_dwell_hip ref_loop_full_alignment_hip ref_loop_full_alignment_hip_blocking ref_loop_full_alignment_hip_decoder ref_loop_pileup_hip; do cp gpurun_out/$f.json profiles/${tag}_$f.json; done
if [ -s gpurun_out/worker_throughput_30.txt ]; then
tail -1 gpurun_out/worker_throughput_30.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
try:
    e=json.loads(open('gpurun_out/worker_throughput_60_decoder.txt').read().strip().splitlines()[-1]); d['full_alignment_240k_windows_decoder_columns_only']=e['full_alignment']
except Exception as ex: print('no 60-file run:', ex)
d['what']='tests/diag/worker_throughput.py 4000 30 8 on one MI355X box (16 host cores): the reference\'s own stage-B worker command (clair3.py CallVariantsFromCffi, 8 decode processes, VCF out) over 30 tensor files, on libc3hip / on libc3hip with decoder columns / on the reference\'s modules (CPU); seeded-random weights and adversarial alt_info, i.e. decode rows that reject several candidates each; loop_seconds is the worker\'s own Total time elapsed (model load and ~0.6 s of start-up included); full_alignment_240k_windows_decoder_columns_only: C3_WT_ONLY=full_alignment C3_WT_LEGS=libc3hip_decoder_columns ... 4000 60 8 (the same command on 60 files, that leg alone)'
json.dump(d,open('profiles/${tag}_worker_throughput_30_files.json','w'),indent=1)
"
fi
python tools/roofline_check.py profiles/${tag}_bench.json profiles/${tag}_kernel_stats_fa_one_in_flight.csv profiles/${tag}_kernel_stats_pileup_one_in_flight.csv > profiles/${tag}_roofline_check.md 2>&1 || true
tail -4 profiles/${tag}_roofline_check.md
