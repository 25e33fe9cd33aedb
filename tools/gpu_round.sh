#!/bin/bash
# One gpurun call: diagnostics -> parity tests -> bench -> rocprof kernel trace.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
STAGES="${1:-diag test bench prof}"
for s in $STAGES; do
  case $s in
    diag)  timeout 600 python tests/diag/gpu_diag.py > gpurun_out/diag.txt 2>&1; echo "diag rc=$?" ;;
    test)  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.txt ;;
    testall) timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.txt ;;
    smoke) timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.txt ;;
    bench) timeout 900 python bench.py --gpus 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json | head -c 3000 ;;
    prof)  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-leg > "$OLDPWD/gpurun_out/prof_bench.json" 2> "$OLDPWD/gpurun_out/prof.err"); echo "prof rc=$?"; find gpurun_out/prof -name '*stats*' | head ;;
    prof1) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof1" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-host-leg --streams 1 > "$OLDPWD/gpurun_out/prof1_bench.json" 2> "$OLDPWD/gpurun_out/prof1.err"); echo "prof1 rc=$?"; find gpurun_out/prof1 -name '*stats*' | head ;;
    pmc)   (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmc_fetch" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --workload full_alignment > /dev/null 2> "$OLDPWD/gpurun_out/pmc_fetch.err"); echo "pmc fetch rc=$?"
           (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OLDPWD/gpurun_out/pmc_write" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --workload full_alignment > /dev/null 2> "$OLDPWD/gpurun_out/pmc_write.err"); echo "pmc write rc=$?" ;;
    pmcp)  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OLDPWD/gpurun_out/pmcp_fetch" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --workload pileup > /dev/null 2> "$OLDPWD/gpurun_out/pmcp_fetch.err"); echo "pmcp fetch rc=$?"
           (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OLDPWD/gpurun_out/pmcp_write" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --workload pileup > /dev/null 2> "$OLDPWD/gpurun_out/pmcp_write.err"); echo "pmcp write rc=$?" ;;
    benchfa) for cfg in "0x1b6 0" "0x1b6 0x1b6"; do set -- $cfg; mask=$1; export C3HIP_WINOGRAD_PMASK=$2; echo "== C3HIP_WINOGRAD=$mask qmask=$2"; C3HIP_WINOGRAD=$mask timeout 600 python bench.py --gpus 1 --workload full_alignment --no-cpu-baseline --no-host-leg --streams 1 2> gpurun_out/benchfa.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.0f windows/s  %.4f ms/step' % (d['value'], d['ms_per_step']))
for k,v in d['kernels'].items(): print('   %-9s %7.1f us %6.1f TF' % (k, v['avg_us'], v['tflops'] or 0))
"; done; unset C3HIP_WINOGRAD_PMASK ;;
    wprobe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w tools/wino_probe.hip -o /tmp/wino_probe && timeout 300 /tmp/wino_probe > gpurun_out/wino_probe.txt 2>&1; echo "wprobe rc=$?"; cat gpurun_out/wino_probe.txt ;;
    ptrace) for g in 1; do C3HIP_PROJ_TRACE=1 timeout 300 python bench.py --gpus 1 --workload pileup --no-cpu-baseline --no-host-leg --streams 1 --steps 3 --warmup 1 2>&1 >/dev/null | grep -A24 'proj trace'; done ;;
    torchrun1) timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg > gpurun_out/torchrun1.json 2> gpurun_out/torchrun1.err; echo "torchrun1 rc=$?"; cut -c1-400 gpurun_out/torchrun1.json; tail -3 gpurun_out/torchrun1.err ;;
    sprobe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -w tools/stream_probe.hip -o /tmp/stream_probe && timeout 300 /tmp/stream_probe > gpurun_out/stream_probe.txt 2>&1; echo "sprobe rc=$?"; cat gpurun_out/stream_probe.txt ;;
    coprobe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -w tools/coissue_probe.hip -o /tmp/coissue_probe && timeout 300 /tmp/coissue_probe > gpurun_out/coissue_probe.txt 2>&1; echo "coprobe rc=$?"; cat gpurun_out/coissue_probe.txt ;;
    tail) for v in 0 1; do echo "== C3HIP_TAIL_MFMA=$v"; C3HIP_TAIL_MFMA=$v timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-host-leg --streams 1 2> gpurun_out/bencht.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  FA %.0f windows/s  %.4f ms/step' % (d['value'], d['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items() if k in ('fa.l4','fa.tail','fa.spp')))
p=d['pileup']; print('  pileup %.0f windows/s  %.4f ms/step' % (p['value'], p['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in p['kernels'].items()))
"; done ;;
    batches) for cfg in "full_alignment 256" "full_alignment 1024" "full_alignment 2048" "pileup 1024" "pileup 4096" "pileup 16384"; do set -- $cfg; echo "== $1 B=$2"; timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-host-leg --workload $1 --batch $2 --steps 20 --warmup 3 2> gpurun_out/benchb.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.0f windows/s with %d in flight (%.4f ms/step) | one in flight %.0f windows/s | whole-forward frac %.3f' % (d['value'], d['config']['batches_in_flight'], d['ms_per_step'], d['one_batch_in_flight']['value'], d['roofline']['whole_forward_frac']))
"; done ;;
    bigp) timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-host-leg --workload pileup --batch 16384 --steps 10 --warmup 2 --streams 1 2> gpurun_out/benchb.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  pileup B=16384: %.0f windows/s' % d['value'], ' '.join('%s=%.1f(%.0fTF)' % (k, v['avg_us'], v['tflops'] or 0) for k,v in d['kernels'].items()))
"; timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-host-leg --workload full_alignment --batch 2048 --steps 10 --warmup 2 --streams 1 2> gpurun_out/benchb.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  FA B=2048: %.0f windows/s' % d['value'], ' '.join('%s=%.1f(%.0fTF)' % (k, v['avg_us'], v['tflops'] or 0) for k,v in d['kernels'].items()))
" ;;
    proj2) for v in 0 1; do echo "== C3HIP_PROJ2_STREAM=$v"; for b in 1024 16384; do C3HIP_PROJ2_STREAM=$v timeout 600 python bench.py --gpus 1 --workload pileup --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-host-leg --streams 1 2> gpurun_out/benchp.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  B=%d value %.0f windows/s  %.4f ms/step' % (d['config']['batch_per_gpu'], d['value'], d['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; done; done ;;
    pmcsq2) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OLDPWD/gpurun_out/pmc_sq_a" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --streams 1 > /dev/null 2> "$OLDPWD/gpurun_out/pmc_sq_a.err"); echo "pmc sq a rc=$?"
           (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM -d "$OLDPWD/gpurun_out/pmc_sq_b" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --streams 1 > /dev/null 2> "$OLDPWD/gpurun_out/pmc_sq_b.err"); echo "pmc sq b rc=$?" ;;
    fa1) timeout 600 python tests/diag/gpu_diag.py fa > gpurun_out/diag.txt 2>&1; grep -E "act2|act5|act8|^  y" gpurun_out/diag.txt | cut -c1-120; for i in 1 2; do timeout 600 python bench.py --gpus 1 --workload full_alignment --no-cpu-baseline --no-host-leg --streams 1 2> gpurun_out/benchc1.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.0f windows/s  %.4f ms/step' % (d['value'], d['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; done ;;
    gtrace) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -w tools/gemm_trace.hip -o /tmp/gemm_trace && timeout 300 /tmp/gemm_trace > gpurun_out/gemm_trace.txt 2>&1; echo "gtrace rc=$?"; cut -c1-1200 gpurun_out/gemm_trace.txt ;;
    convbn) for v in 0 0x8 0x40 0x48; do echo "== C3HIP_CONV_BN64MASK=$v"; C3HIP_CONV_BN64MASK=$v timeout 600 python bench.py --gpus 1 --workload full_alignment --no-cpu-baseline --no-host-leg --streams 1 2> gpurun_out/benchc1.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.0f windows/s  %.4f ms/step' % (d['value'], d['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; done ;;
    conv1) for v in 0 1; do echo "== C3HIP_CONV1_DIRECT=$v"; C3HIP_CONV1_DIRECT=$v timeout 600 python bench.py --gpus 1 --workload full_alignment --no-cpu-baseline --no-host-leg --streams 1 2> gpurun_out/benchc1.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.0f windows/s  %.4f ms/step' % (d['value'], d['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; done ;;
    isolate) for cfg in "0 0" "0 1" "0x40 0"; do set -- $cfg; echo "== BN64MASK=$1 CONV1_DIRECT=$2"; C3HIP_CONV_BN64MASK=$1 C3HIP_CONV1_DIRECT=$2 timeout 600 python tests/diag/gpu_diag.py fa > gpurun_out/iso.txt 2>&1; grep -E "act0|act5|act6|act7|^  y" gpurun_out/iso.txt | cut -c1-150; done ;;
    cmpq) timeout 600 python tools/cmp_variants.py "C3HIP_WINOGRAD_PMASK=0" "C3HIP_WINOGRAD_PMASK=0x1b6" 300 > gpurun_out/cmpq.txt 2>&1; echo "cmpq rc=$?"; grep -E "differ|^y" gpurun_out/cmpq.txt ;;
    diagfap) C3HIP_WINOGRAD_PMASK=0x1b6 timeout 600 python tests/diag/gpu_diag.py fa fa9 > gpurun_out/diagp.txt 2>&1; echo "diag(pmask) rc=$?"; grep -E "act|y " gpurun_out/diagp.txt | head -30 ;;
    streams) for st in 1 2 3 4; do echo "== --streams $st"; timeout 600 python bench.py --gpus 1 --no-cpu-baseline --no-host-leg --streams $st 2> gpurun_out/benchs.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  FA %.0f windows/s  %.4f ms/step | pileup %.0f windows/s %.4f ms/step' % (d['value'], d['ms_per_step'], d['pileup']['value'], d['pileup']['ms_per_step']))
"; done ;;
    decode) timeout 600 python tests/diag/gpu_diag.py decode > gpurun_out/decode.txt 2>&1; echo "decode rc=$?"; cat gpurun_out/decode.txt ;;
    host) timeout 600 python tests/diag/gpu_diag.py host > gpurun_out/host.txt 2>&1; echo "host rc=$?"; cat gpurun_out/host.txt ;;
    diagp) timeout 600 python tests/diag/gpu_diag.py pileup pileup32 > gpurun_out/diag.txt 2>&1; echo "diag rc=$?" ;;
    benchp) for v in 0 1; do echo "== C3HIP_LSTM1_FUSED=$v"; C3HIP_LSTM1_FUSED=$v timeout 600 python bench.py --gpus 1 --workload pileup --no-cpu-baseline --no-host-leg 2> gpurun_out/benchp.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.0f windows/s  %.4f ms/step' % (d['value'], d['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; done ;;
    diagfa) timeout 600 python tests/diag/gpu_diag.py fa fa9 > gpurun_out/diag.txt 2>&1; echo "diag rc=$?" ;;
    counters) rocprofv3 -L > gpurun_out/counters.txt 2>&1; echo "counters rc=$?"; wc -l gpurun_out/counters.txt ;;
    pmcsq) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$OLDPWD/gpurun_out/pmc_sq" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg > /dev/null 2> "$OLDPWD/gpurun_out/pmc_sq.err"); echo "pmc sq rc=$?" ;;
    hostsweep) for th in 0 1 3 7; do echo "== C3HIP_STAGE_THREADS=$th"; C3HIP_STAGE_THREADS=$th timeout 600 python bench.py --gpus 1 --workload full_alignment --no-cpu-baseline --no-host-leg --steps 100 2> gpurun_out/hostsweep.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
h=d['host_inclusive']
print('  device one-in-flight %.0f | host B=256 %.0f (%.2f) | host B=1000 %.0f of %.0f (%.2f) | registered %s' % (d['one_batch_in_flight']['value'], h['value'], h['frac_of_device_resident_one_in_flight'], h['batch_1000']['value'], h['batch_1000']['device_resident_one_in_flight'], h['batch_1000']['frac_of_device_resident'], h.get('batch_1000_registered_source')))
"; done ;;
    cprobe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w tools/conv_probe.hip -o /tmp/conv_probe && timeout 300 /tmp/conv_probe > gpurun_out/conv_probe.txt 2>&1; echo "cprobe rc=$?"; cat gpurun_out/conv_probe.txt ;;
    dmab) pyb() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  value %.0f windows/s  %.4f ms/step' % (d['value'], d['ms_per_step']), ' '.join('%s=%.1f' % (k, v['avg_us']) for k,v in d['kernels'].items()))
"; }
          for dm in ${DENSE_MODES:-3 4 3 4}; do echo "== pileup C3HIP_DENSE_MODE=$dm"; C3HIP_DENSE_MODE=$dm timeout 600 python bench.py --gpus 1 --workload pileup --no-cpu-baseline --no-host-leg --streams 1 --steps 100 --warmup 5 2> gpurun_out/dmab.err | pyb; done
          for dm in ${DENSE_MODES:-3 4 3 4}; do echo "== full_alignment C3HIP_DENSE_MODE=$dm"; C3HIP_DENSE_MODE=$dm timeout 600 python bench.py --gpus 1 --workload full_alignment --no-cpu-baseline --no-host-leg --streams 1 --steps 100 --warmup 5 2> gpurun_out/dmab.err | pyb; done ;;
    traces) C3HIP_LSTM_TRACE=4 C3HIP_DENSE_TRACE=4 timeout 600 python bench.py --gpus 1 --workload pileup --no-cpu-baseline --no-host-leg --streams 1 --steps 10 --warmup 2 2>&1 >/dev/null | grep -A40 "trace (" | cut -c1-400 ;;
    dprobe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc tools/dense_probe.hip -o /tmp/dense_probe && timeout 300 /tmp/dense_probe > gpurun_out/dense_probe.txt 2>&1; echo "dprobe rc=$?"; cat gpurun_out/dense_probe.txt ;;
    info)  (rocminfo | grep -E "Name|Compute Unit|Max Clock|Wavefront" | head -40; lscpu | head -20; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null) > gpurun_out/info.txt 2>&1 ;;
  esac
done
cat gpurun_out/diag.txt 2>/dev/null | head -80
