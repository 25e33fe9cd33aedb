#!/bin/bash
# One gpurun call: tools/gpu_round.sh "<stage> <stage> ..."  -- everything lands in gpurun_out/.
#   test      the whole -m gpu suite                         smoke   __graft_entry__.smoke()
#   bench     python bench.py (the driver's line)            bench20 the driver's own command (--steps 20 --warmup 5)
#   prof_fa / prof_p     rocprofv3 --kernel-trace --stats over ONLY the one-batch-in-flight leg of one workload
#   pmc_fa / pmc_p       HBM traffic of the same leg: separate FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py reads them)
#   l2_fa / l2_p         L2 hit rate of the same leg (TCC_HIT_sum / TCC_MISS_sum; tools/pmc_traffic.py adds it to the traffic record)
#   sq_fa / sq_p         SQ busy / MFMA busy / instruction mix of the same leg
#   wprobe    tools/wino_probe.hip: the F(2,3)-along-H convolution against the direct one (agreement, times, ablations)
#   dprobe / cprobe / lprobe   tools/dense_probe.hip / conv_probe.hip / lstm_probe.hip (ablations and phase traces of the dense / convolution / recurrence kernels)
#   info      rocminfo / lscpu / cgroup quota of the box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT="$PWD/gpurun_out"
leg() {  # leg <workload>: the bench flags that run one workload's one-batch-in-flight leg and nothing else
  echo "--gpus 1 --workload $1 --streams 1 --no-host-leg --no-cpu-baseline --no-reference-gpu --steps ${STEPS:-50} --warmup 5 --repeats 3"
}
prof() {  # prof <tag> <workload>: kernel trace + stats of the one-in-flight leg AND its HIP-event pass (the same kernels, the same one
          # batch in flight), so the roofline object of the JSON line comes from the very launches the CSV averages
  (cd /tmp && C3_BENCH_FULL="$OUT/$1_full.json" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$1" -o c3 -- python "$OLDPWD/bench.py" $(leg $2) > "$OUT/$1.json" 2> "$OUT/$1.err"); echo "$1 rc=$?"; find "$OUT/$1" -name '*kernel_stats*' | head -2
}
pmc() {  # pmc <tag> <workload> <counters...>
  local tag=$1 wl=$2; shift 2
  (cd /tmp && C3_BENCH_FULL=/tmp/pmc_full.json timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/$tag" -o c3 -- python "$OLDPWD/bench.py" $(STEPS=5 leg $wl) --no-profiled-pass > /dev/null 2> "$OUT/$tag.err"); echo "$tag rc=$?"
}
for s in ${1:-test bench}; do
  case $s in
    test)    timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.txt ;;
    smoke)   timeout 600 python __graft_entry__.py --smoke > gpurun_out/smoke.txt 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.txt ;;
    bench)   C3_BENCH_FULL="$OUT/bench_full.json" timeout 900 python bench.py --gpus 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; head -c 3000 gpurun_out/bench.json; echo ;;
    bench20) C3_BENCH_FULL="$OUT/bench20_full.json" timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench20.json 2> gpurun_out/bench20.err; echo "bench20 rc=$?"; head -c 3000 gpurun_out/bench20.json; echo ;;
    prof_fa) prof prof_fa full_alignment ;;
    prof_p)  prof prof_p pileup ;;
    pmc_fa)  pmc pmc_fetch_fa full_alignment FETCH_SIZE; pmc pmc_write_fa full_alignment WRITE_SIZE ;;
    pmc_p)   pmc pmc_fetch_p pileup FETCH_SIZE; pmc pmc_write_p pileup WRITE_SIZE ;;
    l2_fa)   pmc pmc_l2_fa full_alignment TCC_HIT_sum TCC_MISS_sum ;;
    l2_p)    pmc pmc_l2_p pileup TCC_HIT_sum TCC_MISS_sum ;;
    sq_fa)   pmc pmc_sq_a_fa full_alignment SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
             pmc pmc_sq_b_fa full_alignment SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM ;;
    sq_p)    pmc pmc_sq_a_p pileup SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
             pmc pmc_sq_b_p pileup SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM ;;
    dprobe)  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc -I tools tools/dense_probe.hip -o /tmp/dense_probe && timeout 300 /tmp/dense_probe > gpurun_out/dense_probe.txt 2>&1; echo "dprobe rc=$?"; cat gpurun_out/dense_probe.txt ;;
    cprobe)  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc tools/conv_probe.hip -o /tmp/conv_probe && timeout 300 /tmp/conv_probe > gpurun_out/conv_probe.txt 2>&1; echo "cprobe rc=$?"; cat gpurun_out/conv_probe.txt ;;
    l4probe) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc tools/l4_probe.hip -o /tmp/l4_probe && timeout 300 /tmp/l4_probe > gpurun_out/l4_probe.txt 2>&1; echo "l4probe rc=$?"; cat gpurun_out/l4_probe.txt ;;
    lprobe)  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc tools/lstm_probe.hip -o /tmp/lstm_probe && timeout 300 /tmp/lstm_probe > gpurun_out/lstm_probe.txt 2>&1; echo "lprobe rc=$?"; cat gpurun_out/lstm_probe.txt ;;
    wprobe)  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc -I tools tools/wino_probe.hip -o /tmp/wino_probe && timeout 300 /tmp/wino_probe ${WPROBE_ARGS:-} > gpurun_out/wino_probe.txt 2>&1; echo "wprobe rc=$?"; cat gpurun_out/wino_probe.txt ;;
    info)    (rocminfo | grep -E "Name|Compute Unit|Max Clock|Wavefront" | head -40; lscpu | head -20; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null) > gpurun_out/info.txt 2>&1 ;;
    *)       echo "unknown stage $s" ;;
  esac
done
