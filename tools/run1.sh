cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
tools/gpu_round.sh "info test"
timeout 300 python tests/diag/sensitive_window.py > gpurun_out/sensitive_window.txt 2>&1; echo "sens rc=$?"; cat gpurun_out/sensitive_window.txt | tail -5
WPROBE_ARGS=split tools/gpu_round.sh wprobe | tail -30
for f in 0 1; do C3HIP_FP32=$f C3_BENCH_FULL=gpurun_out/p_fp32_$f.json timeout 300 python bench.py --gpus 1 --workload pileup --streams 1 --no-host-leg --no-cpu-baseline --no-reference-gpu --steps 50 --warmup 5 --repeats 3 --no-profiled-pass > gpurun_out/p_fp32_$f.line 2> gpurun_out/p_fp32_$f.err; echo "pileup fp32=$f rc=$?"; head -c 600 gpurun_out/p_fp32_$f.line; echo; done
