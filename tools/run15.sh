#!/bin/bash
# round 6, run 17: the GPU suite on the tree with --gvcf rows in c3_vcf_rows and the torch-free worker process
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q 2>&1 > gpurun_out/pytest_gpu_full.txt; tail -5 gpurun_out/pytest_gpu_full.txt
grep -n "^E " gpurun_out/pytest_gpu_full.txt | head -20
