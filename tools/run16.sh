#!/bin/bash
# round 6, run 20: the library on weights after thousands of optimizer steps (trained through PyTorch-ROCm on the box's MI355X)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python tests/diag/long_training_parity.py 2000 2>&1 | grep -v "amdgpu.ids" | tail -8
