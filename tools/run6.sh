#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_sgd_trained_gpu.py tests/test_sgd_trained.py -q -s 2>&1 | grep -v "^$" | tail -30 | cut -c1-900
