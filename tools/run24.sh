#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "ring or fc_chain" 2>&1 | tail -4 | cut -c1-300
