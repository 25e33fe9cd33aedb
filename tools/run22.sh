#!/bin/bash
# round 6, run 41: the blocking call's equal lane-sized pieces (default now) against the growing pieces, then the GPU suite and the soak
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/blocking_call_pieces.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/blocking_call_pieces3.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 | cut -c1-300
timeout 600 python tests/diag/ring_soak.py 2>&1 | tail -1 | cut -c1-300
