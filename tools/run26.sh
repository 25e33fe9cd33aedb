#!/bin/bash
# round 6, run 45: a long soak of the final tree's host side (ten minutes, three seeds) and the reference-modules fuzz
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for seed in 5 6 7; do SEED=$seed timeout 400 python tests/diag/ring_soak.py 180 2>&1 | tail -1 | cut -c1-300; done
  ls tests/diag | head -30; } | tee gpurun_out/long_soak.txt
