#!/bin/bash
# round 6, run 4: when do the workgroups of one launch finish (tools/wino_probe.hip split: start / end stamps of every workgroup + the split-K bound again)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
WPROBE_ARGS=split tools/gpu_round.sh wprobe | grep -v "^  wino16\|^  vs fp64\|^  max|y|" | tail -40
