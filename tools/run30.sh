#!/bin/bash
# round 6, run 50: the GPU suite three times in a row on the final tree (flakiness check), then smoke and the driver's bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -1; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | cut -c1-400
