#!/bin/bash
# round 6, run 47: three or four submits in flight on the ring (C3_HOST_SLOTS = 4), fresh processes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for rep in 1 2; do for sl in 3 4 2; do export RING_SLOTS=$sl
  python tools/ring_fresh.py full_alignment 256 2>&1 | grep -v amdgpu.ids
  python tools/ring_fresh.py full_alignment 1000 2>&1 | grep -v amdgpu.ids
  python tools/ring_fresh.py pileup 1024 2>&1 | grep -v amdgpu.ids
done; done; } | tee gpurun_out/ring_slots.txt
