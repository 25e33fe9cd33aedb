#!/bin/bash
# round 6, run 46: the whole-genome stand-in (BASELINE configs[3]) on one MI355X: a tenth of the 8 M pileup + 1.5 M full-alignment windows, files ->
# rows on rank 0 through clair3_amd/job.py (world of one: the gather is a device copy)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python tools/wgs_job.py --scale 0.1 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/wgs_job.jsonl | cut -c1-700
df -h /tmp | tail -1
