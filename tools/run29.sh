#!/bin/bash
# round 6, run 48: the GPU suite on the tree with --enable_long_indel rows in c3_vcf_rows
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q 2>&1 > gpurun_out/pytest_gpu_full.txt; tail -4 gpurun_out/pytest_gpu_full.txt | cut -c1-300
grep -n "^E " gpurun_out/pytest_gpu_full.txt | head -10
python -c "
import json
d=json.load(open('gpurun_out/ref_loop_full_alignment_hip_longindel.json')); print({k:d[k] for k in ('case','run','records_a','records_b','identical_text','qual_only','loop_seconds','reference_cpu_loop_seconds')}, len(d['call_differs']))"
