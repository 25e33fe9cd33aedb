#!/bin/bash
# device-resident rate of S handles x B windows on one GPU: tools/split_sweep.sh <workload> "B S" "B S" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
wl=$1; shift
for cfg in "$@"; do set -- $cfg
 timeout 300 python bench.py --gpus 1 --workload $wl --no-cpu-baseline --no-reference-gpu --no-host-leg --no-profiled-pass --steps ${STEPS:-200} --batch $1 --streams $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl B=$1 streams=$2: value %.0f  one-in-flight %.0f' % (d['value'], d['one_batch_in_flight']['value']))"
done
