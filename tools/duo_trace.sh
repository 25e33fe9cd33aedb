#!/bin/bash
# kernel timeline of the C3HIP_DUO=1 path (two halves of a micro-batch on two streams): do the halves' kernels overlap?  bash tools/duo_trace.sh [workload]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
wl=${1:-pileup}
OUT="$PWD/gpurun_out"
for duo in 0 1; do
(cd /tmp && C3HIP_DUO=$duo C3_BENCH_FULL=/tmp/duo_full.json timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/duo_trace_$duo" -o c3 -- python "$OLDPWD/bench.py" --gpus 1 --workload $wl --streams 1 --no-host-leg --no-cpu-baseline --no-reference-gpu --no-profiled-pass --steps 5 --warmup 2 --repeats 1 > /dev/null 2> "$OUT/duo_trace_$duo.err"); echo "duo=$duo rc=$?"
done
python3 - <<'PY'
import csv, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out")
for duo in (0, 1):
    p = os.path.join(out, f"duo_trace_{duo}", "c3_kernel_trace.csv")
    rows = [r for r in csv.DictReader(open(p)) if "c3::" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = rows[-24:]
    t0 = int(last[0]["Start_Timestamp"])
    print(f"== C3HIP_DUO={duo}: the last {len(last)} c3 kernels (start us, end us, queue, stream, kernel)")
    for r in last:
        print("  %8.1f %8.1f  q=%s s=%s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), r.get("Stream_Id", "?"),
                                          r["Kernel_Name"].replace("void c3::", "").split("(")[0][:60]))
PY
