#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit rate of the probe's launches (tools/wino_probe.hip): what the F(2,3) transform's loads cost the memory system
# next to the direct kernel's halo loads, per launch and shape.  One gpurun call: bash tools/wino_pmc.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
OUT="$PWD/gpurun_out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fno-gpu-flush-denormals-to-zero -w -I clair3_amd/csrc tools/wino_probe.hip -o /tmp/wino_probe || exit 1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=wp_$(echo $c | cut -d' ' -f1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/$tag" -o c3 -- /tmp/wino_probe big > /dev/null 2> "$OUT/$tag.err"); echo "$tag rc=$?"
done
python3 - <<'PY'
import csv, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out")
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("wp_FETCH_SIZE", "wp_WRITE_SIZE", "wp_TCC_HIT_sum"):
    p = os.path.join(out, tag, "c3_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "conv3x3" not in k:
            continue
        rows[(k.split("(")[0].replace("void c3::", ""), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "wino_pmc.txt"), "w") as fh:
    fh.write("kernel | grid (threads) | launches | FETCH_SIZE x2 MB (gfx950 correction) | WRITE_SIZE MB | L2 hit rate\n")
    for (k, g), v in sorted(rows.items()):
        f = v.get("FETCH_SIZE", [0]); w = v.get("WRITE_SIZE", [0]); h = sum(v.get("TCC_HIT_sum", [0])); m = sum(v.get("TCC_MISS_sum", [0]))
        fh.write("%s | %s | %d | %.1f | %.1f | %s\n" % (k, g, len(f), 2 * sum(f) / len(f) / 1024, sum(w) / len(w) / 1024, ("%.3f" % (h / (h + m))) if h + m else "-"))
print(open(os.path.join(out, "wino_pmc.txt")).read())
PY
