#!/bin/bash
# registers, scratch, occupancy and LDS of every kernel of libc3hip (hipcc -Rpass-analysis=kernel-resource-usage):
#   tools/kernel_resources.sh [name filter]       -- a kernel with scratch > 0 has spilled: its loads no longer prefetch
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -w -fno-gpu-flush-denormals-to-zero '-DC3HIP_SRC_HASH="x"' \
  -Rpass-analysis=kernel-resource-usage clair3_amd/csrc/c3_model.hip -o /tmp/c3_resources.so 2>&1 | c++filt | python3 -c '
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v.replace("void c3::", "").replace("c3::", "")}
        rows.append(cur)
    elif cur is not None:
        cur[k.split()[0]] = v
print("%-78s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r in rows:
    if flt in r["name"]:
        print("%-78s %5s %5s %7s %4s %7s" % (r["name"][:78], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
' "$1"
