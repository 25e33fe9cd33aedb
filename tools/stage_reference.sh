#!/bin/bash
# Stage the reference's Python modules for the hot path into the git-ignored oracle/_ref/ (see oracle/stage_reference.py).
# Called from __graft_entry__.build(); run by hand after a fresh clone: tools/stage_reference.sh [/path/to/Clair3]
set -e
cd "$(dirname "$0")/.."
exec python3 oracle/stage_reference.py --reference "${1:-/root/reference}"
