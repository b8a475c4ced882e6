#!/bin/bash
# gpurun with the provenance stamp: writes build/git_head.txt (hash of the last commit that touched the kernel
# sources + the content hash of those sources) so that measurements taken on the GPU box -- which sees a
# snapshot without .git -- can say which tree they belong to (tools/provenance.py).
#   tools/gpurun.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.." || exit 1
python3 tools/provenance.py > /dev/null || exit 1
exec /usr/local/graft/bin/gpurun "$@"
