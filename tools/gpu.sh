#!/bin/bash
# Run a command on the MI355X box through gpurun, recording which commit (+ dirty flag) the snapshot is:  tools/gpu.sh [--timeout S] -- '<command>'
# (.git does not travel; tools/pmc_traffic.py and bench.py read .git_head so that profiles say which tree they measured)
cd "$(dirname "$0")/.."
h=$(git rev-parse --short HEAD)
[ -n "$(git status --porcelain --untracked-files=no)" ] && h="$h-dirty"
echo "$h" > .git_head
exec /usr/local/graft/bin/gpurun "$@"
