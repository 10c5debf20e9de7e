#!/bin/bash
# build in-tree, then run a command on a GPU box:  tools/dev/gp.sh [gpurun-timeout] '<command>'
set -euo pipefail
cd /root/repo
T=600
if [[ "$1" =~ ^[0-9]+$ ]]; then T=$1; shift; fi
python -m ppbo_amd.build | tail -1
exec timeout $((T + 900)) gpurun --timeout $T -- "$@"
