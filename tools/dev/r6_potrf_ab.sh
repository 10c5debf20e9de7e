#!/bin/bash
# interleaved A/B of two builds of the library: tools/dev/lib_<name>.so copied over ppbo_amd/libppbo_hip.so in turn
cd $GRAFT_REPO_ROOT
export PPBO_SKIP_STAMP_CHECK=1
for i in 1 2 3; do
  for v in "$@"; do
    cp tools/dev/lib_$v.so ppbo_amd/libppbo_hip.so
    echo "$v: $(python tools/dev/r6_potrf_ab.py 2>/dev/null | tail -1)"
  done
done
