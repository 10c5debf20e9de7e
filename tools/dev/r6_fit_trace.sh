#!/bin/bash
# round 6: kernel trace of the C3 fit alone (the fit portion of r6_final_profiles.sh) -> gpurun_out/r6_fit_trace.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6fit
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/fitprof -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 z > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py $OUT/fitprof 400 > gpurun_out/r6_fit_trace.txt
rm -rf $OUT
