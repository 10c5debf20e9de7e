#!/bin/bash
# round 6: everything quoted in profiles/r06_potrf_panel_step.txt, one box
cd $GRAFT_REPO_ROOT
echo "## tools/dev/bcast_bench.bin"; tools/dev/bcast_bench.bin
echo "## tools/dev/slab_dev.bin (the final slab factor alone, one wavefront; others: 0 idle, 1 fp64 vector work, 2 MFMA + LDS work beside it, 3 = 256 workgroups at once)"; tools/dev/slab_dev.bin | grep "rep 2"
echo "## tools/dev/panel_dev0.bin (the kernel at the start of the round)"; tools/dev/panel_dev0.bin | grep "rep 2\|intervals"
echo "## tools/dev/panel_dev.bin (final)"; tools/dev/panel_dev.bin | grep "rep 2\|intervals"
echo "## tools/dev/r6_potrf_ab.sh before after (median of 30 potrf calls incl. the host's launch + wait, of 10 fits), ms"; bash tools/dev/r6_potrf_ab.sh before after
cp tools/dev/lib_after.so ppbo_amd/libppbo_hip.so
