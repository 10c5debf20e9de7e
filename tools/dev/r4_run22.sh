cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4q
for ch in 8 16 24 40 48 56 64 70 80 96; do PPBO_LINE_Y_OPT=1 PPBO_LINE_Y_CHUNK=$ch python tools/dev/r4_liney.py 2>&1 | tail -1; done | tee gpurun_out/r4q/liney2.txt
