# Regenerates every round-6 artefact that is quoted with a csrc digest (run through gpurun; results land in gpurun_out/r6p)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6p
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# the counter passes first: bench.py quotes the traffic of the dominant kernel from the newest profiles/r0*_pmc_hot_kernels.json
# and marks it current only when that capture carries the digest of the library it is running
bash tools/pmc_quadform.sh > $OUT/pmc.log 2>&1
PROG="bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --no-precision-report" KERNELS="fused_score_kernel" TAG=c2 bash tools/dev/r6_pmc.sh > $OUT/pmc_c2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/pmc_c2 > $OUT/pmc_hot_kernels.json
rm -rf gpurun_out/pmc gpurun_out/pmc_c2
cp $OUT/pmc_hot_kernels.json profiles/r06_pmc_hot_kernels.json
python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python bench.py --config c2 --no-secondary > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --config c4 --no-secondary > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --config c5 --no-secondary --steps 20 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python tools/predict_scaling.py > $OUT/scaling.txt 2> $OUT/scaling.err
python tools/fit_only.py c3 z > $OUT/fit_wall.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-precision-report > $OUT/stats_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c2 -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --no-cpu-baseline --no-secondary --no-precision-report > $OUT/stats_bench_c2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/fitprof -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 z > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/dev/trace_summary.py $OUT/fitprof 130 > $OUT/fit_trace.txt
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/stats_c2 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_c2.csv
rm -rf $OUT/fitprof $OUT/stats $OUT/stats_c2
