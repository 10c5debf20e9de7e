OUT=$GRAFT_REPO_ROOT/gpurun_out/r4o
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ord in 130 258 514 1026; do
  export PPBO_QF_ORDER=$ord
  python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-precision-report --steps 30 2>/dev/null | python3 -c "
import json,sys
l=json.loads([x for x in sys.stdin if x.startswith(chr(123))][-1]); print('order $ord', 'quadform ms', l['roofline']['avg_launch_ms'], 'frac', l['roofline']['frac'], 'ms_per_step', l['ms_per_step'])" >> $OUT/orders.txt
  timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f$ord -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-precision-report > /dev/null 2>&1
  timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/h$ord -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-precision-report > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT
  echo "order $ord" >> $OUT/orders.txt
  python3 tools/dev/pmc_any.py $OUT/f$ord quadform_kernel | grep -v "^==" >> $OUT/orders.txt
  python3 tools/dev/pmc_any.py $OUT/h$ord quadform_kernel | grep -v "^==" >> $OUT/orders.txt
  rm -rf $OUT/f$ord $OUT/h$ord
  cd /tmp
done
