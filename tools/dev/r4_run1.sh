set -x
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_sharded.py tests/test_gpu_dist.py tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4a/tests.log
python tools/predict_scaling.py > gpurun_out/r4a/scaling.txt 2>gpurun_out/r4a/scaling.err
python tools/fit_only.py c3 > gpurun_out/r4a/fit_only.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4a/fitprof -- python3 $GRAFT_REPO_ROOT/tools/fit_only.py c3 > $GRAFT_REPO_ROOT/gpurun_out/r4a/fitprof.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/r4a | head -30
python - <<'PY'
import csv,glob
fs=glob.glob('gpurun_out/r4a/fitprof/**/*kernel_trace.csv',recursive=True)
print(fs)
rows=list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last fit: find last gram_mfma kernel
idx=[i for i,r in enumerate(rows) if 'gram_mfma' in r['Kernel_Name']]
start=idx[-1]
t0=int(rows[start]['Start_Timestamp'])
prev=None
out=open('gpurun_out/r4a/fit_timeline.txt','w')
for r in rows[start:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    nm=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0][:50]
    gap=(s-prev)/1e3 if prev else 0
    prev=e
    out.write(f"{(s-t0)/1e3:10.1f} us dur {(e-s)/1e3:8.2f} gap {gap:7.2f} {nm}\n")
PY
tail -5 gpurun_out/r4a/tests.log
