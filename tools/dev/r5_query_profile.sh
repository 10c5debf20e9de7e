OUT=$GRAFT_REPO_ROOT/gpurun_out/r5q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for acq in EI-EXT; do
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$acq -- python3 $GRAFT_REPO_ROOT/tools/dev/r5_query_trace.py c3 $acq > $OUT/wall_$acq.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/dev/trace_summary.py $OUT/tr_$acq 0 > $OUT/summary_$acq.txt
done
python3 - <<'PY'
import csv, glob, os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5q'
for acq in ('EI-EXT',):
    fs=glob.glob(out+f'/tr_{acq}/**/*kernel_trace.csv', recursive=True)
    rows=[]
    for f in fs: rows+=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    # split into phases by gaps > 1.5 ms; keep the last 3 phases (last rep)
    ph=[[]]
    prev=None
    for r in rows:
        s=int(r['Start_Timestamp'])
        if prev and s-prev>1.5e6: ph.append([])
        ph[-1].append(r); prev=int(r['End_Timestamp'])
    with open(out+f'/phases_{acq}.txt','w') as fo:
        for p in ph[-3:]:
            t0=int(p[0]['Start_Timestamp']); t1=int(p[-1]['End_Timestamp'])
            busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in p)
            fo.write(f"=== phase: {len(p)} launches, span {(t1-t0)/1e3:.1f} us, kernel time {busy/1e3:.1f} us\n")
            agg={}
            for r in p:
                nm=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('<')[0][-40:]
                a=agg.setdefault(nm,[0,0.0]); a[0]+=1; a[1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
            for k,(n,d) in sorted(agg.items(), key=lambda kv:-kv[1][1]):
                fo.write(f"{n:6d} {d:10.1f} us  {k}\n")
            if p is ph[-3]:
                with open(out+f'/timeline_update_model_{acq}.txt','w') as ft:
                    pe=None
                    for r in p:
                        st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
                        nm=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('<')[0][-40:]
                        ft.write(f"{(st-t0)/1e3:9.1f} us  dur {(en-st)/1e3:8.2f}  gap {((st-pe)/1e3 if pe else 0):7.2f}  {nm}\n")
                        pe=en
            prev=None
            for r in p:
                st=int(r['Start_Timestamp']); en=int(r['End_Timestamp'])
                if prev is not None and st-prev>15000:
                    nm=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0].split('<')[0][-40:]
                    fo.write(f"   gap {(st-prev)/1e3:7.1f} us before {nm} at {(st-t0)/1e3:8.1f} us\n")
                prev=en
PY
rm -rf $OUT/tr_*
