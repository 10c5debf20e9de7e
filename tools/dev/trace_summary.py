"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name count / avg / min / max duration and avg preceding gap;
optionally the timeline of the last `n` launches.   python tools/dev/trace_summary.py <dir-or-csv> [n]"""
import csv, glob, os, sys
p = sys.argv[1]
fs = [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in fs:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg, prev = {}, None
tl = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    nm = nm.split("<")[0][-44:] + (("<" + r["Kernel_Name"].split("<", 1)[1][:30]) if "<" in r["Kernel_Name"] and "quadform" in nm else "")
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    a = agg.setdefault(nm, [0, 0.0, 1e30, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] = min(a[2], (e - s) / 1e3); a[3] = max(a[3], (e - s) / 1e3); a[4] += gap
    tl.append((s, (e - s) / 1e3, gap, nm, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
print(f"{'count':>7} {'avg us':>10} {'min us':>10} {'max us':>10} {'avg gap':>9}  kernel")
for k, (n, d, mn, mx, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:7d} {d / n:10.2f} {mn:10.2f} {mx:10.2f} {gp / n:9.2f}  {k}")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if n:
    t0 = tl[-n][0]
    for s, d, gp, nm, gx in tl[-n:]:
        print(f"{(s - t0) / 1e3:10.1f} us  dur {d:9.2f}  gap {gp:8.2f}  {nm}")
