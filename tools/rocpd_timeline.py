"""Print the kernel timeline (start offset, duration, gap to previous) from a rocprofv3 rocpd database.
usage: python tools/rocpd_timeline.py <results.db> [name-substring] [max-rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
t0 = rows[0][1]
prev_end = None
agg = {}
for name, s, e, gx, wx in rows:
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][-40:]
    d = (e - s) / 1e3
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    a = agg.setdefault(short, [0, 0.0, 0.0])
    a[0] += 1; a[1] += d; a[2] += gap
    if pat in name and lim > 0:
        print(f"{(s - t0) / 1e3:12.1f} us  dur {d:8.2f}  gap {gap:7.2f}  wgs {gx // max(wx, 1):6d}  {short}")
        lim -= 1
print("---- totals: count, sum dur (us), sum preceding gap (us)")
for k, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:7d} {d:12.1f} {g:12.1f}  {k}")
