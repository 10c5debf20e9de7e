"""Per-kernel statistics (calls, total, average, min, max, share) from a rocprofv3 rocpd database, as CSV --
the same table `rocprofv3 --stats` prints, for the runs whose output format was the database.
usage: python tools/rocpd_stats.py <results.db> > kernel_stats.csv"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    a = agg.setdefault(name, [])
    a.append((e - s) / 1e3)
tot = sum(sum(v) for v in agg.values())
print("Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%.1f,%.3f,%.3f,%.3f,%.2f' % (k.replace('"', "'"), len(v), sum(v), sum(v) / len(v), min(v), max(v), 100.0 * sum(v) / tot))
