"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files under a directory, for kernels whose name
contains one of the given substrings (largest grid of each).  usage: python tools/dev/pmc_any.py <dir> name1 [name2 ...]"""
import csv, glob, os, sys, collections
root, pats = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(dict))
grid = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        nm = row.get("Kernel_Name", "")
        for p in pats:
            if p in nm:
                d = acc[p][row["Counter_Name"]]
                d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
                grid[(p, row["Dispatch_Id"])] = int(row["Grid_Size"])
for p in pats:
    print(f"== {p}")
    for ctr, d in sorted(acc[p].items()):
        gmax = max(grid[(p, k)] for k in d)
        vals = [v for k, v in d.items() if grid[(p, k)] == gmax]
        print(f"  {ctr:36s} {sum(vals) / len(vals):16.1f}   (launches {len(vals)}, grid {gmax})")
