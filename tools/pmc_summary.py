"""Collapse the rocprofv3 --pmc passes written by tools/pmc_quadform.sh into one JSON: per hot kernel the
per-launch average of every counter (dispatches of the same kernel averaged; multi-instance counters summed per
dispatch by rocprofv3 already), plus the derived figures DESIGN.md quotes.
usage: python tools/pmc_summary.py gpurun_out/pmc > profiles/rNN_pmc_hot_kernels_vK.json"""
import csv, glob, json, os, sys, collections

root = sys.argv[1]
HOT = {"quadform": "quadform_kernel", "kstar": "kstar_kernel", "gram_mfma": "gram_mfma_kernel", "score_kernel": "score_kernel"}
out = {k: {} for k in HOT}
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, counter) -> dispatch -> value
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            for key, pat in HOT.items():
                if pat in name:
                    per[(key, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (key, ctr), d in per.items():
        vals = list(d.values())
        if key == "quadform" and len(vals) > 1:
            vals = vals[1:]                      # drop the warm-up launch
        out[key][ctr] = sum(vals) / len(vals)
q = out["quadform"]
if "SQ_INSTS_MFMA" in q:
    d = {}
    d["mfma_flops_per_launch"] = q["SQ_INSTS_MFMA"] * 2048.0          # v_mfma_f64_16x16x4: 16*16*4*2 flops
    if "SQ_VALU_MFMA_BUSY_CYCLES" in q and "SQ_BUSY_CYCLES" in q and "GRBM_GUI_ACTIVE" in q:
        # busy cycles are summed over the 4 SIMDs of 256 CUs; GRBM_GUI_ACTIVE is averaged over 8 XCDs by the pass
        d["mfma_busy_frac"] = q["SQ_VALU_MFMA_BUSY_CYCLES"] / (q["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4) if q["GRBM_GUI_ACTIVE"] else None
    if "FETCH_SIZE" in q:
        d["fabric_read_bytes(FETCH_SIZE KB x1024 x2 gfx950 correction)"] = q["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in q:
        d["write_bytes"] = q["WRITE_SIZE"] * 1024.0
    if "TCC_HIT_sum" in q and "TCC_MISS_sum" in q:
        d["l2_hit_rate"] = q["TCC_HIT_sum"] / (q["TCC_HIT_sum"] + q["TCC_MISS_sum"])
    q["derived"] = d
json.dump(out, sys.stdout, indent=1)
