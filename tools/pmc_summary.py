"""Collapse the rocprofv3 --pmc passes written by tools/pmc_quadform.sh into one JSON: per hot kernel the
per-launch average of every counter (dispatches of the same kernel averaged; multi-instance counters summed per
dispatch by rocprofv3 already), plus the derived figures DESIGN.md quotes.
usage: python tools/pmc_summary.py gpurun_out/pmc [gpurun_out/pmc_c2 ...] > profiles/r06_pmc_hot_kernels.json
(round 6: several roots -- the C3 passes of tools/pmc_quadform.sh and the C2 passes of tools/dev/r6_pmc.sh for the one-launch
scoring kernel; every entry carries the launch shape it was taken at, which bench.py checks before quoting its traffic)"""
import csv, glob, json, os, sys, collections

roots = sys.argv[1:]
HOT = {"quadform": "quadform_kernel", "fused_score": "fused_score_kernel", "kstar": "kstar_kernel", "gram_mfma": "gram_mfma_kernel", "score_kernel": "score_kernel",
       "rff_project": "rff_project_kernel", "potrf_step": "potrf_step_kernel"}
out = {k: {} for k in HOT}
paths = [p for root in roots for p in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)]
for path in paths:
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, counter) -> dispatch -> value
    grid = {}                                                                # (kernel, dispatch) -> grid size
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            for key, pat in HOT.items():
                if pat in name:
                    per[(key, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
                    grid[(key, row["Dispatch_Id"])] = int(row["Grid_Size"])
    for (key, ctr), d in per.items():
        # only the launches of the benchmark's own shape: the LARGEST grid of that kernel in the run (bench.py also
        # scores 256-512 candidate check batches and the Gram kernel runs at three sizes)
        gmax = max(grid[(key, disp)] for disp in d)
        if key == "gram_mfma":             # the C3 design: N = 2048 -> 32 * 33 / 2 = 528 workgroups of 512 threads
            want = 528 * 512                # (the run also builds Gram matrices at N = 512, 4096 and 8192)
            if any(grid[(key, disp)] == want for disp in d):
                gmax = want
            else:
                gmax = min(grid[(key, disp)] for disp in d)
        vals = [v for disp, v in d.items() if grid[(key, disp)] == gmax]
        if key == "quadform" and len(vals) > 1:
            vals = vals[1:]                      # drop the warm-up launch
        out[key][ctr] = sum(vals) / len(vals)
        out[key]["launches_averaged"] = len(vals)
SHAPES = {"quadform": {"N": 2048, "M": 65536}, "fused_score": {"N": 512, "M": 16384}}     # bench.py's C3 / C2 launches
for qkey in ("quadform", "fused_score"):
    q = out[qkey]
    if "SQ_INSTS_MFMA" not in q:
        continue
    q["shape"] = SHAPES[qkey]
    d = {}
    d["mfma_flops_per_launch"] = q["SQ_INSTS_MFMA"] * 2048.0          # v_mfma_f64_16x16x4: 16*16*4*2 flops
    if "SQ_VALU_MFMA_BUSY_CYCLES" in q and "GRBM_GUI_ACTIVE" in q:
        # busy cycles are summed over the 4 SIMDs of 256 CUs; GRBM_GUI_ACTIVE is summed over the 8 XCDs by the pass
        d["mfma_busy_frac"] = q["SQ_VALU_MFMA_BUSY_CYCLES"] / (q["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4) if q["GRBM_GUI_ACTIVE"] else None
    if "SQ_INSTS_VALU" in q:
        d["vector_instructions_besides_mfma"] = q["SQ_INSTS_VALU"] - q["SQ_INSTS_MFMA"]
    if "FETCH_SIZE" in q:
        d["fabric_read_bytes(FETCH_SIZE KB x1024 x2 gfx950 correction)"] = q["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in q:
        d["write_bytes"] = q["WRITE_SIZE"] * 1024.0
    if "TCC_HIT_sum" in q and "TCC_MISS_sum" in q:
        d["l2_hit_rate"] = q["TCC_HIT_sum"] / (q["TCC_HIT_sum"] + q["TCC_MISS_sum"])
    q["derived"] = d
for key in ("gram_mfma", "rff_project", "kstar"):
    k = out.get(key, {})
    if k.get("TCC_EA0_WRREQ_sum"):
        d = k.setdefault("derived", {})
        d["avg_write_latency_cycles(LEVEL/WRREQ)"] = k["TCC_EA0_WRREQ_LEVEL_sum"] / k["TCC_EA0_WRREQ_sum"]
        d["dram_credit_stall_cycles_per_write_request"] = k.get("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", 0.0) / k["TCC_EA0_WRREQ_sum"]
        d["write_requests_64B_fraction"] = k.get("TCC_EA0_WRREQ_64B_sum", 0.0) / k["TCC_EA0_WRREQ_sum"]
        if "WRITE_SIZE" in k:
            d["write_bytes"] = k["WRITE_SIZE"] * 1024.0
        if "FETCH_SIZE" in k:
            d["fabric_read_bytes(FETCH_SIZE KB x1024 x2 gfx950 correction)"] = k["FETCH_SIZE"] * 1024.0 * 2.0
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:                                   # which kernel sources the counters belong to (bench.py compares it with the tree)
    from ppbo_amd.build import _digest
    out["csrc_digest"] = _digest()
except Exception:
    out["csrc_digest"] = None
json.dump(out, sys.stdout, indent=1)
