#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box through gpurun).  Each --pmc set is its
# own rocprofv3 run (gfx950: 8 SQ slots, FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $OUT/$tag.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
# write path of the HBM-bound kernels (Gram, RFF): requests, in-flight level (average latency = LEVEL / WRREQ), credit stalls
run wstall TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
ls $OUT/*/ 
