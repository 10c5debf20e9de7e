#!/bin/bash
# PMC passes for the one-launch scoring kernel at the C2 shape (run on the GPU box through gpurun)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fused
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${FUSED_ARGS:-512 6 31 16384 20}"
run() { tag=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/dev/r6_fused_run.py $ARGS > $OUT/$tag.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE
run sq3 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
run grbm GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_any.py $OUT fused_score_kernel > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
