#!/bin/bash
# PMC passes of one program for named kernels (run on the GPU box through gpurun):
#   PROG="tools/dev/r6_line_time.py c3" KERNELS="line_cov_kernel dgemm_kernel" TAG=line bash tools/dev/r6_pmc.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG:-x}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/$PROG > $OUT/$tag.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE
run grbm GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 $GRAFT_REPO_ROOT/tools/dev/pmc_any.py $OUT $KERNELS > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
