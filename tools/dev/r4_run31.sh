cd $GRAFT_REPO_ROOT
python tools/dev/r4_nq_breakdown.py c3 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" > gpurun_out/nq_breakdown.txt
head -80 gpurun_out/nq_breakdown.txt
