cd $GRAFT_REPO_ROOT
timeout 900 python tools/dev/r4_soak.py 2000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tee /tmp/soak.txt
