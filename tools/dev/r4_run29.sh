cd $GRAFT_REPO_ROOT
for q in 4 8 16; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q python tools/dev/r4_evidence_workers.py c3 2>&1 | grep -E "workers" | tail -4; done
