"""Probe: the per-query phases of bench.py's per_query_breakdown (C2 shape) a few times each, for
rocprofv3 --kernel-trace --stats (which kernels a PPBO query spends its device time in)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
print(bench.per_query_breakdown(torch, reps=5))
