"""Copies what tools/dev/r6_final_profiles.sh left in gpurun_out/r6p into profiles/r06_* (the tracked, judged copies)."""
import os, shutil
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S, P = os.path.join(R, "gpurun_out", "r6p"), os.path.join(R, "profiles")
NOISE = ("RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl path", "/opt/amdgpu", "W2026", "E2026")


def clean(path):
    return "".join(l for l in open(path) if not l.startswith(NOISE))


def last_json_line(path):
    return [l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1] + "\n"


open(os.path.join(P, "r06_bench.json"), "w").write(last_json_line(os.path.join(S, "bench_c3.json")))
for c in ("c2", "c4", "c5"):
    open(os.path.join(P, f"r06_bench_{c}.json"), "w").write(last_json_line(os.path.join(S, f"bench_{c}.json")))
shutil.copy(os.path.join(S, "kernel_stats.csv"), os.path.join(P, "r06_bench_rocprofv3_kernel_stats.csv"))
shutil.copy(os.path.join(S, "kernel_stats_c2.csv"), os.path.join(P, "r06_bench_c2_rocprofv3_kernel_stats.csv"))
shutil.copy(os.path.join(S, "pmc_hot_kernels.json"), os.path.join(P, "r06_pmc_hot_kernels.json"))
open(os.path.join(P, "r06_scaling_prediction.txt"), "w").write(clean(os.path.join(S, "scaling.txt")))
open(os.path.join(P, "r06_fit_kernel_trace.txt"), "w").write(
    "# rocprofv3 --kernel-trace -- python3 tools/fit_only.py c3 z (four ppbo_gp_fit calls from a whitened start: the two-stream form), tools/dev/trace_summary.py: per kernel count / avg / min / max (us) and\n"
    "# the timeline of the LAST fit; wall clock of the same script without the profiler at the end\n"
    + clean(os.path.join(S, "fit_trace.txt")) + "\n# wall clock without the profiler (tools/fit_only.py c3 z):\n" + clean(os.path.join(S, "fit_wall.txt")))
print("collected into", P)
