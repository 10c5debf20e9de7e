"""Copies what tools/dev/r5_final_profiles.sh left in gpurun_out/r5p into profiles/r05_* (the tracked, judged copies)."""
import os, shutil
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
S, P = os.path.join(R, "gpurun_out", "r5p"), os.path.join(R, "profiles")
NOISE = ("RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl path", "/opt/amdgpu", "W2026", "E2026")


def clean(path):
    return "".join(l for l in open(path) if not l.startswith(NOISE))


def last_json_line(path):
    return [l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1] + "\n"


open(os.path.join(P, "r05_bench.json"), "w").write(last_json_line(os.path.join(S, "bench_c3.json")))
for c in ("c2", "c4", "c5"):
    open(os.path.join(P, f"r05_bench_{c}.json"), "w").write(last_json_line(os.path.join(S, f"bench_{c}.json")))
shutil.copy(os.path.join(S, "kernel_stats.csv"), os.path.join(P, "r05_bench_rocprofv3_kernel_stats.csv"))
shutil.copy(os.path.join(S, "pmc_hot_kernels.json"), os.path.join(P, "r05_pmc_hot_kernels.json"))
open(os.path.join(P, "r05_scaling_prediction.txt"), "w").write(clean(os.path.join(S, "scaling.txt")))
open(os.path.join(P, "r05_fit_kernel_trace.txt"), "w").write(
    "# rocprofv3 --kernel-trace -- python3 tools/fit_only.py c3 z (four ppbo_gp_fit calls from a whitened start: the two-stream form), tools/dev/trace_summary.py: per kernel count / avg / min / max (us) and\n"
    "# the timeline of the LAST fit; wall clock of the same script without the profiler at the end\n"
    + clean(os.path.join(S, "fit_trace.txt")) + "\n# wall clock without the profiler (tools/fit_only.py c3 z):\n" + clean(os.path.join(S, "fit_wall.txt")))
open(os.path.join(P, "r05_query_kernel_trace.txt"), "w").write(
    "# rocprofv3 --kernel-trace -- python3 tools/dev/r5_query_trace.py c3 EI-EXT: one whole PPBO query at the C3 shape through the drop-in objects, three times\n"
    "# (update_model = ppbo_gp_fit + mu_star's three trials in one ppbo_mean_search_multi; next_query EI-EXT = 1000 lines; one Hsampler cycle at F = 4096);\n"
    "# per kernel count / avg / min / max (us) over the three repetitions; wall clock per phase with and without the profiler below\n"
    + clean(os.path.join(S, "query_trace.txt")) + "\n# wall clock under the profiler:\n" + clean(os.path.join(S, "query_wall_traced.txt"))
    + "\n# wall clock without the profiler:\n" + clean(os.path.join(S, "query_wall.txt")))
print("collected into", P)
