"""Soak of the host-polled device loops (whitened fit, omega_MAP, sharded search record): many repetitions, every result
compared bit for bit with the first one, the slowest call reported (a call that fell back to its 5 s / 2 s timeout path
would show up here)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ppbo_amd.engine import get_engine, SCORE_POINTWISE_EI
eng = get_engine(0)
G = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
for name in ("smoke", "c2", "c3"):
    g = dict(np.load(os.path.join(G, f"{name}.npz")))
    X, th, m, kern = eng.dev(g["X"]), g["theta"], int(g["m"]), str(g["kernel"])
    f0 = eng.dev(g["f_init"])
    z0 = eng.dev(np.random.default_rng(2).standard_normal(X.shape[0]))
    whitened_start = name == "c3"           # N >= 1024: the two-stream form of ppbo_gp_fit
    ref, worst, n = None, 0.0, (reps if name != "c3" else reps // 4)
    t_all = time.perf_counter()
    for k in range(n):
        t0 = time.perf_counter()
        r = eng.gp_fit(X, th, kern, m, z0, start_is_whitened=True) if whitened_start else eng.gp_fit(X, th, kern, m, f0)
        fm = r["fMAP"].cpu().numpy()
        dt = time.perf_counter() - t0
        worst = max(worst, dt)
        if ref is None: ref = fm
        elif not np.array_equal(ref, fm): raise SystemExit(f"{name}: fit {k} differs from fit 0")
    print(f"{name}: {n} fits bitwise equal, mean {(time.perf_counter() - t_all) / n * 1e3:.3f} ms, slowest {worst * 1e3:.2f} ms", flush=True)
    post = r["post"]
    Xc = eng.dev(np.random.default_rng(1).random((4096, X.shape[1])))
    mustar = float(np.max(g["mu"]))
    ref, worst = None, 0.0
    for k in range(reps * 4):
        t0 = time.perf_counter()
        out = eng.search_sharded(post, Xc, SCORE_POINTWISE_EI, mustar, 7)
        worst = max(worst, time.perf_counter() - t0)
        if ref is None: ref = out
        elif out != ref: raise SystemExit(f"{name}: search {k} differs")
    print(f"{name}: {reps * 4} sharded-search steps equal, slowest {worst * 1e3:.2f} ms", flush=True)
rng = np.random.default_rng(2)
for F, N, m in ((70, 54, 5), (1000, 512, 31), (4096, 2048, 31)):
    Phi = eng.dev(rng.standard_normal((F, N)) * 0.2 * np.sqrt(70.0 / F))
    om0 = rng.standard_normal(F)
    ref, worst, n = None, 0.0, max(reps // 4, 10)
    for k in range(n):
        t0 = time.perf_counter()
        out = eng.rff_omega_map(Phi, om0, m, 0.3, maxiter=500, gtol=1e-6)
        worst = max(worst, time.perf_counter() - t0)
        if ref is None: ref = out
        elif not (np.array_equal(out[0], ref[0]) and out[1:] == ref[1:]): raise SystemExit(f"omega_map F={F}: run {k} differs")
    print(f"omega_map F={F}: {n} runs bitwise equal ({ref[3]} iterations, |g| {ref[2]:.2e}), slowest {worst * 1e3:.2f} ms", flush=True)
print("soak ok")
