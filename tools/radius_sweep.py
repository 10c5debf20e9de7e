"""GPU box: factorizations / iterations of the f_MAP fit per fixture for several initial trust radii."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
for name in sys.argv[1:] or ["smoke", "rq", "cam_small", "c2", "c4", "c3"]:
    g = dict(np.load(f"tests/golden/{name}.npz"))
    m, sig = int(g["m"]), float(g["theta"][0])
    Sinv = eng.pd_inverse(eng.gram(g["X"], g["theta"], str(g["kernel"])))
    fn = float(np.linalg.norm(g["f_init"]))
    for r0 in (1.0, 4.0, 0.25 * fn, 0.5 * fn, fn, 1000.0):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f, st = eng.fit_fmap(Sinv, g["f_init"], m, sig, gtol=1e-4, initial_radius=r0)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        d = np.abs(f.cpu().numpy() - g["fMAP"]).max() / np.abs(g["fMAP"]).max()
        print(f"{name:9s} N={g['X'].shape[0]:5d} |f0|={fn:7.2f} r0={r0:8.2f}: it {st['iterations']:4d} chol {st['n_cholesky']:5d} "
              f"conv {int(st['converged'])} T {st['T']:.6f} |f-fref|/max {d:.1e}  {dt:8.1f} ms")
