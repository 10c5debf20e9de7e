import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppbo_amd.engine import get_engine
eng = get_engine(0)
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
g = dict(np.load(f"tests/golden/{name}.npz"))
S = eng.gram(g["X"], g["theta"], str(g["kernel"]))
Sinv = eng.pd_inverse(S)
f, st = eng.fit_fmap(Sinv, g["f_init"], int(g["m"]), g["theta"][0], gtol=float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6, verbose=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
print(st)
T, gr = eng.T_and_grad(Sinv, f, int(g["m"]), g["theta"][0])
print("gradnorm", float(torch.linalg.norm(gr)), "ref", float(g["gradnorm_fMAP"]), "max|f-fref|", np.abs(f.cpu().numpy() - g["fMAP"]).max())
Tr, grr = eng.T_and_grad(Sinv, g["fMAP"], int(g["m"]), g["theta"][0])
print("at ref fMAP: T", Tr, "gradnorm", float(torch.linalg.norm(grr)))
