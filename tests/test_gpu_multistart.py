"""a-8, the DEFAULT f_MAP path (whitened L-BFGS + trust-region finisher) against the REFERENCE's own trust-exact runs
from several prior draws (src/gp_model.py:372-389: the reference restarts from random vectors and keeps the best,
because T is not concave).  Fixtures: tools/make_golden_r4.py (the reference itself, one trial per stored start).

  multistart_c2 / multistart_c4   BASELINE configs 2 and 4 (sigma / sigma_f = 0.01 / 0.0067): the reference reaches ONE
      maximum from all 8 starts; so must the default path, from every start, at the reference's own tolerance.
  multistart_mm_se / multistart_mm_rq   two small models at sigma / sigma_f ~ 1e-3 where the reference's runs end in
      3 resp. 7 DIFFERENT strict local maxima over the 8 starts: there no local method can be asked for "the" basin;
      what is asked is what the reference's restart rule relies on -- every result is a strict local maximum, and the
      best over the 8 starts is at least the reference's best."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def host(t):
    return t.cpu().numpy()


@pytest.fixture(scope="module")
def eng():
    from ppbo_amd.engine import get_engine
    return get_engine(0)


def _setup(eng, z, kernel):
    S = eng.gram(z["X"], z["theta"], kernel)
    Sinv, L = eng.pd_inverse_chol(S)
    return S, Sinv, L


@pytest.mark.parametrize("method", ["whitened", "trust-region"])
@pytest.mark.parametrize("name", ["c2", "c4"])
def test_every_start_reaches_the_references_maximum(eng, name, method):
    z, g = load_golden(f"multistart_{name}"), load_golden(name)
    assert np.array_equal(z["X"], g["X"])
    kernel, m, sig = str(g["kernel"]), int(z["m"]), float(z["theta"][0])
    _, Sinv, L = _setup(eng, z, kernel)
    K = z["f_init"].shape[0]
    assert K == 8 and np.ptp(z["T"]) <= 1e-8 * abs(z["T"]).max()      # the reference itself: one maximum from all starts
    same_basin = 0
    for k in range(K):
        f, st = eng.fit_fmap(Sinv, z["f_init"][k], m, sig, gtol=1e-4, L=L if method == "whitened" else None)
        fh = host(f)
        assert st["converged"] and st["gradnorm"] < 1e-4, (k, st)                    # the reference's stopping rule
        assert st["T"] >= float(z["T"][k]) - 1e-7 * abs(float(z["T"][k])), (k, st["T"], float(z["T"][k]))
        # same basin: within 1e-5 max|f| plus the two fits' own Newton gaps |P grad| (both stop at |grad| < 1e-4)
        post = eng.posterior(z["X"], z["theta"], kernel, Sinv, fh, m, want_P=True)
        P = host(post.P)
        gaps = 0.0
        for fv in (fh, z["fMAP"][k]):
            _, gr = eng.T_and_grad(Sinv, fv, m, sig)
            gaps += np.abs(P @ host(gr)).max()
        d = np.abs(fh - z["fMAP"][k]).max()
        ok = d <= 1e-5 * np.abs(z["fMAP"][k]).max() + 1.5 * gaps
        same_basin += int(ok)
        print(f"{name} {method} start {k}: T {st['T']:.10f} (ref {float(z['T'][k]):.10f}) |grad| {st['gradnorm']:.2e} "
              f"(ref {float(z['gradnorm'][k]):.2e}) max|f - f_ref| {d:.2e} gaps {gaps:.2e} evals {st['lbfgs_evals']} "
              f"chol {st['n_cholesky']} {'same basin' if ok else 'OTHER BASIN'}")
    print(f"{name} {method}: {same_basin} of {K} starts end in the reference's basin")
    assert same_basin >= 7


@pytest.mark.parametrize("name", ["mm_se", "mm_rq"])
def test_multimodal_regime_restarts_are_as_good_as_the_references(eng, name):
    z = load_golden(f"multistart_{name}")
    kernel, m, sig = str(z["kernel"]), int(z["m"]), float(z["theta"][0])
    _, Sinv, L = _setup(eng, z, kernel)
    K = z["f_init"].shape[0]
    n_ref_maxima = len(np.unique(np.round(z["T"], 6)))
    assert n_ref_maxima >= 3                       # the fixture IS multimodal: the reference's own runs disagree
    ours, agree = [], 0
    for k in range(K):
        f, st = eng.fit_fmap(Sinv, z["f_init"][k], m, sig, gtol=1e-4, L=L)
        assert st["converged"] and st["gradnorm"] < 1e-4, (k, st)
        # a strict local maximum of T: Sigma^-1 - Lambda(f) is positive definite there (ppbo_posterior factors it and
        # raises otherwise)
        eng.posterior(z["X"], z["theta"], kernel, Sinv, f, m)
        ours.append(st["T"])
        agree += int(abs(st["T"] - float(z["T"][k])) <= 1e-6 * max(1.0, abs(float(z["T"][k]))))
        print(f"{name} start {k}: T {st['T']:.8f} (reference {float(z['T'][k]):.8f})")
    print(f"{name}: the reference's 8 runs end in {n_ref_maxima} different maxima; {agree} of 8 default-path runs end where "
          f"the reference's run from the same start does; best of 8: {max(ours):.8f} vs reference {z['T'].max():.8f}")
    # the reference's guard in this regime is best-of-restarts (gp_model.py:385-387): ours must not be worse -- than the
    # best T ANY run of the reference has reached on this model (T_best_known: this file's runs and the earlier,
    # multi-threaded-BLAS ones whose starts differed by 5e-10 and fell into other basins; tools/make_golden_r4.py now pins
    # one BLAS thread and reproduces the file bit for bit)
    bar = float(z["T_best_known"])
    assert bar >= float(z["T"].max())
    assert max(ours) >= bar - 1e-6 * max(1.0, abs(bar)), (max(ours), bar)
    # and every reference maximum is a fixed point of our optimiser: started AT the reference's result it stays there
    for k in range(K):
        f, st = eng.fit_fmap(Sinv, z["fMAP"][k], m, sig, gtol=1e-4, L=L)
        assert abs(st["T"] - float(z["T"][k])) <= 1e-6 * max(1.0, abs(float(z["T"][k]))), (k, st["T"], float(z["T"][k]))
