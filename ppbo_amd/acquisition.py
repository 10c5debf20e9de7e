"""Drop-in next_query(): the reference's acquisition dispatcher (src/acquisition.py:9-65) with the
Monte-Carlo line acquisitions evaluated in batches on the GPU (ppbo_line_acq).

EI / varmax of a projective query (xi, x): 70 noisy-equispaced grid points on the line
(src/acquisition.py:72-81, 170-178), posterior mean + 70x70 covariance, S draws, max over the
line.  The reference evaluates one line per Python call and searches (xi, x) with GPyOpt's
Bayesian optimisation (25 evaluations); here B lines are scored per launch with common random
numbers, and the outer search is a batched random search over the same domain.
"""
from __future__ import annotations

import time

import numpy as np

LINE_POINTS = 70          # acquisition.py:73,171
SEARCH_LINES = 256        # lines per batched outer search (the reference's BO spends 5 + BO_maxiter)


def _line_scores(xis, xs, GP_model, mc_samples):
    """EI and varmax of B lines in one device call.  Grid noise and z come from the global NumPy stream."""
    FP = GP_model.FP
    grids = np.stack([FP.xi_grid(xi=xi, x=x, alpha_grid_distribution="equispaced", alpha_star=None, m=LINE_POINTS,
                                 is_scaled=True) for xi, x in zip(xis, xs)])
    z = np.random.standard_normal((mc_samples, LINE_POINTS))
    sf2 = float(GP_model.theta[2]) ** 2
    ei, vm = GP_model.eng.line_acq(GP_model._post, grids, z, GP_model.mustar, GP_model.COVARIANCE_SHRINKAGE,
                                   jitter=1e-10 * sf2)
    return ei.cpu().numpy(), vm.cpu().numpy()


def EI(xi, x, GP_model, mc_samples):
    return float(_line_scores([xi], [x], GP_model, mc_samples)[0][0])


def varmax(xi, x, GP_model, mc_samples):
    return float(_line_scores([xi], [x], GP_model, mc_samples)[1][0])


def perturbate_zerocoordinates(x, coords):
    x = np.asarray(x, dtype=float)
    sub = x[coords]
    sub[sub == 0] = 1e-7
    x[coords] = sub
    return x


def _search_joint(xi_dims, GP_model, PPBO_settings, which, fixed_x=None):
    """Batched replacement of maximize_EI / maximize_EI_fixed_x / maximize_varmax
    (src/acquisition.py:91-131, 189-206): xi free on xi_dims, x free on the complement."""
    D = GP_model.D
    x_dims = [i for i in range(D) if i not in xi_dims]
    U = np.random.uniform(0, 1, (SEARCH_LINES, D))
    xis = np.zeros((SEARCH_LINES, D))
    xs = np.zeros((SEARCH_LINES, D))
    xis[:, xi_dims] = U[:, xi_dims]
    xs[:, x_dims] = U[:, x_dims] if fixed_x is None else fixed_x[x_dims]
    ei, vm = _line_scores(xis, xs, GP_model, PPBO_settings.mc_samples)
    b = int(np.argmax(ei if which == "ei" else vm))
    return perturbate_zerocoordinates(xis[b], xi_dims), perturbate_zerocoordinates(xs[b], x_dims)


def maximize_EI(xi_dims, GP_model, PPBO_settings):
    return _search_joint(xi_dims, GP_model, PPBO_settings, "ei")


def maximize_EI_fixed_x(xi_dims, GP_model, PPBO_settings):
    return _search_joint(xi_dims, GP_model, PPBO_settings, "ei", fixed_x=GP_model.xstar.copy())


def maximize_varmax(xi_dims, GP_model, PPBO_settings):
    return _search_joint(xi_dims, GP_model, PPBO_settings, "vm")


def maximize_varmax_given_xi(xi, GP_model, PPBO_settings):
    D = GP_model.D
    xs = np.random.uniform(0, 1, (SEARCH_LINES, D))
    xs[:, np.where(np.asarray(xi) != 0)[0]] = 0.0
    _, vm = _line_scores([xi] * SEARCH_LINES, xs, GP_model, PPBO_settings.mc_samples)
    return xs[int(np.argmax(vm))]


def EId_xstar(GP_model, mc_samples):
    """Coordinate direction maximising EI with x = xstar (src/acquisition.py:132-145); all D lines in one launch."""
    D = GP_model.D
    xis = np.eye(D)
    xs = np.tile(GP_model.xstar, (D, 1))
    xs[np.arange(D), np.arange(D)] = 0.0
    ei, _ = _line_scores(xis, xs, GP_model, mc_samples)
    return xis[int(np.argmax(ei))]


def EId_integrate(GP_model, mc_samples):
    """Coordinate direction maximising EI with x integrated out over 50 uniform draws
    (src/acquisition.py:146-163); the D*50 lines are one launch."""
    D, reps = GP_model.D, 50
    xis = np.repeat(np.eye(D), reps, axis=0)
    xs = np.random.uniform(0, 1, (D * reps, D))
    xs[np.arange(D * reps), np.repeat(np.arange(D), reps)] = 0.0
    ei, _ = _line_scores(xis, xs, GP_model, mc_samples)
    return np.eye(D)[int(np.argmax(ei.reshape(D, reps).mean(axis=1)))]


def random_next_xi(PPBO_settings):
    D = PPBO_settings.D
    nz = list(set(np.random.choice(D, D - 1, replace=True)))
    xi = np.zeros(D)
    xi[nz] = np.random.uniform(0, 1, len(nz))
    return xi


def _cycle(PPBO_settings):
    d = int(PPBO_settings.dim_query_prev_iter + 1)
    if d > PPBO_settings.D:
        d = 1
    PPBO_settings.dim_query_prev_iter = d
    return d


def PCD_next_xi(PPBO_settings):
    return np.eye(PPBO_settings.D)[:, _cycle(PPBO_settings) - 1]


def EXT_next_xi(PPBO_settings, GP_model):
    xi = GP_model.xstar.copy()
    xi[xi == 0] = 1e-7
    xi[_cycle(PPBO_settings) - 1] = 0
    return xi


def next_x_given_xi(xi, GP_model, PPBO_settings):
    free = list(np.where(np.asarray(xi) == 0)[0])
    x = np.zeros(PPBO_settings.D)
    mode = PPBO_settings.x_acquisition_function
    if mode == "exploit":
        x[free] = GP_model.xstar[free]
    elif mode == "varmax":
        x = maximize_varmax_given_xi(xi, GP_model, PPBO_settings)
    elif mode == "random":
        x[free] = np.random.uniform(0, 1, len(free))
    else:
        print("Invalid acquisition function selected!")
        return None
    return perturbate_zerocoordinates(x, free)


def next_query(PPBO_settings, GP_model, unscale=True):
    start = time.time()
    acq = PPBO_settings.xi_acquisition_function
    if acq in ("EI", "EXR", "EI-FIXEDX"):
        xi_dims = list((np.array(PPBO_settings.xi_dims_prev_iter) + 1) % PPBO_settings.D)
        PPBO_settings.xi_dims_prev_iter = xi_dims
    if acq == "EI":
        xi_next, x_next = maximize_EI(xi_dims, GP_model, PPBO_settings)
    elif acq == "EI-FIXEDX":
        xi_next, x_next = maximize_EI_fixed_x(xi_dims, GP_model, PPBO_settings)
    elif acq == "EXR":
        xi_next, x_next = maximize_varmax(xi_dims, GP_model, PPBO_settings)
    elif acq in ("EI-EXT-FAST", "EI-VARMAX-FAST"):
        xi_next = EId_xstar(GP_model, PPBO_settings.mc_samples)
        x_next = next_x_given_xi(xi_next, GP_model, PPBO_settings)
    elif acq in ("EI-EXT", "EI-VARMAX"):
        xi_next = EId_integrate(GP_model, PPBO_settings.mc_samples)
        x_next = next_x_given_xi(xi_next, GP_model, PPBO_settings)
    elif acq in ("COORDINATE-VARMAX", "PCD"):
        xi_next = PCD_next_xi(PPBO_settings)
        x_next = next_x_given_xi(xi_next, GP_model, PPBO_settings)
    elif acq == "RAND":
        xi_next = random_next_xi(PPBO_settings)
        x_next = next_x_given_xi(xi_next, GP_model, PPBO_settings)
    elif acq == "EXT":
        xi_next = EXT_next_xi(PPBO_settings, GP_model)
        x_next = next_x_given_xi(xi_next, GP_model, PPBO_settings)
    else:
        print("Invalid acquisition function name!")
        return 0
    if GP_model.verbose:
        print("Evaluation of the acquisition function took " + str(time.time() - start) + " seconds.")
    xi_next = np.abs(xi_next) / np.max(np.abs(xi_next))          # normalise before unscaling (acquisition.py:58)
    if not unscale:
        return xi_next, x_next
    xi_next = GP_model.FP.unscale(xi_next, retain_0_values=True)
    x_next = GP_model.FP.unscale(x_next, retain_0_values=True)
    if GP_model.verbose:
        print("Next query: (xi,x) = " + str((xi_next, x_next)))
    return xi_next, x_next
