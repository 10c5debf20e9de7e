"""Drop-in next_query(): the reference's acquisition dispatcher (src/acquisition.py:9-65) with the
Monte-Carlo line acquisitions evaluated in batches on the GPU (ppbo_line_acq).

EI / varmax of a projective query (xi, x): 70 noisy-equispaced grid points on the line
(src/acquisition.py:72-81, 170-178), posterior mean + 70x70 covariance, S draws, max over the
line.  The reference evaluates one line per Python call and searches (xi, x) with GPyOpt's
Bayesian optimisation (5 + BO_maxiter evaluations); here B lines are scored per launch with common
random numbers, and the outer search is a batched evolutionary search over the same domain whose
number of refinement rounds follows PPBO_settings.BO_maxiter (_batched_search).
"""
from __future__ import annotations

import time

import numpy as np

LINE_POINTS = 70          # acquisition.py:73,171
SEARCH_LINES = 256        # lines of the first (uniform) round of an outer search
REFINE_LINES = 64         # lines per refinement round
REFINE_PARENTS = 8        # incumbents each refinement round perturbs
# the reference's GPyOpt BO spends 5 initial + PPBO_settings.BO_maxiter sequential line evaluations on one search;
# here BO_maxiter buys refinement rounds instead: one per 5 iterations of the reference's budget (20 -> 4 rounds)
ITERS_PER_ROUND = 5
# Monte-Carlo draws per line INSIDE an outer search = SEARCH_DRAW_FACTOR x PPBO_settings.mc_samples: at the
# reference's 150 draws the estimator's noise (~11 % for varmax) exceeds the spread of the acquisition value over
# the search domain, and a maximiser of the 150-draw estimate mostly maximises noise; a launch scores 512 lines x
# 1200 draws in ~35 ms, where one 150-draw line costs the reference ~0.36 s.  EI() / varmax() themselves keep
# mc_samples as given.
SEARCH_DRAW_FACTOR = 8


def _noisy_alpha_rows(B):
    """B independent draws of the reference's 70 noisy-equispaced abscissae on [0, 1] (feedback_processing.py:57-74
    with is_scaled: linspace(0.005, 0.995, 70) + N(0, 0.01), clipped, sorted, redrawn while two coincide) in ONE NumPy
    call: B x 70 normals leave the global stream in the order B sequential xi_grid calls would have taken them (a
    redraw -- two values clipped onto the same boundary: rare -- takes its numbers after the block instead of inside it)."""
    base = np.linspace(0.005, 0.995, LINE_POINTS)
    a = np.sort(np.clip(base[None, :] + np.random.normal(0.0, 0.01, (B, LINE_POINTS)), 0.0, 1.0), axis=1)
    bad = np.where((np.diff(a, axis=1) == 0.0).any(axis=1))[0]
    while bad.size:
        a[bad] = np.sort(np.clip(base[None, :] + np.random.normal(0.0, 0.01, (bad.size, LINE_POINTS)), 0.0, 1.0), axis=1)
        bad = bad[(np.diff(a[bad], axis=1) == 0.0).any(axis=1)]
    return a


def _line_scores(xis, xs, GP_model, mc_samples, z=None, alphas=None):
    """EI and varmax of B lines in one device call (ppbo_line_acq_xi: the 70 grid points of every line are formed on
    the device from (xi, x, alpha); the host hands over B x (2 D + 70) numbers, not a B x 70 x D grid built by B
    Python calls -- 52 ms of host time for the 1000 lines of EI-EXT at D = 20).  Grid noise (and z unless given) come
    from the global NumPy stream; a caller comparing lines passes ONE z -- and ONE set of grid abscissae `alphas` (the
    70 noisy-equispaced alpha of FP.xi_grid, drawn once) -- so that every line sees the same draws (common random
    numbers)."""
    xis, xs = np.atleast_2d(np.asarray(xis, dtype=float)), np.atleast_2d(np.asarray(xs, dtype=float))
    if alphas is None:
        alphas = _noisy_alpha_rows(len(xis))
    if z is None:
        z = np.random.standard_normal((mc_samples, LINE_POINTS))
    sf2 = float(GP_model.theta[2]) ** 2
    ei, vm = GP_model.eng.line_acq_xi(GP_model._post, xis, xs, alphas, z, GP_model.mustar,
                                      GP_model.COVARIANCE_SHRINKAGE, jitter=1e-10 * sf2)
    return ei.cpu().numpy(), vm.cpu().numpy()


def _noisy_alphas():
    """One draw of the reference's 70 noisy-equispaced abscissae on [0, 1] (feedback_processing.py:57-74, is_scaled)."""
    return _noisy_alpha_rows(1)[0]


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


def refinement_rounds(PPBO_settings):
    return int(np.ceil(max(int(getattr(PPBO_settings, "BO_maxiter", 20)), 0) / ITERS_PER_ROUND))


def _batched_search(k, lines_of, GP_model, PPBO_settings, which):
    """Maximise EI (which='ei') or varmax ('vm') over u in [0,1]^k, where lines_of(U[B,k]) -> (xis[B,D], xs[B,D]).
    Replaces the GPyOpt Bayesian optimisation of src/acquisition.py:96-100,119-123,194-198,210-214 (GPyOpt==1.2.6
    is not installable here): a batched evolutionary search that scores whole populations per device
    launch -- SEARCH_LINES uniform draws, then refinement_rounds() rounds of REFINE_LINES Gaussian perturbations of
    the REFINE_PARENTS best lines so far with a width that halves every round (0.2, 0.1, ... of the unit box) --
    all with ONE set of SEARCH_DRAW_FACTOR x mc_samples Monte-Carlo draws (device-generated), so lines are ranked by common random
    numbers (one z, one noisy alpha grid) instead of by 150-draw noise; the best REFINE_PARENTS then meet in a
    play-off on four times as many fresh draws.
    Returns the best u and a log [(round, best value so far)]."""
    draws = SEARCH_DRAW_FACTOR * int(PPBO_settings.mc_samples)
    # the draws are generated on the device (ppbo_randn) from a seed taken off the global NumPy stream: the search
    # stays reproducible under np.random.seed, and 4 x 1200 x 70 normals no longer cost the host 3 ms per search
    z = GP_model.eng.randn(np.random.randint(0, 2 ** 31 - 1), draws, LINE_POINTS)
    alphas = _noisy_alphas()
    pick = 0 if which == "ei" else 1

    def score(U, zz=z, aa=alphas):
        xis, xs = lines_of(U)
        return _line_scores(xis, xs, GP_model, len(zz), z=zz, alphas=aa)[pick]

    U = np.random.uniform(0.0, 1.0, (SEARCH_LINES, k))
    V = score(U)
    log = [(0, float(V.max()))]
    for rnd in range(1, refinement_rounds(PPBO_settings) + 1):
        parents = U[np.argsort(-V)[:REFINE_PARENTS]]
        width = 0.2 * 0.5 ** (rnd - 1)
        kids = parents[np.arange(REFINE_LINES) % len(parents)] + width * np.random.standard_normal((REFINE_LINES, k))
        kids = np.clip(kids, 0.0, 1.0)
        U, V = np.vstack([U, kids]), np.concatenate([V, score(kids)])
        log.append((rnd, float(V.max())))
    # play-off: the incumbents' values are biased upwards by their own selection; the REFINE_PARENTS best are
    # re-scored on an independent set of draws (and fresh grid noise) and the winner is chosen on those alone
    finalists = U[np.argsort(-V)[:REFINE_PARENTS]]
    V2 = score(finalists, GP_model.eng.randn(np.random.randint(0, 2 ** 31 - 1), 4 * draws, LINE_POINTS), _noisy_alphas())
    return finalists[int(np.argmax(V2))], log


def _search_joint(xi_dims, GP_model, PPBO_settings, which, fixed_x=None):
    """maximize_EI / maximize_EI_fixed_x / maximize_varmax (src/acquisition.py:91-131, 189-206): xi free on xi_dims,
    x free on the complement.  With fixed_x (maximize_EI_fixed_x) only xi's coordinates are searched, and the
    objective is the REFERENCE's: EI(xi_, xstar) with xi_ = xstar overwritten on xi_dims and the FULL xstar as x
    (src/acquisition.py:109-113) -- a line through xstar whose direction keeps xstar's own components off xi_dims --
    although what is returned is (xi zero off xi_dims, x = xstar off xi_dims) as in :124-131."""
    D = GP_model.D
    xi_dims = list(xi_dims)
    x_dims = [i for i in range(D) if i not in xi_dims]
    free = xi_dims + ([] if fixed_x is not None else x_dims)

    def result_of(U):
        xis, xs = np.zeros((len(U), D)), np.zeros((len(U), D))
        xis[:, xi_dims] = U[:, :len(xi_dims)]
        xs[:, x_dims] = U[:, len(xi_dims):] if fixed_x is None else np.asarray(fixed_x)[x_dims]
        return xis, xs

    def scored_lines(U):
        if fixed_x is None:
            return result_of(U)
        xis = np.tile(np.asarray(fixed_x, dtype=float), (len(U), 1))
        xis[:, xi_dims] = U[:, :len(xi_dims)]
        return xis, np.tile(np.asarray(fixed_x, dtype=float), (len(U), 1))

    u, log = _batched_search(len(free), scored_lines, GP_model, PPBO_settings, which)
    GP_model.acq_search_log = log
    GP_model.acq_search_scored = tuple(a[0] for a in scored_lines(u[None, :]))     # the line the winning value belongs to
    xis, xs = result_of(u[None, :])
    return perturbate_zerocoordinates(xis[0], xi_dims), perturbate_zerocoordinates(xs[0], x_dims)


def _split(xi_plus_x, xi_dims, x_dims, D):
    v = np.asarray(xi_plus_x, dtype=float)
    v = v[0] if v.ndim == 2 else v              # GPyOpt hands its objective a (1, D) array (src/acquisition.py:85)
    xi, x = np.zeros(D), np.zeros(D)
    xi[list(xi_dims)] = v[list(xi_dims)]
    x[list(x_dims)] = v[list(x_dims)]
    return xi, x


def EI_to_maximize(xi_plus_x, xi_dims, x_dims, GP_model, mc_samples):
    """The single-point objective the reference hands to its optimiser (src/acquisition.py:84-90); the searches
    below score whole batches of such points in one launch instead of calling this in a loop."""
    return EI(*_split(xi_plus_x, xi_dims, x_dims, GP_model.D), GP_model, mc_samples)


def EI_fixed_x_to_maximize(xi, xstar, xi_dims, GP_model, mc_samples):
    """src/acquisition.py:109-113: xi's free coordinates vary, the line passes through xstar."""
    v = np.asarray(xi, dtype=float)
    v = v[0] if v.ndim == 2 else v
    xi_ = np.array(xstar, dtype=float)
    xi_[list(xi_dims)] = v
    return EI(xi_, xstar, GP_model, mc_samples)


def varmax_to_maximize(xi_plus_x, xi_dims, x_dims, GP_model, mc_samples):
    """src/acquisition.py:180-186."""
    return varmax(*_split(xi_plus_x, xi_dims, x_dims, GP_model.D), GP_model, mc_samples)


def maximize_EI(xi_dims, GP_model, PPBO_settings):
    return _search_joint(xi_dims, GP_model, PPBO_settings, "ei")


def maximize_EI_fixed_x(xi_dims, GP_model, PPBO_settings):
    return _search_joint(xi_dims, GP_model, PPBO_settings, "ei", fixed_x=GP_model.xstar.copy())


def maximize_varmax(xi_dims, GP_model, PPBO_settings):
    return _search_joint(xi_dims, GP_model, PPBO_settings, "vm")


def maximize_varmax_given_xi(xi, GP_model, PPBO_settings):
    """x maximising varmax for a given direction (src/acquisition.py:208-218).  As in the reference, ALL D coordinates
    of x are searched (the objective is varmax(xi, x) over the whole box: on xi's support x shifts the line along
    itself) and the coordinates on xi's support are zeroed in the result afterwards (:216-217)."""
    D = GP_model.D
    xi = np.asarray(xi, dtype=float)

    def lines_of(U):
        return np.tile(xi, (len(U), 1)), np.asarray(U, dtype=float)

    u, log = _batched_search(D, lines_of, GP_model, PPBO_settings, "vm")
    GP_model.acq_search_log = log
    GP_model.acq_search_scored = (xi.copy(), np.array(u, dtype=float))
    x_next = np.array(u, dtype=float)
    x_next[np.where(xi != 0)[0]] = 0.0
    return x_next


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
