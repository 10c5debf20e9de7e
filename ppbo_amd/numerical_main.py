"""The loop around the hot path: run_ppbo_loop of the reference's driver (ppbo_numerical_main.py:57-127) on the
drop-in classes.

The reference file does not run as it stands (hard-coded working directory :14, pypet environment :192) and its
test objectives / simulated users (numerical_experiments/test_functions.py) are outside the hot path, so the user is
a callable here: `user(xi, x) -> alpha_star`, the point alpha_star * xi + x being the user's choice on the projective
line.  Everything else keeps the reference's order and flags:

  * initial queries (:71-94): x zeroed on xi's support (:80); the model is created at the first query (:84-85);
    `turn_initialization_off()` before the LAST initial query's update (:76-77) and again after the block (:99);
    ADAPTIVE_INITIALIZATION (:74-75) copies the previous answer into the remaining initial x's;
  * actual queries (:101-124): `set_last_iteration()` when i + 1 == n_initial + n_actual (:104-105 -- with initial
    queries this never fires, exactly as in the reference); `next_query(settings, model, unscale=True)` (:107);
    `mustar_previous_iteration` (:113); hyper-parameter optimisation after query number
    OPTIMIZE_HYPERPARAMETERS_AFTER_ACTUAL_QUERY_NUMBER or after each iteration (:116-119);
  * per query the unscaled x* and mu* are recorded (:90-92, :120-124).
"""
from __future__ import annotations

import numpy as np

from .acquisition import next_query
from .gp_model import GPModel


def run_ppbo_loop(user, initial_queries_xi, initial_queries_x, number_of_actual_queries, PPBO_settings, *,
                  adaptive_initialization=False, optimize_hyperparameters_after_initialization=False,
                  optimize_hyperparameters_after_each_iteration=False,
                  optimize_hyperparameters_after_actual_query_number=999, incremental=False, engine=None, verbose=False,
                  callback=None):
    """Returns (results[n_q, 2D+1], xstar_results[n_q, D], mustar_results[n_q], GP_model) like the reference.
    `incremental=True` switches the model to the bordered-Cholesky / warm-start fit (not a reference feature).
    `callback(i, GP_model)` is called after every model update (the reference prints there)."""
    D = PPBO_settings.D
    xi0 = np.array(initial_queries_xi, dtype=float)
    x0 = np.array(initial_queries_x, dtype=float)
    n_init = len(xi0)
    n_total = n_init + number_of_actual_queries
    mustar_results = [0.0] * n_total
    xstar_results = np.empty((n_total, D))
    results = np.empty((0, 2 * D + 1))
    GP_model = None
    alpha_star = xi = x = None

    def record(k):
        if GP_model.xstar is not None:
            xstar_results[k, :] = GP_model.FP.unscale(GP_model.xstar)
            mustar_results[k] = GP_model.mustar
        else:                                   # skip_xstaroptimization_during_initialization
            xstar_results[k, :] = np.nan
            mustar_results[k] = np.nan
        if callback is not None:
            callback(k, GP_model)

    for i in range(n_init):
        if i != 0 and adaptive_initialization:
            x0[i:, :] = alpha_star * xi + x
        if i == n_init - 1 and GP_model is not None:
            GP_model.turn_initialization_off()
        x = np.array(x0[i])
        xi = np.array(xi0[i])
        x[xi != 0] = 0
        alpha_star = float(user(xi, x))
        results = np.vstack([results, np.concatenate([alpha_star * xi + x, xi, [alpha_star]])])
        if GP_model is None:
            GP_model = GPModel(PPBO_settings, engine=engine, incremental=incremental)
            if n_init == 1:
                GP_model.turn_initialization_off()
        GP_model.update_feedback_processing_object(np.array(results))
        GP_model.update_data()
        GP_model.update_model()
        record(i)
        if verbose:
            print("xstar of the initialization " + str(i + 1) + "/" + str(n_init) + " is " + str(xstar_results[i]))
    if optimize_hyperparameters_after_initialization:
        GP_model.update_model(optimize_theta=True)
    if verbose:
        print("Initialization done! (Acq." + str(PPBO_settings.xi_acquisition_function) + " )")
    GP_model.turn_initialization_off()

    for i in range(number_of_actual_queries):
        if verbose:
            print("Starting query " + str(i + 1) + "/" + str(number_of_actual_queries) + " ...")
        if i + 1 == n_init + number_of_actual_queries:
            GP_model.set_last_iteration()
        xi_next, x_next = next_query(PPBO_settings, GP_model, unscale=True)
        alpha_star = float(user(xi_next, x_next))
        results = np.vstack([results, np.concatenate([alpha_star * xi_next + x_next, xi_next, [alpha_star]])])
        GP_model.update_feedback_processing_object(np.array(results))
        GP_model.mustar_previous_iteration = GP_model.mustar
        GP_model.update_data()
        if i + 1 == optimize_hyperparameters_after_actual_query_number:
            GP_model.update_model(optimize_theta=True)
        else:
            GP_model.update_model(optimize_theta=optimize_hyperparameters_after_each_iteration)
        record(n_init + i)
        if verbose:
            print("xstar of the iteration: " + str(xstar_results[n_init + i]))
    if verbose:
        print("Run done! (Acq." + str(PPBO_settings.xi_acquisition_function) + " )")
    return results, xstar_results, mustar_results, GP_model


def line_search_user(objective, lower, upper, points=4001):
    """A simulated user for a MINIMISATION test objective: the alpha in [alpha_min, alpha_max] (the part of the line
    inside the box, misc.alpha_bounds) with the smallest objective, by a dense scan.  (The reference's users run
    differential evolution on the same one-dimensional problem, test_functions.py:10-61.)"""
    from .misc import alpha_bounds
    lower, upper = np.asarray(lower, dtype=float), np.asarray(upper, dtype=float)

    def user(xi, x):
        a0, a1 = alpha_bounds(xi, lower, upper)
        al = np.linspace(a0, a1, points)
        pts = al[:, None] * np.asarray(xi, dtype=float)[None, :] + np.asarray(x, dtype=float)[None, :]
        return float(al[int(np.argmin(objective(pts)))])
    return user
