"""Settings bag of the drop-in.  The reference's PPBO_settings (src/ppbo_settings.py:8-79) is a plain
attribute container; the drop-in keeps its constructor signature and every attribute other modules read,
but derives the strategy-dependent fields from one table instead of an if/elif chain."""
from __future__ import annotations

# strategy -> (how x is chosen given xi, needs a coordinate counter, needs the cyclic xi dimension list)
_STRATEGY_TABLE = {
    "PCD": ("exploit", True, False),
    "EXT": ("exploit", True, False),
    "RAND": ("random", False, False),
    "EI": ("none", False, True),
    "EI-FIXEDX": ("none", False, True),
    "EXR": ("none", False, True),
    "EI-EXT": ("exploit", False, False),
    "EI-EXT-FAST": ("exploit", False, False),
    "EI-VARMAX": ("varmax", False, False),
    "EI-VARMAX-FAST": ("varmax", False, False),
    "COORDINATE-VARMAX": ("varmax", True, False),
}
STRATEGIES = tuple(_STRATEGY_TABLE)

_FIXED = dict(
    fMAP_optimizer="trust-exact",          # :41  (the HIP fit is a trust-region Newton as well)
    TGN_speed=0.4,                         # :51
    n_gausshermite_sample_points=200,      # :52  (closed form on the device; kept for Hsampler's constructor)
)


class PPBO_settings:
    def __init__(self, D, bounds, xi_acquisition_function, theta_initial=None, user_feedback_grid_size=100, m=25,
                 verbose=True, EI_EXR_mc_samples=150, EI_EXR_BO_maxiter=20, mustar_finding_trials=3,
                 kernel="SE_kernel", skip_computations_during_initialization=True,
                 skip_xstaroptimization_during_initialization=False, alpha_grid_distribution="equispaced",
                 fMAP_method="whitened"):
        """fMAP_method (not a reference option): "whitened" = L-BFGS in z = L^-1 f finished by the trust
        region (the default; O(N^2) per iteration), "trust-region" = the exact Newton trust region on f alone, which
        follows SciPy trust-exact's iteration rules (the reference's optimiser class, src/gp_model.py:382-384) -- the
        parity mode for flows that want the reference's basin behaviour at sigma << sigma_f."""
        if fMAP_method not in ("whitened", "trust-region"):
            raise ValueError("fMAP_method must be 'whitened' or 'trust-region'")
        vars(self).update(_FIXED)
        self.fMAP_method = fMAP_method
        vars(self).update(
            D=D, original_bounds=bounds, verbose=verbose, kernel=kernel,
            user_feedback_grid_size=user_feedback_grid_size,
            theta_initial=[1, 0.1, 8] if theta_initial is None else theta_initial,
            n_pseudoobservations=m, alpha_grid_distribution=alpha_grid_distribution,
            mustar_finding_trials=mustar_finding_trials,
            mc_samples=EI_EXR_mc_samples, BO_maxiter=EI_EXR_BO_maxiter,
            skip_computations_during_initialization=skip_computations_during_initialization,
            skip_xstaroptimization_during_initialization=skip_xstaroptimization_during_initialization,
            xi_acquisition_function=xi_acquisition_function,
        )
        row = _STRATEGY_TABLE.get(xi_acquisition_function)
        if row is None:
            print("Unknown acquisition function!")
            return
        self.x_acquisition_function, counter, cyclic = row
        if counter:
            self.dim_query_prev_iter = D               # coordinate cycling starts at dimension 1 (:60,77)
        if cyclic:
            self.xi_dims_prev_iter = [0, 1] if D > 2 else [1]   # :66-69
