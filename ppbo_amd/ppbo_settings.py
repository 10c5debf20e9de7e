"""Settings bag of the drop-in (same constructor and derived attributes as the reference's
src/ppbo_settings.py:8-79, which is a plain attribute container)."""
from __future__ import annotations

_EXPLOIT = ("PCD", "EXT", "EI-EXT", "EI-EXT-FAST")
_VARMAX = ("EI-VARMAX", "EI-VARMAX-FAST", "COORDINATE-VARMAX")
_JOINT = ("EI", "EI-FIXEDX", "EXR")
STRATEGIES = _EXPLOIT + _VARMAX + _JOINT + ("RAND",)


class PPBO_settings:
    def __init__(self, D, bounds, xi_acquisition_function, theta_initial=None, user_feedback_grid_size=100, m=25,
                 verbose=True, EI_EXR_mc_samples=150, EI_EXR_BO_maxiter=20, mustar_finding_trials=3,
                 kernel="SE_kernel", skip_computations_during_initialization=True,
                 skip_xstaroptimization_during_initialization=False, alpha_grid_distribution="equispaced"):
        self.verbose = verbose
        self.user_feedback_grid_size = user_feedback_grid_size
        self.skip_computations_during_initialization = skip_computations_during_initialization
        self.skip_xstaroptimization_during_initialization = skip_xstaroptimization_during_initialization
        self.D = D
        self.original_bounds = bounds
        self.fMAP_optimizer = "trust-exact"          # ppbo_settings.py:41 (the HIP fit is a trust-region Newton too)
        self.mustar_finding_trials = mustar_finding_trials
        self.kernel = kernel
        self.theta_initial = [1, 0.1, 8] if theta_initial is None else theta_initial
        self.n_pseudoobservations = m
        self.alpha_grid_distribution = alpha_grid_distribution
        self.TGN_speed = 0.4                         # ppbo_settings.py:51
        self.n_gausshermite_sample_points = 200      # ppbo_settings.py:52 (closed form on device)
        self.mc_samples = EI_EXR_mc_samples
        self.BO_maxiter = EI_EXR_BO_maxiter
        self.xi_acquisition_function = xi_acquisition_function
        acq = xi_acquisition_function
        if acq in ("PCD", "EXT"):
            self.dim_query_prev_iter = D             # coordinate cycling starts at dimension 1
            self.x_acquisition_function = "exploit"
        elif acq == "RAND":
            self.x_acquisition_function = "random"
        elif acq in _JOINT:
            self.x_acquisition_function = "none"
            self.xi_dims_prev_iter = [0, 1] if D > 2 else [1]
        elif acq in ("EI-EXT", "EI-EXT-FAST"):
            self.x_acquisition_function = "exploit"
        elif acq in _VARMAX:
            self.x_acquisition_function = "varmax"
            if acq == "COORDINATE-VARMAX":
                self.dim_query_prev_iter = D
        else:
            print("Unknown acquisition function!")
