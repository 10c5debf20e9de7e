"""Thin Python layer over the C-ABI: torch-ROCm tensors own the device buffers,
every computation is a call into libppbo_hip.so.  No NumPy math, no fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import KERNEL_IDS, SCORE_MEAN, SCORE_POINTWISE_EI, SCORE_VARIANCE, PPBO_ERR_NOT_PD  # noqa: F401

SHRINKAGE = 1e-6  # COVARIANCE_SHRINKAGE of the reference (gp_model.py:26)


class NotPositiveDefinite(RuntimeError):
    def __init__(self, msg, info=0):
        super().__init__(msg)
        self.info = info


@dataclass
class Posterior:
    """Device-resident posterior state consumed by predict / predict_cov / line_acq."""
    kernel: str
    theta: tuple
    m: int
    X: torch.Tensor          # [N, D]
    alpha: torch.Tensor      # [N]   Sigma^-1 f_MAP
    lam_diag: torch.Tensor   # [N]   Lambda_MAP (star form)
    lam_off: torch.Tensor    # [N]
    G: torch.Tensor | None   # [N, N] R Lambda
    P: torch.Tensor | None = None  # posterior covariance (optional)
    Gt: torch.Tensor | None = None  # transpose of G in the one-launch scoring kernel's layout (formed on first use)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    """One ppbo_ctx bound to one GPU."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise RuntimeError("ppbo_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)   # every allocation / stream below names it explicitly
        ctx = C.c_void_p()
        rc = self.lib.ppbo_ctx_create(device, C.byref(ctx))
        if rc != 0:
            raise RuntimeError(f"ppbo_ctx_create failed with code {rc}")
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.ppbo_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers -------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _err(self):
        buf = C.create_string_buffer(512)
        self.lib.ppbo_last_error(self.ctx, buf, 512)
        return buf.value.decode(errors="replace")

    def _check(self, rc, what, info=0):
        if rc == 0:
            return
        if rc == PPBO_ERR_NOT_PD:
            raise NotPositiveDefinite(f"{what}: {self._err()}", info)
        raise RuntimeError(f"{what} failed (code {rc}): {self._err()}")

    def dev(self, a, dtype=torch.float64):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=self.device)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.float64, device=self.device)

    @staticmethod
    def _theta(theta):
        return (C.c_double * 3)(float(theta[0]), float(theta[1]), float(theta[2]))

    def _model(self, post: Posterior, with_var=True, kstar_fp32=False):
        N, D = post.X.shape
        md = _lib.Model()
        md.kernel_id = KERNEL_IDS[post.kernel]
        md.N, md.D, md.m = N, D, post.m
        md.theta = self._theta(post.theta)
        md.d_X = post.X.data_ptr()
        md.d_alpha = post.alpha.data_ptr()
        md.d_lam_diag = post.lam_diag.data_ptr() if post.lam_diag is not None else 0
        md.d_lam_off = post.lam_off.data_ptr() if post.lam_off is not None else 0
        md.d_G = post.G.data_ptr() if (with_var and post.G is not None) else 0
        md.kstar_fp32 = int(bool(kstar_fp32))
        md.d_Gt = 0
        if md.d_G and N <= 1024 and post.kernel != "camphor_copper_kernel":
            # models the one-launch scoring kernel takes: its matrix-core loop reads G transposed -- formed ONCE per
            # posterior here (the library would otherwise do it in a workspace on every call)
            if post.Gt is None or post.Gt.device != post.G.device:
                post.Gt = self.transposed_G(post.G)
                # once per posterior: the transpose is complete before ANY stream (another engine's, a side thread's) can
                # be handed the pointer through this Posterior
                torch.cuda.current_stream(self.device).synchronize()
            md.d_Gt = post.Gt.data_ptr()
        return md

    def transposed_G(self, G):
        """G [N, N] -> its transpose in the layout of ppbo_model.d_Gt (ppbo_transposed_G)."""
        N = G.shape[0]
        rows, ld = C.c_int(0), C.c_int(0)
        self._check(self.lib.ppbo_transposed_G_shape(N, C.byref(rows), C.byref(ld)), "ppbo_transposed_G_shape")
        Gt = self.empty(rows.value, ld.value)
        self._check(self.lib.ppbo_transposed_G(self.ctx, _ptr(G), N, _ptr(Gt), self._stream()), "ppbo_transposed_G")
        return Gt

    # ---- the path's collective behind the C-ABI (RCCL; no torch.distributed needed) -----------------
    def dist_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._check(self.lib.ppbo_dist_unique_id(self.ctx, buf), "ppbo_dist_unique_id")
        return buf.raw

    def dist_init(self, unique_id: bytes, rank: int, world: int):
        self._check(self.lib.ppbo_dist_init(self.ctx, C.c_char_p(unique_id), int(rank), int(world)), "ppbo_dist_init")

    def argmax_allgather(self, local_val: float, local_global_idx: int):
        bv, bi = C.c_double(0.0), C.c_int64(-1)
        rc = self.lib.ppbo_argmax_allgather(self.ctx, float(local_val), int(local_global_idx), C.byref(bv), C.byref(bi),
                                            self._stream())
        self._check(rc, "ppbo_argmax_allgather")
        return bv.value, bi.value

    def argmax_combine(self, records):
        """records: device float64 [world, 2] of (value, global index) -> (best value, its index); reduced on the
        device by one wavefront, one 16-byte read-back."""
        bv, bi = C.c_double(0.0), C.c_int64(-1)
        rc = self.lib.ppbo_argmax_combine(self.ctx, _ptr(records), int(records.numel() // 2), C.byref(bv), C.byref(bi),
                                          self._stream())
        self._check(rc, "ppbo_argmax_combine")
        return bv.value, bi.value

    def dist_destroy(self):
        self._check(self.lib.ppbo_dist_destroy(self.ctx), "ppbo_dist_destroy")

    def argmax_allgather_record(self, record):
        """record: device float64 [2] as written by predict_record -> job-wide (best value, global index) through the
        ctx's RCCL communicator (ppbo_argmax_allgather_record: all-gather, reduction, one 16-byte read-back)."""
        bv, bi = C.c_double(0.0), C.c_int64(-1)
        rc = self.lib.ppbo_argmax_allgather_record(self.ctx, _ptr(record), C.byref(bv), C.byref(bi), self._stream())
        self._check(rc, "ppbo_argmax_allgather_record")
        return bv.value, bi.value

    def predict_record(self, post, Xc, score=SCORE_MEAN, mustar=0.0, index_offset=0, out=None, kstar_fp32=False):
        """One shard of a sharded search, enqueue only (ppbo_predict_record): out[2] (device) = (best score,
        index_offset + first row index as a float64); nothing is read back, nothing synchronises."""
        Xc = self.dev(Xc)
        md = self._model(post, score != SCORE_MEAN, kstar_fp32)
        rec = self.empty(2) if out is None else out
        rc = self.lib.ppbo_predict_record(self.ctx, C.byref(md), _ptr(Xc), Xc.shape[0], int(score), float(mustar),
                                          int(index_offset), _ptr(rec), self._stream())
        self._check(rc, "ppbo_predict_record")
        return rec

    def search_sharded(self, post, Xc, score=SCORE_MEAN, mustar=0.0, index_offset=0, kstar_fp32=False):
        """One whole sharded search step in ONE library call (ppbo_search_sharded): score this rank's rows, RCCL
        all-gather of the 16-byte records (when dist_init has run on this ctx), reduction, one read-back."""
        Xc = self.dev(Xc)
        md = self._model(post, score != SCORE_MEAN, kstar_fp32)
        bv, bi = C.c_double(0.0), C.c_int64(-1)
        rc = self.lib.ppbo_search_sharded(self.ctx, C.byref(md), _ptr(Xc), Xc.shape[0], int(score), float(mustar),
                                          int(index_offset), C.byref(bv), C.byref(bi), self._stream())
        self._check(rc, "ppbo_search_sharded")
        return bv.value, bi.value

    # ---- per-kernel event timing ---------------------------------------------
    def profile(self, on=True):
        self.lib.ppbo_profile_enable(self.ctx, int(on))
        self.lib.ppbo_profile_reset(self.ctx)

    def profile_reset(self):
        self.lib.ppbo_profile_reset(self.ctx)

    def profile_read(self, name):
        tot, cnt = C.c_double(0.0), C.c_int(0)
        rc = self.lib.ppbo_profile_read(self.ctx, name.encode(), C.byref(tot), C.byref(cnt))
        self._check(rc, "ppbo_profile_read")
        return tot.value, cnt.value

    # ---- K1 / K2 --------------------------------------------------------
    def gram(self, X, theta, kernel="SE_kernel", shrink=SHRINKAGE, out=None):
        X = self.dev(X)
        N, D = X.shape
        S = self.empty(N, N) if out is None else out
        rc = self.lib.ppbo_gram(self.ctx, KERNEL_IDS[kernel], _ptr(X), N, D, self._theta(theta), shrink, _ptr(S),
                                self._stream())
        self._check(rc, "ppbo_gram")
        return S

    def store_floor(self, out):
        """Write-only pass over an N x N device matrix (ppbo_store_floor): the Gram kernel's ceiling at that N."""
        rc = self.lib.ppbo_store_floor(self.ctx, _ptr(out), out.shape[0], self._stream())
        self._check(rc, "ppbo_store_floor")
        return out

    def cross_cov(self, X1, X2, theta, kernel="SE_kernel"):
        X1, X2 = self.dev(X1), self.dev(X2)
        n1, D = X1.shape
        n2 = X2.shape[0]
        K = self.empty(n1, n2)
        rc = self.lib.ppbo_cross_cov(self.ctx, KERNEL_IDS[kernel], _ptr(X1), n1, _ptr(X2), n2, D, self._theta(theta),
                                     _ptr(K), n2, self._stream())
        self._check(rc, "ppbo_cross_cov")
        return K

    # ---- K6 ----------------------------------------------------------------
    def potrf_(self, A):
        """In-place lower Cholesky of a square device tensor."""
        N = A.shape[0]
        info = C.c_int(0)
        rc = self.lib.ppbo_potrf(self.ctx, _ptr(A), N, A.stride(0), C.byref(info), self._stream())
        self._check(rc, "ppbo_potrf", info.value)
        return A

    def pd_inverse(self, A):
        A = self.dev(A)
        N = A.shape[0]
        out = self.empty(N, N)
        info = C.c_int(0)
        rc = self.lib.ppbo_pd_inverse(self.ctx, _ptr(A), N, _ptr(out), C.byref(info), self._stream())
        self._check(rc, "ppbo_pd_inverse", info.value)
        return out

    def pd_inverse_factors(self, A):
        """(A^-1, L^-1) with A = L L^T: what pd_inverse_append borders."""
        A = self.dev(A)
        N = A.shape[0]
        out, linv = self.empty(N, N), self.empty(N, N)
        info = C.c_int(0)
        rc = self.lib.ppbo_pd_inverse_factors(self.ctx, _ptr(A), N, _ptr(out), _ptr(linv), C.byref(info), self._stream())
        self._check(rc, "ppbo_pd_inverse_factors", info.value)
        return out, linv

    def pd_inverse_chol(self, A):
        """(A^-1, L) with A = L L^T (lower triangle of L valid): what fit_fmap_whitened iterates with."""
        A = self.dev(A)
        N = A.shape[0]
        out, L = self.empty(N, N), self.empty(N, N)
        info = C.c_int(0)
        rc = self.lib.ppbo_pd_inverse_ex(self.ctx, _ptr(A), N, _ptr(out), _ptr(L), None, C.byref(info), self._stream())
        self._check(rc, "ppbo_pd_inverse_ex", info.value)
        return out, L

    def pd_inverse_factors3(self, A):
        """(A^-1, L^-1, L): everything the bordered append extends."""
        A = self.dev(A)
        N = A.shape[0]
        out, linv, L = self.empty(N, N), self.empty(N, N), self.empty(N, N)
        info = C.c_int(0)
        rc = self.lib.ppbo_pd_inverse_ex(self.ctx, _ptr(A), N, _ptr(out), _ptr(L), _ptr(linv), C.byref(info), self._stream())
        self._check(rc, "ppbo_pd_inverse_ex", info.value)
        return out, linv, L

    def pd_inverse_append(self, A, A11inv, L11inv, L11=None):
        """(A^-1, L^-1) of A[N,N] given those of its leading N1 x N1 block (one appended query, f-4); with L11 (the
        factor of that block) the bordered factor L comes back as a third result."""
        A, A11inv, L11inv = self.dev(A), self.dev(A11inv), self.dev(L11inv)
        N, N1 = A.shape[0], A11inv.shape[0]
        out, linv = self.empty(N, N), self.empty(N, N)
        info = C.c_int(0)
        if L11 is None:
            rc = self.lib.ppbo_pd_inverse_append(self.ctx, _ptr(A), N, _ptr(A11inv), _ptr(L11inv), N1, _ptr(out),
                                                 _ptr(linv), C.byref(info), self._stream())
            self._check(rc, "ppbo_pd_inverse_append", info.value)
            return out, linv
        L11 = self.dev(L11)
        L = self.empty(N, N)
        rc = self.lib.ppbo_pd_inverse_append_ex(self.ctx, _ptr(A), N, _ptr(A11inv), _ptr(L11inv), _ptr(L11), N1, _ptr(out),
                                                _ptr(linv), _ptr(L), C.byref(info), self._stream())
        self._check(rc, "ppbo_pd_inverse_append_ex", info.value)
        return out, linv, L

    def dgemm(self, A, B, transA=False, transB=False, alpha=1.0, beta=0.0, C_out=None):
        A, B = self.dev(A), self.dev(B)
        M = A.shape[1] if transA else A.shape[0]
        K = A.shape[0] if transA else A.shape[1]
        Nn = B.shape[0] if transB else B.shape[1]
        Cc = C_out if C_out is not None else self.empty(M, Nn)
        rc = self.lib.ppbo_dgemm(self.ctx, int(transA), int(transB), M, Nn, K, alpha, _ptr(A), A.stride(0), _ptr(B),
                                 B.stride(0), beta, _ptr(Cc), Cc.stride(0), self._stream())
        self._check(rc, "ppbo_dgemm")
        return Cc

    def lu_slogdet_(self, A):
        """LU with partial pivoting in place; returns (prod sign(u_ii), sum log|u_ii|)."""
        N = A.shape[0]
        sg, ld, info = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        rc = self.lib.ppbo_lu_slogdet(self.ctx, _ptr(A), N, A.stride(0), C.byref(sg), C.byref(ld), C.byref(info),
                                      self._stream())
        self._check(rc, "ppbo_lu_slogdet")
        return sg.value, ld.value, info.value

    def laplace_logdet(self, Sigma, lam_diag, lam_off, m):
        N = Sigma.shape[0]
        sg, ld, info = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        rc = self.lib.ppbo_laplace_logdet(self.ctx, _ptr(Sigma), _ptr(lam_diag), _ptr(lam_off), N, m, C.byref(sg),
                                          C.byref(ld), C.byref(info), self._stream())
        self._check(rc, "ppbo_laplace_logdet")
        return sg.value, ld.value, info.value

    def dgemv(self, A, x, trans=False, lower=False):
        A, x = self.dev(A), self.dev(x).reshape(-1)
        N = A.shape[0]
        if A.dim() != 2 or A.shape[1] != N or x.numel() != N:
            raise ValueError(f"dgemv: A is {tuple(A.shape)}, x has {x.numel()} entries (a square A and len(x) == N are required)")
        y = self.empty(N)
        rc = self.lib.ppbo_dgemv(self.ctx, int(trans), int(lower), N, _ptr(A), A.stride(0), _ptr(x), _ptr(y),
                                 self._stream())
        self._check(rc, "ppbo_dgemv")
        return y

    # ---- K5 / fit -------------------------------------------------------------
    def laplace_terms(self, f, m, sigma):
        f = self.dev(f).reshape(-1)
        N = f.numel()
        T = self.empty(1)
        beta, ld, lo = self.empty(N), self.empty(N), self.empty(N)
        rc = self.lib.ppbo_laplace_terms(self.ctx, _ptr(f), N, m, float(sigma), _ptr(T), _ptr(beta), _ptr(ld),
                                         _ptr(lo), self._stream())
        self._check(rc, "ppbo_laplace_terms")
        return float(T.item()), beta, ld, lo

    def sum_phi(self, f, m, sigma, order):
        """Per-query sums of src/gp_model.py:206-218 (device tensor [N / (m+1)])."""
        f = self.dev(f).reshape(-1)
        N = f.numel()
        out = self.empty(N // (m + 1))
        rc = self.lib.ppbo_sum_phi(self.ctx, _ptr(f), N, m, float(sigma), int(order), _ptr(out), self._stream())
        self._check(rc, "ppbo_sum_phi")
        return out

    def regularize_covariance(self, K, reg_level=1e-4, pos_diag=True, jitter=1e-7):
        """src/misc.py:71-88 on a device copy of K (the caller's matrix is left alone, as the reference's callers
        use the returned value)."""
        K = self.dev(K).clone()
        N = K.shape[0]
        rc = self.lib.ppbo_regularize_covariance(self.ctx, _ptr(K), N, K.stride(0), float(reg_level), int(bool(pos_diag)),
                                                 float(jitter), self._stream())
        self._check(rc, "ppbo_regularize_covariance")
        return K

    def T_and_grad(self, Sigma_inv, f, m, sigma):
        f = self.dev(f).reshape(-1)
        N = f.numel()
        grad = self.empty(N)
        T = C.c_double(0.0)
        rc = self.lib.ppbo_T_and_grad(self.ctx, _ptr(Sigma_inv), _ptr(f), N, m, float(sigma), C.byref(T), _ptr(grad),
                                      self._stream())
        self._check(rc, "ppbo_T_and_grad")
        return T.value, grad

    def fit_fmap(self, Sigma_inv, f_init, m, sigma, gtol=1e-4, maxiter=0, verbose=False, initial_radius=0.0, L=None,
                 lbfgs_max_evals=0):
        """f_MAP from one start vector.  L = None: trust-region Newton on f (ppbo_fit_fmap, the reference's
        algorithm class); L = Cholesky factor of Sigma: whitened L-BFGS finished by that trust region
        (ppbo_fit_fmap_whitened) -- same optimum, tens of O(N^2) evaluations instead of O(N^3) factorizations."""
        f0 = self.dev(f_init).reshape(-1)
        N = f0.numel()
        out = self.empty(N)
        opts = _lib.FitOpts(float(gtol), int(maxiter), int(verbose), float(initial_radius), int(lbfgs_max_evals), 0, 0)
        st = _lib.FitStats()
        if L is None:
            rc = self.lib.ppbo_fit_fmap(self.ctx, _ptr(Sigma_inv), N, m, float(sigma), _ptr(f0), C.byref(opts),
                                        _ptr(out), C.byref(st), self._stream())
            self._check(rc, "ppbo_fit_fmap")
        else:
            rc = self.lib.ppbo_fit_fmap_whitened(self.ctx, _ptr(L), L.stride(0), _ptr(Sigma_inv), N, m, float(sigma),
                                                 _ptr(f0), C.byref(opts), _ptr(out), C.byref(st), self._stream())
            self._check(rc, "ppbo_fit_fmap_whitened")
        stats = dict(iterations=st.iterations, n_cholesky=st.n_cholesky, converged=bool(st.converged), T=st.T,
                     gradnorm=st.gradnorm, lbfgs_iterations=st.lbfgs_iterations, lbfgs_evals=st.lbfgs_evals,
                     lbfgs_status=st.lbfgs_status)
        return out, stats

    def gp_fit(self, X, theta, kernel, m, f_init, shrink=SHRINKAGE, gtol=1e-4, maxiter=0, verbose=0, lbfgs_max_evals=0,
               start_is_whitened=False, want_Sigma=True, want_Linv=False, want_posterior=True):
        """One whole GP fit in one library call (ppbo_gp_fit): Sigma, its Cholesky factor and inverse, f_MAP from one
        start by the whitened search, and the posterior state -- the work of update_Sigma + update_Sigma_inv +
        update_fMAP + the posterior (src/gp_model.py:91-117), everything enqueued behind each other on one stream with
        ONE host wait.  start_is_whitened: f_init holds z0 and the start is the prior draw L z0.
        Returns dict(Sigma, Sigma_inv, L, Linv, fMAP, post, stats, info); info = 2 (with post = None) when
        Sigma^-1 - Lambda_MAP is not positive definite (raises NotPositiveDefinite when Sigma itself is not)."""
        X = self.dev(X)
        N, D = X.shape
        f0 = self.dev(f_init).reshape(-1)
        if f0.numel() != N:
            raise ValueError(f"gp_fit: the start vector has {f0.numel()} entries, the design {N} rows")
        Sigma = self.empty(N, N) if want_Sigma else None
        Sinv, L = self.empty(N, N), self.empty(N, N)
        Linv = self.empty(N, N) if want_Linv else None
        fmap = self.empty(N)
        if want_posterior:
            alpha, ld, lo, G = self.empty(N), self.empty(N), self.empty(N), self.empty(N, N)
        else:
            alpha = ld = lo = G = None
        opts = _lib.FitOpts(float(gtol), int(maxiter), int(verbose), 0.0, int(lbfgs_max_evals), 0, int(bool(start_is_whitened)))
        st = _lib.FitStats()
        info = C.c_int(0)
        rc = self.lib.ppbo_gp_fit(self.ctx, KERNEL_IDS[kernel], _ptr(X), N, D, self._theta(theta), float(shrink), int(m),
                                  _ptr(f0), C.byref(opts), _ptr(Sigma), _ptr(Sinv), _ptr(L), _ptr(Linv), _ptr(fmap),
                                  _ptr(alpha), _ptr(ld), _ptr(lo), _ptr(G), C.byref(st), C.byref(info), self._stream())
        if rc == PPBO_ERR_NOT_PD and info.value == 2:
            post = None
        else:
            self._check(rc, "ppbo_gp_fit", info.value)
            post = Posterior(kernel, tuple(float(t) for t in theta), m, X, alpha, ld, lo, G, None) if want_posterior else None
        stats = dict(iterations=st.iterations, n_cholesky=st.n_cholesky, converged=bool(st.converged), T=st.T,
                     gradnorm=st.gradnorm, lbfgs_iterations=st.lbfgs_iterations, lbfgs_evals=st.lbfgs_evals,
                     lbfgs_status=st.lbfgs_status)
        return dict(Sigma=Sigma, Sigma_inv=Sinv, L=L, Linv=Linv, fMAP=fmap, post=post, stats=stats, info=info.value)

    def posterior(self, X, theta, kernel, Sigma_inv, fMAP, m, want_P=False) -> Posterior:
        X = self.dev(X)
        f = self.dev(fMAP).reshape(-1)
        N = f.numel()
        alpha, ld, lo = self.empty(N), self.empty(N), self.empty(N)
        G = self.empty(N, N)
        P = self.empty(N, N) if want_P else None
        info = C.c_int(0)
        rc = self.lib.ppbo_posterior(self.ctx, _ptr(Sigma_inv), _ptr(f), N, m, float(theta[0]), _ptr(alpha), _ptr(ld),
                                     _ptr(lo), _ptr(G), _ptr(P), C.byref(info), self._stream())
        self._check(rc, "ppbo_posterior", info.value)
        return Posterior(kernel, tuple(float(t) for t in theta), m, X, alpha, ld, lo, G, P)

    # ---- prediction ---------------------------------------------------------------
    def predict(self, post: Posterior, Xc, score=SCORE_MEAN, mustar=0.0, want_mu=True, want_var=True,
                want_score=False, want_best=True, kstar_fp32=False):
        Xc = self.dev(Xc)
        M = Xc.shape[0]
        with_var = want_var or score != SCORE_MEAN
        md = self._model(post, with_var, kstar_fp32)
        mu = self.empty(M) if want_mu else None
        var = self.empty(M) if (want_var and with_var) else None
        sc = self.empty(M) if want_score else None
        bv, bi = C.c_double(0.0), C.c_int64(-1)
        rc = self.lib.ppbo_predict(self.ctx, C.byref(md), _ptr(Xc), M, int(score), float(mustar), _ptr(mu), _ptr(var),
                                   _ptr(sc), C.byref(bv) if want_best else None, C.byref(bi) if want_best else None,
                                   self._stream())
        self._check(rc, "ppbo_predict")
        return dict(mu=mu, var=var, score=sc, best_val=bv.value, best_idx=bi.value)

    def predict_cov(self, post: Posterior, Xc, shrink=SHRINKAGE):
        Xc = self.dev(Xc)
        M = Xc.shape[0]
        md = self._model(post, True)
        mu, cov = self.empty(M), self.empty(M, M)
        rc = self.lib.ppbo_predict_cov(self.ctx, C.byref(md), _ptr(Xc), M, float(shrink), _ptr(mu), _ptr(cov),
                                       self._stream())
        self._check(rc, "ppbo_predict_cov")
        return mu, cov

    def mean_grad(self, post: Posterior, Xc):
        """mu[M] and d mu / d x [M,D] at the rows of Xc (ppbo_mean_grad)."""
        Xc = self.dev(Xc)
        M, D = Xc.shape
        md = self._model(post, False)
        mu, grad = self.empty(M), self.empty(M, D)
        rc = self.lib.ppbo_mean_grad(self.ctx, C.byref(md), _ptr(Xc), M, _ptr(mu), _ptr(grad), self._stream())
        self._check(rc, "ppbo_mean_grad")
        return mu, grad

    def mean_search(self, post: Posterior, cand, K=32, sep=0.05, iters=100, tol=1e-9, sync=True):
        """Device-resident maximiser of the posterior mean over the rows of `cand` (ppbo_mean_search): returns the
        refined maxima x[found, D], mu[found] as NumPy arrays.  sync=False only enqueues (h_found = NULL: nothing
        synchronises) and returns the device tensors x[K, D], mu[K] -- rows that found no start carry mu = -inf -- so
        that several searches can be queued behind each other and read back together."""
        cand = self.dev(cand)
        M, D = cand.shape
        md = self._model(post, False)
        xs, mus = self.empty(K, D), self.empty(K)
        found = C.c_int(0)
        rc = self.lib.ppbo_mean_search(self.ctx, C.byref(md), _ptr(cand), M, int(K), float(sep), int(iters), float(tol),
                                       _ptr(xs), _ptr(mus), C.byref(found) if sync else None, self._stream())
        self._check(rc, "ppbo_mean_search")
        if not sync:
            return xs, mus
        n = found.value
        return xs[:n].cpu().numpy(), mus[:n].cpu().numpy()

    def mean_search_multi(self, post: Posterior, pool, shifts, extra=None, xprev=None, K=32, sep=0.05, iters=100, tol=1e-9,
                          screen_fp32=True):
        """All trials of one mu_star call in one enqueue (ppbo_mean_search_multi): trial t searches frac(pool + shifts[t]);
        trial 0 also the rows of `extra` ("design": the posterior's own design points, nothing is copied) and the point
        `xprev`.  Returns the device tensors x[T, K, D], mu[T, K] (mu = -inf where a trial found fewer than K starts);
        nothing synchronises."""
        pool = self.dev(pool)
        M, D = pool.shape
        if D != post.X.shape[1]:
            # the library strides pool, shifts and extra by the MODEL's D: a mismatch would read out of bounds on the device
            raise ValueError(f"mean_search_multi: pool has {D} columns, the model {post.X.shape[1]}")
        sh = np.ascontiguousarray(np.atleast_2d(np.asarray(shifts, dtype=np.float64)))
        T = sh.shape[0]
        if sh.shape[1] != D:
            raise ValueError("mean_search_multi: shifts must be [T, D]")
        E, ex_ptr = 0, None
        if isinstance(extra, str):
            if extra != "design":
                raise ValueError("mean_search_multi: extra is an array of points or the string 'design'")
            E = post.X.shape[0]
        elif extra is not None:
            extra = self.dev(extra)
            if extra.dim() != 2 or extra.shape[1] != D:
                raise ValueError("mean_search_multi: extra must be [E, D]")
            E, ex_ptr = extra.shape[0], extra
        xp = None
        if xprev is not None:
            xp = np.ascontiguousarray(np.asarray(xprev, dtype=np.float64).reshape(-1))
            if xp.size != D:
                raise ValueError("mean_search_multi: xprev must have D entries")
        md = self._model(post, False)
        xs, mus = self.empty(T, K, D), self.empty(T, K)
        dp = C.POINTER(C.c_double)
        rc = self.lib.ppbo_mean_search_multi(self.ctx, C.byref(md), _ptr(pool), M, sh.ctypes.data_as(dp), T, _ptr(ex_ptr), E,
                                             xp.ctypes.data_as(dp) if xp is not None else None, int(K), float(sep),
                                             int(iters), float(tol), int(bool(screen_fp32)), _ptr(xs), _ptr(mus),
                                             self._stream())
        self._check(rc, "ppbo_mean_search_multi")
        return xs, mus

    def mean_ascent(self, post: Posterior, starts, iters=100, tol=1e-9):
        starts = self.dev(starts)
        K, D = starts.shape
        md = self._model(post, False)
        xs, mus = self.empty(K, D), self.empty(K)
        its = torch.zeros(K, dtype=torch.int32, device=self.device)
        rc = self.lib.ppbo_mean_ascent(self.ctx, C.byref(md), _ptr(starts), K, int(iters), float(tol), _ptr(xs), _ptr(mus),
                                       _ptr(its), self._stream())
        self._check(rc, "ppbo_mean_ascent")
        return xs, mus, its

    def shift_points(self, pool, shift, out=None):
        """out = frac(pool + shift) row-wise (ppbo_shift_points)."""
        pool = self.dev(pool)
        M, D = pool.shape
        out = self.empty(M, D) if out is None else out
        sh = (C.c_double * D)(*[float(v) for v in shift])
        rc = self.lib.ppbo_shift_points(self.ctx, _ptr(pool), M, D, sh, _ptr(out), self._stream())
        self._check(rc, "ppbo_shift_points")
        return out

    def line_acq(self, post: Posterior, grid, z, mustar, shrink=SHRINKAGE, jitter=0.0):
        grid = self.dev(grid)
        B, G, _ = grid.shape
        z = self.dev(z)
        S = z.shape[0]
        md = self._model(post, True)
        ei, vm = self.empty(B), self.empty(B)
        rc = self.lib.ppbo_line_acq(self.ctx, C.byref(md), _ptr(grid), B, G, float(shrink), _ptr(z), S, float(mustar),
                                    float(jitter), _ptr(ei), _ptr(vm), self._stream())
        self._check(rc, "ppbo_line_acq")
        return ei, vm

    def line_acq_xi(self, post: Posterior, xis, xs, alphas, z, mustar, shrink=SHRINKAGE, jitter=0.0):
        """EI and varmax of the B lines {alpha * xis[b] + xs[b]} (ppbo_line_acq_xi): the grid points are formed on the
        device.  alphas: [G] (shared by all lines) or [B, G]."""
        xis, xs, alphas, z = self.dev(xis), self.dev(xs), self.dev(alphas), self.dev(z)
        B, D = xis.shape
        if xs.shape != (B, D):
            raise ValueError("line_acq_xi: xis and xs must both be [B, D]")
        per_line = alphas.dim() == 2
        G = alphas.shape[-1]
        if per_line and alphas.shape[0] != B:
            raise ValueError("line_acq_xi: per-line abscissae must be [B, G]")
        S = z.shape[0]
        md = self._model(post, True)
        ei, vm = self.empty(B), self.empty(B)
        rc = self.lib.ppbo_line_acq_xi(self.ctx, C.byref(md), _ptr(xis), _ptr(xs), _ptr(alphas), int(per_line), B, G,
                                       float(shrink), _ptr(z), S, float(mustar), float(jitter), _ptr(ei), _ptr(vm),
                                       self._stream())
        self._check(rc, "ppbo_line_acq_xi")
        return ei, vm

    # ---- RFF -------------------------------------------------------------------------
    def rff_project(self, X, W, b, sigma_f, out=None):
        X, W, b = self.dev(X), self.dev(W), self.dev(b).reshape(-1)
        N, D = X.shape
        F = W.shape[0]
        Phi = self.empty(F, N) if out is None else out
        rc = self.lib.ppbo_rff_project(self.ctx, _ptr(X), N, D, _ptr(W), F, _ptr(b), float(sigma_f), _ptr(Phi),
                                       self._stream())
        self._check(rc, "ppbo_rff_project")
        return Phi

    def rff_score(self, Xc, W, b, sigma_f, omega, want_score=True):
        Xc, W, b, omega = self.dev(Xc), self.dev(W), self.dev(b).reshape(-1), self.dev(omega).reshape(-1)
        M, D = Xc.shape
        F = W.shape[0]
        sc = self.empty(M) if want_score else None
        bv, bi = C.c_double(0.0), C.c_int64(-1)
        rc = self.lib.ppbo_rff_score(self.ctx, _ptr(Xc), M, D, _ptr(W), F, _ptr(b), float(sigma_f), _ptr(omega),
                                     _ptr(sc), C.byref(bv), C.byref(bi), self._stream())
        self._check(rc, "ppbo_rff_score")
        return sc, bv.value, bi.value

    def rff_omega_map(self, Phi, omega0, m, sigma, maxiter=500, gtol=1e-6):
        """Device-resident maximiser of S (ppbo_rff_omega_map): returns (omega_MAP as NumPy, S, |grad S|, iterations)."""
        Phi = self.dev(Phi)
        om = self.dev(omega0).reshape(-1).clone()
        F, N = Phi.shape
        S, gn, it = C.c_double(0.0), C.c_double(0.0), C.c_int(0)
        rc = self.lib.ppbo_rff_omega_map(self.ctx, _ptr(Phi), F, N, int(m), float(sigma), _ptr(om), int(maxiter), float(gtol),
                                         C.byref(S), C.byref(gn), C.byref(it), self._stream())
        self._check(rc, "ppbo_rff_omega_map")
        return om.cpu().numpy(), S.value, gn.value, it.value

    def randn(self, seed, *shape):
        """Standard normal draws generated on the device (ppbo_randn): a pure function of (seed, index)."""
        out = self.empty(*shape)
        rc = self.lib.ppbo_randn(self.ctx, int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(out), out.numel(), self._stream())
        self._check(rc, "ppbo_randn")
        return out

    def rff_search(self, cand, W, b, sigma_f, omega, K=32, sep=0.05, iters=200, tol=1e-10):
        """Device-resident maximiser of phi(x)^T omega over the rows of `cand` (ppbo_rff_search): refined maxima
        x[found, D], values[found] as NumPy arrays."""
        cand, W, b, omega = self.dev(cand), self.dev(W), self.dev(b).reshape(-1), self.dev(omega).reshape(-1)
        M, D = cand.shape
        F = W.shape[0]
        xs, vals = self.empty(K, D), self.empty(K)
        found = C.c_int(0)
        rc = self.lib.ppbo_rff_search(self.ctx, _ptr(cand), M, D, _ptr(W), F, _ptr(b), float(sigma_f), _ptr(omega), int(K),
                                      float(sep), int(iters), float(tol), _ptr(xs), _ptr(vals), C.byref(found), self._stream())
        self._check(rc, "ppbo_rff_search")
        n = found.value
        return xs[:n].cpu().numpy(), vals[:n].cpu().numpy()

    def rff_terms(self, Phi, omega, m, sigma):
        Phi, omega = self.dev(Phi), self.dev(omega).reshape(-1)
        F, N = Phi.shape
        g, h = self.empty(F), self.empty(F)
        S = C.c_double(0.0)
        rc = self.lib.ppbo_rff_terms(self.ctx, _ptr(Phi), F, N, m, float(sigma), _ptr(omega), C.byref(S), _ptr(g),
                                     _ptr(h), self._stream())
        self._check(rc, "ppbo_rff_terms")
        return S.value, g, h


_default = {}


def get_engine(device: int = 0) -> Engine:
    if device not in _default:
        _default[device] = Engine(device)
    return _default[device]
