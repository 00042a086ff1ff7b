"""Device-side GP engine: owns the PyTorch-ROCm buffers and drives libmi_gp.so through ctypes.

Mirrors what the reference keeps inside its PyMC model object ``m``/``gp`` (gpmcmc.py:178,401):
the training inputs, the kernel structure, and a way to evaluate logp / dlogp / the conditional."""
import ctypes

import numpy as np
import torch

from . import _lib


def parse_kernel(kernel):
    """Kernel-string grammar of GPMCMC.change_model (gpmcmc.py:497-505): names joined by + or *."""
    import re

    kerns = re.split(r"[+*]", kernel)
    ops = [ch for ch in kernel if ch in "+*"]
    for k in kerns:
        if k not in _lib.KERNEL_IDS:
            raise Exception(f"Error: kernel string must contain only {list(_lib.KERNEL_IDS)}")
    if len(kerns) > _lib.MAX_KERN:
        raise Exception(f"Error: at most {_lib.MAX_KERN} kernel components are supported")
    return kerns, ops


def pack_theta(ls, kv, gv, jitter, alpha=None):
    """C-ABI theta layout: [ls(nkern*d), kv(nkern), alpha(nkern), gv, jitter]."""
    ls = np.atleast_2d(np.asarray(ls, dtype=np.float64))
    kv = np.atleast_1d(np.asarray(kv, dtype=np.float64))
    alpha = np.ones_like(kv) if alpha is None else np.atleast_1d(np.asarray(alpha, dtype=np.float64))
    return np.concatenate([ls.ravel(), kv, alpha, [float(gv), float(jitter)]])


class MiGP:
    """One GP data set + kernel structure bound to one MI355X and one HIP stream."""

    def __init__(self, X, y, kernel="RBF", device=0, panel_tiles=0, need_grad=True):
        if not torch.cuda.is_available():
            raise RuntimeError("MiGP needs a ROCm GPU: the GP hot path has no CPU implementation")
        self.lib = _lib.load()
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        if X.ndim != 2 or X.shape[0] != y.shape[0]:
            raise ValueError("X must be (n,d) and y (n,)")
        if not (np.isfinite(X).all() and np.isfinite(y).all()):
            # the device exp clamps its argument (migp_math.h): a NaN input would come out as a finite covariance entry,
            # where the reference's PyTensor graph propagates NaN and the "posdef" check rejects the point
            raise ValueError("X and y must be finite")
        self._bad_data = False
        self.n, self.d = X.shape
        self.kerns, self.ops = parse_kernel(kernel)
        self.nkern = len(self.kerns)
        self.device = int(device)
        self.dev = torch.device("cuda", self.device)
        cfg = _lib.MiGpConfig()
        cfg.n, cfg.d, cfg.nkern = self.n, self.d, self.nkern
        for i, k in enumerate(self.kerns):
            cfg.kernel_ids[i] = _lib.KERNEL_IDS[k]
        for i, o in enumerate(self.ops):
            cfg.ops[i] = _lib.OP_IDS[o]
        cfg.device = self.device
        cfg.panel_tiles = int(panel_tiles)
        h = ctypes.c_void_p()
        r = self.lib.mi_gp_create(ctypes.byref(cfg), ctypes.byref(h))
        if r != 0:
            raise RuntimeError(f"mi_gp_create failed ({r}): {self.lib.mi_gp_last_global_error().decode()}")
        self.h = h
        self.np_ = int(self.lib.mi_gp_padded_n(h))
        self.ntheta = int(self.lib.mi_gp_num_theta(h))
        # leading dimension: padded n plus 16 doubles so that column panels are not power-of-two strided
        self.lda = self.np_ + 16
        with torch.cuda.device(self.dev):
            self.X_t = torch.from_numpy(X).to(self.dev)
            self.y_t = torch.from_numpy(y).to(self.dev)
            self.K_t = torch.empty((self.np_ + 128, self.lda), dtype=torch.float64, device=self.dev)
            self.Z_t = self.W_t = None
            if need_grad:
                self.Z_t = torch.zeros((self.np_, self.lda), dtype=torch.float64, device=self.dev)
                self.W_t = torch.zeros((self.np_, self.lda), dtype=torch.float64, device=self.dev)
            torch.cuda.synchronize(self.dev)
        b = _lib.MiGpBuffers()
        b.X_dev, b.y_dev, b.K_dev = self.X_t.data_ptr(), self.y_t.data_ptr(), self.K_t.data_ptr()
        b.lda = self.lda
        b.Z_dev = self.Z_t.data_ptr() if need_grad else None
        b.W_dev = self.W_t.data_ptr() if need_grad else None
        self._check(self.lib.mi_gp_set_data(h, ctypes.byref(b)), "mi_gp_set_data")

    def _check(self, r, what):
        if r < 0:
            raise RuntimeError(f"{what} failed ({r}): {self.lib.mi_gp_last_error(self.h).decode()}")
        return r

    def _theta(self, theta):
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if theta.shape != (self.ntheta,):
            raise ValueError(f"theta must have {self.ntheta} entries")
        return theta, theta.ctypes.data_as(ctypes.POINTER(ctypes.c_double))

    def lml(self, theta):
        """LML at natural-scale theta; -inf if K is not positive definite (info > 0)."""
        theta, tp = self._theta(theta)
        out = ctypes.c_double()
        self._factored_ok = False
        if self._bad_data:  # non-finite warped data: what PyMC turns into logp = -inf (update_data)
            self.info = 1
            return -np.inf
        self.info = self._check(self.lib.mi_gp_lml(self.h, tp, ctypes.byref(out)), "mi_gp_lml")
        return out.value

    def lml_parts(self):
        a, b = ctypes.c_double(), ctypes.c_double()
        self.lib.mi_gp_lml_parts(self.h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def lml_grad(self, theta):
        """(LML, dLML/dtheta) at natural-scale theta; (-inf, zeros) if K is not positive definite."""
        if self.Z_t is None:
            raise RuntimeError("this MiGP was created with need_grad=False")
        theta, tp = self._theta(theta)
        out = ctypes.c_double()
        grad = np.zeros(self.ntheta)
        gp_ = grad.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        self._factored_ok = False
        if self._bad_data:
            self.info = 1
            return -np.inf, grad
        self.info = self._check(self.lib.mi_gp_lml_grad(self.h, tp, ctypes.byref(out), gp_), "mi_gp_lml_grad")
        return out.value, grad

    # ------------------------------------------------------------------ batched evaluation
    def _ensure_batch(self, k, need_grad):
        """Device buffers for k problems in lockstep (mi_gp_set_batch): K (+ Z, W for the gradient) as ONE tensor each."""
        have = getattr(self, "_batch_k", 0)
        if have >= k and (not need_grad or self._bZ is not None):
            return
        k = max(k, have)
        with torch.cuda.device(self.dev):
            self._bK = torch.empty((k, self.np_ + 128, self.lda), dtype=torch.float64, device=self.dev)
            self._bZ = self._bW = None
            if need_grad or getattr(self, "_batch_grad", False):
                self._bZ = torch.zeros((k, self.np_, self.lda), dtype=torch.float64, device=self.dev)
                self._bW = torch.zeros((k, self.np_, self.lda), dtype=torch.float64, device=self.dev)
                self._batch_grad = True
            torch.cuda.synchronize(self.dev)
        b = _lib.MiGpBatchBuffers()
        b.K_dev = self._bK.data_ptr()
        b.Z_dev = self._bZ.data_ptr() if self._bZ is not None else None
        b.W_dev = self._bW.data_ptr() if self._bW is not None else None
        b.stride_k = (self.np_ + 128) * self.lda
        b.stride_zw = self.np_ * self.lda
        b.count = k
        self._check(self.lib.mi_gp_set_batch(self.h, ctypes.byref(b)), "mi_gp_set_batch")
        self._batch_k = k

    def _thetas(self, thetas):
        th = np.ascontiguousarray(thetas, dtype=np.float64)
        if th.ndim != 2 or th.shape[1] != self.ntheta:
            raise ValueError(f"thetas must be (k, {self.ntheta})")
        return th

    def lml_batch(self, thetas):
        """LML at k hyper-parameter vectors of the same data in ONE lockstep evaluation (mi_gp_lml_batch): every launch
        carries blockIdx.z = problem.  Returns (k,) values, -inf where the covariance is not positive definite; the
        values are those lml() returns one at a time."""
        th = self._thetas(thetas)
        k = th.shape[0]
        self._ensure_batch(k, False)
        self._factored_ok = False
        out = np.empty(k)
        info = np.zeros(k, dtype=np.int32)
        if self._bad_data:
            return np.full(k, -np.inf)
        dpt, ipt = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        self._check(self.lib.mi_gp_lml_batch(self.h, k, th.ctypes.data_as(dpt), out.ctypes.data_as(dpt), info.ctypes.data_as(ipt)),
                    "mi_gp_lml_batch")
        self.batch_info = info
        return out

    def lml_grad_batch(self, thetas):
        """(LML (k,), dLML/dtheta (k, ntheta)) at k hyper-parameter vectors in one lockstep evaluation
        (mi_gp_lml_grad_batch); rows of non-positive-definite problems are (-inf, zeros)."""
        th = self._thetas(thetas)
        k = th.shape[0]
        self._ensure_batch(k, True)
        self._factored_ok = False
        out = np.empty(k)
        grad = np.zeros((k, self.ntheta))
        info = np.zeros(k, dtype=np.int32)
        if self._bad_data:
            return np.full(k, -np.inf), grad
        dpt, ipt = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        self._check(self.lib.mi_gp_lml_grad_batch(self.h, k, th.ctypes.data_as(dpt), out.ctypes.data_as(dpt),
                                                  grad.ctypes.data_as(dpt), info.ctypes.data_as(ipt)), "mi_gp_lml_grad_batch")
        self.batch_info = info
        return out, grad

    def lml_grad_data(self, theta, want_x=True):
        """(LML, dLML/dtheta, dLML/dy, dLML/dX) -- the data-side gradients drive the chain rule through
        warps (cwgp / iwgp) and free input rows (inverse_opt).  dLML/dX is None unless want_x."""
        val, grad = self.lml_grad(theta)
        if not np.isfinite(val):
            return val, grad, np.zeros(self.n), (np.zeros((self.n, self.d)) if want_x else None)
        alpha = np.empty(self.n)
        self._check(self.lib.mi_gp_alpha(self.h, alpha.ctypes.data_as(ctypes.POINTER(ctypes.c_double))), "mi_gp_alpha")
        gx = None
        if want_x:
            with torch.cuda.device(self.dev):
                if getattr(self, "_gx_t", None) is None:
                    self._gx_t = torch.empty((self.n, self.d), dtype=torch.float64, device=self.dev)
                    torch.cuda.synchronize(self.dev)
                self._check(self.lib.mi_gp_grad_x(self.h, self._gx_t.data_ptr()), "mi_gp_grad_x")
                gx = self._gx_t.cpu().numpy()
        return val, grad, -alpha, gx

    def update_data(self, X=None, y=None):
        """Overwrite the resident inputs / outputs in place (same shapes): warped data change at every
        posterior evaluation while the buffers, the handle and its streams stay.  Non-finite data (an overflowing warp)
        are not uploaded: until the next finite update lml / lml_grad return -inf like a non-positive-definite covariance
        (the reference's graph would produce NaN and PyMC's checks reject the point)."""
        self._factored_ok = False
        # Each array carries its own state: a non-finite one is not uploaded and stays "bad" until a finite replacement
        # arrives, the finite member of a pair IS uploaded (round 4 dropped it: a later update of the other array then
        # evaluated against stale data).
        if X is not None:
            self._bad_X = not np.isfinite(np.asarray(X, dtype=np.float64)).all()
            if self._bad_X:
                X = None
        if y is not None:
            self._bad_y = not np.isfinite(np.asarray(y, dtype=np.float64)).all()
            if self._bad_y:
                y = None
        self._bad_data = bool(getattr(self, "_bad_X", False) or getattr(self, "_bad_y", False))
        if X is None and y is None:
            return
        with torch.cuda.device(self.dev):
            if X is not None:
                X = np.ascontiguousarray(X, dtype=np.float64)
                if X.shape != (self.n, self.d):
                    raise ValueError("X must keep its shape")
                self.X_t.copy_(torch.from_numpy(X))
            if y is not None:
                y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
                if y.shape != (self.n,):
                    raise ValueError("y must keep its shape")
                self.y_t.copy_(torch.from_numpy(y))
            torch.cuda.synchronize(self.dev)

    def set_diag(self, diag=None):
        """Per-point diagonal added to K at assembly (None removes it)."""
        self._factored_ok = False
        with torch.cuda.device(self.dev):
            if diag is None:
                self._diag_t = None
                self._check(self.lib.mi_gp_set_diag(self.h, None), "mi_gp_set_diag")
                return
            diag = np.ascontiguousarray(np.asarray(diag, dtype=np.float64).reshape(-1))
            if diag.shape != (self.n,):
                raise ValueError("diag must have n entries")
            self._diag_t = torch.from_numpy(diag).to(self.dev)
            torch.cuda.synchronize(self.dev)
            self._check(self.lib.mi_gp_set_diag(self.h, self._diag_t.data_ptr()), "mi_gp_set_diag")

    def factor(self, theta):
        """Factorise for prediction (conditional form); returns LAPACK-style info."""
        theta, tp = self._theta(theta)
        self._factored_ok = False
        self._u_theta_ok = False
        self._pred_count = 0
        if self._bad_data:  # the resident arrays are not the caller's data (update_data refused a non-finite array)
            raise FloatingPointError("the last update_data carried non-finite values: nothing to factorise")
        self.info = self._check(self.lib.mi_gp_factor(self.h, tp), "mi_gp_factor")
        if self.info == 0:
            self._factored_ok, self._factored_theta = True, theta.copy()
        return self.info

    def _ensure_factored(self, theta):
        """Factorise unless the resident factor already belongs to this theta (BO sweeps, DE populations and
        refinement steps predict thousands of times at fixed hyper-parameters)."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if getattr(self, "_factored_ok", False) and np.array_equal(theta, self._factored_theta):
            return
        if self.factor(theta) != 0:
            raise FloatingPointError(f"covariance not positive definite at pivot {self.info}")

    def predict(self, theta, Xnew, pred_noise=True, chunk=16384, via_inverse=None):
        """Posterior mean and diagonal variance at Xnew (converted inputs), chunked over points.
        ``via_inverse``: use U = L^-T and one GEMM per chunk (mi_gp_predict_u) instead of the blocked triangular
        solve; default: when the gradient buffers exist and the sweep is large enough to pay for forming U
        (or U is already resident from an earlier sweep at this theta)."""
        self._ensure_factored(theta)
        Xnew = np.ascontiguousarray(Xnew, dtype=np.float64)
        if Xnew.ndim != 2 or Xnew.shape[1] != self.d:
            raise ValueError("Xnew must be (m, d)")
        m = Xnew.shape[0]
        if via_inverse is None:
            # U pays off for a large sweep, when it is already there, or from the third sweep on the same factor
            # (differential-evolution generations, refinement steps)
            self._pred_count = getattr(self, "_pred_count", 0) + 1
            via_inverse = self.Z_t is not None and (getattr(self, "_u_theta_ok", False) or 3 * m >= self.n
                                                    or self._pred_count >= 3)
        if via_inverse and self.Z_t is None:
            raise RuntimeError("this MiGP was created with need_grad=False")
        mean = np.empty(m)
        var = np.empty(m)
        with torch.cuda.device(self.dev):
            for s in range(0, m, chunk):
                mc = min(chunk, m - s)
                mp = (mc + 127) // 128 * 128
                rows = 2 * mp if via_inverse else mp
                if getattr(self, "_work", None) is None or self._work.shape[0] < rows:
                    self._work = torch.empty((rows, self.lda), dtype=torch.float64, device=self.dev)
                    torch.cuda.synchronize(self.dev)
                if mc <= self.PINNED_IO_MAX_POINTS:
                    # A few points (BO refinement steps, acquisition sweeps of small populations): the points go in and the
                    # moments come out through PINNED host memory the kernels address directly -- no torch copies, no torch
                    # synchronisation, one stream synchronisation inside the call (110 -> ~45 us per call at N = 512).  Same
                    # kernels on the same values: same bits as the device-buffer route below.
                    io = self._pinned_io(mc * self.d + 2 * mc)
                    io_np = io.numpy()
                    io_np[: mc * self.d] = Xnew[s : s + mc].ravel()
                    base = io.data_ptr()
                    fn = self.lib.mi_gp_predict_u if via_inverse else self.lib.mi_gp_predict
                    self._check(fn(self.h, base, mc, self._work.data_ptr(), self.lda, base + 8 * mc * self.d,
                                   base + 8 * (mc * self.d + mc), 1 if pred_noise else 0),
                                "mi_gp_predict_u" if via_inverse else "mi_gp_predict")
                    if via_inverse:
                        self._u_theta_ok = True
                    mean[s : s + mc] = io_np[mc * self.d : mc * self.d + mc]
                    var[s : s + mc] = io_np[mc * self.d + mc : mc * self.d + 2 * mc]
                    continue
                xn = torch.from_numpy(Xnew[s : s + mc]).to(self.dev)
                mu_t = torch.empty(mc, dtype=torch.float64, device=self.dev)
                var_t = torch.empty(mc, dtype=torch.float64, device=self.dev)
                torch.cuda.synchronize(self.dev)
                fn = self.lib.mi_gp_predict_u if via_inverse else self.lib.mi_gp_predict
                self._check(fn(self.h, xn.data_ptr(), mc, self._work.data_ptr(), self.lda, mu_t.data_ptr(),
                               var_t.data_ptr(), 1 if pred_noise else 0),
                            "mi_gp_predict_u" if via_inverse else "mi_gp_predict")
                if via_inverse:
                    self._u_theta_ok = True
                mean[s : s + mc] = mu_t.cpu().numpy()
                var[s : s + mc] = var_t.cpu().numpy()
        return mean, var

    def predict_grad(self, theta, Xnew, pred_noise=True, refactor=True):
        """Posterior mean / variance at a few points and their gradients w.r.t. the points:
        (mean[m], var[m], dmean[m,d], dvar[m,d]).  ``refactor=False`` reuses the resident factorisation
        (same theta as the previous predict / predict_grad call)."""
        if self.Z_t is None:
            raise RuntimeError("this MiGP was created with need_grad=False")
        if refactor:
            self._factored_ok = False
        self._ensure_factored(theta)
        Xnew = np.ascontiguousarray(Xnew, dtype=np.float64)
        if Xnew.ndim != 2 or Xnew.shape[1] != self.d:
            raise ValueError("Xnew must be (m, d)")
        m = Xnew.shape[0]
        mp = (m + 127) // 128 * 128
        with torch.cuda.device(self.dev):
            if getattr(self, "_work2", None) is None or self._work2.shape[0] < 2 * mp:
                self._work2 = torch.empty((2 * mp, self.lda), dtype=torch.float64, device=self.dev)
                torch.cuda.synchronize(self.dev)
            if m <= self.PINNED_IO_MAX_POINTS:
                # (pinned I/O as in predict(): this is the call BO's refinement steps make once per optimiser iteration)
                nout = 2 * m + 2 * m * self.d
                io = self._pinned_io(m * self.d + nout)
                io_np = io.numpy()
                io_np[: m * self.d] = Xnew.ravel()
                base = io.data_ptr()
                o_p = base + 8 * m * self.d
                self._check(self.lib.mi_gp_predict_grad(self.h, base, m, self._work2.data_ptr(), self.lda, o_p, o_p + 8 * m,
                                                        1 if pred_noise else 0, o_p + 16 * m, o_p + 16 * m + 8 * m * self.d),
                            "mi_gp_predict_grad")
                o = io_np[m * self.d : m * self.d + nout].copy()
                return (o[:m], o[m : 2 * m], o[2 * m : 2 * m + m * self.d].reshape(m, self.d),
                        o[2 * m + m * self.d :].reshape(m, self.d))
            xn = torch.from_numpy(Xnew).to(self.dev)
            out = torch.empty((2 * m + 2 * m * self.d,), dtype=torch.float64, device=self.dev)
            torch.cuda.synchronize(self.dev)
            mu_p, var_p = out.data_ptr(), out.data_ptr() + 8 * m
            dmu_p, dvar_p = out.data_ptr() + 16 * m, out.data_ptr() + 16 * m + 8 * m * self.d
            self._check(self.lib.mi_gp_predict_grad(self.h, xn.data_ptr(), m, self._work2.data_ptr(), self.lda, mu_p, var_p,
                                                    1 if pred_noise else 0, dmu_p, dvar_p), "mi_gp_predict_grad")
            o = out.cpu().numpy()
        return (o[:m], o[m : 2 * m], o[2 * m : 2 * m + m * self.d].reshape(m, self.d),
                o[2 * m + m * self.d :].reshape(m, self.d))

    PINNED_IO_MAX_POINTS = 256  # predict / predict_grad: up to this many points travel through pinned host memory

    def _pinned_io(self, nelem):
        """A pinned (page-locked, device-visible) host buffer of at least nelem doubles, grown on demand."""
        buf = getattr(self, "_pin_io", None)
        if buf is None or buf.numel() < nelem:
            buf = torch.empty(max(int(nelem), 4096), dtype=torch.float64).pin_memory()
            self._pin_io = buf
        return buf

    def set_option(self, what, value):
        """Per-handle tuning knobs (include/mi_gp.h: 0 look-ahead, 2 super-panel width, 7 small-tile threshold,
        8 one-workgroup-per-CU bulk updates, 14 tile order); unknown ids raise."""
        self._check(self.lib.mi_gp_set_option(self.h, int(what), int(value)), "mi_gp_set_option")

    def get_option(self, what, default=None):
        """Current value of a knob on this handle, the library's own defaults included (mi_gp_get_option; 40: 1 once the handle
        has switched its cross-stream edges to events by itself).  `default` for ids the library does not know."""
        out = ctypes.c_int()
        if self.lib.mi_gp_get_option(self.h, int(what), ctypes.byref(out)) != 0:
            return default
        return out.value

    def last_error(self):
        """Text of the handle's last error or notice (a demotion of the cross-stream edges is reported here once)."""
        return self.lib.mi_gp_last_error(self.h).decode()

    def set_profiling(self, level):
        self.lib.mi_gp_set_profiling(self.h, int(level))

    def timers(self):
        out = (ctypes.c_double * 14)()
        self.lib.mi_gp_timers(self.h, out, 14)
        keys = ["assemble_ms", "cholesky_ms", "reduce_ms", "total_ms", "gemm_ms", "gemm_flops", "gemm_launches",
                "trtri_ms", "lauum_ms", "contract_ms", "gemm_b_ms", "gemm_b_flops", "gemm_b_launches", "enqueue_ms"]
        return dict(zip(keys, list(out)))

    def close(self):
        if getattr(self, "h", None) is not None:
            self.lib.mi_gp_destroy(self.h)  # synchronises the handle's streams first
            self.h = None
        # the device buffers the handle borrowed (a batch holds K-fold copies of K, U, K^-1)
        for name in ("_bK", "_bZ", "_bW", "K_t", "Z_t", "W_t", "_work", "_work2", "_gx_t", "_pin_io"):
            if hasattr(self, name):
                setattr(self, name, None)
        self._batch_k = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
