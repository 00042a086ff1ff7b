"""One log-marginal-likelihood evaluation sharded over the GPUs of a node (second
row; BASELINE config 4): 1-D block-cyclic distribution of 512- or 1024-column super-panels of the covariance
over the ranks, right-looking Cholesky in which the owner factors a panel and broadcasts it
(torch.distributed: RCCL over xGMI with backend "nccl") and every rank updates the panels it owns.
One look-ahead step: the owner of the next panel updates and factors it first and its broadcast is
posted before the bulk updates, so the transfer overlaps them.

Only the exchange step is a collective (one broadcast per panel, two scalars all-reduced at the
end); assembly is local (X is replicated, 8*N*d bytes).  All arithmetic runs in the same HIP kernels
as the single-GPU path, reached through the block-level C-ABI entry points
(mi_gp_assemble_block, mi_gp_chol_panel, mi_gp_gemm_f64, mi_gp_lml_partial).

The gradient (section 8e, third row) shards K^-1 the same way.  Every rank keeps the panels it receives, so the
complete factor L is resident everywhere at no extra traffic (288 GB of HBM: 34 GB at N = 65536).  Then
  1. rows J of U = L^-T for the owned panels J: a right-side triangular solve against L with rows J of the
     identity (mi_gp_trsm_block) -- N^3/3 flops in total, split evenly by the block-cyclic ownership, no exchange;
  2. ONE exchange: each row panel of U is broadcast by its owner (upper part only, 4*N^2 bytes in total);
  3. the owned column slabs of K^-1 = U U^T (one triangular-k GEMM per slab, N^3/3 flops in total), alpha = U beta
     (replicated, bandwidth-bound), and the slab's share of 1/2 tr((alpha alpha^T - K^-1) dK/dtheta)
     (mi_gp_grad_contract_block);
  4. one all-reduce of the ntheta partial sums."""
import ctypes
import math

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .backend import parse_kernel

MINV = 128 * 128  # doubles per leaf inverse (mi_gp_chol_panel writes one per 128-column tile)
DINV_ROWS = 128  # pwt * MINV leaf-inverse doubles appended to a broadcast panel (flat view of 128 rows of pw + 16 doubles)


def panel_tiles(ntc, world):
    """Super-panel width in 128-column tiles.  One or two ranks: 1024 columns (k = 1024 updates, half as many and twice
    as large broadcasts; one rank, N = 16384: LML 42.7 -> 36.0 ms, LML + gradient 101 -> 91 ms) once every rank still owns
    at least four panels, else 512.  Four ranks and more: 512 -- not measured (no multi-GPU box yet), a model: with the
    bulk updates split over the ranks the serial chain `owner updates the next panel -> factors it -> broadcasts it`
    becomes the critical path, and its update + factor work per column grows with the panel width (N = 65536 on 8 ranks:
    ~5.7 ms per 1024-column panel x 64 against ~2 ms per 512-column panel x 128, for ~0.17 s of bulk updates per rank)."""
    if world >= 4:
        return 4
    return 8 if ntc >= 32 * world else 4


class DistGP:
    """A GP data set whose covariance is column-panel sharded over the ranks of the default process
    group (or a single process when torch.distributed is not initialised)."""

    def __init__(self, X, y, kernel="RBF", device=None, panel_width_tiles=None):
        if not torch.cuda.is_available():
            raise RuntimeError("DistGP needs ROCm GPUs: the GP hot path has no CPU implementation")
        self.lib = _lib.load()
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # With a process group the exchange steps are always issued (a one-rank group included: the RCCL call path is
        # then the one a multi-GPU run takes); without one there is nothing to exchange.
        self.collective = dist.is_initialized()
        self.bytes_broadcast = 0
        X = np.ascontiguousarray(X, dtype=np.float64)
        y = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        self.n, self.d = X.shape
        self.kerns, self.ops = parse_kernel(kernel)
        self.nkern = len(self.kerns)
        self.dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.np_ = (self.n + 127) // 128 * 128
        self.ntc = self.np_ // 128
        self.pwt = panel_tiles(self.ntc, self.world) if panel_width_tiles is None else int(panel_width_tiles)
        self.npan = (self.ntc + self.pwt - 1) // self.pwt
        self.pw = self.pwt * 128
        self.own = [j for j in range(self.npan) if j % self.world == self.rank]
        self.local_index = {j: i for i, j in enumerate(self.own)}
        self.ntheta = self.nkern * self.d + 2 * self.nkern + 2
        rows = self.np_ + 128  # + the y^T row block (forward solve folded into the factorisation)
        self.ld = max(len(self.own), 1) * self.pw + 16
        self.ldbuf = self.pw + 16
        with torch.cuda.device(self.dev):
            self.X_t = torch.from_numpy(X).to(self.dev)
            self.y_t = torch.from_numpy(y).to(self.dev)
            self.K = torch.zeros((rows, self.ld), dtype=torch.float64, device=self.dev)
            self.P = [torch.zeros((rows + DINV_ROWS, self.ldbuf), dtype=torch.float64, device=self.dev) for _ in range(2)]
            self.theta_t = torch.zeros(self.ntheta, dtype=torch.float64, device=self.dev)
            self.dinv = torch.zeros(self.pwt * MINV, dtype=torch.float64, device=self.dev)
            self.info = torch.zeros(4, dtype=torch.int32, device=self.dev)
            self.out = torch.zeros(16, dtype=torch.float64, device=self.dev)
            # look-ahead inside the rank: the owner of the next panel updates, factors and stages it on this stream while
            # its bulk updates with the current panel run on the main stream (round 2; one rank, N = 65536: see DESIGN.md)
            self.side = torch.cuda.Stream(device=self.dev)
        self.kids = (ctypes.c_int * _lib.MAX_KERN)(*[_lib.KERNEL_IDS[k] for k in self.kerns] + [0] * (_lib.MAX_KERN - self.nkern))
        self.opids = (ctypes.c_int * _lib.MAX_KERN)(*[_lib.OP_IDS[o] for o in self.ops] + [0] * (_lib.MAX_KERN - len(self.ops)))

    # ------------------------------------------------------------------ helpers
    def _w(self, j):
        return min(self.pwt, self.ntc - j * self.pwt)

    def _check(self, r, what):
        if r != 0:
            raise RuntimeError(f"{what} failed ({r}): {self.lib.mi_gp_last_global_error().decode()}")

    def _ptr(self, t, row, col):
        return t.data_ptr() + 8 * (row * t.stride(0) + col)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _assemble(self, j, noise_form):
        li, w = self.local_index[j], self._w(j)
        r0 = c0 = j * self.pw
        nrows, ncols = max(0, self.n - r0), max(0, min(self.n - c0, w * 128))
        self._check(self.lib.mi_gp_assemble_block(
            self.d, self.nkern, self.kids, self.opids, self.theta_t.data_ptr(),
            self.X_t.data_ptr() + 8 * r0 * self.d, nrows, self.X_t.data_ptr() + 8 * c0 * self.d, ncols, r0, c0,
            self._ptr(self.K, r0, li * self.pw), self.ld, self.np_ - r0, w * 128, noise_form, self._stream()),
            "mi_gp_assemble_block")
        yr = self.K[self.np_:, li * self.pw: li * self.pw + w * 128]
        yr.zero_()
        if ncols > 0:
            yr[0, :ncols] = self.y_t[c0: c0 + ncols]

    def _factor(self, j):
        li, w = self.local_index[j], self._w(j)
        r0 = j * self.pw
        self._check(self.lib.mi_gp_chol_panel(self._ptr(self.K, r0, li * self.pw), self.ld, (self.np_ + 128 - r0) // 128, w,
                                              self.dinv.data_ptr(), self.info.data_ptr(), r0, self._stream()),
                    "mi_gp_chol_panel")

    def _update(self, jt, j, buf):
        """panel jt (owned) -= P_j[rows >= jt] P_j[rows of jt]^T, lower trapezoid only."""
        li, wt, wj = self.local_index[jt], self._w(jt), self._w(j)
        rt = jt * self.pw
        m = self.np_ + 128 - rt
        off = rt - j * self.pw  # row of panel jt's diagonal block inside the broadcast buffer
        a_ptr = self._ptr(buf, off, 0)
        self._check(self.lib.mi_gp_gemm_f64(0, 1, m, wt * 128, wj * 128, -1.0, a_ptr, self.ldbuf, a_ptr, self.ldbuf, 1.0,
                                            self._ptr(self.K, rt, li * self.pw), self.ld, 1, 0, 1, 0, 0, 0, self._stream()),
                    "mi_gp_gemm_f64")

    def _stage(self, j, buf):
        li, w = self.local_index[j], self._w(j)
        r0 = j * self.pw
        rows = self.np_ + 128 - r0
        buf[:rows, : w * 128].copy_(self.K[r0:, li * self.pw: li * self.pw + w * 128])
        # the leaf inverses travel with the panel (128 more rows): the gradient's triangular solves need them everywhere
        buf[rows: rows + DINV_ROWS].view(-1)[: w * MINV].copy_(self.dinv[: w * MINV])

    # ------------------------------------------------------------------ evaluation
    def lml(self, theta, noise_form=0, _keep=False):
        """LML at natural-scale theta (C-ABI layout); -inf if the covariance is not positive definite."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if theta.shape != (self.ntheta,):
            raise ValueError(f"theta must have {self.ntheta} entries")
        if _keep:
            self._alloc_grad_buffers()
        with torch.cuda.device(self.dev):
            self.theta_t.copy_(torch.from_numpy(theta))
            self.info.fill_(0x7F7F7F7F)
            for j in self.own:
                self._assemble(j, noise_form)
            owner = lambda j: j % self.world  # noqa: E731
            if owner(0) == self.rank:
                self._factor(0)
                self._stage(0, self.P[0])
            work = self._bcast(0)
            main = torch.cuda.current_stream(self.dev)
            staged = None  # event behind the side stream's staging of the panel the next step consumes
            for j in range(self.npan):
                buf = self.P[j % 2]
                if work is not None:
                    work.wait()
                if staged is not None:
                    main.wait_event(staged)
                    staged = None
                if _keep:
                    self._keep_panel(j, buf)
                jn = j + 1
                work = None
                if jn < self.npan:
                    if owner(jn) == self.rank:
                        # everything queued on the main stream so far (the previous step's updates of panel jn, the reads
                        # of the buffer that is about to be restaged) precedes the side stream's work
                        ready = torch.cuda.Event()
                        ready.record(main)
                        with torch.cuda.stream(self.side):
                            self.side.wait_event(ready)
                            self._update(jn, j, buf)
                            self._factor(jn)
                            self._stage(jn, self.P[jn % 2])
                            staged = torch.cuda.Event()
                            staged.record(self.side)
                            work = self._bcast(jn)  # posted behind the staging: RCCL orders itself after the side stream
                    else:
                        work = self._bcast(jn)  # posted before the bulk updates so that it overlaps them
                for jt in self.own:
                    if jt > jn:
                        self._update(jt, j, buf)
            main.wait_stream(self.side)
            # local pieces of sum log L_ii and |beta|^2, then one small all-reduce
            acc = torch.zeros(3, dtype=torch.float64, device=self.dev)
            for j in self.own:
                li, w = self.local_index[j], self._w(j)
                c0 = j * self.pw
                nv = max(0, min(self.n - c0, w * 128))
                if nv == 0:
                    continue
                self._check(self.lib.mi_gp_lml_partial(self._ptr(self.K, c0, li * self.pw), self.ld,
                                                       self._ptr(self.K, self.np_, li * self.pw), nv, self.out.data_ptr(),
                                                       self._stream()), "mi_gp_lml_partial")
                acc[:2] += self.out[1:3]
            info = self.info[:1].clone()
            if self.collective:
                dist.all_reduce(acc, op=dist.ReduceOp.SUM)
                dist.all_reduce(info, op=dist.ReduceOp.MIN)
            logdet, quad = acc[0].item(), acc[1].item()
            self.info_value = int(info.item())
        if self.info_value != 0x7F7F7F7F:
            return -math.inf
        self.logdet, self.quad = logdet, quad
        return -0.5 * self.n * math.log(2.0 * math.pi) - 0.5 * quad - logdet

    def _bcast(self, j):
        if not self.collective:
            return None
        rows = self.np_ + 128 - j * self.pw + DINV_ROWS
        view = self.P[j % 2][:rows]  # contiguous leading rows of the panel buffer
        self.bytes_broadcast += view.numel() * 8
        return dist.broadcast(view, src=j % self.world, async_op=True)

    # ------------------------------------------------------------------ gradient
    def _alloc_grad_buffers(self):
        if getattr(self, "Lf", None) is not None:
            return
        with torch.cuda.device(self.dev):
            self.ldf = self.np_ + 16
            self.Lf = torch.zeros((self.np_, self.ldf), dtype=torch.float64, device=self.dev)   # complete factor
            self.Uf = torch.zeros((self.np_, self.ldf), dtype=torch.float64, device=self.dev)   # complete U = L^-T
            self.dinv_f = torch.zeros(self.ntc * MINV, dtype=torch.float64, device=self.dev)
            self.beta_f = torch.zeros(self.np_, dtype=torch.float64, device=self.dev)
            self.alpha_f = torch.zeros(self.np_, dtype=torch.float64, device=self.dev)
            self.W = torch.zeros((self.np_, self.ldbuf), dtype=torch.float64, device=self.dev)  # one K^-1 column slab
            self.S = torch.zeros((max(len(self.own), 1) * self.pw, self.ldf), dtype=torch.float64, device=self.dev)
            self.UP = [torch.zeros(self.pw * self.np_, dtype=torch.float64, device=self.dev) for _ in range(2)]
            nsc = self.lib.mi_gp_grad_contract_block_scratch(self.n, 0, self.pw, self.ntheta)
            self.part = torch.zeros(max(int(nsc), 1), dtype=torch.float64, device=self.dev)
            self.gslab = torch.zeros(self.ntheta, dtype=torch.float64, device=self.dev)

    def _keep_panel(self, j, buf):
        w = self._w(j)
        r0 = j * self.pw
        rows = self.np_ - r0
        self.Lf[r0:, r0: r0 + w * 128].copy_(buf[:rows, : w * 128])
        self.dinv_f[j * self.pwt * MINV: (j * self.pwt + w) * MINV].copy_(
            buf[rows + 128: rows + 128 + DINV_ROWS].view(-1)[: w * MINV])
        nv = max(0, min(self.n - r0, w * 128))  # beta = L^-1 y rides in the first row of the panel's y block
        self.beta_f[r0: r0 + nv].copy_(buf[rows, :nv])

    def _u_owned(self):
        """Rows of U = L^-T for the owned panels, stacked in ascending panel order in self.S: X L^T = (those rows of I).
        Right-looking over the column panels c: the rows whose panel index is <= c are a prefix of the stack, so every
        step is one diagonal solve (one panel of columns) and ONE update GEMM over all active rows -- exact staircase flops
        (N^3/3 over all ranks) without tall-skinny launches."""
        S = self.S
        S.zero_()
        for li, j in enumerate(self.own):
            w = self._w(j)
            torch.diagonal(S[li * self.pw: li * self.pw + w * 128], offset=j * self.pw).fill_(1.0)
        m = 0
        nxt = 0
        for c in range(self.npan):
            if nxt < len(self.own) and self.own[nxt] == c:
                m += self._w(c) * 128
                nxt += 1
            if m == 0:
                continue
            wc, c0 = self._w(c), c * self.pw
            self._check(self.lib.mi_gp_trsm_block(self.Lf.data_ptr(), self.ldf, self.dinv_f.data_ptr(), c * self.pwt, wc,
                                                  self._ptr(S, 0, c0), self.ldf, m, self._stream()), "mi_gp_trsm_block")
            nrem = self.np_ - c0 - wc * 128
            if nrem > 0:
                self._check(self.lib.mi_gp_gemm_f64(0, 1, m, nrem, wc * 128, -1.0, self._ptr(S, 0, c0), self.ldf,
                                                    self._ptr(self.Lf, c0 + wc * 128, c0), self.ldf, 1.0,
                                                    self._ptr(S, 0, c0 + wc * 128), self.ldf, 0, 0, 1, 0, 0, 0,
                                                    self._stream()), "mi_gp_gemm_f64")
        for li, j in enumerate(self.own):
            w, r0 = self._w(j), j * self.pw
            self.Uf[r0: r0 + w * 128, r0: self.np_].copy_(S[li * self.pw: li * self.pw + w * 128, r0: self.np_])

    def lml_grad(self, theta):
        """(LML, dLML/dtheta) at natural-scale theta, C-ABI parameter order; (-inf, zeros) if K is not positive definite.
        Same values on every rank."""
        val = self.lml(theta, 0, _keep=True)
        grad = np.zeros(self.ntheta)
        if not np.isfinite(val):
            return val, grad
        with torch.cuda.device(self.dev):
            # 1. owned row panels of U
            self._u_owned()
            # 2. the exchange: upper part of each row panel, packed, from its owner (two buffers in flight)
            if self.collective:
                pending = []
                for j in range(self.npan):
                    w, r0 = self._w(j), j * self.pw
                    cols = self.np_ - r0
                    pk = self.UP[j % 2][: w * 128 * cols].view(w * 128, cols)
                    if len(pending) == 2:
                        self._finish_u(*pending.pop(0))
                    if j % self.world == self.rank:
                        pk.copy_(self.Uf[r0: r0 + w * 128, r0: self.np_])
                    self.bytes_broadcast += pk.numel() * 8
                    pending.append((j, pk, dist.broadcast(pk, src=j % self.world, async_op=True)))
                while pending:
                    self._finish_u(*pending.pop(0))
            # 3. alpha = U beta (replicated), then slab by slab: K^-1[:, J] = U[J:, :] U[J, :]^T and its share of the trace
            self._check(self.lib.mi_gp_trmv_upper(self.Uf.data_ptr(), self.ldf, self.beta_f.data_ptr(), self.n,
                                                  self.alpha_f.data_ptr(), self._stream()), "mi_gp_trmv_upper")
            gsum = torch.zeros(self.ntheta, dtype=torch.float64, device=self.dev)
            for j in self.own:
                w, r0 = self._w(j), j * self.pw
                if r0 >= self.n:
                    continue  # padding only
                m = self.np_ - r0
                a_ptr = self._ptr(self.Uf, r0, r0)
                self._check(self.lib.mi_gp_gemm_f64(0, 1, m, w * 128, m, 1.0, a_ptr, self.ldf, a_ptr, self.ldf, 0.0,
                                                    self.W.data_ptr(), self.ldbuf, 1, 3, 1, 0, 0, 0, self._stream()),
                            "mi_gp_gemm_f64")
                self._check(self.lib.mi_gp_grad_contract_block(
                    self.d, self.nkern, self.kids, self.opids, self.theta_t.data_ptr(), self.X_t.data_ptr(), self.n,
                    self.W.data_ptr(), self.ldbuf, r0, r0, w * 128, self.alpha_f.data_ptr(), self.part.data_ptr(),
                    self.part.numel(), self.gslab.data_ptr(), self._stream()), "mi_gp_grad_contract_block")
                gsum += self.gslab
            # 4. one small all-reduce
            if self.collective:
                dist.all_reduce(gsum, op=dist.ReduceOp.SUM)
            grad = gsum.cpu().numpy()
        return val, grad

    def _finish_u(self, j, pk, work):
        work.wait()
        if j % self.world != self.rank:
            w, r0 = self._w(j), j * self.pw
            self.Uf[r0: r0 + w * 128, r0: self.np_].copy_(pk)
