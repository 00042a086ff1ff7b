"""One log-marginal-likelihood evaluation sharded over the GPUs of a node (second
row; BASELINE config 4): 1-D block-cyclic distribution of 512- or 1024-column super-panels of the covariance
over the ranks, right-looking Cholesky in which the owner factors a panel and broadcasts it
(torch.distributed: RCCL over xGMI with backend "nccl") and every rank updates the panels it owns.
One look-ahead step: the owner of the next panel updates and factors it first and its broadcast is
posted before the bulk updates, so the transfer overlaps them.  The broadcast is pipelined: a panel buffer is
piece-major (one contiguous piece per 128-column tile column) and the owner sends tile column c as soon as its strip is
done, while columns c + 1 .. are still being factored -- only the last piece's transfer is exposed (include/mi_gp.h,
"sharded factorisation"); the owner itself never waits for its own sends, only before it re-uses their buffer.

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
DINV_ROWS = 128  # the tile column's leaf inverse rides behind its rows in a piece (128 rows of 128 doubles)


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


class _Works:
    """The work handles of one panel's pieces: wait() orders the current stream behind all of them."""

    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()


class _Arrived:
    """Stand-in for a collective's work handle (emulation): wait() orders the current stream behind the panel's arrival."""

    def __init__(self, dev, ev):
        self.dev, self.ev = dev, ev

    def wait(self):
        torch.cuda.current_stream(self.dev).wait_event(self.ev)


class DistGP:
    """A GP data set whose covariance is column-panel sharded over the ranks of the default process
    group (or a single process when torch.distributed is not initialised).

    ``emulate=(world, rank)``: this single process plays rank ``rank`` of a ``world``-rank job -- it owns, updates and
    factors exactly that rank's panels; the panels other ranks would broadcast are copied in from a complete factor
    (``set_factor_source``) on a stand-in "link" stream.  tools/emulate_rank.py uses it to measure a rank's compute path
    and owner chain on one GPU before a multi-GPU node is available."""

    def __init__(self, X, y, kernel="RBF", device=None, panel_width_tiles=None, emulate=None):
        if not torch.cuda.is_available():
            raise RuntimeError("DistGP needs ROCm GPUs: the GP hot path has no CPU implementation")
        self.lib = _lib.load()
        self.emulate = emulate is not None
        if self.emulate:
            self.world, self.rank = int(emulate[0]), int(emulate[1])
            self.collective = False
        else:
            self.rank = dist.get_rank() if dist.is_initialized() else 0
            self.world = dist.get_world_size() if dist.is_initialized() else 1
            # With a process group the exchange steps are always issued (a one-rank group included: the RCCL call path is
            # then the one a multi-GPU run takes); without one there is nothing to exchange.
            self.collective = dist.is_initialized()
        self.bytes_broadcast = 0
        self.lazy_sends = None  # None: by world size (see lml); True / False: the owner never / always waits for its panel sends
        # "bcast": one dist.broadcast per piece (RCCL's ring / tree out of the owner).  "mesh": the owner sends 1/(W-1) of a
        # piece to every peer over that peer's own xGMI link and the peers all-gather among themselves -- two transfers of
        # bytes/(W-1) per link instead of one of `bytes` over one link (see _mesh).  UNVERIFIED on hardware (no multi-GPU box
        # in rounds 1-5): opt-in, covered by the gloo multi-rank tests through host staging.
        self.exchange = "bcast"
        self.peer_groups = None
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
            # piece-major panel buffers: piece c = tile column c of a panel (its rows at a stride of 128, then its leaf inverse)
            # (+ world rows: the mesh exchange pads a piece to a multiple of world - 1 chunks)
            self.P = [torch.zeros((self.pwt, rows + DINV_ROWS + self.world, 128), dtype=torch.float64, device=self.dev) for _ in range(2)]
            self.xfer = torch.cuda.Stream(device=self.dev)  # the mesh exchange's receive -> all-gather sequence
            self.theta_t = torch.zeros(self.ntheta, dtype=torch.float64, device=self.dev)
            self.info = torch.zeros(4, dtype=torch.int32, device=self.dev)
            self.out = torch.zeros(16, dtype=torch.float64, device=self.dev)
            # look-ahead inside the rank: the owner of the next panel updates, factors and stages it on this stream while
            # its bulk updates with the current panel run on the main stream
            self.side = torch.cuda.Stream(device=self.dev)
            self.link = torch.cuda.Stream(device=self.dev) if self.emulate else None  # stands in for RCCL's own stream
        self.kids = (ctypes.c_int * _lib.MAX_KERN)(*[_lib.KERNEL_IDS[k] for k in self.kerns] + [0] * (_lib.MAX_KERN - self.nkern))
        self.opids = (ctypes.c_int * _lib.MAX_KERN)(*[_lib.OP_IDS[o] for o in self.ops] + [0] * (_lib.MAX_KERN - len(self.ops)))
        # everything of a panel step except the exchange sits behind one C entry (mi_gp_shard_step)
        cfg = _lib.MiGpShardConfig()
        cfg.n, cfg.d, cfg.nkern = self.n, self.d, self.nkern
        for i in range(_lib.MAX_KERN):
            cfg.kernel_ids[i], cfg.ops[i] = self.kids[i], self.opids[i]
        cfg.panel_tiles, cfg.world, cfg.rank, cfg.device = self.pwt, self.world, self.rank, self.dev.index
        cfg.X_dev, cfg.y_dev, cfg.K_dev, cfg.ldk = self.X_t.data_ptr(), self.y_t.data_ptr(), self.K.data_ptr(), self.ld
        cfg.P_dev[0], cfg.P_dev[1], cfg.ldp = self.P[0].data_ptr(), self.P[1].data_ptr(), self.P[0].stride(0)
        cfg.theta_dev, cfg.info_dev, cfg.out_dev = self.theta_t.data_ptr(), self.info.data_ptr(), self.out.data_ptr()
        self.sh = ctypes.c_void_p()
        r = self.lib.mi_gp_shard_create(ctypes.byref(cfg), ctypes.byref(self.sh))
        if r != 0:
            raise RuntimeError(f"mi_gp_shard_create failed ({r}): {self.lib.mi_gp_shard_last_error(None).decode()}")
        self.source = None
        self.link_ms = 0.0

    def close(self):
        if getattr(self, "sh", None):
            self.lib.mi_gp_shard_destroy(self.sh)
            self.sh = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    # ------------------------------------------------------------------ helpers
    def _w(self, j):
        return min(self.pwt, self.ntc - j * self.pwt)

    def _check(self, r, what):
        if r != 0:
            raise RuntimeError(f"{what} failed ({r}): {self.lib.mi_gp_last_global_error().decode()}")

    def _scheck(self, r, what):
        if r != 0:
            raise RuntimeError(f"{what} failed ({r}): {self.lib.mi_gp_shard_last_error(self.sh).decode()}")

    def _ptr(self, t, row, col):
        return t.data_ptr() + 8 * (row * t.stride(0) + col)

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def set_option(self, what, value):
        """mi_gp_shard_set_option: 0 bulk updates one workgroup per CU beside the chain, 1 per-step events, 2 early update,
        3 chain on the main stream ahead of the bulk update (default on several ranks), 4 tiles of a bulk update that run one
        workgroup per CU beside this rank's chain, the rest two per CU (default 2048, 0: all), 5 a chain on the main stream
        stages (and this class sends) every tile column behind its strip (1; 2, the default: the first column is also
        updated first and alone; 0: the panel behind its last column)."""
        self._scheck(self.lib.mi_gp_shard_set_option(self.sh, int(what), int(value)), "mi_gp_shard_set_option")

    def step_times(self):
        """[npan + 1][4] ms: update / factor / stage (side stream) and bulk (main stream) of every step of the last
        evaluation (needs set_option(1, 1)); row npan holds panel 0's factorisation and staging."""
        buf = (ctypes.c_double * (4 * (self.npan + 1)))()
        n = self.lib.mi_gp_shard_times(self.sh, buf, self.npan + 1)
        return np.array(buf[: 4 * max(n, 0)]).reshape(-1, 4)

    def piece_times(self):
        """[npan + 1][pwt] ms from the start of a step's chain to the end of the staging of each tile column of the panel
        it produces (needs set_option(1, 1)); row npan: panel 0."""
        buf = (ctypes.c_double * (self.pwt * (self.npan + 1)))()
        n = self.lib.mi_gp_shard_piece_times(self.sh, buf, self.npan + 1)
        return np.array(buf[: self.pwt * max(n, 0)]).reshape(-1, self.pwt)

    def set_factor_source(self, K_full, ld_full):
        """emulation: a complete factor (padded n + 128 rows: L in the lower triangle, beta^T in row np) that the panels
        of other ranks are copied from."""
        self.source = (K_full, ld_full)

    # ------------------------------------------------------------------ evaluation
    def lml(self, theta, noise_form=0, _keep=False):
        """LML at natural-scale theta (C-ABI layout); -inf if the covariance is not positive definite."""
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if theta.shape != (self.ntheta,):
            raise ValueError(f"theta must have {self.ntheta} entries")
        if _keep:
            self._alloc_grad_buffers()
        lib, sh = self.lib, self.sh
        owner = lambda j: j % self.world  # noqa: E731
        with torch.cuda.device(self.dev):
            main = torch.cuda.current_stream(self.dev)
            ms, ss = ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(self.side.cuda_stream)
            self.theta_t.copy_(torch.from_numpy(theta))
            self._scheck(lib.mi_gp_shard_begin(sh, noise_form, ms, ss), "mi_gp_shard_begin")
            # sends[b]: the broadcasts this rank posted as the OWNER out of buffer b.  The owner has the panel already: it
            # never waits for them on its compute streams, only before it stages into that buffer again (a receive into it
            # is ordered behind them on the transport's own stream anyway).
            sends = [None, None]
            # (Only RCCL runs a communicator's collectives strictly one after the other.  gloo -- the tests' transport -- may
            # run two at once: there a later receive into a buffer could overtake this rank's own send out of it to a slow
            # peer, so with any other backend the owner waits for its sends at once, as it did until round 4.)
            # Lazy sends on a real multi-rank RCCL job have never run on hardware (no multi-GPU box in rounds 1-5): they are
            # opt-in there (``lazy_sends = True``) until a 2-GPU run has shown lazy and eager results bit-equal; a one-rank
            # communicator (every panel is the rank's own) and the emulation keep them.
            if not self.collective:
                lazy_sends = True
            elif self.lazy_sends is not None:
                lazy_sends = bool(self.lazy_sends) and dist.get_backend() == "nccl"
            else:
                lazy_sends = dist.get_backend() == "nccl" and self.world == 1
            work = self._exchange(0)  # panel 0 was staged on the main stream by its owner
            if owner(0) == self.rank:
                if work is not None and not lazy_sends:
                    work.wait()
                    work = None
                sends[0], work = work, None
            # the stream the owner's chain runs on (mi_gp_shard option 3): the side stream beside the bulk update on one
            # rank, the main stream ahead of it on several
            chain_stream = main if lib.mi_gp_shard_chain_stream(sh) == 1 else self.side
            for j in range(self.npan):
                jn = j + 1
                mine = jn < self.npan and owner(jn) == self.rank
                if work is not None:
                    work.wait()  # the main stream waits for panel j ...
                    if mine and chain_stream is not main:
                        with torch.cuda.stream(self.side):
                            work.wait()  # ... and so does the side stream, whose chain reads it first
                work = None
                if chain_stream is not main and owner(j) == self.rank:
                    # this rank's side stream produced panel j: its chain READ P[(j - 1) % 2], the buffer the next exchange
                    # (posted under the main stream just below) writes.  Nothing else orders that write behind the read.
                    main.wait_stream(self.side)
                if mine and sends[jn % 2] is not None:
                    with torch.cuda.stream(chain_stream):
                        sends[jn % 2].wait()  # this step stages panel jn into the buffer those sends read
                    sends[jn % 2] = None
                if jn < self.npan and not mine:
                    work = self._exchange(jn)  # posted before this step's launches so that it overlaps them
                    sends[jn % 2] = None  # (ordered behind this rank's earlier sends out of that buffer by the transport)
                self._scheck(lib.mi_gp_shard_step(sh, j, ms, ss), "mi_gp_shard_step")
                if mine:
                    with torch.cuda.stream(self.side):
                        # piece by piece behind its staging (mi_gp_shard_wait_piece), not behind the later columns'
                        # factorisation or the bulk update
                        sends[jn % 2] = self._exchange(jn)
                        if sends[jn % 2] is not None and not lazy_sends:
                            sends[jn % 2].wait()
                            sends[jn % 2] = None
                if _keep:
                    self._keep_panel(j, self.P[j % 2])
                    if chain_stream is not main:
                        # the copy above reads P[j % 2] on the main stream AFTER this step's ready / bulk events were
                        # recorded, and the side stream's next staging (step j + 1, panel j + 2) overwrites that very
                        # buffer: order it behind the copy (nothing else did; a write-after-read race)
                        kept = torch.cuda.Event()
                        kept.record(main)
                        self.side.wait_event(kept)
            for b in range(2):
                if sends[b] is not None:
                    sends[b].wait()  # the next evaluation re-uses the buffers
            self._scheck(lib.mi_gp_shard_finish(sh, ms, ss), "mi_gp_shard_finish")
            # local pieces of sum log L_ii and |beta|^2, then one small all-reduce
            acc = self.out[1:3].clone()
            info = self.info[:1].clone()
            if self.collective:
                dist.all_reduce(acc, op=dist.ReduceOp.SUM)
                dist.all_reduce(info, op=dist.ReduceOp.MIN)
            logdet, quad = acc[0].item(), acc[1].item()
            self.info_value = int(info.item())
        if self.info_value != 0x7F7F7F7F:
            return -math.inf
        self.logdet, self.quad = logdet, quad
        if self.emulate:
            return float("nan")  # a single emulated rank holds only its share of the two sums
        return -0.5 * self.n * math.log(2.0 * math.pi) - 0.5 * quad - logdet

    def set_exchange(self, mode):
        """"bcast" (default) or "mesh"; collective: every rank has to make the same call (the peer groups are created here)."""
        if mode not in ("bcast", "mesh"):
            raise ValueError(mode)
        if mode == "mesh" and self.collective and self.world > 2 and self.peer_groups is None:
            # peer_groups[o]: everybody but rank o (the ranks that all-gather a piece owner o scattered)
            self.peer_groups = [dist.new_group([r for r in range(self.world) if r != o]) for o in range(self.world)]
        if mode == "mesh" and self.collective and self.world > 1 and dist.get_backend() == "nccl" and not getattr(self, "_mesh_checked", False):
            self._mesh_selfcheck()
        self.exchange = mode

    def _mesh_selfcheck(self):
        """the RCCL path of the mesh exchange (grouped point-to-point sends, peer all-gather on per-owner subgroups)
        never ran on hardware in rounds 1-6.  Before the first real panel travels that way, every owner in turn sends a small
        pattern through exactly those primitives and through dist.broadcast; any rank that sees a difference raises (on
        every rank: the verdict is all-reduced), so a transport problem shows up here and not as a wrong factor."""
        W = self.world
        npeer, cr = W - 1, 64
        bad = torch.zeros(1, dtype=torch.int32, device=self.dev)
        with torch.cuda.device(self.dev):
            for o in range(W):
                ref = torch.zeros((cr * npeer, 128), dtype=torch.float64, device=self.dev)
                got = torch.zeros_like(ref)
                if o == self.rank:
                    ref.copy_(torch.arange(ref.numel(), dtype=torch.float64, device=self.dev).reshape(ref.shape) * (o + 1) + 0.25)
                    got.copy_(ref)
                dist.broadcast(ref, src=o)
                peers = [r for r in range(W) if r != o]
                chunks = [got[g * cr: (g + 1) * cr] for g in range(npeer)]
                if o == self.rank:
                    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, chunks[g], peers[g]) for g in range(npeer)]):
                        w.wait()
                else:
                    g = peers.index(self.rank)
                    for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, chunks[g], o)]):
                        w.wait()
                    if npeer > 1:
                        dist.all_gather_into_tensor(got, chunks[g].clone(), group=self.peer_groups[o])
                torch.cuda.synchronize(self.dev)
                if not torch.equal(ref, got):
                    bad += 1
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) != 0:
            raise RuntimeError("DistGP.set_exchange('mesh'): the mesh exchange's transport self-check disagrees with dist.broadcast "
                               "on at least one rank; staying with the broadcast exchange")
        self._mesh_checked = True

    def _exchange(self, j):
        """Post the exchange of panel j under the current stream, one piece (tile column) at a time: the RCCL broadcasts
        from its owner -- who orders each behind that piece's staging -- or, emulating another rank's panel, copies from
        the complete factor on the stand-in link stream.  Returns something with .wait() (makes the then-current stream
        wait for the whole panel) or None when there is nothing to wait for."""
        if self.collective:
            return self._mesh(j) if (self.exchange == "mesh" and self.world > 1) else self._bcast(j)
        if self.emulate and j % self.world != self.rank:
            return self._standin(j)
        return None  # this rank's own panel: nothing arrives, and nothing is sent in an emulation

    def _piece_rows(self, j):
        return self.np_ + 128 - j * self.pw  # rows of panel j in a piece (the y^T block included)

    def _standin(self, j):
        src, _ = self.source
        w, r0, rows = self._w(j), j * self.pw, self._piece_rows(j)
        cur = torch.cuda.current_stream(self.dev)
        self.link.wait_stream(cur)  # like the transport: ordered behind the stream the exchange was posted under
        with torch.cuda.stream(self.link):
            for c in range(w):
                self.P[j % 2][c, :rows].copy_(src[r0: r0 + rows, r0 + c * 128: r0 + (c + 1) * 128])
            ev = torch.cuda.Event()
            ev.record(self.link)
        self.bytes_broadcast += w * (rows + DINV_ROWS) * 128 * 8
        return _Arrived(self.dev, ev)

    def _bcast(self, j):
        if not self.collective:
            return None
        src, rows = j % self.world, self._piece_rows(j) + DINV_ROWS
        works = []
        for c in range(self._w(j)):
            view = self.P[j % 2][c, :rows]  # contiguous: the column's rows, then its leaf inverse
            if src == self.rank:
                self._scheck(self.lib.mi_gp_shard_wait_piece(self.sh, c, self._stream()), "mi_gp_shard_wait_piece")
            self.bytes_broadcast += view.numel() * 8
            works.append(dist.broadcast(view, src=src, async_op=True))
        return _Works(works)

    def _mesh(self, j):
        """Panel j over the xGMI mesh, piece by piece: the owner sends chunk g (1 / (W - 1) of the piece's rows) to peer g --
        W - 1 point-to-point sends in ONE grouped launch, one per outgoing link -- and the peers all-gather the chunks among
        themselves (peer_groups[owner]), each over its links to the other peers.  Per link a piece costs 2 bytes / (W - 1)
        instead of `bytes` for a broadcast that leaves the owner over one link (3.5 x at W = 8).  The receive and the
        all-gather run on the transfer stream, ordered behind the stream the exchange was posted under (like _bcast); the
        returned handle makes the then-current stream wait for the last all-gather.  With gloo (the tests' transport, no GPU
        point-to-point) the chunks are staged through host memory, synchronously."""
        W, o = self.world, j % self.world
        rows = self._piece_rows(j) + DINV_ROWS
        npeer = W - 1
        cr = -(-rows // npeer)  # rows per chunk; the padding rows of the last chunk travel too (the buffers have room)
        peers = [r for r in range(W) if r != o]
        nccl = dist.get_backend() == "nccl"
        cur = torch.cuda.current_stream(self.dev)
        works = []
        for c in range(self._w(j)):
            padded = self.P[j % 2][c, : cr * npeer]
            chunks = [padded[g * cr: (g + 1) * cr] for g in range(npeer)]
            self.bytes_broadcast += padded.numel() * 8
            if o == self.rank:
                self._scheck(self.lib.mi_gp_shard_wait_piece(self.sh, c, self._stream()), "mi_gp_shard_wait_piece")
                if nccl:
                    works += dist.batch_isend_irecv([dist.P2POp(dist.isend, chunks[g], peers[g]) for g in range(npeer)])
                else:
                    for g in range(npeer):
                        dist.send(chunks[g].cpu(), peers[g])  # (.cpu() is ordered behind the staging on the current stream)
            else:
                g = peers.index(self.rank)
                if nccl:
                    self.xfer.wait_stream(cur)
                    with torch.cuda.stream(self.xfer):
                        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, chunks[g], o)]):
                            w.wait()  # the transfer stream waits for the chunk (the compute streams do not)
                        if npeer > 1:
                            works.append(dist.all_gather_into_tensor(padded, chunks[g], group=self.peer_groups[o], async_op=True))
                        else:
                            ev = torch.cuda.Event()
                            ev.record(self.xfer)
                            works.append(_Arrived(self.dev, ev))
                else:
                    mine = torch.empty(chunks[g].shape, dtype=torch.float64)
                    dist.recv(mine, o)
                    parts = [mine]
                    if npeer > 1:
                        parts = [torch.empty_like(mine) for _ in range(npeer)]
                        dist.all_gather(parts, mine, group=self.peer_groups[o])
                    for q in range(npeer):
                        chunks[q].copy_(parts[q])  # stream-ordered behind the earlier readers of this buffer
        return _Works(works) if works else None

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
        w, r0, rows = self._w(j), j * self.pw, self._piece_rows(j)
        for c in range(w):
            c0 = r0 + c * 128
            self.Lf[r0:, c0: c0 + 128].copy_(buf[c, : self.np_ - r0])
            self.dinv_f[(j * self.pwt + c) * MINV: (j * self.pwt + c + 1) * MINV].copy_(buf[c, rows: rows + DINV_ROWS].reshape(-1))
            nv = max(0, min(self.n - c0, 128))  # beta = L^-1 y rides in the first row of the piece's y block
            self.beta_f[c0: c0 + nv].copy_(buf[c, self.np_ - r0, :nv])

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
