/* mi_gp.h -- C-ABI of libmi_gp.so, the MI355X (gfx950) GP log-marginal-likelihood backend.
 *
 * The reference (andrew-angus/andvaranaut) has no FFI for this path: the seam is the PyMC model
 * built in GPMCMC.__fit (gpmcmc.py:189-323) and consumed by pm.find_MAP / pm.sample / gp.predict
 * (gpmcmc.py:345,351,593).  In PyMC terms that seam is one value-and-gradient callable
 * theta -> (logp, dlogp) and one conditional (theta, X*) -> (mu*, var*); the entry points below
 * are exactly those two, plus the block-level operations they are built from.  Each entry point
 * names the reference call site it replaces.
 *
 * Conventions
 *  - all arrays are float64, row-major; pointers named *_dev are device (HBM) addresses, e.g.
 *    torch.Tensor.data_ptr(); the library borrows them and never frees them;
 *  - theta is a HOST array in natural (untransformed) scale:
 *        [ ls(nkern*d) | kv(nkern) | alpha(nkern, RatQuad only) | gv | jitter ]
 *    host Python applies priors, log/interval transforms and their Jacobians (gpmcmc.py:193-208);
 *  - every function returns int: 0 ok; >0 LAPACK-style info = 1-based index of the first
 *    non-positive pivot (the LML is then -inf, as PyMC's "posdef" check gives);
 *    <0 bad argument (-1) or HIP/RCCL failure (-2); text via mi_gp_last_error().  Never aborts.
 *  - a handle is bound to one device and one HIP stream and is NOT thread-safe; calls return after
 *    the stream has been synchronised (scalar outputs are valid on return).
 */
#ifndef MI_GP_H
#define MI_GP_H
#ifdef __cplusplus
extern "C" {
#endif

enum { MI_GP_RBF = 0, MI_GP_MATERN52 = 1, MI_GP_MATERN32 = 2, MI_GP_EXPONENTIAL = 3, MI_GP_RATQUAD = 4 };
enum { MI_GP_OP_ADD = 0, MI_GP_OP_MUL = 1 };
#define MI_GP_MAX_KERN 8

/* libmi_gp.so is built with -fvisibility=hidden: these entry points are the only symbols it exports */
#define MI_GP_API __attribute__((visibility("default")))

typedef struct mi_gp_handle mi_gp_handle;

/* kernel string of GPMCMC.change_model (gpmcmc.py:497-515) already split into ids and ops */
typedef struct mi_gp_config {
  int n;                          /* training points */
  int d;                          /* input dimensions (nx) */
  int nkern;                      /* kernel components, <= MI_GP_MAX_KERN */
  int kernel_ids[MI_GP_MAX_KERN]; /* MI_GP_RBF ... */
  int ops[MI_GP_MAX_KERN];        /* ops[i] joins component i and i+1, applied left to right */
  int device;                     /* HIP device ordinal */
  int panel_tiles;                /* Cholesky super-panel width in 128-column tiles; 0 = default */
} mi_gp_config;

/* device buffers, allocated by the caller (PyTorch tensors in the Python host) */
typedef struct mi_gp_buffers {
  const double* X_dev; /* n x d   converted inputs  (xin of gpmcmc.py:235-237) */
  const double* y_dev; /* n       converted outputs (yin of gpmcmc.py:279)     */
  double* K_dev;       /* (np+128) x lda, np = mi_gp_padded_n(): covariance, overwritten by its
                          lower Cholesky factor; the extra 128 rows carry y^T -> beta^T = (L^-1 y)^T */
  long lda;            /* leading dimension of K/Z/W in elements: even, >= np */
  double* Z_dev;       /* np x lda  U = L^-T, upper triangular (mi_gp_lml_grad / mi_gp_predict_grad only; else may be NULL) */
  double* W_dev;       /* np x lda  K^-1 (lower triangle) and GEMM scratch (same entry points; else may be NULL) */
} mi_gp_buffers;

MI_GP_API const char* mi_gp_last_global_error(void);
MI_GP_API const char* mi_gp_last_error(mi_gp_handle* h);

MI_GP_API int mi_gp_create(const mi_gp_config* cfg, mi_gp_handle** out);
MI_GP_API int mi_gp_destroy(mi_gp_handle* h);
MI_GP_API long mi_gp_padded_n(const mi_gp_handle* h);   /* n rounded up to a multiple of 128 */
MI_GP_API int mi_gp_num_theta(const mi_gp_handle* h);   /* nkern*d + 2*nkern + 2 */
MI_GP_API void* mi_gp_stream(const mi_gp_handle* h);    /* the handle's hipStream_t */
MI_GP_API int mi_gp_set_data(mi_gp_handle* h, const mi_gp_buffers* buffers);

/* LML(theta) = -1/2 |L^-1 y|^2 - sum log L_ii - n/2 log 2pi with L = chol(K(theta) + gv I + jitter I).
 * Replaces the logp evaluation of gp.marginal_likelihood (gpmcmc.py:321-323; explicit form :311-318). */
MI_GP_API int mi_gp_lml(mi_gp_handle* h, const double* theta_host, double* lml_out);
/* sum log L_ii and |L^-1 y|^2 of the last factorisation */
MI_GP_API int mi_gp_lml_parts(mi_gp_handle* h, double* logdet_out, double* quad_out);

/* LML and its gradient w.r.t. the natural parameters, grad_out[mi_gp_num_theta()] in theta order
 * (d/d jitter equals d/d gv):  dLML/dtheta_k = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta_k), K^-1 = L^-T L^-1
 * formed on the fp64 MFMA GEMM.  Replaces the dlogp half of model.logp_dlogp_function that
 * pm.find_MAP (gpmcmc.py:332,345,357) and pm.sample / NUTS (gpmcmc.py:351) call per step. */
MI_GP_API int mi_gp_lml_grad(mi_gp_handle* h, const double* theta_host, double* lml_out, double* grad_out);

/* Batched evaluation: `count` covariances of the SAME inputs, one theta each, factorised in lockstep (every kernel launch
 * of the evaluation carries blockIdx.z = problem).  One evaluation below N ~ 10^4 is bound by its serial panel chain and
 * leaves most of the chip idle; the reference evaluates the same data at several theta at once in its MAP restarts
 * (gpmcmc.py:328-343) and in the chains of pm.sample (gpmcmc.py:351).  Buffers (PyTorch tensors, borrowed): problem p uses
 * K_dev + p * stride_k ((np + 128) x lda each, the lda of mi_gp_set_data) and, for the gradient, Z_dev / W_dev +
 * p * stride_zw (np x lda each); strides in elements, even.  Results are those of mi_gp_lml / mi_gp_lml_grad at the same
 * theta (same arithmetic per element).  info_out[p] (optional): 0, or the 1-based index of problem p's first non-positive
 * pivot (lml_out[p] = -inf, its gradient 0).  The handle's single-evaluation state (factor, K^-1) is invalidated. */
typedef struct mi_gp_batch_buffers {
  double* K_dev;
  double* Z_dev; /* may be NULL: mi_gp_lml_batch only */
  double* W_dev; /* may be NULL: mi_gp_lml_batch only */
  long stride_k;
  long stride_zw;
  int count;
} mi_gp_batch_buffers;
MI_GP_API int mi_gp_set_batch(mi_gp_handle* h, const mi_gp_batch_buffers* buffers);
MI_GP_API int mi_gp_lml_batch(mi_gp_handle* h, int k, const double* thetas_host, double* lml_out, int* info_out);
MI_GP_API int mi_gp_lml_grad_batch(mi_gp_handle* h, int k, const double* thetas_host, double* lml_out, double* grads_out, int* info_out);

/* Data-side gradients at the theta of the last successful mi_gp_lml_grad (K^-1 and alpha still resident):
 *   mi_gp_alpha  : alpha = K^-1 y (n doubles to the host); dLML/dy = -alpha
 *   mi_gp_grad_x : dLML/dX (n x d row-major, device), dLML/dx_im = sum_j (alpha_i alpha_j - Kinv_ij) dK_ij/dx_im
 * They carry the chain rule through the output / input warps whose parameters the reference samples together
 * with the hyper-parameters (cwgp / iwgp, gpmcmc.py:211-279, Jacobian term :319) and through the free
 * observation rows of inverse_opt (gpmcmc.py:1096-1101,1156-1165); PyMC obtains both by autodiff.  Any d (beyond 128 input
 * dimensions mi_gp_grad_x runs one pass per window of 128 output dimensions). */
MI_GP_API int mi_gp_alpha(mi_gp_handle* h, double* alpha_host);
MI_GP_API int mi_gp_grad_x(mi_gp_handle* h, double* gx_dev);
/* Optional per-point diagonal (n doubles, device, borrowed; NULL removes it) added to K on top of theta's
 * (gv, jitter): the noise vector `ynoise` of inverse_opt (gpmcmc.py:1134-1158, K += diag(ynoise)). */
MI_GP_API int mi_gp_set_diag(mi_gp_handle* h, const double* diag_dev);

/* Factorise K(theta) + jitter I + gv I (the conditional's form, [3P] Marginal._build_conditional) and
 * keep L and beta = L^-1 y on the device for mi_gp_predict.  Replaces the first half of
 * gp.predict(x, point=hyps, diag=True, pred_noise=True) at gpmcmc.py:593-594. */
MI_GP_API int mi_gp_factor(mi_gp_handle* h, const double* theta_host);
/* Posterior mean and diagonal variance at m new points (Xnew_dev m x d, converted inputs):
 * A = L^-1 K(X, X*), mu = A^T beta, var = kdiag - colsum(A o A) (+ gv if pred_noise); the same algebra
 * is written out in-tree at gpmcmc.py:766-778.  work_dev is caller scratch of
 * ceil(m/128)*128 rows x ldw (ldw even, >= mi_gp_padded_n()); mean_dev / var_dev receive m doubles.
 * Xnew_dev, mean_dev and var_dev (and dmean_dev / dvar_dev below) may be any DEVICE-VISIBLE address: device memory, or pinned
 * host memory (hipHostMalloc) -- for a few points the Python host passes the latter and skips every copy around the call
 * (up to 256 points: 110 -> 48 us per call at N = 512); the call returns with the stream synchronised either way. */
MI_GP_API int mi_gp_predict(mi_gp_handle* h, const double* Xnew_dev, int m, double* work_dev, long ldw, double* mean_dev,
                  double* var_dev, int pred_noise);
/* The same conditional through U = L^-T (formed once per mi_gp_factor, N^3/3 flops): A = K(X*, X) U is one GEMM with
 * a triangular k-range instead of the blocked triangular solve -- the path for sweeps of many points at fixed
 * hyper-parameters (BO's 10 000-point proposals at gpmcmc.py:691-697).  Needs Z_dev / W_dev; work_dev must hold
 * 2 * ceil(m/128)*128 rows x ldw. */
MI_GP_API int mi_gp_predict_u(mi_gp_handle* h, const double* Xnew_dev, int m, double* work_dev, long ldw, double* mean_dev,
                    double* var_dev, int pred_noise);
/* The same plus d mu / d x* and d var / d x* (m x d each, device): d mu = sum_i alpha_i dk(x_i,x*)/dx*,
 * d var = -2 sum_i w_i dk(x_i,x*)/dx* with w = K^-1 k(X,x*).  Replaces the PyTensor graph of the single-point
 * predictive that BO's refinement differentiates (gpmcmc.py:766-801).  Needs Z_dev / W_dev; work_dev must hold
 * 2 * ceil(m/128)*128 rows x ldw (the upper half receives the w rows); (nkern + 1) * d doubles must fit 60 KB of LDS
 * (d <= 1536 with four components, 3840 with one). */
MI_GP_API int mi_gp_predict_grad(mi_gp_handle* h, const double* Xnew_dev, int m, double* work_dev, long ldw, double* mean_dev,
                       double* var_dev, int pred_noise, double* dmean_dev, double* dvar_dev);

/* tuning knobs (benchmarks / A-B tests), ALL per handle -- nothing here is process-wide:
 *   0  look-ahead: factor the next super-panel on a second stream while the trailing update runs; 0 never, 1 by size
 *      (default: from 20 tile columns = N > 2432 on, where the overlap beats the cross-stream hand-offs -- and from 4 tile
 *      columns = N > 384 on for problems that run in column mode from the start, options 37 and 45), 2 always
 *   2  super-panel width in 128-column tiles (default 0 = by trailing size, options 4-6)
 *   4-6  trailing sizes (tile columns) above which the super-panel is 16 / 8 / 4 tiles wide (below the last: 2);
 *        defaults: never 16, else 8; with look-ahead active, problems of up to 64 tile columns use at most 4
 *   7  GEMM launches with fewer 128x128 tiles than this run on 64x64 tiles (default 1024)
 *   8  trailing size at or below which look-ahead bulk updates run one workgroup per CU (default: always)
 *   9  128x128-tile GEMM launches with uniform k hand the tiles beyond their last full round of 512 to the 64x64-tile
 *      kernel (default 1: a last round with few tiles costs a whole round; sharded N=65536 on one rank 1.52 -> 1.47 s)
 *   14 band height (tile rows) of the band-column-major tile order of uniform-k trapezoid launches (default 8, 0 = row-major)
 *   16 panel-stream GEMM launches raise their waves' issue priority (s_setprio 3) against the bulk update's (default 1)
 *   18 tiles of a bulk update that run one workgroup per CU beside the panel chain, the rest two per CU (default 1536, 0: no split)
 *   19 ... only when at least this many tiles remain for the second part (default 1024)
 *   20 trailing tile columns from which the next super-panel's update rides at the head of the trailing update's tile
 *      enumeration instead of in launches of its own (default 72, 0: never)
 *   21 trailing tile columns at or below which a two-stream factorisation continues on the main stream alone (default 8)
 *   26 cross-stream edges of the factorisation: 0 hipEventRecord + hipStreamWaitEvent; 1 stream memory operations --
 *      hipStreamWriteValue32 behind the producer's work, hipStreamWaitValue32 in front of the consumer's (4-5 us per edge
 *      instead of 11-12 on MI355X; N = 6144 3.40 -> 3.26 ms); 2 (default) the same protocol with the PANEL stream's halves
 *      folded into launches of the library: its write + wait at a super-panel boundary is one one-lane launch, its wait
 *      for the next-panel update is a poll at the end of the leaf in front of the first reader (N = 4096 1.995 -> 1.965 ms).
 *      Modes 1 and 2 need hipDeviceAttributeCanUseStreamWaitValue: mi_gp_create queries it and falls back to 0 (the
 *      option then stays 0 whatever is set).  The panel stream's polls are enqueued AHEAD of the main-stream writes they
 *      wait for, which needs the two streams' kernels to be dispatched concurrently.  Where they are not -- rocprofv3
 *      --pmc, AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING, more handles evaluating at once than the device has hardware
 *      queues for (six per device are tested) -- such a poll ends only through its limit.  The library deals with that
 *      itself (the reference's evaluations never fail for reasons of scheduling, gpmcmc.py:331-339): mi_gp_create probes
 *      the dispatch once (a one-lane poll with a limit of a few ms) and starts with mode 0 where it is serialised; and an
 *      evaluation whose poll gives up (every later poll of it then returns at once) makes the handle switch to mode 0 for
 *      good and is evaluated AGAIN before the call returns -- the caller sees the result, mi_gp_last_error names the
 *      demotion once, mi_gp_get_option(40) reads 1.  Setting option 26 to 1 or 2 again re-arms the polls.  Only a second
 *      time-out in the same call returns -2 ("a cross-stream signal ... was not seen within its poll limit"); the handle
 *      stays usable.  Same bits in every mode.
 *   27 log2 of the number of sleeps after which such a poll gives up (default 22 = seconds; 4 .. 30)
 *   28 test hook: the next two-stream evaluation leaves one main-stream signal unwritten (its poll must give up); needs
 *      option 26 = 2 (refused otherwise: a runtime wait has no limit) and is cleared by the next evaluation whatever its schedule
 *   30 gradient evaluations of 64 tile columns and more: the leaf blocks and the block-doubling levels of U = L^-T with nodes of
 *      up to this many tiles start on the main stream inside the factorisation's chain-bound last steps instead of behind it
 *      (default 16, 0: never; same launches per tile, bit-identical gradients; N = 16384 LML + gradient 69.8 -> 69.4 ms)
 *   31 ... in the steps with at most this many trailing tile columns, half as many new columns per step (default 48)
 *   32 in-panel updates (between two leaves of a super-panel) with k = 128 over at most two tile columns and at most this many
 *      16-row x 128-column slices run on the thin kernel (default 2048; 0: never).  Regroups sums (agreement to rounding); the
 *      choice depends on the update's shape alone, so every schedule and a batch return the same bits.
 *   35 extended super-panels: a super-panel with at most this many tile rows below it (default 32; 0: never) also applies its
 *      in-panel updates to the NEXT super-panel's first tile column, level by level, instead of one update of that column
 *      behind the panel (problems of 20 tile columns or more, not the last 8 columns).  Regroups that column's sums; a rule of
 *      the shape alone as well.
 *   37 column mode: the last this-many tile columns (default 24; 0: never) are factored column by column -- leaf, strip and one
 *      k = 256 thin update of the next column on the panel stream; older columns reach a column through k = 128 updates on the
 *      main stream, a column behind the chain.  Problems of up to that many tile columns run in it from the start, on two
 *      streams from 4 tile columns on (round 6; it was 8 before option 45).  Regroups sums (agreement to rounding); a rule of the shape alone: one stream, two
 *      streams and a batch return the same bits.  N = 2048 0.665 -> 0.619 ms, 3072 0.981 -> 0.920, 4096 1.471 -> 1.443.
 *   46 problems of up to this many tile columns run in column mode from the START even where option 37 would put panels in
 *      front of it (default 31 = N <= 3968; ignored when 37 is 0).  Regroups sums like 37, a rule of the shape alone.
 *      N = 3200 0.994 -> 0.946 ms, 3968 1.292 -> 1.267; at 32 tile columns it turns (N = 4096 1.416 -> 1.454).
 *   38 column mode of a BATCH: the main stream applies its k = 128 updates to the columns behind the chain's next one in
 *      k-segmented launches of this many columns (default 8; 1: one launch per column as for a single evaluation).  The tile
 *      takes every 128-column partial sum as a launch of its own would round it: same bits, scheduling only.
 *   45 two-stream evaluations queue their first two kernels (theta / y rows, assembly) on the PANEL stream, so that the first
 *      leaf follows them in stream order instead of behind a cross-stream edge (default 1; N = 1024 -7 %, 2048 -4 %, from 32 tile
 *      columns on 0.1-0.4 %); scheduling only
 *   47 the HOST's wait at the end of an evaluation: its last kernel publishes the evaluation's sequence number in the pinned result
 *      buffer, behind the values it stands for, and the call spins on that word for up to this many microseconds (default 2000;
 *      0: never) before it falls back to hipStreamSynchronize -- a stream synchronisation costs 6-8 us behind the kernel's end
 *      (N = 128 LML + gradient 0.101 -> 0.090 ms, 1024 0.310 -> 0.291, 4096 1.424 -> 1.387).  An evaluation that outlasts the budget
 *      makes the handle's next 15 calls skip the spin (a long evaluation costs a core 2 ms in 16 calls); every 256th call
 *      synchronises the stream all the same.  The word is the LAST thing the evaluation's last kernel does: when a call returns
 *      through the spin every device-side read and write of the evaluation is complete (that kernel writes nothing but the
 *      pinned result buffer) and only its retirement may be outstanding -- the handle's buffers may be read or overwritten
 *      from any stream, as after a synchronisation.  Same bits.
 * 8, 14, 16, 18, 19, 21, 26, 27, 30, 31, 38, 45 and 47 only change scheduling (bit-identical results); 20 moves tiles between the
 * two GEMM kernels (same k order); 2, 4-7, 9, 32, 35, 37 and 46 regroup sums (agreement to rounding), and so does 0 where it changes
 * the super-panel width (20 to 60 tile columns).
 * Unknown ids return -1.  (Round 1's options 1, 3, 10-13 -- GEMM variants, hipGraph replay, persistent bulk kernels,
 * exclusive leaf, fused leaf + strip -- and round 5's 29, 33, 34, 36, 39 -- forms that lost their A/B: the next-panel update's
 * occupancy as a knob, strided thin updates, that update in pieces, 64x128 tiles -- and round 3's 24, the assembly in two parts,
 * which stopped paying in round 6, are gone with the code they selected.) */
MI_GP_API int mi_gp_set_option(mi_gp_handle* h, int what, int value);
/* current value of a knob, the library's own defaults included; 40 (read-only): 1 once the handle has switched its cross-stream
 * edges to events by itself (option 26) */
MI_GP_API int mi_gp_get_option(mi_gp_handle* h, int what, int* value);

/* profiling: level 0 none, 1 per-phase HIP events, 2 additionally per-GEMM-launch HIP events */
MI_GP_API int mi_gp_set_profiling(mi_gp_handle* h, int level);
/* out[0..13] = assemble_ms, cholesky_ms, reduce_ms, total_ms, gemm_ms (sum over launches),
 *             gemm_flops (algorithmic), number of gemm launches, trtri_ms, lauum_ms, contract_ms,
 *             then the same three GEMM figures for the 128x128-tile kernel (gemm_f64_kernel_b) alone
 *             -- of the last evaluation (profiling level >= 1);
 *   out[13] = host time spent enqueueing the last single evaluation's launches, ms (measured at every profiling level) */
MI_GP_API int mi_gp_timers(mi_gp_handle* h, double* out, int n);

/* ---- block-level operations (also used by the multi-GPU driver and the parity tests) ---- */

/* C = beta*C + alpha*op(A)*op(B) in fp64 on v_mfma_f64_16x16x4_f64; row-major, m,n multiples of
 * 128, k multiple of 32.  transa=0: A is m x k; 1: A is k x m.  transb=0: B is k x n; 1: B is n x k.
 * tri=1 computes only tiles on/below the block diagonal (the element-wise lower triangle is
 * guaranteed, the strict upper part of diagonal blocks is unspecified); kmode restricts k per tile for triangular
 * operands (0 full, 1 k>=col-tile start, 2 k<row-tile end, 3 k>=row-tile start, 4 k<col-tile end).
 * Replaces the OpenBLAS dgemm/dsyrk calls inside LAPACK dpotrf/dtrtri/dlauum that PyTensor's
 * Cholesky Op reaches (gpmcmc.py:313; scipy.linalg.cholesky). */
MI_GP_API int mi_gp_gemm_f64(int transa, int transb, int m, int n, int k, double alpha, const double* A_dev, long lda,
                   const double* B_dev, long ldb, double beta, double* C_dev, long ldc, int tri, int kmode,
                   int batch, long strideA, long strideB, long strideC, void* hip_stream);

/* The same product (batch 1) with the launcher's knobs per call: launches with fewer than small_below 128x128 tiles run on
 * 64x64 tiles (handle option 7), tail_small (option 9), band height of the trapezoid tile order (option 14), one workgroup
 * per CU (option 8's effect).  For A/B measurements of single launches and the GEMM tests. */
MI_GP_API int mi_gp_gemm_f64_tuned(int transa, int transb, int m, int n, int k, double alpha, const double* A_dev, long lda,
                         const double* B_dev, long ldb, double beta, double* C_dev, long ldc, int tri, int kmode,
                         int small_below, int tail_small, int band, int one_per_cu, void* hip_stream);

/* C = beta*C + alpha*A*B^T (tri as above) with both operands stored as k-segments: segment g (kseg columns, a multiple of
 * 128) of A at A_dev + g * kseg_stride with rows lda apart, the same for B.  This is how the GEMM kernels read the sharded
 * driver's piece-major panel buffers (one contiguous piece per tile column, sent as soon as it is final).  small_below < 0:
 * the launcher's default. */
MI_GP_API int mi_gp_gemm_nt_kseg(int m, int n, int k, double alpha, const double* A_dev, long lda, const double* B_dev, long ldb,
                       int kseg, long kseg_stride, double beta, double* C_dev, long ldc, int tri, int small_below,
                       void* hip_stream);

/* K(Xrows, Xcols) for one rectangular block of a (distributed) covariance: rows row0.. and columns
 * col0.. of the global matrix; noise + jitter go on the global diagonal, identity in the padding
 * (rows >= nrows / columns >= ncols of the padded block).  Same kernel as the single-GPU assembly
 * (gpmcmc.py:282-312). */
MI_GP_API int mi_gp_assemble_block(int d, int nkern, const int* kernel_ids, const int* ops, const double* theta_dev,
                         const double* Xrows_dev, int nrows, const double* Xcols_dev, int ncols, int row0, int col0,
                         double* K_dev, long ldk, int rows_pad, int cols_pad, int noise_form, void* hip_stream);

/* Factor the w_tiles leading 128-column tiles of a (row_tiles x w_tiles)-tile lower trapezoid in place:
 * diagonal leaves, strip solves of all rows below, in-panel updates.  dinv_dev: w_tiles * 16384 doubles that receive
 * the explicit 128x128 inverses of the panel's diagonal blocks (lower triangular, in an internal tile order; the strip solves are
 * products with them, and mi_gp_trsm_block reuses them); *info_dev receives atomicMin(col_base + bad pivot index + 1).
 * LAPACK dpotrf panel step. */
MI_GP_API int mi_gp_chol_panel(double* A_dev, long lda, int row_tiles, int w_tiles, double* dinv_dev, int* info_dev,
                     int col_base, void* hip_stream);

/* out_dev[1] = sum_i log L[i][i], out_dev[2] = sum_i beta[i]^2 over n entries (one workgroup). */
MI_GP_API int mi_gp_lml_partial(const double* L_dev, long ld, const double* beta_dev, int n, double* out_dev, void* hip_stream);

/* ---- sharded gradient ("gradient at C4 scale needs distributed K^-1"): the reference gets dLML/dtheta from
 * pytensor.grad through Cholesky.L_op behind pm.find_MAP / pm.sample (gpmcmc.py:345,351); the sharded driver
 * (andvaranaut_amd/distributed.py) builds it from these blocks. */

/* X L^T = B in place (X = B L^-T) for the 128-column tiles [c0_tiles, c0_tiles + w_tiles) of a complete lower factor
 * L_dev (element (0,0) first, leading dimension ldl); dinv_dev: the 16384-double leaf inverses mi_gp_chol_panel wrote,
 * indexed by global tile; B_dev points at the first of those columns of the m-row right-hand side (m multiple of 128).
 * With B = rows J of the identity this yields rows J of U = L^-T.  LAPACK dtrsm('R','L','T','N'). */
MI_GP_API int mi_gp_trsm_block(const double* L_dev, long ldl, const double* dinv_dev, int c0_tiles, int w_tiles, double* B_dev,
                     long ldb, int m, void* hip_stream);

/* out = U x for an upper-triangular n x n U (alpha = L^-T beta, gpmcmc.py:315). */
MI_GP_API int mi_gp_trmv_upper(const double* U_dev, long ld, const double* x_dev, int n, double* out_dev, void* hip_stream);

/* grad_dev[ntheta] = 1/2 sum (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta over the column slab [col0, col0 + cols) of the
 * lower triangle (rows >= col0; diagonal counted once).  W_dev points at element (row0, col0) of K^-1 (row0 <= col0,
 * all multiples of 64); part_dev: mi_gp_grad_contract_block_scratch() doubles.  Slab sums add up to mi_gp_lml_grad's. */
MI_GP_API long mi_gp_grad_contract_block_scratch(int n, int col0, int cols, int ntheta);
MI_GP_API int mi_gp_grad_contract_block(int d, int nkern, const int* kernel_ids, const int* ops, const double* theta_dev,
                              const double* X_dev, int n, const double* W_dev, long ldw, int row0, int col0, int cols,
                              const double* alpha_dev, double* part_dev, long part_len, double* grad_dev,
                              void* hip_stream);

/* ---- sharded factorisation, one call per panel step (SURVEY 8e second row; BASELINE config 4).  The covariance is
 * distributed over `world` ranks in 1-D block-cyclic column panels of panel_tiles * 128 columns (panel j belongs to rank
 * j % world); a rank stores its panels side by side (local column li * pw for its li-th panel) with GLOBAL rows.  The
 * exchange itself (one broadcast per tile column of the panel, out of P_dev[j % 2]) stays with the caller (torch.distributed /
 * RCCL); these entries enqueue everything else of a step behind one call:
 *   begin : assemble the owned panels (+ their y^T rows); the owner of panel 0 factors it and stages it into P_dev[0]
 *   step j: panel j is complete in P_dev[j % 2] and visible to main_stream AND side_stream.  The owner of panel j+1 updates it with
 *           panel j, factors it and stages it into P_dev[(j+1) % 2] on side_stream (the caller broadcasts it under that
 *           stream); then ONE GEMM launch on main_stream updates every owned panel > j+1 (panel-list mode of the kernel)
 *   finish: main_stream waits for side_stream; out_dev[1] = sum log L_ii, out_dev[2] = sum beta_i^2 over the owned panels
 * A panel buffer is PIECE-major: piece c (tile column c of the panel) starts at P_dev[b] + c * ldp and holds rows
 * r0 = j * pw .. np + 127 of that column at a row stride of 128 doubles (np = padded n; rows np.. = the y^T block whose
 * first row becomes beta^T), then the column's 128 x 128 leaf inverse (the sharded gradient needs it everywhere):
 * (np + 128 - r0 + 128) * 128 contiguous doubles = ONE broadcast; ldp >= (np + 256) * 128.  On several ranks the owner
 * stages piece c right behind column c's strip (option 5) and the caller sends it -- mi_gp_shard_wait_piece(s, c, stream)
 * orders `stream` behind that staging -- while columns c + 1 .. are still being factored: only the last piece's
 * transfer is exposed.
 * world / rank need not be a process group's: a single process can play rank r of W (tools/emulate_rank.py).
 * Replaces the per-panel LAPACK dpotrf steps behind pt.slinalg.cholesky (gpmcmc.py:313) at a size one GPU need not hold. */
typedef struct mi_gp_shard mi_gp_shard;
typedef struct {
  int n, d, nkern;
  int kernel_ids[MI_GP_MAX_KERN];
  int ops[MI_GP_MAX_KERN];
  int panel_tiles;         /* panel width in 128-column tiles */
  int world, rank;
  int device;
  const double* X_dev;     /* n x d, replicated */
  const double* y_dev;     /* n */
  double* K_dev;           /* (np + 128) x ldk, ldk >= owned panels * pw, even */
  long ldk;
  double* P_dev[2];        /* panel buffers, panel_tiles pieces of ldp doubles each */
  long ldp;                /* piece stride, >= (np + 256) * 128 */
  const double* theta_dev; /* C-ABI theta on the device (the caller uploads it before begin) */
  int* info_dev;           /* [1] bad-pivot word: reset by begin, atomicMin(global column + 1) */
  double* out_dev;         /* [4] scalars of finish */
} mi_gp_shard_config;
MI_GP_API int mi_gp_shard_create(const mi_gp_shard_config* cfg, mi_gp_shard** out);
MI_GP_API int mi_gp_shard_destroy(mi_gp_shard* s);
MI_GP_API int mi_gp_shard_begin(mi_gp_shard* s, int noise_form, void* main_stream, void* side_stream);
MI_GP_API int mi_gp_shard_step(mi_gp_shard* s, int j, void* main_stream, void* side_stream);
MI_GP_API int mi_gp_shard_finish(mi_gp_shard* s, void* main_stream, void* side_stream);
/* `stream` waits until tile column c of the panel this rank factored last (begin: panel 0; step j: panel j + 1) is staged */
MI_GP_API int mi_gp_shard_wait_piece(mi_gp_shard* s, int c, void* hip_stream);
/* options: 0 bulk updates at one workgroup per CU while this rank's side stream factors the next panel (default 1);
 *          1 record per-step HIP events (update / factor / stage on the side stream, bulk on the main stream);
 *          2 update the panel this rank factors in the NEXT step first and alone, so that its chain does not wait for the
 *            whole bulk update (default 1; that panel's update may run on the other tile size: agreement to rounding);
 *          3 the owner's chain (update + factor + stage of the next panel) runs on main_stream AHEAD of its bulk update
 *            instead of beside it on side_stream (default: 1 when world > 1 -- every other rank waits for that chain, and
 *            alone on the chip it is 2-3x shorter than next to a bulk update; 0 on one rank).  Either way side_stream is
 *            ordered behind the staging when the call returns: the caller broadcasts the panel under side_stream.
 * what = 4: tiles of a bulk update that run one workgroup per CU beside this rank's own chain, the rest two per CU (2048;
 *            0: the whole update one per CU).
 * what = 5: a chain on main_stream stages every tile column behind its strip (1), and the previous panel's update takes the
 *            first tile column alone and first so that it is final, staged and sent early (2, the default); 0: the panel is
 *            staged behind its last column.  0 and 1 give the same bits; 2 regroups launches (agreement to rounding).
 * Reproducibility: the sharded factorisation is bit-reproducible for a FIXED world size, panel width and option set.
 * Options 2 and 3 and the world size change which launches update a panel (the early next-panel update goes through the
 * ordinary launcher and may run on 64x64 tiles where the panel-list launch uses 128x128), i.e. they regroup sums: results
 * then agree to rounding (1e-11 relative on the LML in the tests), not bit for bit. */
MI_GP_API int mi_gp_shard_set_option(mi_gp_shard* s, int what, int value);
/* per-step phase times of the last evaluation (option 1; call after synchronising): out[4 * j + 0..3] =
 * update_ms, factor_ms, stage_ms, bulk_ms of step j (0 where the step had no such phase; factor of panel 0 is in
 * out[4 * npanels + 1], its staging in out[4 * npanels + 2]); returns the number of steps written, < 0 on error */
MI_GP_API int mi_gp_shard_times(mi_gp_shard* s, double* out, int max_steps);
/* out[j * panel_tiles + c] = ms from the start of step j's chain to the end of the staging of tile column c of the panel it
 * produces (row npanels: panel 0, from the start of its factorisation); option 1; returns the number of steps written */
MI_GP_API int mi_gp_shard_piece_times(mi_gp_shard* s, double* out, int max_steps);
/* 0: the chain runs on side_stream (it must then wait for panel j as well), 1: on main_stream (option 3) */
MI_GP_API int mi_gp_shard_chain_stream(const mi_gp_shard* s);
MI_GP_API const char* mi_gp_shard_last_error(mi_gp_shard* s);

#ifdef __cplusplus
}
#endif
#endif
