// Handle-level C-ABI (include/mi_gp.h): covariance assembly -> blocked right-looking Cholesky ->
// log marginal likelihood.  Replaces what pm.find_MAP / pm.sample evaluate per step through
// pm.gp.Marginal.marginal_likelihood (gpmcmc.py:321-323, 345, 351).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "migp_kernels.h"
#include "../../include/mi_gp.h"

using namespace migp;

constexpr int SIG_SLOTS = 1024;  // cross-stream edges of one evaluation (a 16384-point factorisation has ~60)

struct mi_gp_handle {
  mi_gp_config cfg;
  KernSpec spec;
  int n, np, ntc;       // points, padded points, 128-column tiles
  int ntheta;
  int device;
  hipStream_t stream;    // trailing updates, assembly, reductions
  hipStream_t pstream;   // look-ahead panel factorisation (higher priority)
  std::vector<hipEvent_t> ev_pool;  // one event per cross-stream hand-off of an evaluation (never re-recorded inside one
  size_t ev_next;                   // evaluation: a captured DAG then holds one node pair per hand-off)
  hipEvent_t wait_ev;               // recorded on the main stream behind the (a2) update of tile columns wait_col + 1 ..:
  int wait_col;                     // the panel stream waits for it after the leaf + strip of tile column wait_col
  // Cross-stream edges as stream memory operations (round 4, option 26): `from` writes the evaluation's epoch into a slot of
  // sig_dev behind its work (hipStreamWriteValue32), `to` waits for slot >= epoch (hipStreamWaitValue32).  Measured on
  // MI355X (ping-pong of short kernels over two streams): +4-5 us per edge against +11-12 us with hipEventRecord +
  // hipStreamWaitEvent.  One slot per edge of an evaluation, never re-used inside it; the epoch grows by one per
  // factorisation, so a slot's old value never satisfies a new wait.
  unsigned* sig_dev;                // [SIG_SLOTS]
  unsigned sig_epoch;
  int sig_next;
  int wait_slot;                    // the slot that stands in for wait_ev
  int wait2_col, wait2_slot;        // the leaf of tile column wait2_col ends only once this slot is written (everything queued on the
                                    // main stream before the super-panel's chain: the first in-panel update behind that leaf writes
                                    // the next super-panel's first column), -1: none
  bool smo_supported;               // hipDeviceAttributeCanUseStreamWaitValue
  int poll_limit_log2;              // option 27: an in-kernel poll gives up after 2^this sleeps (default 22: seconds)
  int test_drop_signal;             // option 28 (tests): the next evaluation leaves one main-stream signal unwritten
  bool demoted;                     // a poll ran into its limit (or mi_gp_create's probe found kernel dispatch serialised): the edges
                                    // are events from then on (use_smo = 0) -- mi_gp_get_option(40)
  int u_early_max_s;                // option 30: block-doubling levels of U = L^-T (node sizes up to this many tiles) that start inside the
                                    // factorisation's chain-bound tail on the main stream (gradient evaluations; 0: none)
  int u_early_cols;                 // option 31: ... from this many trailing tile columns on, and at most this many new columns per step
  bool u_early;                     // this evaluation takes part (set by enqueue_all)
  int u_leaf_done, u_node_done[12]; // tile columns whose leaf block of U is done / full nodes done per level
  int thin_max_wg;                  // option 32: in-panel updates of at most this many 16-row x 128-column slices (k = 128, at most
                                    // THIN_MAX_COLS tile columns) run on the thin kernel (thin_f64.hip); 0: never
  int start_on_panel;               // option 45: see enqueue_factor (default 1; scheduling only)
  int spin_us;                      // option 47: wait_evaluation() spins on the evaluation's sequence word for up to this many us (0: never)
  double eval_seq;                  // sequence number of the evaluation in flight (published by its last kernel)
  int spin_backoff;                 // evaluations left that go straight to hipStreamSynchronize (the last spin ran into its budget)
  unsigned spin_hits;               // spin waits that saw the word (every 256th synchronises the stream all the same)
  int rl_group;                     // option 38: column mode of a BATCH applies the main stream's k = 128 updates to the far columns in
                                    // k-segmented launches of this many columns (same bits, the trailing matrices read and written once per group)
  int rl_cols;                      // option 37: the last rl_cols tile columns are factored COLUMN BY COLUMN (cholesky(): column mode); 0: never
  int rl_whole;                     // option 46: problems of up to this many tile columns run in column mode from the start (whole_columns())
  int ext_rows;                     // option 35: a super-panel with at most this many tile rows below it also applies its updates to
                                    // the NEXT super-panel's first tile column, level by level (chol_panel's nx); 0: never
  int done_col, done_slot;          // the update behind the strip of tile column done_col raises this slot ("super-panel done")
  int use_smo;                      // option 26: 0 events, 1 runtime stream memory operations, 2 (default) the panel stream's
                                    // halves folded into one-lane launches of the library / the end of a leaf
  // tuning options (mi_gp_set_option), all per handle
  int tail_small;   // option 9: 128x128-tile launches finish their last partial round on 64x64 tiles (default 1)
  int chain_prio;   // s_setprio(3) in the GEMM launches of the panel stream (option 16; the leaf and strip kernels always raise it)
  int lookahead;    // 0 never, 1 by size (default: from LOOKAHEAD_MIN_TILES tile columns on), 2 always
  int lowocc_thr;   // trailing sizes (tile columns) at or below which bulk updates run one workgroup per CU
  int w_thr[3];     // trailing sizes (tile columns) above which the super-panel is 16 / 8 / 4 tiles wide
  int small_below;  // GEMM launches with fewer 128x128 tiles than this run on 64x64 tiles
  int band_rows;    // band height of the band-column-major tile order of uniform-k trapezoid launches
  int split_tiles;  // option 18: tiles of a bulk update that run one workgroup per CU beside the chain; the rest two per CU (0: no split)
  int split_min_rest;  // option 19: ... only when at least this many tiles remain for the second part
  bool asm_on_panel;   // this evaluation's set_yrows + assembly were queued on the PANEL stream (column mode from the start on two
                       // streams: the first leaf follows them in stream order, no cross-stream edge in front of the chain)
  int single_below;    // option 21: trailing tile columns at or below which a two-stream factorisation continues on one stream (0: never)
  int merge_min_tiles; // option 20: trailing sizes (tile columns) from which the next super-panel's update rides at the head of the
                       // trailing update's enumeration instead of in launches of its own (0: never)
  mi_gp_buffers buf;
  bool have_data;
  // handle-owned small scratch
  double* theta_dev;    // [ntheta]
  double* dinv_dev;     // [ntc][128][128] explicit inverses of the diagonal blocks of L (leaf output, strip operand)
  double* alpha_dev;    // [np] K^-1 y
  double* part_dev;     // [grad_contract_blocks(n)][ntheta]
  double* gxs_dev;      // [grad_x_splits][n][d] partial dLML/dX (allocated on first mi_gp_grad_x)
  double* grad_host;    // pinned [ntheta]
  int* info_dev;
  double* lr_part_dev;  // [2 * LML_REDUCE_BLOCKS] slice sums of lml_reduce_kernel
  unsigned* lr_sync_dev;  // its ticket (zero between evaluations)
  double* out_host;     // pinned [16]
  double* theta_host;   // pinned
  // profiling
  int prof_level;
  hipEvent_t ev[8];
  std::vector<hipEvent_t> gemm_ev;  // pairs
  std::vector<char> gemm_ev_big;    // per pair: 1 if the 128x128-tile kernel ran
  std::vector<double> gemm_ev_flops;
  size_t gemm_ev_used;
  double gemm_flops_acc;
  double t_assemble_ms, t_chol_ms, t_reduce_ms, t_gemm_ms, t_total_ms, gemm_flops, n_gemm;
  double t_trtri_ms, t_lauum_ms, t_contract_ms;
  double t_enqueue_ms;  // host time of enqueueing the last single evaluation (always measured: two clock reads)
  double t_gemm_big_ms, gemm_big_flops, n_gemm_big;  // the 128x128-tile kernel only
  // batched evaluation (mi_gp_set_batch / mi_gp_lml_batch / mi_gp_lml_grad_batch): while a batch runs, buf / the scratch
  // pointers above point at the batch's arrays and bt carries the strides; nullptr / nb = 1 otherwise
  Batch bt;
  const Batch* btp;        // &bt while a batch is being enqueued, nullptr otherwise (what the launchers get)
  mi_gp_batch_buffers bbuf;
  int batch_cap;           // problems the batch scratch below is sized for
  double *b_theta_dev, *b_dinv_dev, *b_alpha_dev, *b_part_dev, *b_grad_host, *b_out_host, *b_theta_host;
  int* b_info_dev;
  double* b_lr_part_dev;
  unsigned* b_lr_sync_dev;
  bool factored;
  bool have_u;             // Z_dev holds U = L^-T and alpha_dev = K^-1 y of the last mi_gp_factor (mi_gp_predict_grad)
  bool have_kinv;          // W_dev holds K^-1 (lower) and alpha_dev = K^-1 y of the last mi_gp_lml_grad
  const double* diag_dev;  // optional per-point diagonal added at assembly (mi_gp_set_diag)
  char err[256];
};

static int hfail(mi_gp_handle* h, hipError_t e, const char* where) {
  snprintf(h->err, sizeof(h->err), "%s: %s", where, hipGetErrorString(e));
  return -2;
}
#define HCK(call, where)                          \
  do {                                            \
    hipError_t e__ = (call);                      \
    if (e__ != hipSuccess) return hfail(h, e__, where); \
  } while (0)

extern "C" const char* mi_gp_last_error(mi_gp_handle* h) { return h ? h->err : "null handle"; }

// frees whatever a (possibly half-built) handle owns; every member is null / empty until it is created
static void release_handle(mi_gp_handle* h) {
  (void)hipSetDevice(h->device);
  if (h->pstream) (void)hipStreamSynchronize(h->pstream);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  (void)hipFree(h->theta_dev); (void)hipFree(h->dinv_dev); (void)hipFree(h->info_dev); (void)hipFree(h->sig_dev);
  (void)hipFree(h->alpha_dev); (void)hipFree(h->part_dev); (void)hipFree(h->gxs_dev);
  (void)hipFree(h->lr_part_dev); (void)hipFree(h->lr_sync_dev); (void)hipFree(h->b_lr_part_dev); (void)hipFree(h->b_lr_sync_dev);
  (void)hipFree(h->b_theta_dev); (void)hipFree(h->b_dinv_dev); (void)hipFree(h->b_alpha_dev); (void)hipFree(h->b_part_dev);
  (void)hipFree(h->b_info_dev);
  if (h->b_grad_host) (void)hipHostFree(h->b_grad_host);
  if (h->b_out_host) (void)hipHostFree(h->b_out_host);
  if (h->b_theta_host) (void)hipHostFree(h->b_theta_host);
  if (h->grad_host) (void)hipHostFree(h->grad_host);
  if (h->out_host) (void)hipHostFree(h->out_host);
  if (h->theta_host) (void)hipHostFree(h->theta_host);
  for (int i = 0; i < 8; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  for (auto& ev : h->gemm_ev) (void)hipEventDestroy(ev);
  for (auto& ev : h->ev_pool) (void)hipEventDestroy(ev);
  if (h->pstream) (void)hipStreamDestroy(h->pstream);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// The default edges (option 26 = 2) enqueue a poll on the panel stream AHEAD of the main-stream write it waits for.  That ends
// only through its limit where kernel dispatch is serialised: rocprofv3 --pmc, AMD_SERIALIZE_KERNEL / HIP_LAUNCH_BLOCKING, a
// debugger.  One probe at creation -- a one-lane poll on the panel stream with a short limit (2^14 sleeps: a few ms), the write
// behind it on the main stream -- finds that out before an evaluation can stall for seconds; such a handle uses events
// (gpmcmc.py:331-339: the reference's evaluations never fail for reasons of scheduling).  ~40 us where dispatch is concurrent.
static hipError_t probe_dispatch(mi_gp_handle* h) {
  hipError_t e = hipMemset(h->info_dev, 0x7f, sizeof(int) * 4);
  // (both streams have launched before: the probe does not time the first launch's code-object load)
  if (e == hipSuccess) e = launch_signal_write_wait(h->sig_dev + SIG_SLOTS - 1, nullptr, 0u, h->info_dev, h->pstream);
  if (e == hipSuccess) e = launch_signal_write_wait(h->sig_dev + SIG_SLOTS - 1, nullptr, 0u, h->info_dev, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->pstream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = launch_signal_write_wait(nullptr, h->sig_dev + SIG_SLOTS - 1, 1u, h->info_dev, h->pstream, 1, 0, 14);
  if (e == hipSuccess) e = launch_signal_write_wait(h->sig_dev + SIG_SLOTS - 1, nullptr, 1u, h->info_dev, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->pstream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  int info = 0;
  if (e == hipSuccess) e = hipMemcpy(&info, h->info_dev, sizeof(int), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemset(h->sig_dev + SIG_SLOTS - 1, 0, sizeof(unsigned));
  if (e == hipSuccess && info == SIGNAL_TIMEOUT_INFO) {
    h->use_smo = 0;
    h->demoted = true;
    snprintf(h->err, sizeof(h->err), "kernel dispatch is serialised on this device: cross-stream edges are events (option 26 = 0)");
  }
  return e;
}

extern "C" int mi_gp_create(const mi_gp_config* cfg, mi_gp_handle** out) {
  if (!cfg || !out) { set_global_error("mi_gp_create: null argument"); return -1; }
  if (cfg->n <= 0 || cfg->d <= 0 || cfg->nkern <= 0 || cfg->nkern > MAX_KERN) {
    set_global_error("mi_gp_create: n, d must be positive and 1 <= nkern <= 8");
    return -1;
  }
  for (int i = 0; i < cfg->nkern; ++i)
    if (cfg->kernel_ids[i] < 0 || cfg->kernel_ids[i] > KID_RATQUAD) { set_global_error("mi_gp_create: unknown kernel id"); return -1; }
  mi_gp_handle* h = new mi_gp_handle();  // value-initialised: every pointer / stream / event starts null
  memset(h->err, 0, sizeof(h->err));
  h->cfg = *cfg;
  h->spec.nkern = cfg->nkern;
  h->spec.d = cfg->d;
  for (int i = 0; i < MAX_KERN; ++i) {
    h->spec.kid[i] = i < cfg->nkern ? cfg->kernel_ids[i] : 0;
    h->spec.op[i] = i < cfg->nkern ? cfg->ops[i] : 0;
  }
  h->n = cfg->n;
  h->np = (cfg->n + 127) / 128 * 128;
  h->ntc = h->np / 128;
  h->ntheta = cfg->nkern * cfg->d + 2 * cfg->nkern + 2;
  h->device = cfg->device;
  h->have_data = false;
  h->factored = false;
  h->have_kinv = false;
  h->have_u = false;
  h->diag_dev = nullptr;
  h->gxs_dev = nullptr;
  h->t_trtri_ms = h->t_lauum_ms = h->t_contract_ms = 0.0;
  h->t_gemm_big_ms = h->gemm_big_flops = h->n_gemm_big = 0.0;
  h->prof_level = 0;
  h->gemm_ev_used = 0;
  hipError_t e = hipSetDevice(h->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    e = hipStreamCreateWithPriority(&h->pstream, hipStreamNonBlocking, hi);
  }
  h->ev_next = 0;
  h->lookahead = 1;
  h->tail_small = 1;
  h->chain_prio = 1;  // N = 8192: 6.06 -> 5.94 ms, N = 16384: 28.11 -> 27.74 ms (interleaved A/B)
  h->small_below = GemmParams().small_below;  // (768 looked 1 % better at N = 6144 .. 12288 while the 64x64-tile kernel carried the k-flush branch; without it: level)
  h->band_rows = GemmParams().band;
  h->split_tiles = 1536;  // (2048 until the chain got shorter -- stream memory operations, strip kernel: N = 16384 26.21 -> 25.96 ms,
                          // 1024: 26.21, 1280: 26.09, 1792: 26.08; N = 12288 flat)
  h->split_min_rest = 1024;
  h->merge_min_tiles = 72;
  h->single_below = 8;  // (16 with event hand-offs; with option 26: N = 4096 1.983 -> 1.958 ms, 8192 5.50 -> 5.49, 16384 26.84 -> 26.73)
  h->use_smo = 2;
  {
    // hipStreamWriteValue32 / hipStreamWaitValue32 need driver support: without it every two-stream evaluation would fail,
    // so the edges fall back to events (option 26 = 0; mi_gp_set_option refuses 1 and 2 then)
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, h->device) != hipSuccess) can = 0;
    (void)hipGetLastError();
    h->smo_supported = can != 0;
    if (!h->smo_supported) h->use_smo = 0;
  }
  h->poll_limit_log2 = 22;
  h->test_drop_signal = 0;
  h->demoted = false;
  h->thin_max_wg = 2048;
  h->rl_cols = 24;
  h->rl_whole = 31;
  h->start_on_panel = 1;
  h->spin_us = 2000;
  h->eval_seq = 0.0;
  h->spin_backoff = 0;
  h->spin_hits = 0;
  h->asm_on_panel = false;
  h->rl_group = 8;
  h->ext_rows = 32;
  h->done_col = h->done_slot = -1;
  h->u_early_max_s = 16;
  h->u_early_cols = 48;
  h->u_early = false;
  h->sig_epoch = 0;
  h->sig_next = 0;
  h->wait_slot = -1;
  h->wait2_col = h->wait2_slot = -1;
  // round-2 A/B (tools/dev_ab_opts.py, interleaved in one process): bulk updates at one workgroup per CU whenever the
  // panel chain runs beside them (N = 16384: 29.99 -> 28.97 ms) and 8-tile super-panels at every size (N = 2048 1.045 ->
  // 1.017 ms, 4096 2.470 -> 2.388, 8192 6.417 -> 6.348, 16384 28.59 -> 28.39 against the 8 / 4 split of round 1)
  h->lowocc_thr = 1 << 20;
  h->w_thr[0] = 1 << 20; h->w_thr[1] = 0; h->w_thr[2] = 0;
  if (e == hipSuccess) e = hipMalloc(&h->theta_dev, sizeof(double) * h->ntheta);
  // (+ 4 blocks behind the leaf inverses: the strips' operand-order copies of their first 256 rows for the thin updates, two
  // buffers of two blocks -- column mode reads the previous column's copy beside the current one's)
  if (e == hipSuccess) e = hipMalloc(&h->dinv_dev, sizeof(double) * MINV_ELEMS * (size_t)(h->ntc + 4));
  if (e == hipSuccess) e = hipMalloc(&h->alpha_dev, sizeof(double) * h->np);
  if (e == hipSuccess) e = hipMalloc(&h->part_dev, sizeof(double) * (size_t)grad_contract_blocks(h->n) * h->ntheta);
  if (e == hipSuccess) e = hipHostMalloc(&h->grad_host, sizeof(double) * h->ntheta);
  if (e == hipSuccess) e = hipMalloc(&h->info_dev, sizeof(int) * 4);
  if (e == hipSuccess) e = hipMalloc(&h->sig_dev, sizeof(unsigned) * SIG_SLOTS);
  if (e == hipSuccess) e = hipMemset(h->sig_dev, 0, sizeof(unsigned) * SIG_SLOTS);
  if (e == hipSuccess) e = hipMalloc(&h->lr_part_dev, sizeof(double) * 2 * LML_REDUCE_BLOCKS);
  if (e == hipSuccess) e = hipMalloc(&h->lr_sync_dev, sizeof(unsigned) * 2);  // [0] lml_reduce's ticket, [1] grad_final's
  if (e == hipSuccess) e = hipMemset(h->lr_sync_dev, 0, sizeof(unsigned) * 2);
  if (e == hipSuccess) e = hipHostMalloc(&h->out_host, sizeof(double) * 16);
  if (e == hipSuccess) memset(h->out_host, 0, sizeof(double) * 16);  // (the sequence words: wait_evaluation())
  if (e == hipSuccess) e = hipHostMalloc(&h->theta_host, sizeof(double) * h->ntheta);
  for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreate(&h->ev[i]);
  if (e == hipSuccess) e = gemm_f64_enable_lds();
  if (e == hipSuccess) e = leaf_enable_lds();
  if (e == hipSuccess && h->use_smo >= 2) e = probe_dispatch(h);
  if (e != hipSuccess) {
    char msg[200];
    snprintf(msg, sizeof(msg), "mi_gp_create: %s", hipGetErrorString(e));
    set_global_error(msg);
    release_handle(h);
    return -2;
  }
  *out = h;
  return 0;
}

extern "C" int mi_gp_destroy(mi_gp_handle* h) {
  if (!h) return 0;
  release_handle(h);
  return 0;
}

extern "C" long mi_gp_padded_n(const mi_gp_handle* h) { return h ? h->np : -1; }
extern "C" int mi_gp_num_theta(const mi_gp_handle* h) { return h ? h->ntheta : -1; }
extern "C" void* mi_gp_stream(const mi_gp_handle* h) { return h ? (void*)h->stream : nullptr; }

extern "C" int mi_gp_set_data(mi_gp_handle* h, const mi_gp_buffers* b) {
  if (!h || !b || !b->X_dev || !b->y_dev || !b->K_dev) return -1;
  if (b->lda < h->np || (b->lda & 1)) {
    snprintf(h->err, sizeof(h->err), "mi_gp_set_data: lda must be even and >= padded n (%d)", h->np);
    return -1;
  }
  h->buf = *b;
  h->have_data = true;
  return 0;
}

// tuning knobs, all per handle (include/mi_gp.h lists them)
extern "C" int mi_gp_set_option(mi_gp_handle* h, int what, int value) {
  if (!h) return -1;
  if (what == 0) h->lookahead = value < 0 ? 0 : value > 2 ? 2 : value;
  else if (what == 2) h->cfg.panel_tiles = value;
  else if (what >= 4 && what <= 6) h->w_thr[what - 4] = value;
  else if (what == 7) h->small_below = value;
  else if (what == 8) h->lowocc_thr = value;
  else if (what == 14) h->band_rows = value;
  else if (what == 16) h->chain_prio = value;
  else if (what == 18) h->split_tiles = value;
  else if (what == 19) h->split_min_rest = value;
  else if (what == 20) h->merge_min_tiles = value;
  else if (what == 21) h->single_below = value;
  else if (what == 26) {
    h->use_smo = !h->smo_supported ? 0 : value < 0 ? 0 : value > 2 ? 2 : value;
    if (h->use_smo != 0) h->demoted = false;  // (the caller asks for polls again: the next time-out demotes again)
  }
  else if (what == 27) h->poll_limit_log2 = value < 4 ? 4 : value > 30 ? 30 : value;
  else if (what == 28) {
    // (with option 26 = 1 the waiter is a runtime hipStreamWaitValue32 without a limit: the hook would hang the process)
    if (value && h->use_smo < 2) {
      snprintf(h->err, sizeof(h->err), "mi_gp_set_option: option 28 needs option 26 = 2 (a bounded in-kernel poll)");
      return -1;
    }
    h->test_drop_signal = value ? 1 : 0;
  }
  else if (what == 30) h->u_early_max_s = value < 0 ? 0 : value;
  else if (what == 31) h->u_early_cols = value < 8 ? 8 : value;
  else if (what == 32) h->thin_max_wg = value < 0 ? 0 : value;
  else if (what == 35) h->ext_rows = value < 0 ? 0 : value;
  else if (what == 37) h->rl_cols = value < 0 ? 0 : value;
  else if (what == 46) h->rl_whole = value < 0 ? 0 : value;
  else if (what == 38) h->rl_group = value < 1 ? 1 : value > 8 ? 8 : value;
  else if (what == 45) h->start_on_panel = value ? 1 : 0;
  else if (what == 47) { h->spin_us = value < 0 ? 0 : value; h->spin_backoff = 0; }
  else if (what == 9) h->tail_small = value ? 1 : 0;
  else {
    snprintf(h->err, sizeof(h->err), "mi_gp_set_option: unknown option %d", what);
    return -1;
  }
  return 0;
}

// current value of a knob (the library's own defaults included); 40: 1 once the handle has demoted its edges to events
extern "C" int mi_gp_get_option(mi_gp_handle* h, int what, int* value) {
  if (!h || !value) return -1;
  switch (what) {
    case 0: *value = h->lookahead; break;
    case 2: *value = h->cfg.panel_tiles; break;
    case 4: case 5: case 6: *value = h->w_thr[what - 4]; break;
    case 7: *value = h->small_below; break;
    case 8: *value = h->lowocc_thr; break;
    case 9: *value = h->tail_small; break;
    case 14: *value = h->band_rows; break;
    case 16: *value = h->chain_prio; break;
    case 18: *value = h->split_tiles; break;
    case 19: *value = h->split_min_rest; break;
    case 20: *value = h->merge_min_tiles; break;
    case 21: *value = h->single_below; break;
    case 26: *value = h->use_smo; break;
    case 27: *value = h->poll_limit_log2; break;
    case 28: *value = h->test_drop_signal; break;
    case 30: *value = h->u_early_max_s; break;
    case 31: *value = h->u_early_cols; break;
    case 32: *value = h->thin_max_wg; break;
    case 35: *value = h->ext_rows; break;
    case 37: *value = h->rl_cols; break;
    case 46: *value = h->rl_whole; break;
    case 38: *value = h->rl_group; break;
    case 45: *value = h->start_on_panel; break;
    case 47: *value = h->spin_us; break;
    case 40: *value = h->demoted ? 1 : 0; break;
    default:
      snprintf(h->err, sizeof(h->err), "mi_gp_get_option: unknown option %d", what);
      return -1;
  }
  return 0;
}

extern "C" int mi_gp_set_profiling(mi_gp_handle* h, int level) {
  if (!h) return -1;
  h->prof_level = level;
  return 0;
}

// ---------------------------------------------------------------- driver pieces
static hipError_t prof_gemm(mi_gp_handle* h, const GemmParams& p, int ak, int bk, int batch, double flops,
                            hipStream_t st) {
  if (h->prof_level >= 2) {
    // one event pair per kernel launch: a split product (gemm_tail_tiles) is two launches, its flops divided by tiles;
    // `flops` are those of the WHOLE product, a sub-range launch (p.tile0 / p.tile_cnt) is credited its share of tiles
    const int tail = gemm_tail_tiles(p, batch);
    const int tiles = p.tri ? p.nt * (p.nt + 1) / 2 + (p.mt - p.nt) * p.nt : p.mt * p.nt;
    const int t0 = p.tile0, t1 = p.tile_cnt > 0 ? (t0 + p.tile_cnt < tiles ? t0 + p.tile_cnt : tiles) : tiles;
    const int big_end = t1 < tiles - tail ? t1 : tiles - tail;
    hipError_t r = hipSuccess;
    for (int part = 1; part <= 2 && r == hipSuccess; ++part) {
      const int mine = part == 1 ? (gemm_uses_small_tiles(p, batch) ? (t0 == 0 ? tiles : 0) : big_end - t0)
                                 : ((tail > 0 && t1 == tiles) ? tail : 0);
      if (mine <= 0) continue;
      if (h->gemm_ev_used + 2 > h->gemm_ev.size()) {
        for (int i = 0; i < 64; ++i) {
          hipEvent_t e;
          r = hipEventCreate(&e);
          if (r != hipSuccess) return r;
          h->gemm_ev.push_back(e);
        }
      }
      (void)hipEventRecord(h->gemm_ev[h->gemm_ev_used], st);
      r = launch_gemm_f64(p, ak, bk, batch, st, part);
      (void)hipEventRecord(h->gemm_ev[h->gemm_ev_used + 1], st);
      const size_t pair = h->gemm_ev_used / 2;
      if (h->gemm_ev_big.size() <= pair) { h->gemm_ev_big.resize(pair + 64); h->gemm_ev_flops.resize(pair + 64); }
      h->gemm_ev_big[pair] = (part == 1 && !gemm_uses_small_tiles(p, batch)) ? 1 : 0;
      h->gemm_ev_flops[pair] = flops * (double)mine / (double)tiles;
      h->gemm_flops_acc += flops * (double)mine / (double)tiles;
      h->gemm_ev_used += 2;
    }
    return r;
  }
  return launch_gemm_f64(p, ak, bk, batch, st);
}

// trapezoid update  A[r0:, c0:c0+nc] -= P P_c^T  with P = A[r0:, k0:k0+kw] (tile units)
// In-panel updates only (their shapes do not depend on the schedule), by SHAPE alone -- not the batch size, not a scheduling
// option: a batch returns the single evaluation's bits, and so does every schedule.
constexpr int THIN_MAX_COLS = 2;  // (the strip in front of such an update hands it its B operand in operand order: 2 x 128 rows)
static bool thin_shape(const mi_gp_handle* h, int mt, int nc, int kw) {
  return h->thin_max_wg > 0 && nc <= THIN_MAX_COLS && kw == 1 && (long)mt * 8 * nc <= h->thin_max_wg;
}

// wr (in-panel updates only): raised to the evaluation's epoch once everything queued on `st` before this update is done
static hipError_t syrk_trapezoid(mi_gp_handle* h, double* A, long lda, int ntr, int r0, int nc, int k0, int kw,
                                 hipStream_t st, int one_per_cu = 0, int tile0 = 0, int tile_cnt = 0, int fc = 0,
                                 bool in_panel = false, unsigned* wr = nullptr, bool lsw = false, int kflush = 0) {
  // (lsw: the strip in front of this update has written its first rows in operand order -- chol_panel decides both by the same rule)
  if (in_panel && lsw && thin_shape(h, ntr - r0, nc, kw))
    return launch_syrk_thin(A + (long)r0 * 128 * lda + (long)k0 * 128, A + (long)r0 * 128 * lda + (long)r0 * 128, lda, ntr - r0, nc,
                            kw * 128, st, h->btp, wr, h->sig_epoch, h->dinv_dev + (size_t)h->ntc * MINV_ELEMS);
  if (wr != nullptr) {  // (the 64x64-tile kernel has no such hook: a one-lane launch in front of it)
    hipError_t we = launch_signal_write_wait(wr, nullptr, h->sig_epoch, h->info_dev, st);
    if (we != hipSuccess) return we;
  }
  GemmParams p;
  p.one_per_cu = one_per_cu;
  p.tile0 = tile0;
  p.tile_cnt = tile_cnt;
  p.fc = fc;
  p.hiprio = (st == h->pstream && h->chain_prio) ? 1 : 0;
  p.small_below = h->small_below;
  p.tail_small = h->tail_small;
  p.band = h->band_rows;
  p.A = A + (long)r0 * 128 * lda + (long)k0 * 128;
  p.B = p.A;
  p.C = A + (long)r0 * 128 * lda + (long)r0 * 128;
  p.lda = p.ldb = p.ldc = lda;
  p.strideA = p.strideB = p.strideC = h->btp ? h->btp->sK : 0;
  p.mt = ntr - r0;
  p.nt = nc;
  p.k = kw * 128;
  p.tri = 1;
  p.kmode = 0;
  p.alpha = -1.0;
  p.beta = 1.0;
  p.kflush = kflush;
  p.dead_last_half = 1;  // (every trapezoid of the factorisation ends in the y^T tile row)
  // algorithmic flops (nb*m^2 for the lower-triangle SYRK, 2*nb*rows*cols for the block
  // below it, one y^T row for the folded-in forward solve); the MFMA work issued is slightly larger
  // (full diagonal tiles, a 128-row tile for the y row).
  const double c = nc * 128.0, rows_real = (p.mt - 1) * 128.0;
  const double flops = (double)p.k * (c * (c + 1.0) + 2.0 * (rows_real - c) * c + 2.0 * c);
  return prof_gemm(h, p, 0, 0, h->btp ? h->btp->nb : 1, flops, st);
}

// factor tile columns [c0, c0+w) of the (ntr x ntc)-tile trapezoid, recursively halving w; nx (0 / 1): every level's update
// also covers the nx tile columns behind the panel, so that they are up to date when the panel's last strip is.
// follow: number of tile columns of the k = 128 update that the CALLER runs right behind this (one-column) panel's strip
static hipError_t chol_panel(mi_gp_handle* h, double* A, long lda, int ntr, int c0, int w, hipStream_t st, int nx = 0, int follow = 0) {
  hipError_t e;
  if (w == 1) {
    // the update behind this column's strip is a k = 128 one over `fol` columns: on the thin kernel the strip hands it its
    // B operand (the first fol x 128 rows of the strip) in operand order
    const int fol = nx > 0 ? nx : follow;
    const bool sw = fol > 0 && fol <= 2 && thin_shape(h, ntr - c0 - 1, fol, 1);
    double* lsw = h->dinv_dev + (size_t)h->ntc * MINV_ELEMS;
    double* blk = A + (long)c0 * 128 * lda + (long)c0 * 128;
    double* dinv = h->dinv_dev + (size_t)c0 * MINV_ELEMS;
    const int m = (ntr - c0 - 1) * 128;
    // (the trapezoid's last tile row is the y^T block: below the last tile column there is nothing else, and the leaf
    // solves that one row itself)
    // the super-panel's other columns are being updated on the main stream ((a2)); their first reader is the in-panel update
    // behind this column's strip.  Option 26 = 2: this leaf polls for that update's signal before it ends (it is done by
    // then as a rule: it started with (a1)); otherwise a runtime wait behind the strip.
    const bool waits = c0 == h->wait_col;
    // the other edge a leaf may carry: everything the main stream had queued before this panel's chain (wait2; see cholesky()).
    // That slot is written behind the (a2) signal, so where both fall on one leaf it stands for both.
    const bool waits2 = c0 == h->wait2_col && h->wait2_slot >= 0 && h->use_smo >= 2;
    const bool folded = waits2 || (waits && h->wait_slot >= 0 && h->use_smo >= 2);
    if (c0 == h->wait2_col) h->wait2_col = -1;
    e = launch_potrf_leaf128(blk, lda, dinv, c0 * 128, h->info_dev, st, m == 128 ? blk + 128 * lda : nullptr, h->btp,
                             waits2 ? h->sig_dev + h->wait2_slot : folded ? h->sig_dev + h->wait_slot : nullptr,
                             h->sig_epoch, h->poll_limit_log2);
    if (e == hipSuccess && m > 128)
      e = launch_trsm_strip128(dinv, blk + 128 * lda, lda, m, st, h->btp, h->btp ? h->btp->sK : 0, sw ? lsw : nullptr, 8 * fol);
    if (e == hipSuccess && waits) {
      h->wait_col = -1;
      if (!folded)
        e = h->wait_slot >= 0 ? hipStreamWaitValue32(st, h->sig_dev + h->wait_slot, h->sig_epoch, hipStreamWaitValueGte, 0xffffffffu)
                              : hipStreamWaitEvent(st, h->wait_ev, 0);
    }
    if (e == hipSuccess && nx > 0) {
      unsigned* wr = nullptr;
      if (c0 == h->done_col && h->done_slot >= 0) wr = h->sig_dev + h->done_slot;
      if (c0 == h->done_col) h->done_col = -1;
      e = syrk_trapezoid(h, A, lda, ntr, c0 + 1, nx, c0, 1, st, 0, 0, 0, 0, true, wr, sw);
    }
    return e;
  }
  const int w1 = w / 2, w2 = w - w1;
  e = chol_panel(h, A, lda, ntr, c0, w1, st, 0, w1 == 1 ? w2 + nx : 0);
  if (e != hipSuccess) return e;
  e = syrk_trapezoid(h, A, lda, ntr, c0 + w1, w2 + nx, c0, w1, st, 0, 0, 0, 0, true, nullptr,
                     w1 == 1 && w2 + nx <= 2 && thin_shape(h, ntr - c0 - w1, w2 + nx, 1));
  if (e != hipSuccess) return e;
  return chol_panel(h, A, lda, ntr, c0 + w1, w2, st, nx);
}

// Right-looking blocked Cholesky of the (ntr x ntc)-tile lower trapezoid with one super-panel of
// look-ahead: while the trailing update of super-panel J runs on the main stream, the next
// super-panel (whose columns were updated first) is factored on the high-priority panel stream.
// super-panel width (128-column tiles) for a trailing matrix of `rem` tile columns: wide panels while
// the trailing update is long enough to hide their factorisation (k = 1024 runs the GEMM at ~64
// TFLOP/s instead of ~57 at k = 512), narrower ones once the panel chain is the critical path
// Super-panel width in tiles for `rem` remaining tile columns.  `cap` (0: none) limits the size-derived width: the
// two-stream driver factors problems of up to 64 tile columns in 4-tile super-panels (8 vs 4, interleaved A/B at the end
// of round 2: N = 4608 2.510 vs 2.475 ms, 5120 2.840 vs 2.725, 6144 3.602 vs 3.504, 7168 4.628 vs 4.538, 8192 5.742 vs
// 5.678; 9216 equal, 10240 9.05 vs 9.14, 16384 27.4 vs 28.8 -- and 4-tile panels only for the last 52 / 64 columns of
// larger problems lose 1-2 %).  An explicit panel_tiles (option 2) overrides everything.
constexpr int NARROW_PANELS_MAX_TILES = 60;  // (64 until the end of round 4: with the cheaper cross-stream edges N = 8192 runs 5.41 vs 5.33 ms
                                           // on 4- vs 8-tile panels; 7168: 4.16 vs 4.18, 6144: 3.16 vs 3.26, 4096: 1.94 vs 2.01)
static int pick_w(const mi_gp_handle* h, int rem, int cap) {
  int W = h->cfg.panel_tiles;
  if (W <= 0) {
    W = (rem > h->w_thr[0]) ? 16 : (rem > h->w_thr[1]) ? 8 : (rem > h->w_thr[2]) ? 4 : 2;
    if (cap > 0 && W > cap) W = cap;
  }
  return rem < W ? rem : W;
}

static hipError_t next_event(mi_gp_handle* h, hipEvent_t* out) {
  if (h->ev_next == h->ev_pool.size()) {
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    h->ev_pool.push_back(ev);
  }
  *out = h->ev_pool[h->ev_next++];
  return hipSuccess;
}

// a fresh signal slot of this evaluation, written behind everything queued on `from` so far; -1: none left / events in use
static int signal_from(mi_gp_handle* h, hipStream_t from, hipError_t* e) {
  *e = hipSuccess;
  if (!h->use_smo || h->sig_next >= SIG_SLOTS) return -1;
  const int slot = h->sig_next++;
  *e = hipStreamWriteValue32(from, h->sig_dev + slot, h->sig_epoch, 0);
  return slot;
}

// `to` waits for everything queued on `from` so far
static hipError_t hand_off(mi_gp_handle* h, hipStream_t from, hipStream_t to) {
  hipError_t se;
  const int slot = signal_from(h, from, &se);
  if (slot >= 0) return se != hipSuccess ? se : hipStreamWaitValue32(to, h->sig_dev + slot, h->sig_epoch, hipStreamWaitValueGte, 0xffffffffu);
  hipEvent_t ev;
  hipError_t e = next_event(h, &ev);
  if (e != hipSuccess) return e;
  e = hipEventRecord(ev, from);
  if (e != hipSuccess) return e;
  return hipStreamWaitEvent(to, ev, 0);
}

// Below this many tile columns one stream is faster than two: the cross-stream hand-offs cost more than the overlap
// returns (one stream vs two, end of round 2: N = 2048 0.94 vs 1.01 ms, N = 4096 2.235 vs 2.252, N = 4608 2.513 vs 2.472,
// N = 5120 2.849 vs 2.820, N = 6144 3.81 vs 3.59, N = 8192 6.35 vs 5.67).
constexpr int LOOKAHEAD_MIN_TILES = 20;  // round 4: with the single-stream tail (option 21) two streams won from 28 tile columns on
                                       // (N = 3584 1.735 -> 1.670 ms, 4096 2.099 -> 2.054); with the hand-offs as stream memory
                                       // operations (option 26) from 20 (one stream vs two: N = 2048 0.875 vs 0.908 ms, 2304 1.026 vs
                                       // 1.022, 2560 1.145 vs 1.119, 2816 1.268 vs 1.251, 3072 1.373 vs 1.352, 3328 1.532 vs 1.484,
                                       // 3584 1.736 vs 1.606)

// ... and from COLUMN_MODE_MIN_TILES on when the whole problem runs in column mode (round 5: three launches per column on the
// panel stream, the rest on the main stream: one stream vs two at N = 1024 0.348 vs 0.336 ms, 1536 0.512 vs 0.475, 2048 0.665 vs
// 0.619; in panel mode two streams still lose there: N = 2048 0.665 vs 0.699).  Round 6: with the evaluation STARTING on the
// panel stream (option 45: no hand-off ahead of the first leaf) two streams win from 4 tile columns on (one stream vs two:
// N = 384 0.124 vs 0.124 ms, 512 0.168 vs 0.158, 640 0.211 vs 0.195, 768 0.258 vs 0.236, 896 0.305 vs 0.273); it was 8.
constexpr int COLUMN_MODE_MIN_TILES = 4;
static int lookahead_min_tiles(const mi_gp_handle* h, int ntc);
// Column mode for the WHOLE problem: up to rl_cols tile columns by the tail rule itself, and (round 6, option 46) up to rl_whole = 31:
// for 25 .. 31 tile columns a first panel of 1 .. 7 columns with its entry stall costs more than the main stream's lag behind
// the chain in the first columns (24 / 31: N = 3200 0.994 -> 0.946 ms, 3456 1.101 -> 1.059, 3584 1.156 -> 1.141, 3840 1.234 -> 1.209,
// 3968 1.292 -> 1.267; batches of 8 -1.7 .. -3.3 %).  At 32 columns it turns: N = 4096 1.416 -> 1.454 (a batch of 8 would still
// gain 2.7 %, but the rule is one of the shape alone, and the single evaluation decides it).
static bool whole_columns(const mi_gp_handle* h, int ntc) {
  return h->rl_cols > 0 && (ntc <= h->rl_cols || ntc <= h->rl_whole);
}

static hipError_t u_levels(mi_gp_handle* h, int final_cols, int max_s);

// COLUMN MODE (round 5, option 37): tile columns [cs, ntc) one by one.  The chain-bound end of a factorisation -- and all of
// a small one -- pays a fixed ~5-8 us per launch on the panel stream, so the fewest, shortest launches per column win: leaf,
// strip, and ONE thin update of the next column by the two columns before it (k = 256, both B operands from the strips'
// operand-order copies); everything older reaches a column through the main stream, which applies column p to the columns
// from p + 3 on (k = 128, 64x64 tiles) a column behind the chain:
//   panel stream:  leaf j [start: S_j -- strips <= j-1 are done | end: polls T_(j-2)]  strip j  thin(col j+1 <- cols j-1, j)
//   main stream:   wait S_j   update(cols >= j+2 <- col j-1)   signal T_(j-1)
// The main stream's update starts when leaf j HAS its CU (it would otherwise fill the chip in front of it) and has until the
// end of leaf j+1 -- ~58 us for ~20.  Which kernel updates a tile with which k is a matter of the column alone (not of the
// streams: on one stream the same launches run in program order), so every schedule returns the same bits.
// t_pending: something queued on the main stream writes columns > cs (the previous super-panel's update): leaf cs polls for it.
static hipError_t chol_columns(mi_gp_handle* h, double* A, long lda, int ntr, int ntc, int cs, hipStream_t T, hipStream_t P,
                               bool t_pending) {
  hipError_t e = hipSuccess;
#define CKC(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
  const bool two = P != T;
  const bool smo = two && h->use_smo >= 2;
  int tslot[3] = {-1, -1, -1};        // tslot[p % 3]: the slot behind the main stream's update by column p (or the entry update)
  hipEvent_t tev[3];
  bool tev_set[3] = {false, false, false};
  auto t_signal = [&](int idx) -> hipError_t {
    tslot[idx] = -1;
    tev_set[idx] = false;
    hipError_t se = hipSuccess;
    if (smo) tslot[idx] = signal_from(h, T, &se);
    if (se != hipSuccess) return se;
    if (tslot[idx] < 0) {
      se = next_event(h, &tev[idx]);
      if (se == hipSuccess) se = hipEventRecord(tev[idx], T);
      tev_set[idx] = true;
    }
    return se;
  };
  double* lsw0 = h->dinv_dev + (size_t)h->ntc * MINV_ELEMS;
  const int group = (h->btp && h->btp->nb > 1) ? h->rl_group : 1;
  int seg0 = cs;  // grouped schedule: first column (k-segment) the columns behind the chain's next one have not had yet
  if (two && t_pending) CKC(t_signal((cs + 1) % 3));  // polled by leaf cs: the index leaf j polls is (j - 2) mod 3 = (j + 1) mod 3
  for (int j = cs; j < ntc; ++j) {
    double* blk = A + (long)j * 128 * lda + (long)j * 128;
    double* dinv = h->dinv_dev + (size_t)j * MINV_ELEMS;
    const int m = (ntr - j - 1) * 128;
    // (the thin update of column c reads the strips of columns c - 2 and c - 1: a strip writes its operand-order copy when the
    // NEXT column's update is a thin one as well -- the limit is monotone in the column, so that covers this column's)
    auto thin_at = [&](int c) { return h->thin_max_wg > 0 && (long)(ntr - c - 1) * 8 <= h->thin_max_wg; };
    const bool thin_ok = thin_at(j);
    const bool lsw_out = thin_ok || (j + 2 < ntc && thin_at(j + 1));
    double* lswj = lsw0 + (size_t)(2 * (j & 1)) * MINV_ELEMS;
    // main stream's work of this step: column j - 1 (final since strip j - 1) updates the columns from j + 2 on
    const bool t_work = j - 1 >= cs && j + 2 < ntc;
    const int pidx = (j + 1) % 3;  // = (j - 2) mod 3
    const bool polls = two && (tslot[pidx] >= 0 || tev_set[pidx]);
    int sslot = -1;
    if (two && t_work) {
      if (smo && h->sig_next < SIG_SLOTS) sslot = h->sig_next++;
      else CKC(hand_off(h, P, T));  // (behind the previous step's thin update: strip j - 1 is done)
    }
    CKC(launch_potrf_leaf128(blk, lda, dinv, j * 128, h->info_dev, P, m == 128 ? blk + 128 * lda : nullptr, h->btp,
                             polls && tslot[pidx] >= 0 ? h->sig_dev + tslot[pidx] : nullptr, h->sig_epoch, h->poll_limit_log2,
                             sslot >= 0 ? h->sig_dev + sslot : nullptr));
    if (m > 128) CKC(launch_trsm_strip128(dinv, blk + 128 * lda, lda, m, P, h->btp, h->btp ? h->btp->sK : 0, lsw_out ? lswj : nullptr, 16));
    if (polls && tslot[pidx] < 0) CKC(hipStreamWaitEvent(P, tev[pidx], 0));
    tslot[pidx] = -1;
    tev_set[pidx] = false;
    if (j + 1 < ntc) {
      // the next column <- this one and (from the second column of the mode on) the one before it
      const bool k2 = j - 1 >= cs;
      const int k0 = k2 ? j - 1 : j, kw = k2 ? 2 : 1, mt = ntr - j - 1;
      double* Pp = A + (long)(j + 1) * 128 * lda + (long)k0 * 128;
      double* Cc = A + (long)(j + 1) * 128 * lda + (long)(j + 1) * 128;
      if (thin_ok) {
        const double* la = k2 ? lsw0 + (size_t)(2 * ((j - 1) & 1) + 1) * MINV_ELEMS : lswj;  // column j-1: its strip's SECOND block
        CKC(launch_syrk_thin(Pp, Cc, lda, mt, 1, kw * 128, P, h->btp, nullptr, h->sig_epoch, la, k2 ? lswj : nullptr));
      } else {
        CKC(syrk_trapezoid(h, A, lda, ntr, j + 1, 1, k0, kw, P));
      }
    }
    if (t_work) {
      if (sslot >= 0) CKC(hipStreamWaitValue32(T, h->sig_dev + sslot, h->sig_epoch, hipStreamWaitValueGte, 0xffffffffu));
      if (group <= 1) {
        CKC(syrk_trapezoid(h, A, lda, ntr, j + 2, ntc - j - 2, j - 1, 1, T));
        if (two) CKC(t_signal((j - 1) % 3));
      } else {
        // A batch is bound by the main stream's updates, not by the chain, and a k = 128 update reads and writes the trailing
        // matrices for 128 columns of k.  Same arithmetic, grouped: the column the chain needs next takes the segments it
        // has not had yet (k-segmented launch: the tile takes each 128-column partial sum as a launch of its own would), the
        // columns behind it take `group` segments at a time.  Invariant: every column >= j + 3 has exactly the segments < seg0.
        CKC(syrk_trapezoid(h, A, lda, ntr, j + 2, 1, seg0, j - seg0, T, 0, 0, 0, 0, false, nullptr, false, j - seg0 > 1 ? 128 : 0));
        if (two) CKC(t_signal((j - 1) % 3));
        if (j - seg0 >= group && j + 3 < ntc) {
          CKC(syrk_trapezoid(h, A, lda, ntr, j + 3, ntc - j - 3, seg0, j - seg0, T, 0, 0, 0, 0, false, nullptr, false, 128));
          seg0 = j;
        }
      }
    }
    if (two && h->u_early && j > cs && (j - cs) % 4 == 0) {
      // gradient evaluations: U = L^-T over the columns that are final (strips <= j - 1), behind the main stream's update
      const int upto = h->u_leaf_done + h->u_early_cols / 2 < j ? h->u_leaf_done + h->u_early_cols / 2 : j;
      if (!t_work && sslot < 0) CKC(hand_off(h, P, T));
      CKC(u_levels(h, upto, h->u_early_max_s));
    }
  }
#undef CKC
  return e;
}

static int lookahead_min_tiles(const mi_gp_handle* h, int ntc) {
  return whole_columns(h, ntc) ? COLUMN_MODE_MIN_TILES : LOOKAHEAD_MIN_TILES;
}

static hipError_t cholesky_enqueue(mi_gp_handle* h, double* A, long lda, int ntr, int ntc);
static hipError_t cholesky(mi_gp_handle* h, double* A, long lda, int ntr, int ntc) {
  const hipError_t e = cholesky_enqueue(h, A, lda, ntr, ntc);
  h->test_drop_signal = 0;  // (option 28 is for ONE evaluation, whether or not its schedule had the edge the hook drops)
  return e;
}

static hipError_t cholesky_enqueue(mi_gp_handle* h, double* A, long lda, int ntr, int ntc) {
  // A batched evaluation (blockIdx.z = problem) carries nb times the work per launch, so the look-ahead pays from smaller
  // problems on (nb = 8: N = 2560 +5 %, 3072 +10 %, 4096 +7 %; nb = 2 from 3072 on).  The super-panel widths stay those of
  // the single evaluation of the same size, so that a batch returns the single entry points' bits.
  const int nb = h->btp ? h->btp->nb : 1;
  const bool la_single = h->lookahead == 2 || (h->lookahead == 1 && ntc >= lookahead_min_tiles(h, ntc));
  const bool la = la_single || (h->lookahead == 1 && nb >= 2 && ntc >= (nb >= 8 ? 20 : 24));
  hipStream_t T = h->stream, P = la ? h->pstream : h->stream;
  hipError_t e;
#define CKE(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
  h->ev_next = 0;
  h->wait_col = -1;
  h->sig_next = 0;
  h->wait_slot = -1;
  h->wait2_col = h->wait2_slot = -1;
  h->done_col = h->done_slot = -1;
  if (++h->sig_epoch == 0xffffffffu) {  // (4e9 factorisations on one handle: start over)
    CKE(hipStreamSynchronize(h->stream));
    CKE(hipStreamSynchronize(h->pstream));
    CKE(hipMemset(h->sig_dev, 0, sizeof(unsigned) * SIG_SLOTS));
    h->sig_epoch = 1;
  }
  // the panel stream starts after what is queued on the main stream (assembly)
  if (P != T) {
    if (h->asm_on_panel && P == h->pstream) {}  // (the assembly is in front of the chain on this very stream)
    else CKE(hand_off(h, T, P));
  } else if (h->asm_on_panel) {
    CKE(hand_off(h, h->pstream, T));  // (cannot happen: enqueue_factor decides by the same rule; kept for safety)
  }
  h->asm_on_panel = false;
  const int wcap = (la_single && ntc <= NARROW_PANELS_MAX_TILES) ? 4 : 0;
  int w = pick_w(h, ntc, wcap);
  // EXTENDED super-panels (round 5, option 35): in the chain-bound part of a factorisation the panel's own in-panel updates
  // also cover the next super-panel's first tile column (chol_panel's nx = 1), level by level.  The separate update of that
  // column behind the panel ((a1): k = the panel's width, 23-33 us on the chain at N = 4096, and a one-lane launch for the
  // two edges in front of it, 8 us) becomes one k = 128 update behind the last strip, whose first workgroup also tells the
  // main stream that the panel is done.  A rule of the SHAPE alone (every schedule applies it, so the bits do not depend on
  // the schedule): at most ext_rows tile rows below the panel, more than EXT_MIN_REST tile columns behind it (the last
  // columns run on one stream, where it would only add a launch), problems of LOOKAHEAD_MIN_TILES tile columns or more.
  // While the trailing update is the critical path it would be wrong: the panel then waits for the main stream's bulk update
  // in its MIDDLE (the first in-panel update that touches the next column), and the main stream idles for the other half.
  constexpr int EXT_MIN_REST = 8;
  auto ext = [&](int c0, int wp) {
    const int m1 = c0 + wp;
    return (h->ext_rows > 0 && ntc >= LOOKAHEAD_MIN_TILES && ntr - m1 <= h->ext_rows && ntc - m1 > EXT_MIN_REST) ? 1 : 0;
  };
  int done_slot_cur = -1;  // the slot super-panel J's last in-panel update raises (extended panels on two streams)
  // edges of an extended panel [c0, c0 + wp) that is about to be queued on the panel stream: its first update of the next
  // column (behind the leaf of column c0 + wp / 2 - 1) needs everything queued on the main stream so far
  auto ext_edges = [&](int c0, int wp) -> hipError_t {
    done_slot_cur = -1;
    h->done_col = -1;
    if (P == T) return hipSuccess;
    if (h->use_smo >= 2 && h->sig_next + 2 <= SIG_SLOTS) {
      const int slot = h->sig_next++;
      hipError_t we = hipStreamWriteValue32(T, h->sig_dev + slot, h->sig_epoch, 0);
      h->wait2_col = c0 + (wp >= 2 ? wp / 2 : 1) - 1;
      h->wait2_slot = slot;
      done_slot_cur = h->done_slot = h->sig_next++;
      h->done_col = c0 + wp - 1;
      return we;
    }
    return hand_off(h, T, P);
  };
  // column mode (chol_columns) for the last rl_cols tile columns -- a rule of the shape alone, like the extended panels
  auto rl = [&](int c0) { return h->rl_cols > 0 && c0 < ntc && (ntc - c0 <= h->rl_cols || (c0 == 0 && whole_columns(h, ntc))); };
  if (rl(0)) {
    CKE(chol_columns(h, A, lda, ntr, ntc, 0, T, P, false));
    if (P != T) CKE(hand_off(h, P, T));
    return hipSuccess;
  }
  int nx_cur = ext(0, w);
  if (nx_cur) CKE(ext_edges(0, w));
  CKE(chol_panel(h, A, lda, ntr, 0, w, P, nx_cur));
  for (int J = 0; J < ntc;) {
    const int n1 = J + w;  // first tile column right of this super-panel
    // The panel stream's edges at a super-panel boundary: it tells the main stream that super-panel J is done (the main
    // stream may read it from here on) and, when it goes on to the next panel on its own stream, it waits for the main
    // stream's previous update of that panel's first column (the T -> P edge further down).  With option 26 = 2 the two are
    // ONE one-lane launch on the panel stream (write, then poll) instead of two runtime kernels; the main stream's halves
    // stay runtime stream memory operations.
    int tp_slot = -1;  // >= 0: the panel stream already waits for this slot; the T -> P edge below only has to write it
    if (P != T && h->u_early && J > 0 && ntc - J <= h->u_early_cols) {
      // Gradient evaluations: in the chain-bound last steps the main stream would now idle until the panel stream has
      // factored super-panel J.  The leaf blocks and the first block-doubling levels of U = L^-T over the columns that are
      // final (everything left of J) run here instead of behind the factorisation (same launches on the same tiles, only
      // grouped differently over the node batches: same bits).
      const int upto = h->u_leaf_done + h->u_early_cols / 2 < J ? h->u_leaf_done + h->u_early_cols / 2 : J;
      CKE(u_levels(h, upto, h->u_early_max_s));
    }
    if (P != T) {
      const bool stays_two = n1 < ntc && (rl(n1) || !(ntc - n1 <= h->single_below / nb));
      bool tp_edge = false;
      if (stays_two && !nx_cur) {
        const int wn_ = pick_w(h, ntc - n1, wcap);
        const bool merged_ = n1 + wn_ < ntc && h->merge_min_tiles > 0 && ntc - n1 >= h->merge_min_tiles;
        tp_edge = merged_ || J > 0;
      }
      if (nx_cur && done_slot_cur >= 0) {
        // (an extended panel: its last in-panel update raised the slot -- nothing to launch on the panel stream)
        CKE(hipStreamWaitValue32(T, h->sig_dev + done_slot_cur, h->sig_epoch, hipStreamWaitValueGte, 0xffffffffu));
      } else if (h->use_smo >= 2 && tp_edge && h->sig_next + 2 <= SIG_SLOTS) {
        const int a = h->sig_next++;
        tp_slot = h->sig_next++;
        CKE(launch_signal_write_wait(h->sig_dev + a, h->sig_dev + tp_slot, h->sig_epoch, h->info_dev, P, h->btp ? h->btp->nb : 1,
                                     h->btp ? h->btp->sinfo : 0, h->poll_limit_log2));
        CKE(hipStreamWaitValue32(T, h->sig_dev + a, h->sig_epoch, hipStreamWaitValueGte, 0xffffffffu));
      } else {
        CKE(hand_off(h, P, T));
      }
    }
    if (n1 >= ntc) break;
    if (rl(n1)) {
      // the rest column by column: column n1 <- super-panel J on the panel stream ((a1); an extended panel has done it),
      // the columns behind it <- super-panel J on the main stream, polled for by the first leaf
      if (P != T) {
        if (!nx_cur) {
          if (J > 0) {
            if (tp_slot >= 0) CKE(hipStreamWriteValue32(T, h->sig_dev + tp_slot, h->sig_epoch, 0));
            else CKE(hand_off(h, T, P));
          }
          CKE(syrk_trapezoid(h, A, lda, ntr, n1, 1, J, w, P));
        }
        if (ntc - n1 - 1 > 0) CKE(syrk_trapezoid(h, A, lda, ntr, n1 + 1, ntc - n1 - 1, J, w, T));
      } else if (ntc - n1 - nx_cur > 0) {
        CKE(syrk_trapezoid(h, A, lda, ntr, n1 + nx_cur, ntc - n1 - nx_cur, J, w, T));
      }
      CKE(chol_columns(h, A, lda, ntr, ntc, n1, T, P, P != T && ntc - n1 - 1 > 0));
      if (P != T) CKE(hand_off(h, P, T));
      break;
    }
    // The END of a large factorisation is a small one: below LOOKAHEAD_MIN_TILES trailing columns the cross-stream hand-offs
    // cost more than the overlap returns (that is why small problems run on one stream), so the rest runs on the main
    // stream alone (round 4, option 21; the super-panel widths stay what they were, so the arithmetic does not change).
    if (P != T && ntc - n1 <= h->single_below / nb) P = T;  // (a batch's launches carry nb times the work)
    const int wn = pick_w(h, ntc - n1, wcap);
    const bool bulk = n1 + wn < ntc;
    // tiles of the trailing update of columns [n1 + wn, ntc) / of the whole trailing trapezoid [n1, ntc)
    const int bc = ntc - n1 - wn, br = ntr - n1 - wn;
    const int btiles = bc * (bc + 1) / 2 + (br - bc) * bc;
    const int low = ntc - n1 <= h->lowocc_thr ? 1 : 0;
    const int nxn = ext(n1, wn);  // the panel queued in this step
    if (P != T && bulk && !nx_cur && h->merge_min_tiles > 0 && ntc - n1 >= h->merge_min_tiles) {
      // BULK-BOUND super-panels (round 4): the panel stream idles for most of such a step, so the next super-panel need not
      // be updated by launches of its own ((a1) on the panel stream + (a2) on the main stream, 64x64 tiles, ~55 TFLOP/s, a
      // last partial round each).  The whole trailing trapezoid [n1, ntc) is ONE enumeration on the 128x128-tile kernel with
      // the next super-panel's wn columns first; a prefix of full rounds that covers them runs two workgroups per CU with
      // nothing beside it, the panel stream starts behind it, and the rest follows as below (one per CU beside the chain,
      // then two per CU).  Same tiles and k order per tile as the split form.
      const int ac = ntc - n1, ar = ntr - n1;
      const int atiles = ac * (ac + 1) / 2 + (ar - ac) * ac;
      const int ft = wn * (wn + 1) / 2 + (ar - wn) * wn;
      int x1 = (ft + 511) / 512 * 512;
      if (x1 > atiles) x1 = atiles;
      CKE(syrk_trapezoid(h, A, lda, ntr, n1, ac, J, w, T, 0, 0, x1, wn));
      if (tp_slot >= 0) CKE(hipStreamWriteValue32(T, h->sig_dev + tp_slot, h->sig_epoch, 0));
      else CKE(hand_off(h, T, P));
      int done = x1;
      if (low && h->split_tiles > 0 && atiles - done >= h->split_tiles + h->split_min_rest) {
        CKE(syrk_trapezoid(h, A, lda, ntr, n1, ac, J, w, T, 1, done, h->split_tiles, wn));
        done += h->split_tiles;
        CKE(syrk_trapezoid(h, A, lda, ntr, n1, ac, J, w, T, 0, done, atiles, wn));
      } else if (atiles > done) {
        CKE(syrk_trapezoid(h, A, lda, ntr, n1, ac, J, w, T, low, done, atiles, wn));
      }
      if (nxn) CKE(ext_edges(n1, wn));
      CKE(chol_panel(h, A, lda, ntr, n1, wn, P, nxn));
      J = n1;
      w = wn;
      nx_cur = nxn;
      continue;
    }
    if (P != T) {
      // (a1) the next super-panel's FIRST tile column on the panel stream itself: the chain goes on to its leaf without
      //      waiting for the other wn - 1 columns (round 1 updated all wn columns on the main stream first: 40-80 us on
      //      the critical path per super-panel).  That column was last touched by the previous step's bulk update (b)
      //      on the main stream: wait for it first.
      // (a2) the other columns on the main stream meanwhile; the panel stream waits for them after that leaf + strip
      if (!nx_cur) {  // (an extended panel has updated column n1 itself, behind the main stream's earlier updates of it)
        if (J > 0) {
          if (tp_slot >= 0) CKE(hipStreamWriteValue32(T, h->sig_dev + tp_slot, h->sig_epoch, 0));
          else CKE(hand_off(h, T, P));
        }
        CKE(syrk_trapezoid(h, A, lda, ntr, n1, 1, J, w, P));
      }
      if (wn > 1) {
        // One workgroup per CU for problems of up to 48 tile columns: the chain's next leaf needs a CU to itself, and with two
        // 64x64-tile workgroups on every CU none empties before this grid drains (the first leaf of a super-panel waits 70-160 us
        // at N = 8192).  Beyond that the update itself takes so much longer at half occupancy that N >= 8192 loses 1.5-2 % (the
        // chain waits for THIS launch at those steps, not for the leaf); N <= 6144 gains 0.7-1 %.  Scheduling only.
        const int a2low = ntc <= 48 ? 1 : 0;
        CKE(syrk_trapezoid(h, A, lda, ntr, n1 + 1, wn - 1, J, w, T, a2low));
        if (h->test_drop_signal && h->use_smo >= 2 && h->sig_next < SIG_SLOTS) {
          // test hook (option 28): this edge's slot is never written -- the panel stream's poll has to give up
          h->test_drop_signal = 0;
          h->wait_slot = h->sig_next++;
          e = hipSuccess;
        } else {
          h->wait_slot = signal_from(h, T, &e);
        }
        if (e != hipSuccess) return e;
        if (h->wait_slot < 0) {
          CKE(next_event(h, &h->wait_ev));
          CKE(hipEventRecord(h->wait_ev, T));
        }
        h->wait_col = n1;
      }
    } else if (wn - nx_cur > 0) {
      CKE(syrk_trapezoid(h, A, lda, ntr, n1 + nx_cur, wn - nx_cur, J, w, T));
    }
    // (b) the rest of the trailing matrix, concurrently with that panel factorisation; once the panel chain is the
    // critical path the bulk update runs one workgroup per CU so that a leaf / strip workgroup fits beside it everywhere.
    // Enqueued BEFORE the chain's ~25 launches: when the host runs only just ahead of the device (under rocprofv3 it does:
    // 150-200 us of idle main stream per super-panel at N = 8192) the bulk update is already queued when (a2) ends.
    // (On a single stream the order cannot matter for the schedule; there the bulk update stays behind the chain, where
    // it measures 1.6 % faster -- 1.771 vs 1.800 ms per launch at N = 16384, same box, interleaved: it then starts after
    // ~0.5 ms of a mostly idle chip instead of straight after the next-panel update.)
    bool ext_done = false;
    if (bulk && P != T && nxn && h->use_smo >= 2) {
      // an extended panel follows: its chain polls for the bulk update of column n1 + wn in its middle -- that column first,
      // the signal, then the rest (the same tiles on the same kernels as one launch would give them: same bits)
      CKE(syrk_trapezoid(h, A, lda, ntr, n1 + wn, 1, J, w, T, low));
      CKE(ext_edges(n1, wn));
      ext_done = true;
      if (bc > 1) CKE(syrk_trapezoid(h, A, lda, ntr, n1 + wn + 1, bc - 1, J, w, T, low));
    } else if (bulk && P != T) {
      // Early super-panels are bound by the bulk update, not by the chain (the panel stream idles for most of it): only the
      // first split_tiles tiles run one workgroup per CU -- the mode that leaves every CU room for the chain's leaf /
      // strip / in-panel workgroups (and costs the kernel 5 % even alone) -- and the rest runs two per CU once the chain is through
      // (same tiles, same kernels: bit-identical results).  split_tiles ~ what the update gets done while a chain runs.
      if (low && h->split_tiles > 0 && btiles >= h->split_tiles + h->split_min_rest) {
        CKE(syrk_trapezoid(h, A, lda, ntr, n1 + wn, bc, J, w, T, 1, 0, h->split_tiles));
        CKE(syrk_trapezoid(h, A, lda, ntr, n1 + wn, bc, J, w, T, 0, h->split_tiles, btiles));
      } else {
        CKE(syrk_trapezoid(h, A, lda, ntr, n1 + wn, bc, J, w, T, low));
      }
    }
    // (an extended panel writes column n1 + wn: in every schedule BEHIND this step's bulk update of that column)
    if (bulk && P == T && nxn) CKE(syrk_trapezoid(h, A, lda, ntr, n1 + wn, ntc - n1 - wn, J, w, T, 0));
    if (nxn && !ext_done) CKE(ext_edges(n1, wn));
    CKE(chol_panel(h, A, lda, ntr, n1, wn, P, nxn));
    if (bulk && P == T && !nxn) CKE(syrk_trapezoid(h, A, lda, ntr, n1 + wn, ntc - n1 - wn, J, w, T, 0));
    J = n1;
    w = wn;
    nx_cur = nxn;
  }
#undef CKE
  return hipSuccess;
}

// Kernels of one evaluation: assembly, factorisation of the augmented trapezoid [[K],[y^T]] (L ends
// up in K_dev, beta = L^-1 y in row np), reduction.
static int enqueue_factor(mi_gp_handle* h, int noise_form, bool prof) {
  // Two-stream evaluations (round 6): the evaluation's first two kernels go to the PANEL stream, so that the first leaf follows the
  // assembly in stream order instead of behind a cross-stream edge (~10 us: N = 1024 0.335 -> 0.308 ms, 2048 0.651 -> 0.605; from 32
  // tile columns on, where a panel and not a column comes first, it is 0.1-0.4 %: N = 4096 1.411 -> 1.405, LML + gradient 2.515 -> 2.473).
  // The main stream's first launch waits for the panel stream anyway (a leaf's start signal in column mode, the first panel's
  // end otherwise), and every API call ends with both streams drained.  Same launches: scheduling only.  (The rule is
  // cholesky_enqueue's.)
  {
    const int nb_ = h->btp ? h->btp->nb : 1;
    const bool la_ = h->lookahead == 2 || (h->lookahead == 1 && h->ntc >= lookahead_min_tiles(h, h->ntc)) ||
                     (h->lookahead == 1 && nb_ >= 2 && h->ntc >= (nb_ >= 8 ? 20 : 24));
    h->asm_on_panel = la_ && h->start_on_panel;
  }
  const hipStream_t s0 = h->asm_on_panel ? h->pstream : h->stream;
  if (prof) (void)hipEventRecord(h->ev[0], s0);
  // first kernel of the evaluation: y rows, the bad-pivot word, and theta from the pinned host buffer to theta_dev
  HCK(launch_set_yrows(h->buf.K_dev, h->buf.lda, h->np, h->np, h->buf.y_dev, h->n, s0, h->info_dev, h->theta_host,
                       h->theta_dev, h->ntheta, h->btp), "set_yrows");
  // (Until round 6 evaluations of 96 tile columns and more assembled the first super-panel's columns first and the rest one
  // workgroup per CU beside its factorisation, option 24: with the faster assembly it measured level to 0.5 % behind one launch at
  // N = 12288 .. 20480 and 0.8 % behind at N = 8192, profiles/NOTES_r06.md -- removed.)
  HCK(launch_assemble(h->spec, h->theta_dev, h->buf.X_dev, h->n, h->buf.X_dev, h->n, h->buf.K_dev, h->buf.lda, h->np,
                      h->np, 1, noise_form, s0, 0, h->diag_dev, h->btp), "assemble");
  if (prof) (void)hipEventRecord(h->ev[1], s0);
  HCK(cholesky(h, h->buf.K_dev, h->buf.lda, h->ntc + 1, h->ntc), "cholesky");
  if (prof) (void)hipEventRecord(h->ev[2], h->stream);
  // the scalars go straight to the pinned host buffer (device-visible): no download launch behind the reduction
  h->eval_seq += 1.0;  // (exact in a double for 2^53 evaluations)
  HCK(launch_lml_reduce(h->buf.K_dev, h->buf.lda, h->buf.K_dev + (long)h->np * h->buf.lda, h->n, h->out_host, h->stream, h->info_dev,
                        h->btp, h->lr_part_dev, h->lr_sync_dev, h->eval_seq), "lml_reduce");
  if (prof) (void)hipEventRecord(h->ev[3], h->stream);
  return 0;
}

static int enqueue_gradient(mi_gp_handle* h, bool prof);
static hipError_t inverse_transpose(mi_gp_handle* h);

static int enqueue_all(mi_gp_handle* h, int what, bool prof) {
  h->u_leaf_done = 0;
  for (int& v : h->u_node_done) v = 0;
  // (from 64 tile columns on: N = 8192 LML + gradient 11.17 -> 10.98 ms, N = 16384 69.81 -> 69.40; at N = 4096 the main stream
  // has no idle time to fill in those steps: 2.74 -> 2.81)
  h->u_early = what == 2 && !h->btp && h->u_early_max_s > 0 && h->ntc >= 64 && h->buf.Z_dev && h->buf.W_dev;
  if (int r = enqueue_factor(h, what == 1 ? 1 : 0, prof)) return r;
  if (what == 2) return enqueue_gradient(h, prof);
  return 0;
}

// Run `what` (0 factor marginal form, 1 factor conditional form, 2 factor + gradient) as plain launches on the handle's
// stream(s).  Round 1 replayed a captured hipGraph per evaluation; measured again in round 2 (tools/time_sizes.py) replay
// is 1-4 % faster than plain launches from N = 4096 on and SLOWER below (N = 128: 0.104 vs 0.087 ms), its keep / drop
// heuristic made the timing depend on the instantiation, and the HIP runtime of this stack crashes in
// hip::Graph::UpdateStreams when executable graphs of two-stream captures come and go
// (profiles/r02_hipgraph_updatestreams_segv.txt; tools/stress_handles.py reproduced it in seconds) -- removed.
static int run_evaluation(mi_gp_handle* h, int what) {
  const bool prof = h->prof_level >= 1;
  h->gemm_ev_used = 0;
  h->gemm_flops_acc = 0.0;
  // theta travels inside the first kernel (set_yrows_kernel); the scalars and the gradient are written to pinned host
  // memory by the kernels that produce them: no copy launches
  return enqueue_all(h, what, prof);
}

// A poll of the last evaluation ran into its limit: the factor is unsynchronised garbage.  The reference's evaluations never
// fail for reasons of scheduling (a failed one is swallowed inside the optimiser loop, gpmcmc.py:331-339), so the handle gives
// up the protocol that needs concurrent dispatch -- polls enqueued ahead of the writes they wait for -- for event edges, says
// so once through mi_gp_last_error, and the caller's loop evaluates the same theta again (attempt 1).  A second time-out (event
// edges have no polls: the test hook, or a caller who re-armed option 26 in between) is the caller's error -2.
static int poll_timeout(mi_gp_handle* h, int attempt) {
  HCK(hipStreamSynchronize(h->pstream), "panel stream sync");  // (its remaining launches ran through: every later poll gave up at once)
  if (attempt == 0 && h->use_smo != 0) {
    h->use_smo = 0;
    h->demoted = true;
    snprintf(h->err, sizeof(h->err), "a cross-stream signal was not seen within its poll limit: this handle's cross-stream edges are "
                                     "events from now on (option 26 = 0), the evaluation was repeated");
    return 0;
  }
  snprintf(h->err, sizeof(h->err), "a cross-stream signal of the factorisation was not seen within its poll limit");
  return -2;
}

// The host's end of an evaluation.  The evaluation's last kernel -- lml_reduce, or the gradient's final reduction -- publishes
// the evaluation's sequence number in the pinned result buffer (out[4] / out[5] of every problem), released at system scope
// behind the scalars it stands for.  hipStreamSynchronize costs a round trip of 6-8 us behind that kernel's end (a completion
// signal and a blocked wait); spinning on the word sees it within the PCIe write latency: N = 128 0.056 -> 0.051 ms, 512 0.161 ->
// 0.153, 1024 0.310 -> 0.291, 4096 1.424 -> 1.387.  The spin has a budget (option 47, default 2 ms); an evaluation that runs
// into it synchronises the stream as before and the handle's next 15 evaluations do not spin at all, so a long evaluation
// costs a core 2 ms in 16 calls, not its run time.  The word is the LAST thing the evaluation's last kernel does, and that
// kernel writes nothing but the pinned result buffer: after a successful spin every device-side read and write of the
// evaluation is complete and only the kernel's RETIREMENT may be outstanding.  Everything this library does next goes through
// the same stream (in order); the paths that need idle streams (time-outs, the epoch wrap, profiling events, destruction)
// synchronise them themselves; every 256th spin synchronises the stream all the same (keeps the runtime's bookkeeping of
// completed launches short).  The panel stream is idle by then: the main stream's last kernels wait for it.
static hipError_t wait_evaluation(mi_gp_handle* h, int what, int k, bool prof) {
  bool seen = false;
  if (h->spin_us > 0 && !prof && h->spin_backoff == 0) {
    long long want;
    memcpy(&want, &h->eval_seq, sizeof(want));
    const double* f = h->out_host + (what == 2 ? 5 : 4);
    const auto t0 = std::chrono::steady_clock::now();
    int p = 0;
    for (unsigned it = 1;; ++it) {
      while (p < k && __atomic_load_n(reinterpret_cast<const long long*>(f + 16 * p), __ATOMIC_ACQUIRE) == want) ++p;
      if (p == k) { seen = true; break; }
      __builtin_ia32_pause();
      if ((it & 63u) == 0 && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > h->spin_us) break;
    }
    if (!seen) h->spin_backoff = 15;
    else if ((++h->spin_hits & 255u) == 0) seen = false;
  } else if (h->spin_backoff > 0) {
    --h->spin_backoff;
  }
  return seen ? hipSuccess : hipStreamSynchronize(h->stream);
}

static int factor_internal(mi_gp_handle* h, const double* theta, int what) {
  h->factored = false;
  h->have_kinv = false;
  h->have_u = false;
  if (!h->have_data) { snprintf(h->err, sizeof(h->err), "mi_gp_set_data has not been called"); return -1; }
  HCK(hipSetDevice(h->device), "hipSetDevice");
  for (int i = 0; i < h->ntheta; ++i) {
    if (!std::isfinite(theta[i])) { snprintf(h->err, sizeof(h->err), "theta[%d] is not finite", i); return -1; }
    h->theta_host[i] = theta[i];
  }
  const bool prof = h->prof_level >= 1;
  for (int attempt = 0;; ++attempt) {
    const auto t_enq0 = std::chrono::steady_clock::now();
    if (int r = run_evaluation(h, what)) return r;
    h->t_enqueue_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq0).count();
    HCK(wait_evaluation(h, what, 1, prof), "stream sync");
    if ((int)h->out_host[3] != SIGNAL_TIMEOUT_INFO) break;
    if (int r = poll_timeout(h, attempt)) return r;
  }
  if (prof) {
    float ms;
    (void)hipEventElapsedTime(&ms, h->ev[0], h->ev[1]); h->t_assemble_ms = ms;
    (void)hipEventElapsedTime(&ms, h->ev[1], h->ev[2]); h->t_chol_ms = ms;
    (void)hipEventElapsedTime(&ms, h->ev[2], h->ev[3]); h->t_reduce_ms = ms;
    (void)hipEventElapsedTime(&ms, h->ev[0], h->ev[3]); h->t_total_ms = ms;
    double g = 0.0, gb = 0.0, fb = 0.0, nb = 0.0;
    for (size_t i = 0; i + 1 < h->gemm_ev_used; i += 2) {
      (void)hipEventElapsedTime(&ms, h->gemm_ev[i], h->gemm_ev[i + 1]);
      g += ms;
      if (h->gemm_ev_big[i / 2]) { gb += ms; fb += h->gemm_ev_flops[i / 2]; nb += 1.0; }
    }
    h->t_gemm_ms = g;
    h->t_gemm_big_ms = gb;
    h->gemm_big_flops = fb;
    h->n_gemm_big = nb;
    h->gemm_flops = h->gemm_flops_acc;
    h->n_gemm = (double)(h->gemm_ev_used / 2);
    if (what == 2) {
      (void)hipEventElapsedTime(&ms, h->ev[4], h->ev[5]); h->t_trtri_ms = ms;
      (void)hipEventElapsedTime(&ms, h->ev[5], h->ev[6]); h->t_lauum_ms = ms;
      (void)hipEventElapsedTime(&ms, h->ev[6], h->ev[7]); h->t_contract_ms = ms;
    }
  }
  const int info = (int)h->out_host[3];  // forwarded by lml_reduce_kernel (reset by set_yrows_kernel)
  if (info != 0x7f7f7f7f) return info;  // 1-based index of the first bad pivot
  return 0;
}

extern "C" int mi_gp_lml(mi_gp_handle* h, const double* theta, double* lml_out) {
  if (!h || !theta || !lml_out) return -1;
  const int r = factor_internal(h, theta, 0);
  if (r < 0) return r;
  if (r > 0) { *lml_out = -INFINITY; return r; }
  *lml_out = h->out_host[0];
  return 0;
}

extern "C" int mi_gp_lml_parts(mi_gp_handle* h, double* logdet, double* quad) {
  if (!h) return -1;
  if (logdet) *logdet = h->out_host[1];
  if (quad) *quad = h->out_host[2];
  return 0;
}

// out: [assemble_ms, chol_ms, reduce_ms, total_ms, gemm_ms, gemm_flops, n_gemm_launches,
//       trtri_ms, lauum_ms, contract_ms, gemm_b_ms, gemm_b_flops, n_gemm_b_launches, enqueue_ms (host, any profiling level)]
extern "C" int mi_gp_timers(mi_gp_handle* h, double* out, int n) {
  if (!h || !out) return -1;
  const double v[14] = {h->t_assemble_ms, h->t_chol_ms, h->t_reduce_ms, h->t_total_ms, h->t_gemm_ms, h->gemm_flops,
                        h->n_gemm, h->t_trtri_ms, h->t_lauum_ms, h->t_contract_ms, h->t_gemm_big_ms,
                        h->gemm_big_flops, h->n_gemm_big, h->t_enqueue_ms};
  for (int i = 0; i < n && i < 14; ++i) out[i] = v[i];
  return 0;
}

// ---------------------------------------------------------------- gradient (K7)
// U = L^-T (upper triangular, row-major in Z_dev) by leaf solves + level-batched block doubling:
//   [[L11, 0], [L21, L22]]^-T = [[U11, -U11 L21^T U22], [0, U22]]
// then Kinv = U U^T (lower tiles, W_dev), alpha = U beta, and the contraction kernel.
static hipError_t gemm_call(mi_gp_handle* h, int ak, int bk, const double* A, long lda, long sA, const double* B, long ldb,
                            long sB, double* C, long ldc, long sC, int mt, int nt, int k, int tri, int kmode,
                            double alpha, double beta, int batch, long zA = 0, long zB = 0, long zC = 0) {
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.strideA = sA; p.strideB = sB; p.strideC = sC;
  p.mt = mt; p.nt = nt; p.k = k; p.tri = tri; p.kmode = kmode; p.alpha = alpha; p.beta = beta;
  p.small_below = h->small_below; p.band = h->band_rows; p.tail_small = h->tail_small;
  if (h->btp) {  // batched evaluation: the problems are the second batch level (zA / zB / zC: the strides of the matrices A, B, C live in)
    p.batch1 = batch;
    p.strideA2 = zA; p.strideB2 = zB; p.strideC2 = zC;
    batch *= h->btp->nb;
  }
  return launch_gemm_f64(p, ak, bk, batch, h->stream);
}

// Leaf blocks and FULL nodes of the block-doubling levels of U over tile columns [0, final_cols) of L, as far as they are
// not done yet (u_leaf_done / u_node_done): level s (nodes of 2 s tiles, li = log2 s) needs its nodes' halves -- full nodes of
// level s / 2 -- done.  Called with growing final_cols inside the factorisation's tail (cholesky()) and once with everything
// from inverse_transpose(); the node batches are split differently, the per-tile arithmetic is the same.
static hipError_t u_levels(mi_gp_handle* h, int final_cols, int max_s) {
  const double* L = h->buf.K_dev;
  double* U = h->buf.Z_dev;
  double* T = h->buf.W_dev;
  const long ld = h->buf.lda;
  const int ntc = h->ntc;
  const long zK = h->btp ? h->btp->sK : 0, zZ = h->btp ? h->btp->sZ : 0, zW = h->btp ? h->btp->sW : 0;
  hipError_t e;
  if (final_cols > ntc) final_cols = ntc;
  if (final_cols > h->u_leaf_done) {
    if (h->u_leaf_done == 0) {
      e = launch_set_identity_blocks(U, ld, ntc, h->stream, h->btp);
      if (e != hipSuccess) return e;
    }
    // leaves: X L_kk^T = I  ->  X = L_kk^-T
    const int c0 = h->u_leaf_done;
    e = launch_trsm_strip128_batched(h->dinv_dev + (size_t)c0 * MINV_ELEMS, U + (long)c0 * (128 * ld + 128), ld, 128 * ld + 128, 128,
                                     final_cols - c0, h->stream, h->btp, zZ);
    if (e != hipSuccess) return e;
    h->u_leaf_done = final_cols;
  }
  int li = 0;
  for (int s = 1; s < ntc && s <= max_s; s *= 2, ++li) {
    const int child_cols = li == 0 ? h->u_leaf_done : h->u_node_done[li - 1] * s;  // columns covered by finished halves
    const int avail = child_cols / (2 * s);  // (<= ntc / (2 s): only full nodes)
    const int done = h->u_node_done[li];
    if (avail <= done) continue;
    const long node = (long)2 * s * 128 * (ld + 1);
    const long off = (long)done * node;
    const int batch = avail - done;
    const double* U11 = U + off;
    const double* U22 = U + off + (long)s * 128 * (ld + 1);
    const double* L21 = L + off + (long)s * 128 * ld;
    double* P = T + off + (long)s * 128;
    double* U12 = U + off + (long)s * 128;
    // P = U11 L21^T   (U11 upper triangular: k >= row tile)
    e = gemm_call(h, 0, 0, U11, ld, node, L21, ld, node, P, ld, node, s, s, s * 128, 0, 3, 1.0, 0.0, batch, zZ, zK, zW);
    if (e != hipSuccess) return e;
    // U12 = -P U22    (U22 upper triangular: k <= column tile)
    e = gemm_call(h, 0, 1, P, ld, node, U22, ld, node, U12, ld, node, s, s, s * 128, 0, 4, -1.0, 0.0, batch, zW, zZ, zZ);
    if (e != hipSuccess) return e;
    h->u_node_done[li] = avail;
  }
  return hipSuccess;
}

static hipError_t inverse_transpose(mi_gp_handle* h) {
  const double* L = h->buf.K_dev;
  double* U = h->buf.Z_dev;
  double* T = h->buf.W_dev;
  const long ld = h->buf.lda;
  const int ntc = h->ntc;
  const long zK = h->btp ? h->btp->sK : 0, zZ = h->btp ? h->btp->sZ : 0, zW = h->btp ? h->btp->sW : 0;
  // every full node of every level (what the factorisation's tail has not done already), then the trailing partial nodes
  // level by level: a partial node's first half is a full node of the level below, its second half is built by the partial
  // nodes of the levels below
  hipError_t e = u_levels(h, ntc, 1 << 30);
  if (e != hipSuccess) return e;
  for (int s = 1; s < ntc; s *= 2) {
    const int nfull = ntc / (2 * s);             // nodes whose second half is complete
    const int rem = ntc - nfull * 2 * s;         // tiles left for a trailing partial node
    if (rem <= s) continue;
    const long node = (long)2 * s * 128 * (ld + 1);
    const int s2 = rem - s;
    const long off = (long)nfull * node;
    const double* U11 = U + off;
    const double* U22 = U + off + (long)s * 128 * (ld + 1);
    const double* L21 = L + off + (long)s * 128 * ld;
    double* P = T + off + (long)s * 128;
    double* U12 = U + off + (long)s * 128;
    e = gemm_call(h, 0, 0, U11, ld, node, L21, ld, node, P, ld, node, s, s2, s * 128, 0, 3, 1.0, 0.0, 1, zZ, zK, zW);
    if (e != hipSuccess) return e;
    e = gemm_call(h, 0, 1, P, ld, node, U22, ld, node, U12, ld, node, s, s2, s2 * 128, 0, 4, -1.0, 0.0, 1, zW, zZ, zZ);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// everything after the factorisation: U = L^-T, Kinv = U U^T, alpha = U beta, contraction, download
static int enqueue_gradient(mi_gp_handle* h, bool prof) {
  if (prof) (void)hipEventRecord(h->ev[4], h->stream);
  HCK(inverse_transpose(h), "inverse_transpose");
  if (prof) (void)hipEventRecord(h->ev[5], h->stream);
  const long ld = h->buf.lda;
  // Kinv = U U^T, lower tiles only, k >= row tile
  const long zZ = h->btp ? h->btp->sZ : 0, zW = h->btp ? h->btp->sW : 0;
  HCK(gemm_call(h, 0, 0, h->buf.Z_dev, ld, 0, h->buf.Z_dev, ld, 0, h->buf.W_dev, ld, 0, h->ntc, h->ntc, h->np, 1, 3, 1.0,
                0.0, 1, zZ, zZ, zW), "lauum");
  if (prof) (void)hipEventRecord(h->ev[6], h->stream);
  HCK(launch_trmv_upper(h->buf.Z_dev, ld, h->buf.K_dev + (long)h->np * ld, h->n, h->alpha_dev, h->stream, h->btp), "trmv");
  // the final reduction writes the gradient straight into the handle's pinned host buffer (device-visible)
  HCK(launch_grad_contract(h->spec, h->theta_dev, h->buf.X_dev, h->n, h->buf.W_dev, ld, h->alpha_dev, h->part_dev,
                           h->grad_host, h->stream, h->btp, h->lr_sync_dev + (h->btp ? h->btp->nb : 1), h->out_host + 5, h->eval_seq),
      "grad_contract");
  if (prof) (void)hipEventRecord(h->ev[7], h->stream);
  return 0;
}

extern "C" int mi_gp_lml_grad(mi_gp_handle* h, const double* theta, double* lml_out, double* grad_out) {
  if (!h || !theta || !lml_out || !grad_out) return -1;
  if (!h->buf.Z_dev || !h->buf.W_dev) {
    snprintf(h->err, sizeof(h->err), "mi_gp_lml_grad needs Z_dev and W_dev in mi_gp_set_data");
    return -1;
  }
  // the gradient kernels run unconditionally behind the factorisation (one captured DAG); on a
  // non-positive-definite K their output is discarded
  const int r = factor_internal(h, theta, 2);
  if (r < 0) return r;
  for (int i = 0; i < h->ntheta; ++i) grad_out[i] = 0.0;
  if (r > 0) { *lml_out = -INFINITY; return r; }
  *lml_out = h->out_host[0];
  for (int i = 0; i < h->ntheta; ++i) grad_out[i] = h->grad_host[i];
  h->have_kinv = true;
  return 0;
}

// Data-side gradients of the LML at the theta of the last successful mi_gp_lml_grad (whose K^-1 and alpha are
// still resident): dLML/dy = -alpha and dLML/dX.  They feed the chain rule through the reference's output and
// input warps (cwgp / iwgp, gpmcmc.py:211-279) and through the free observation rows of inverse_opt
// (gpmcmc.py:1096-1101), which PyMC differentiates by autodiff through the same Cholesky.
extern "C" int mi_gp_alpha(mi_gp_handle* h, double* alpha_host) {
  if (!h || !alpha_host) return -1;
  if (!h->have_kinv) { snprintf(h->err, sizeof(h->err), "mi_gp_alpha: call mi_gp_lml_grad first"); return -1; }
  HCK(hipSetDevice(h->device), "hipSetDevice");
  HCK(hipMemcpyAsync(alpha_host, h->alpha_dev, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->stream), "alpha download");
  HCK(hipStreamSynchronize(h->stream), "stream sync");
  return 0;
}

extern "C" int mi_gp_grad_x(mi_gp_handle* h, double* gx_dev) {
  if (!h || !gx_dev) return -1;
  if (!h->have_kinv) { snprintf(h->err, sizeof(h->err), "mi_gp_grad_x: call mi_gp_lml_grad first"); return -1; }
  HCK(hipSetDevice(h->device), "hipSetDevice");
  const int nsplit = grad_x_splits(h->n, h->cfg.d);
  if (nsplit > 1 && !h->gxs_dev)
    HCK(hipMalloc(&h->gxs_dev, sizeof(double) * (size_t)nsplit * h->n * h->cfg.d), "grad_x scratch");
  HCK(launch_grad_x(h->spec, h->theta_dev, h->buf.X_dev, h->n, h->buf.W_dev, h->buf.lda, h->alpha_dev, gx_dev,
                    nsplit > 1 ? h->gxs_dev : nullptr, h->stream), "grad_x");
  HCK(hipStreamSynchronize(h->stream), "stream sync");
  return 0;
}

// Optional per-point diagonal (n doubles on the device, borrowed; nullptr removes it) added to K at assembly on
// top of the (gv, jitter) terms of theta: the observation-noise vector of inverse_opt (gpmcmc.py:1134-1158).
extern "C" int mi_gp_set_diag(mi_gp_handle* h, const double* diag_dev) {
  if (!h) return -1;
  h->diag_dev = diag_dev;
  h->factored = false;
  h->have_kinv = false;
  return 0;
}

// ---------------------------------------------------------------- batched evaluation
// K covariances of the SAME inputs (one theta each) factorised in lockstep: every launch of the evaluation carries
// blockIdx.z = problem.  One evaluation below N ~ 10^4 is bound by its serial panel chain (leaf -> strip -> update per 128
// columns) and leaves most of the chip idle; MAP restarts (gpmcmc.py:328-343) and the NUTS chains that share a GPU
// (gpmcmc.py:351) evaluate the same data at different theta, so their chains can run side by side inside the same launches
// instead of on separate handles and streams (which stops paying at the fourth handle: hardware queues).
extern "C" int mi_gp_set_batch(mi_gp_handle* h, const mi_gp_batch_buffers* b) {
  if (!h || !b || !b->K_dev || b->count < 1) return -1;
  const long need_k = (long)(h->np + 128) * h->buf.lda, need_z = (long)h->np * h->buf.lda;
  if (!h->have_data) { snprintf(h->err, sizeof(h->err), "mi_gp_set_batch: call mi_gp_set_data first (lda is taken from it)"); return -1; }
  if (b->stride_k < need_k || ((b->Z_dev || b->W_dev) && b->stride_zw < need_z) || (b->stride_k & 1) || (b->stride_zw & 1)) {
    snprintf(h->err, sizeof(h->err), "mi_gp_set_batch: strides must be even and >= (np + 128) * lda = %ld (K), np * lda = %ld (Z, W)", need_k, need_z);
    return -1;
  }
  HCK(hipSetDevice(h->device), "hipSetDevice");
  if (b->count > h->batch_cap) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->b_theta_dev); (void)hipFree(h->b_dinv_dev); (void)hipFree(h->b_alpha_dev); (void)hipFree(h->b_part_dev);
    (void)hipFree(h->b_info_dev); (void)hipFree(h->b_lr_part_dev); (void)hipFree(h->b_lr_sync_dev);
    h->b_lr_part_dev = nullptr; h->b_lr_sync_dev = nullptr;
    if (h->b_grad_host) (void)hipHostFree(h->b_grad_host);
    if (h->b_out_host) (void)hipHostFree(h->b_out_host);
    if (h->b_theta_host) (void)hipHostFree(h->b_theta_host);
    h->b_theta_dev = h->b_dinv_dev = h->b_alpha_dev = h->b_part_dev = h->b_grad_host = h->b_out_host = h->b_theta_host = nullptr;
    h->b_info_dev = nullptr;
    h->batch_cap = 0;
    const size_t k = (size_t)b->count;
    HCK(hipMalloc(&h->b_theta_dev, sizeof(double) * k * h->ntheta), "batch scratch");
    HCK(hipMalloc(&h->b_dinv_dev, sizeof(double) * k * MINV_ELEMS * (size_t)(h->ntc + 4)), "batch scratch");
    HCK(hipMalloc(&h->b_alpha_dev, sizeof(double) * k * h->np), "batch scratch");
    HCK(hipMalloc(&h->b_part_dev, sizeof(double) * k * (size_t)grad_contract_blocks(h->n) * h->ntheta), "batch scratch");
    HCK(hipMalloc(&h->b_info_dev, sizeof(int) * 4 * k), "batch scratch");
    HCK(hipMalloc(&h->b_lr_part_dev, sizeof(double) * 2 * LML_REDUCE_BLOCKS * k), "batch scratch");
    HCK(hipMalloc(&h->b_lr_sync_dev, sizeof(unsigned) * 2 * k), "batch scratch");  // [0, k) lml_reduce's tickets, [k, 2k) grad_final's
    HCK(hipMemset(h->b_lr_sync_dev, 0, sizeof(unsigned) * 2 * k), "batch scratch");
    HCK(hipHostMalloc(&h->b_grad_host, sizeof(double) * k * h->ntheta), "batch scratch");
    HCK(hipHostMalloc(&h->b_out_host, sizeof(double) * 16 * k), "batch scratch");
    memset(h->b_out_host, 0, sizeof(double) * 16 * k);
    HCK(hipHostMalloc(&h->b_theta_host, sizeof(double) * k * h->ntheta), "batch scratch");
    h->batch_cap = b->count;
  }
  h->bbuf = *b;
  return 0;
}

// what: 0 LML, 2 LML + gradient.  info_out[p]: 0, or the 1-based index of problem p's first bad pivot (its LML is -inf then).
static int batch_internal(mi_gp_handle* h, int k, const double* thetas, int what, double* lml_out, double* grad_out, int* info_out) {
  if (!h->have_data || h->batch_cap < 1) { snprintf(h->err, sizeof(h->err), "call mi_gp_set_data and mi_gp_set_batch first"); return -1; }
  if (k < 1 || k > h->bbuf.count) { snprintf(h->err, sizeof(h->err), "batch of %d problems, buffers for %d", k, h->bbuf.count); return -1; }
  if (what == 2 && (!h->bbuf.Z_dev || !h->bbuf.W_dev)) { snprintf(h->err, sizeof(h->err), "mi_gp_lml_grad_batch needs Z_dev and W_dev in mi_gp_set_batch"); return -1; }
  HCK(hipSetDevice(h->device), "hipSetDevice");
  for (int i = 0; i < k * h->ntheta; ++i) {
    if (!std::isfinite(thetas[i])) { snprintf(h->err, sizeof(h->err), "theta[%d] of problem %d is not finite", i % h->ntheta, i / h->ntheta); return -1; }
    h->b_theta_host[i] = thetas[i];
  }
  h->factored = h->have_kinv = h->have_u = false;  // the single-evaluation state of the handle is not touched, but K_dev may alias
  // point the evaluation at the batch's arrays, run the ordinary enqueue code with blockIdx.z = problem, restore
  const mi_gp_buffers buf0 = h->buf;
  double *theta_dev0 = h->theta_dev, *dinv0 = h->dinv_dev, *alpha0 = h->alpha_dev, *part0 = h->part_dev, *grad0 = h->grad_host,
         *out0 = h->out_host, *thost0 = h->theta_host;
  int* info0 = h->info_dev;
  double* lrp0 = h->lr_part_dev;
  unsigned* lrs0 = h->lr_sync_dev;
  h->lr_part_dev = h->b_lr_part_dev; h->lr_sync_dev = h->b_lr_sync_dev;
  h->buf.K_dev = h->bbuf.K_dev; h->buf.Z_dev = h->bbuf.Z_dev; h->buf.W_dev = h->bbuf.W_dev;
  h->theta_dev = h->b_theta_dev; h->dinv_dev = h->b_dinv_dev; h->alpha_dev = h->b_alpha_dev; h->part_dev = h->b_part_dev;
  h->grad_host = h->b_grad_host; h->out_host = h->b_out_host; h->theta_host = h->b_theta_host; h->info_dev = h->b_info_dev;
  h->bt.nb = k;
  h->bt.sK = h->bbuf.stride_k; h->bt.sZ = h->bt.sW = h->bbuf.stride_zw;
  h->bt.sdinv = (long)MINV_ELEMS * (h->ntc + 4); h->bt.salpha = h->np;
  h->bt.spart = (long)grad_contract_blocks(h->n) * h->ntheta;
  h->bt.stheta = h->ntheta; h->bt.sinfo = 4; h->bt.sout = 16;
  h->btp = &h->bt;
  const int prof0 = h->prof_level;
  h->prof_level = 0;
  int r = 0;
  for (int attempt = 0;; ++attempt) {
    r = run_evaluation(h, what);
    if (r == 0) {
      const hipError_t e = wait_evaluation(h, what, k, false);
      if (e != hipSuccess) r = hfail(h, e, "stream sync");
    }
    if (r != 0) break;
    // a cross-stream poll that gave up leaves unsynchronised data behind in EVERY problem: the whole batch is evaluated again
    // with event edges (poll_timeout), or fails as a whole
    bool timed_out = false;
    for (int p = 0; p < k; ++p) timed_out = timed_out || (int)h->out_host[16 * p + 3] == SIGNAL_TIMEOUT_INFO;
    if (!timed_out) break;
    r = poll_timeout(h, attempt);
    if (r != 0) break;
  }
  h->prof_level = prof0;
  h->btp = nullptr;
  h->bt = Batch();
  h->buf = buf0;
  h->theta_dev = theta_dev0; h->dinv_dev = dinv0; h->alpha_dev = alpha0; h->part_dev = part0; h->grad_host = grad0;
  h->out_host = out0; h->theta_host = thost0; h->info_dev = info0;
  h->lr_part_dev = lrp0; h->lr_sync_dev = lrs0;
  if (r != 0) return r;
  for (int p = 0; p < k; ++p) {
    const int info = (int)h->b_out_host[16 * p + 3];
    const bool ok = info == 0x7f7f7f7f;
    if (info_out) info_out[p] = ok ? 0 : info;
    lml_out[p] = ok ? h->b_out_host[16 * p] : -INFINITY;
    if (grad_out)
      for (int i = 0; i < h->ntheta; ++i) grad_out[(size_t)p * h->ntheta + i] = ok ? h->b_grad_host[(size_t)p * h->ntheta + i] : 0.0;
  }
  return 0;
}

extern "C" int mi_gp_lml_batch(mi_gp_handle* h, int k, const double* thetas, double* lml_out, int* info_out) {
  if (!h || !thetas || !lml_out) return -1;
  return batch_internal(h, k, thetas, 0, lml_out, nullptr, info_out);
}

extern "C" int mi_gp_lml_grad_batch(mi_gp_handle* h, int k, const double* thetas, double* lml_out, double* grad_out, int* info_out) {
  if (!h || !thetas || !lml_out || !grad_out) return -1;
  return batch_internal(h, k, thetas, 2, lml_out, grad_out, info_out);
}

// ---------------------------------------------------------------- conditional (K8)
extern "C" int mi_gp_factor(mi_gp_handle* h, const double* theta) {
  if (!h || !theta) return -1;
  const int r = factor_internal(h, theta, 1);
  h->factored = (r == 0);
  return r;
}

// solve X L^T = B in place for tile columns [c0, c0+w) of the mp x np work matrix
static hipError_t trsm_rec(mi_gp_handle* h, double* Bw, long ldw, int mp, int c0, int w) {
  const double* L = h->buf.K_dev;
  const long lda = h->buf.lda;
  if (w == 1) {
    return launch_trsm_strip128(h->dinv_dev + (size_t)c0 * MINV_ELEMS, Bw + (long)c0 * 128, ldw, mp, h->stream);
  }
  const int w1 = w / 2, w2 = w - w1;
  hipError_t e = trsm_rec(h, Bw, ldw, mp, c0, w1);
  if (e != hipSuccess) return e;
  // B[:, c0+w1 : c0+w) -= X[:, c0 : c0+w1) * L[c0+w1 : c0+w, c0 : c0+w1)^T
  e = gemm_call(h, 0, 0, Bw + (long)c0 * 128, ldw, 0, L + (long)(c0 + w1) * 128 * lda + (long)c0 * 128, lda, 0,
                Bw + (long)(c0 + w1) * 128, ldw, 0, mp / 128, w2, w1 * 128, 0, 0, -1.0, 1.0, 1);
  if (e != hipSuccess) return e;
  return trsm_rec(h, Bw, ldw, mp, c0 + w1, w2);
}

extern "C" int mi_gp_predict(mi_gp_handle* h, const double* Xnew_dev, int m, double* work_dev, long ldw,
                             double* mean_dev, double* var_dev, int pred_noise) {
  if (!h || !Xnew_dev || !work_dev || !mean_dev || !var_dev || m <= 0) return -1;
  if (!h->factored) { snprintf(h->err, sizeof(h->err), "mi_gp_predict: call mi_gp_factor first"); return -1; }
  if (ldw < h->np || (ldw & 1)) { snprintf(h->err, sizeof(h->err), "mi_gp_predict: ldw must be even and >= padded n"); return -1; }
  HCK(hipSetDevice(h->device), "hipSetDevice");
  const int mp = (m + 127) / 128 * 128;
  // K(Xnew, X): one prediction point per row, zeros in the padding
  HCK(launch_assemble(h->spec, h->theta_dev, Xnew_dev, m, h->buf.X_dev, h->n, work_dev, ldw, mp, h->np, 0, 0, h->stream),
      "assemble cross");
  HCK(trsm_rec(h, work_dev, ldw, mp, 0, h->ntc), "trsm");
  // Stationary.diag == 1: the composite diagonal is the +/* fold of kv; pred_noise adds sqrt(gv)^2
  const int nk = h->spec.nkern, d = h->spec.d;
  const double* th = h->theta_host;
  double kd = th[nk * d];
  for (int c = 1; c < nk; ++c) kd = (h->spec.op[c - 1] == 0) ? kd + th[nk * d + c] : kd * th[nk * d + c];
  const double sg = std::sqrt(th[nk * d + 2 * nk]);
  HCK(launch_predict_reduce(work_dev, ldw, h->buf.K_dev + (long)h->np * h->buf.lda, h->n, m, kd,
                            pred_noise ? sg * sg : 0.0, mean_dev, var_dev, h->stream), "predict_reduce");
  HCK(hipStreamSynchronize(h->stream), "stream sync");
  return 0;
}

// The same conditional through U = L^-T: A = K(X*, X) U is ONE triangular-k GEMM (k < (tj+1)*128, ~70 TFLOP/s) instead
// of the blocked triangular solve (~250 launches, ~30 TFLOP/s on tall-skinny right-hand sides).  U costs N^3/3 flops
// once per factorisation, so this is the path for sweeps of many points at fixed hyper-parameters (BO's 10 000-point
// proposals, differential-evolution generations).  Needs Z_dev / W_dev; work_dev must hold 2 * ceil(m/128)*128 rows.
extern "C" int mi_gp_predict_u(mi_gp_handle* h, const double* Xnew_dev, int m, double* work_dev, long ldw,
                               double* mean_dev, double* var_dev, int pred_noise) {
  if (!h || !Xnew_dev || !work_dev || !mean_dev || !var_dev || m <= 0) return -1;
  if (!h->factored) { snprintf(h->err, sizeof(h->err), "mi_gp_predict_u: call mi_gp_factor first"); return -1; }
  if (!h->buf.Z_dev || !h->buf.W_dev) {
    snprintf(h->err, sizeof(h->err), "mi_gp_predict_u needs Z_dev and W_dev in mi_gp_set_data");
    return -1;
  }
  if (ldw < h->np || (ldw & 1)) { snprintf(h->err, sizeof(h->err), "mi_gp_predict_u: ldw must be even and >= padded n"); return -1; }
  HCK(hipSetDevice(h->device), "hipSetDevice");
  const long ld = h->buf.lda;
  if (!h->have_u) {
    HCK(inverse_transpose(h), "inverse_transpose");
    HCK(launch_trmv_upper(h->buf.Z_dev, ld, h->buf.K_dev + (long)h->np * ld, h->n, h->alpha_dev, h->stream), "trmv");
    h->have_u = true;
  }
  const int mp = (m + 127) / 128 * 128;
  double* krows = work_dev + (long)mp * ldw;
  HCK(launch_assemble(h->spec, h->theta_dev, Xnew_dev, m, h->buf.X_dev, h->n, krows, ldw, mp, h->np, 0, 0, h->stream),
      "assemble cross");
  GemmParams p;
  p.A = krows; p.B = h->buf.Z_dev; p.C = work_dev;
  p.lda = ldw; p.ldb = ld; p.ldc = ldw;
  p.strideA = p.strideB = p.strideC = 0;
  p.mt = mp / 128; p.nt = h->ntc; p.k = h->np; p.tri = 0; p.kmode = 4; p.alpha = 1.0; p.beta = 0.0;
  p.small_below = h->small_below; p.band = h->band_rows; p.tail_small = h->tail_small;
  HCK(launch_gemm_f64(p, 0, 1, 1, h->stream), "K* U");
  const int nk = h->spec.nkern, d = h->spec.d;
  const double* th = h->theta_host;
  double kd = th[nk * d];
  for (int c = 1; c < nk; ++c) kd = (h->spec.op[c - 1] == 0) ? kd + th[nk * d + c] : kd * th[nk * d + c];
  const double sg = std::sqrt(th[nk * d + 2 * nk]);
  HCK(launch_predict_reduce(work_dev, ldw, h->buf.K_dev + (long)h->np * h->buf.lda, h->n, m, kd,
                            pred_noise ? sg * sg : 0.0, mean_dev, var_dev, h->stream), "predict_reduce");
  HCK(hipStreamSynchronize(h->stream), "stream sync");
  return 0;
}

// Posterior mean / variance at m points AND their gradients w.r.t. the (converted) points: the differentiable
// predictive of BO's refinement (gpmcmc.py:766-801).  Needs Z_dev / W_dev (U = L^-T is formed once per
// mi_gp_factor, N^3/3 flops on the GEMM kernel) and work_dev with 2 * ceil(m/128)*128 rows: the second half
// receives w_p = K^-1 k(X, x*_p) = U (L^-1 k*_p), one row per point.
extern "C" int mi_gp_predict_grad(mi_gp_handle* h, const double* Xnew_dev, int m, double* work_dev, long ldw,
                                  double* mean_dev, double* var_dev, int pred_noise, double* dmean_dev,
                                  double* dvar_dev) {
  if (!h || !dmean_dev || !dvar_dev) return -1;
  if (!h->buf.Z_dev || !h->buf.W_dev) {
    snprintf(h->err, sizeof(h->err), "mi_gp_predict_grad needs Z_dev and W_dev in mi_gp_set_data");
    return -1;
  }
  if ((size_t)(h->cfg.nkern + 1) * h->cfg.d * sizeof(double) > 61440) {
    snprintf(h->err, sizeof(h->err), "mi_gp_predict_grad: (nkern + 1) * d must fit 61440 bytes of LDS (d <= %d here)", 7680 / (h->cfg.nkern + 1));
    return -1;
  }
  if (!h->factored) { snprintf(h->err, sizeof(h->err), "mi_gp_predict_grad: call mi_gp_factor first"); return -1; }
  if (!Xnew_dev || !work_dev || !mean_dev || !var_dev || m <= 0) return -1;
  if (ldw < h->np || (ldw & 1)) { snprintf(h->err, sizeof(h->err), "mi_gp_predict_grad: ldw must be even and >= padded n"); return -1; }
  HCK(hipSetDevice(h->device), "hipSetDevice");
  const long ld = h->buf.lda;
  if (!h->have_u) {
    HCK(inverse_transpose(h), "inverse_transpose");
    HCK(launch_trmv_upper(h->buf.Z_dev, ld, h->buf.K_dev + (long)h->np * ld, h->n, h->alpha_dev, h->stream), "trmv");
    h->have_u = true;
  }
  if (m <= 16) {
    // few points (BO refinement): with U resident, A_p = L^-1 k*_p = U^T k*_p is one pass over U per point instead
    // of the ~250-launch blocked triangular solve
    const int mp = (m + 127) / 128 * 128;
    double* krows = work_dev + (long)mp * ldw;  // K(X*, X) rows; overwritten by the w rows below
    HCK(launch_assemble(h->spec, h->theta_dev, Xnew_dev, m, h->buf.X_dev, h->n, krows, ldw, mp, h->np, 0, 0, h->stream),
        "assemble cross");
    for (int p = 0; p < m; ++p)
      HCK(launch_trmv_upper_t(h->buf.Z_dev, ld, krows + (long)p * ldw, h->n, work_dev + (long)p * ldw, h->stream), "trmv_t");
    const int nk = h->spec.nkern, d = h->spec.d;
    const double* th = h->theta_host;
    double kd = th[nk * d];
    for (int c = 1; c < nk; ++c) kd = (h->spec.op[c - 1] == 0) ? kd + th[nk * d + c] : kd * th[nk * d + c];
    const double sg = std::sqrt(th[nk * d + 2 * nk]);
    HCK(launch_predict_reduce(work_dev, ldw, h->buf.K_dev + (long)h->np * h->buf.lda, h->n, m, kd,
                              pred_noise ? sg * sg : 0.0, mean_dev, var_dev, h->stream), "predict_reduce");
  } else {
    const int r = mi_gp_predict(h, Xnew_dev, m, work_dev, ldw, mean_dev, var_dev, pred_noise);
    if (r != 0) return r;
  }
  const int mp = (m + 127) / 128 * 128;
  double* wrows = work_dev + (long)mp * ldw;
  for (int p = 0; p < m; ++p)
    HCK(launch_trmv_upper(h->buf.Z_dev, ld, work_dev + (long)p * ldw, h->n, wrows + (long)p * ldw, h->stream), "trmv w");
  HCK(launch_predict_grad(h->spec, h->theta_dev, h->buf.X_dev, h->n, Xnew_dev, m, h->alpha_dev, wrows, ldw, dmean_dev,
                          dvar_dev, h->stream), "predict_grad");
  HCK(hipStreamSynchronize(h->stream), "stream sync");
  return 0;
}
