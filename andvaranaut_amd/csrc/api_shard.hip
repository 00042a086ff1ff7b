// Sharded factorisation, one C call per panel step (include/mi_gp.h "sharded factorisation"; SURVEY.md section 8e,
// second row; BASELINE config 4).  1-D block-cyclic column panels over `world` ranks; the exchange (one broadcast per
// panel) stays with the caller (torch.distributed = RCCL over xGMI), everything else of a step is enqueued here:
// the next panel's update + factorisation + staging on a side stream, and ONE panel-list GEMM launch for all owned
// trailing panels on the main stream (round 2 issued one launch per owned panel from Python: 16 launches per step at 8
// ranks, each below the 1024-tile threshold of the 128x128-tile kernel).
// Round 4: a panel buffer is PIECE-major -- one contiguous piece per tile column (its rows x 128 doubles, then that
// column's 128 x 128 leaf inverse) -- and a piece is staged as soon as its column is final (behind its strip), with an
// event per piece, so that the caller can broadcast tile column c while columns c + 1 .. are still being factored.  The
// GEMM kernels read such a buffer through their k-segmented operand form (GemmParams::kseg).
#include <cstdio>
#include <vector>
#include "migp_kernels.h"
#include "../../include/mi_gp.h"

using namespace migp;

namespace {

typedef double double2_t __attribute__((ext_vector_type(2)));

// one piece of a panel buffer in ONE launch: rows x 128 doubles of a tile column (row stride lds) -> dst (row stride 128), then the
// column's 128 x 128 leaf inverse behind them (a separate hipMemcpyAsync cost a second boundary on the owner's chain)
__global__ void copy_piece_kernel(double* __restrict__ dst, const double* __restrict__ src, long lds, int rows,
                                  const double* __restrict__ dinv) {
  const long body = (long)rows * 64, total = body + MINV_ELEMS / 2;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    if (e < body) {
      const int r = (int)(e >> 6), c = (int)(e & 63);
      reinterpret_cast<double2_t*>(dst + (long)r * 128)[c] = reinterpret_cast<const double2_t*>(src + (long)r * lds)[c];
    } else {
      reinterpret_cast<double2_t*>(dst + (long)rows * 128)[e - body] = reinterpret_cast<const double2_t*>(dinv)[e - body];
    }
  }
}

// out[1] = sum log L_ii, out[2] = sum beta_i^2 over the owned panels: one workgroup, fixed order (bit-reproducible).
// tab[e] = {cum, g, ccol, w}; panel e's diagonal entry i sits at K[(g*128 + i) * ldk + ccol*128 + i], beta at row np.
__global__ __launch_bounds__(256) void shard_reduce_kernel(const double* __restrict__ K, long ldk, const int4* __restrict__ tab,
                                                           int nown, int n, int np, double* __restrict__ out) {
  __shared__ double s1[256], s2[256];
  double a = 0.0, b = 0.0;
  for (int e = 0; e < nown; ++e) {
    const int4 d = tab[e];
    const int c0 = d.y * 128, nv = min(n - c0, d.w * 128);
    for (int i = threadIdx.x; i < nv; i += 256) {
      a += log(K[(long)(c0 + i) * ldk + d.z * 128 + i]);
      const double bv = K[(long)np * ldk + d.z * 128 + i];
      b += bv * bv;
    }
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      s1[threadIdx.x] += s1[threadIdx.x + w];
      s2[threadIdx.x] += s2[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[1] = s1[0];
    out[2] = s2[0];
  }
}

}  // namespace

struct mi_gp_shard {
  mi_gp_shard_config cfg;
  KernSpec spec;
  int np, ntc, ntr, pwt, pw, npan, nown;
  std::vector<int> own;      // global panel index of the li-th owned panel
  std::vector<int4> table;   // panel-list table of the GEMM kernel (+ closing entry)
  int4* table_dev = nullptr;
  double* dinv_dev = nullptr;  // pwt leaf inverses of the panel being factored
  hipEvent_t ev_ready = nullptr, ev_staged = nullptr, ev_side_done = nullptr, ev_bulk = nullptr;
  std::vector<hipEvent_t> ev_piece;  // recorded behind the staging of tile column c of the panel being factored
  bool staged_pending = false, ready_valid = false, bulk_valid = false;
  int bulk_one_per_cu = 1;
  int split_tiles = 2048;  // option 4: tiles of a bulk update that run one workgroup per CU beside this rank's chain (0: all)
  int early_next = 1;  // option 2: update the panel this rank factors next step first and alone (see mi_gp_shard_step)
  int rest_from = -1;  // first column first: the panel whose update of tile columns 1.. of the panel being factored is still due
  int pipelined = 2;   // option 5: a chain on the main stream stages each tile column behind its strip (0: behind the whole panel);
                       // 2: and the previous panel's update takes tile column 0 first, so that it is final -- and leaves -- early
  int chain_on_main = 0;  // option 3: the owner chain runs on the main stream ahead of the bulk update (default: world > 1)
  int prof = 0;
  std::vector<hipEvent_t> pev;  // per step: side e0..e3 (before update, after update, after factor, after stage), main b0, b1
  std::vector<unsigned char> pmask;
  std::vector<hipEvent_t> ppiece;  // per step and tile column: behind the staging of that piece (option 1)
  int prof_step = -1;              // the step whose chain is being enqueued (row npan: panel 0 in begin)
  char err[256] = "";
};

static thread_local char g_shard_err[256] = "";

static int sfail(mi_gp_shard* s, hipError_t e, const char* where) {
  snprintf(s->err, sizeof(s->err), "%s: %s", where, hipGetErrorString(e));
  return -2;
}
#define SCK(call, where)                                  \
  do {                                                    \
    hipError_t e__ = (call);                              \
    if (e__ != hipSuccess) return sfail(s, e__, where);   \
  } while (0)

extern "C" const char* mi_gp_shard_last_error(mi_gp_shard* s) { return s ? s->err : g_shard_err; }

static int panel_w(const mi_gp_shard* s, int j) { return std::min(s->pwt, s->ntc - j * s->pwt); }

extern "C" int mi_gp_shard_destroy(mi_gp_shard* s) {
  if (!s) return 0;
  (void)hipSetDevice(s->cfg.device);
  (void)hipDeviceSynchronize();
  (void)hipFree(s->table_dev);
  (void)hipFree(s->dinv_dev);
  if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
  if (s->ev_staged) (void)hipEventDestroy(s->ev_staged);
  if (s->ev_side_done) (void)hipEventDestroy(s->ev_side_done);
  if (s->ev_bulk) (void)hipEventDestroy(s->ev_bulk);
  for (auto& e : s->ev_piece) if (e) (void)hipEventDestroy(e);
  for (auto& e : s->pev) if (e) (void)hipEventDestroy(e);
  for (auto& e : s->ppiece) if (e) (void)hipEventDestroy(e);
  delete s;
  return 0;
}

extern "C" int mi_gp_shard_create(const mi_gp_shard_config* cfg, mi_gp_shard** out) {
  if (!cfg || !out) { snprintf(g_shard_err, sizeof(g_shard_err), "mi_gp_shard_create: null argument"); return -1; }
  if (cfg->n <= 0 || cfg->d <= 0 || cfg->nkern <= 0 || cfg->nkern > MAX_KERN || cfg->panel_tiles <= 0 || cfg->world <= 0 ||
      cfg->rank < 0 || cfg->rank >= cfg->world || !cfg->X_dev || !cfg->y_dev || !cfg->K_dev || !cfg->P_dev[0] || !cfg->P_dev[1] ||
      !cfg->theta_dev || !cfg->info_dev || !cfg->out_dev || (cfg->ldk & 1) || (cfg->ldp & 1)) {
    snprintf(g_shard_err, sizeof(g_shard_err), "mi_gp_shard_create: bad argument (positive sizes, rank < world, even leading dimensions, no null buffers)");
    return -1;
  }
  mi_gp_shard* s = new mi_gp_shard();
  s->cfg = *cfg;
  s->spec.nkern = cfg->nkern;
  s->spec.d = cfg->d;
  for (int i = 0; i < MAX_KERN; ++i) {
    s->spec.kid[i] = i < cfg->nkern ? cfg->kernel_ids[i] : 0;
    s->spec.op[i] = i < cfg->nkern ? cfg->ops[i] : 0;
  }
  s->np = (cfg->n + 127) / 128 * 128;
  s->ntc = s->np / 128;
  s->ntr = s->ntc + 1;
  s->pwt = cfg->panel_tiles;
  s->pw = s->pwt * 128;
  s->npan = (s->ntc + s->pwt - 1) / s->pwt;
  for (int j = cfg->rank; j < s->npan; j += cfg->world) s->own.push_back(j);
  s->nown = (int)s->own.size();
  s->chain_on_main = cfg->world > 1 ? 1 : 0;
  if (cfg->ldk < (long)std::max(s->nown, 1) * s->pw || cfg->ldp < (long)(s->np + 256) * 128) {
    snprintf(g_shard_err, sizeof(g_shard_err), "mi_gp_shard_create: ldk must hold the %d owned panels of %d columns, ldp one piece ((np + 256) * 128 doubles)",
             s->nown, s->pw);
    delete s;
    return -1;
  }
  int cum = 0;
  for (int li = 0; li < s->nown; ++li) {
    const int j = s->own[li], w = panel_w(s, j), g = j * s->pwt;
    s->table.push_back(make_int4(cum, g, li * s->pwt, w));
    cum += w * (w + 1) / 2 + (s->ntr - g - w) * w;
  }
  s->table.push_back(make_int4(cum, 0, 0, 0));
  hipError_t e = hipSetDevice(cfg->device);
  if (e == hipSuccess) e = hipMalloc(&s->table_dev, sizeof(int4) * s->table.size());
  if (e == hipSuccess) e = hipMemcpy(s->table_dev, s->table.data(), sizeof(int4) * s->table.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&s->dinv_dev, sizeof(double) * MINV_ELEMS * (size_t)s->pwt);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_staged, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_side_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_bulk, hipEventDisableTiming);
  s->ev_piece.assign(s->pwt, nullptr);
  for (int c = 0; c < s->pwt && e == hipSuccess; ++c) e = hipEventCreateWithFlags(&s->ev_piece[c], hipEventDisableTiming);
  if (e == hipSuccess && ensure_kernel_attributes() != 0) e = hipErrorUnknown;
  if (e != hipSuccess) {
    snprintf(g_shard_err, sizeof(g_shard_err), "mi_gp_shard_create: %s", hipGetErrorString(e));
    mi_gp_shard_destroy(s);
    return -2;
  }
  *out = s;
  return 0;
}

extern "C" int mi_gp_shard_chain_stream(const mi_gp_shard* s) { return s ? s->chain_on_main : -1; }

extern "C" int mi_gp_shard_set_option(mi_gp_shard* s, int what, int value) {
  if (!s) return -1;
  if (what == 0) s->bulk_one_per_cu = value ? 1 : 0;
  else if (what == 1) s->prof = value ? 1 : 0;
  else if (what == 2) s->early_next = value ? 1 : 0;
  else if (what == 3) s->chain_on_main = value ? 1 : 0;
  else if (what == 4) s->split_tiles = value;
  else if (what == 5) s->pipelined = value < 0 ? 0 : value > 2 ? 2 : value;
  else { snprintf(s->err, sizeof(s->err), "mi_gp_shard_set_option: unknown option %d", what); return -1; }
  return 0;
}

// ---------------------------------------------------------------- pieces
static hipError_t prof_mark(mi_gp_shard* s, int step, int slot, hipStream_t st) {
  if (!s->prof) return hipSuccess;
  const size_t need = (size_t)(s->npan + 1) * 6;
  if (s->pev.size() < need) { s->pev.resize(need, nullptr); s->pmask.assign(s->npan + 1, 0); }
  hipEvent_t& ev = s->pev[(size_t)step * 6 + slot];
  if (!ev) {
    hipError_t e = hipEventCreate(&ev);
    if (e != hipSuccess) return e;
  }
  s->pmask[step] |= (unsigned char)(1u << slot);
  return hipEventRecord(ev, st);
}

// tile column c of panel j (owned) of K -> piece c of buf: rows r0 .. np + 127 at row stride 128, then the column's leaf inverse
static hipError_t stage_piece(mi_gp_shard* s, int j, int li, int c, double* buf, hipStream_t st) {
  const int r0 = j * s->pw, rows = s->np + 128 - r0;
  const double* src = s->cfg.K_dev + (long)r0 * s->cfg.ldk + (long)li * s->pw + (long)c * 128;
  double* dst = buf + (long)c * s->cfg.ldp;
  const long total = (long)rows * 64 + MINV_ELEMS / 2;
  int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  copy_piece_kernel<<<blocks, 256, 0, st>>>(dst, src, s->cfg.ldk, rows, s->dinv_dev + (size_t)c * MINV_ELEMS);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (s->prof && s->prof_step >= 0) {
    const size_t need = (size_t)(s->npan + 1) * s->pwt;
    if (s->ppiece.size() < need) s->ppiece.resize(need, nullptr);
    hipEvent_t& pe = s->ppiece[(size_t)s->prof_step * s->pwt + c];
    if (!pe) {
      e = hipEventCreate(&pe);
      if (e != hipSuccess) return e;
    }
    e = hipEventRecord(pe, st);
    if (e != hipSuccess) return e;
  }
  return hipEventRecord(s->ev_piece[c], st);
}

// owned panel jt (local index li) -= P_j[rows >= jt] P_j[rows of jt]^T with panel j in buf; cn > 0: only tile columns
// [cb, cb + cn) of the panel (their trapezoid starts cb tile rows further down)
static hipError_t update_panel(mi_gp_shard* s, int jt, int li, int j, const double* buf, hipStream_t st, int cb = 0, int cn = 0) {
  const int wt = cn > 0 ? cn : panel_w(s, jt), wj = panel_w(s, j), rt = jt * s->pw + cb * 128;
  GemmParams p;
  p.A = buf + (long)(rt - j * s->pw) * 128;  // piece-major buffer: rows 128 doubles apart, tile column c at c * ldp
  p.B = p.A;
  p.C = s->cfg.K_dev + (long)rt * s->cfg.ldk + (long)li * s->pw + (long)cb * 128;
  p.lda = p.ldb = 128;
  p.kseg = 128;
  p.kseg_stride = s->cfg.ldp;
  p.ldc = s->cfg.ldk;
  p.strideA = p.strideB = p.strideC = 0;
  p.mt = (s->np + 128 - rt) / 128;
  p.nt = wt;
  p.k = wj * 128;
  p.tri = 1;
  p.kmode = 0;
  p.alpha = -1.0;
  p.beta = 1.0;
  p.hiprio = 1;
  return launch_gemm_f64(p, 0, 0, 1, st);
}

// factor tile columns [c0, c0 + w) of owned panel j (recursive halving, as migp::chol_panel_blocks) and stage every
// column behind its strip
static hipError_t factor_stage_rec(mi_gp_shard* s, int j, int li, double* A, int ntr, int c0, int w, double* buf, hipStream_t st) {
  const long lda = s->cfg.ldk;
  hipError_t e;
  if (w == 1) {
    double* blk = A + (long)c0 * 128 * lda + (long)c0 * 128;
    double* dinv = s->dinv_dev + (size_t)c0 * MINV_ELEMS;
    e = launch_potrf_leaf128(blk, lda, dinv, j * s->pw + c0 * 128, s->cfg.info_dev, st);
    if (e != hipSuccess) return e;
    e = launch_trsm_strip128(dinv, blk + 128 * lda, lda, (ntr - c0 - 1) * 128, st);
    if (e != hipSuccess) return e;
    e = stage_piece(s, j, li, c0, buf, st);
    if (e == hipSuccess && c0 == 0 && s->rest_from >= 0) {
      // first column first: the previous panel's update of tile columns 1.. was held back so that column 0 could be
      // factored, staged and SENT before it
      const int jp = s->rest_from;
      s->rest_from = -1;
      if (panel_w(s, j) > 1) e = update_panel(s, j, li, jp, s->cfg.P_dev[jp & 1], st, 1, panel_w(s, j) - 1);
    }
    return e;
  }
  const int w1 = w / 2, w2 = w - w1;
  e = factor_stage_rec(s, j, li, A, ntr, c0, w1, buf, st);
  if (e != hipSuccess) return e;
  GemmParams p;
  p.A = A + (long)(c0 + w1) * 128 * lda + (long)c0 * 128;
  p.B = p.A;
  p.C = A + (long)(c0 + w1) * 128 * lda + (long)(c0 + w1) * 128;
  p.lda = p.ldb = p.ldc = lda;
  p.strideA = p.strideB = p.strideC = 0;
  p.mt = ntr - c0 - w1; p.nt = w2; p.k = w1 * 128; p.tri = 1; p.kmode = 0; p.alpha = -1.0; p.beta = 1.0;
  e = launch_gemm_f64(p, 0, 0, 1, st);
  if (e != hipSuccess) return e;
  return factor_stage_rec(s, j, li, A, ntr, c0 + w1, w2, buf, st);
}

// Factor owned panel j and stage it into buf.  before_stage == nullptr: PIPELINED, every tile column is staged (and its
// event recorded) right behind its strip.  Otherwise the buffer may still be the operand of main-stream updates that run
// beside this chain: the whole panel is factored first, then the stream waits for `before_stage` and stages the pieces.
static hipError_t factor_stage_panel(mi_gp_shard* s, int j, int li, double* buf, bool pipelined, hipEvent_t before_stage,
                                     hipStream_t st) {
  const int w = panel_w(s, j), r0 = j * s->pw;
  double* A = s->cfg.K_dev + (long)r0 * s->cfg.ldk + (long)li * s->pw;
  s->prof_step = j == 0 ? s->npan : j - 1;  // the step that produces panel j
  if (pipelined) return factor_stage_rec(s, j, li, A, (s->np + 128 - r0) / 128, 0, w, buf, st);
  hipError_t e = chol_panel_blocks(A, s->cfg.ldk, (s->np + 128 - r0) / 128, w, s->dinv_dev, s->cfg.info_dev, r0, st);
  if (e == hipSuccess && before_stage) e = hipStreamWaitEvent(st, before_stage, 0);
  for (int c = 0; c < w && e == hipSuccess; ++c) e = stage_piece(s, j, li, c, buf, st);
  return e;
}

// every owned panel with local index >= li0 in ONE launch (panel-list mode)
static hipError_t update_bulk(mi_gp_shard* s, int li0, int j, const double* buf, int one_per_cu, hipStream_t st) {
  if (li0 >= s->nown) return hipSuccess;
  GemmParams p;
  p.A = buf;
  p.B = buf;
  p.C = s->cfg.K_dev;
  p.lda = p.ldb = 128;
  p.kseg = 128;
  p.kseg_stride = s->cfg.ldp;
  p.ldc = s->cfg.ldk;
  p.strideA = p.strideB = p.strideC = 0;
  p.mt = p.nt = 0;
  p.k = panel_w(s, j) * 128;
  p.tri = 1;
  p.kmode = 0;
  p.alpha = -1.0;
  p.beta = 1.0;
  p.one_per_cu = one_per_cu;
  p.pl = s->table_dev;
  p.pl_first = li0;
  p.pl_n = s->nown - li0;
  p.pl_abase = j * s->pwt;
  p.pl_rows = s->ntr;
  p.pl_tiles = s->table[s->nown].x - s->table[li0].x;
  // beside this rank's own chain only the first split_tiles tiles run one workgroup per CU, the rest two per CU once the
  // chain is through (the single-GPU driver's split, api_gp.hip cholesky(); same tiles, same kernels)
  if (one_per_cu && s->split_tiles > 0 && p.pl_tiles >= s->split_tiles + 1024) {
    p.tile_cnt = s->split_tiles;
    hipError_t e = launch_gemm_f64(p, 0, 0, 1, st);
    if (e != hipSuccess) return e;
    p.one_per_cu = 0;
    p.tile0 = s->split_tiles;
    p.tile_cnt = p.pl_tiles;
  }
  return launch_gemm_f64(p, 0, 0, 1, st);
}

// ---------------------------------------------------------------- evaluation
extern "C" int mi_gp_shard_begin(mi_gp_shard* s, int noise_form, void* main_stream, void* side_stream) {
  if (!s) return -1;
  (void)side_stream;
  hipStream_t M = (hipStream_t)main_stream;
  SCK(hipSetDevice(s->cfg.device), "hipSetDevice");
  s->staged_pending = s->ready_valid = s->bulk_valid = false;
  s->rest_from = -1;
  if (s->prof) s->pmask.assign(s->npan + 1, 0);
  const int n = s->cfg.n, d = s->cfg.d;
  bool info_reset = false;
  for (int li = 0; li < s->nown; ++li) {
    const int j = s->own[li], w = panel_w(s, j), r0 = j * s->pw;
    const int nrows = std::max(0, n - r0), ncols = std::max(0, std::min(n - r0, w * 128));
    double* blk = s->cfg.K_dev + (long)r0 * s->cfg.ldk + (long)li * s->pw;
    SCK(launch_assemble(s->spec, s->cfg.theta_dev, s->cfg.X_dev + (long)r0 * d, nrows, s->cfg.X_dev + (long)r0 * d, ncols, blk,
                        s->cfg.ldk, s->np - r0, w * 128, 0, noise_form, M, 0), "assemble");
    SCK(launch_set_yrows(s->cfg.K_dev + (long)li * s->pw, s->cfg.ldk, s->np, w * 128, s->cfg.y_dev + r0, ncols, M,
                         info_reset ? nullptr : s->cfg.info_dev), "set_yrows");
    info_reset = true;
  }
  if (!info_reset) SCK(hipMemsetAsync(s->cfg.info_dev, 0x7f, sizeof(int), M), "info reset");
  if (s->cfg.rank == 0 && s->nown > 0) {  // owner of panel 0
    SCK(prof_mark(s, s->npan, 1, M), "event");
    SCK(factor_stage_panel(s, 0, 0, s->cfg.P_dev[0], true, nullptr, M), "factor + stage panel 0");
    SCK(prof_mark(s, s->npan, 2, M), "event");
    SCK(prof_mark(s, s->npan, 3, M), "event");
  }
  return 0;
}

// Step j.  Streams and events:
//   side  : [waits ev_ready] update panel jn with panel j -> factor it -> [waits ev_bulk] stage it -> ev_staged
//   main  : [waits ev_staged if panel j came from this rank's side stream]
//           update the panel this rank factors NEXT step (jn + 1, if owned) with panel j FIRST and alone -> ev_ready,
//           so that next step's chain does not wait for the whole bulk update (the single-GPU driver's (a1) hand-over);
//           then ONE panel-list launch for every other owned panel > jn -> ev_bulk (its reads of P[j % 2] are done: the
//           next step's staging overwrites that buffer's partner only after them)
// The caller makes BOTH streams wait for the arrival of panel j before this call (work.wait() under each stream).
extern "C" int mi_gp_shard_step(mi_gp_shard* s, int j, void* main_stream, void* side_stream) {
  if (!s || j < 0 || j >= s->npan) return -1;
  hipStream_t M = (hipStream_t)main_stream, S = (hipStream_t)side_stream;
  SCK(hipSetDevice(s->cfg.device), "hipSetDevice");
  const double* buf = s->cfg.P_dev[j & 1];
  if (s->staged_pending) {  // panel j was staged by this rank's side stream
    SCK(hipStreamWaitEvent(M, s->ev_staged, 0), "wait staged");
    s->staged_pending = false;
  }
  const int jn = j + 1, world = s->cfg.world, rank = s->cfg.rank;
  bool chain = false;
  if (jn < s->npan && jn % world == rank && s->chain_on_main) {
    // option 3: the chain runs on the main stream AHEAD of this rank's bulk update -- alone on the chip it is 2-3x
    // shorter than beside a bulk update (fp64 VALU and MFMA share the DP pipe), and on several ranks the chain, not this
    // rank's bulk update, is what every other rank waits for
    const int li = (jn - rank) / world;
    SCK(prof_mark(s, j, 0, M), "event");
    if (s->pipelined == 2 && panel_w(s, jn) > 1) {
      SCK(update_panel(s, jn, li, j, buf, M, 0, 1), "update next panel, first tile column");
      s->rest_from = j;  // the other columns follow behind column 0's leaf, strip and staging (factor_stage_rec)
    } else {
      SCK(update_panel(s, jn, li, j, buf, M), "update next panel");
    }
    SCK(prof_mark(s, j, 1, M), "event");
    SCK(factor_stage_panel(s, jn, li, s->cfg.P_dev[jn & 1], s->pipelined != 0, nullptr, M), "factor + stage next panel");
    SCK(prof_mark(s, j, 2, M), "event");
    SCK(prof_mark(s, j, 3, M), "event");
    // the caller broadcasts the pieces under side_stream, each behind mi_gp_shard_wait_piece: ordered behind that piece's
    // staging, NOT behind the later columns' factorisation or the bulk update that follows on the main stream
  } else if (jn < s->npan && jn % world == rank) {
    const int li = (jn - rank) / world;
    // the previous step recorded ev_ready right behind its update of panel jn; without one (first step) everything
    // queued on the main stream so far precedes the side stream's work
    if (!s->ready_valid) SCK(hipEventRecord(s->ev_ready, M), "record ready");
    SCK(hipStreamWaitEvent(S, s->ev_ready, 0), "wait ready");
    SCK(prof_mark(s, j, 0, S), "event");
    SCK(update_panel(s, jn, li, j, buf, S), "update next panel");
    SCK(prof_mark(s, j, 1, S), "event");
    // P[jn % 2] was the operand of the previous step's main-stream updates, which run beside this chain: staged at the end
    SCK(factor_stage_panel(s, jn, li, s->cfg.P_dev[jn & 1], false, s->bulk_valid ? s->ev_bulk : nullptr, S), "factor + stage next panel");
    SCK(prof_mark(s, j, 2, S), "event");
    SCK(prof_mark(s, j, 3, S), "event");
    SCK(hipEventRecord(s->ev_staged, S), "record staged");
    s->staged_pending = true;
    chain = true;
  }
  s->ready_valid = false;
  s->bulk_valid = false;
  // owned panels > jn: local indices from li0 on
  int li0 = 0;
  while (li0 < s->nown && s->own[li0] <= jn) ++li0;
  if (li0 < s->nown) {
    SCK(prof_mark(s, j, 4, M), "event");
    if (s->early_next && !s->chain_on_main && s->own[li0] == jn + 1) {  // this rank factors panel jn + 1 in the next step
      SCK(update_panel(s, jn + 1, li0, j, buf, M), "update the panel after next");
      SCK(hipEventRecord(s->ev_ready, M), "record ready");
      s->ready_valid = true;
      ++li0;
    }
    SCK(update_bulk(s, li0, j, buf, (chain && s->bulk_one_per_cu) ? 1 : 0, M), "bulk update");
    SCK(prof_mark(s, j, 5, M), "event");
    SCK(hipEventRecord(s->ev_bulk, M), "record bulk");
    s->bulk_valid = true;
  }
  return 0;
}

extern "C" int mi_gp_shard_wait_piece(mi_gp_shard* s, int c, void* stream) {
  if (!s || c < 0 || c >= s->pwt) return -1;
  SCK(hipSetDevice(s->cfg.device), "hipSetDevice");
  SCK(hipStreamWaitEvent((hipStream_t)stream, s->ev_piece[c], 0), "wait piece");
  return 0;
}

extern "C" int mi_gp_shard_finish(mi_gp_shard* s, void* main_stream, void* side_stream) {
  if (!s) return -1;
  hipStream_t M = (hipStream_t)main_stream, S = (hipStream_t)side_stream;
  SCK(hipSetDevice(s->cfg.device), "hipSetDevice");
  SCK(hipEventRecord(s->ev_side_done, S), "record side");
  SCK(hipStreamWaitEvent(M, s->ev_side_done, 0), "wait side");
  s->staged_pending = false;
  shard_reduce_kernel<<<1, 256, 0, M>>>(s->cfg.K_dev, s->cfg.ldk, s->table_dev, s->nown, s->cfg.n, s->np, s->cfg.out_dev);
  SCK(hipGetLastError(), "shard_reduce");
  return 0;
}

// out[step * panel_tiles + c] = ms from the start of step `step`'s chain (its update of the next panel; row npanels:
// the factorisation of panel 0) to the end of the staging of tile column c (option 1; 0 where there was none)
extern "C" int mi_gp_shard_piece_times(mi_gp_shard* s, double* out, int max_steps) {
  if (!s || !out) return -1;
  const int rows = std::min(max_steps, s->npan + 1);
  for (int i = 0; i < s->pwt * rows; ++i) out[i] = 0.0;
  if (s->pev.empty() || s->ppiece.empty()) return 0;
  for (int j = 0; j < rows; ++j) {
    const int slot0 = j == s->npan ? 1 : 0;
    if (!((s->pmask[j] >> slot0) & 1)) continue;
    for (int c = 0; c < s->pwt; ++c) {
      hipEvent_t pe = s->ppiece[(size_t)j * s->pwt + c];
      float ms = 0.f;
      if (pe && hipEventElapsedTime(&ms, s->pev[(size_t)j * 6 + slot0], pe) == hipSuccess && ms > 0.f) out[j * s->pwt + c] = ms;
    }
  }
  return rows;
}

extern "C" int mi_gp_shard_times(mi_gp_shard* s, double* out, int max_steps) {
  if (!s || !out) return -1;
  const int rows = std::min(max_steps, s->npan + 1);
  for (int i = 0; i < 4 * rows; ++i) out[i] = 0.0;
  if (s->pev.empty()) return 0;
  auto el = [&](int step, int a, int b) -> double {
    if (!((s->pmask[step] >> a) & 1) || !((s->pmask[step] >> b) & 1)) return 0.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s->pev[(size_t)step * 6 + a], s->pev[(size_t)step * 6 + b]) != hipSuccess) return 0.0;
    return (double)ms;
  };
  for (int j = 0; j < rows; ++j) {
    out[4 * j + 0] = el(j, 0, 1);
    out[4 * j + 1] = el(j, 1, 2);
    out[4 * j + 2] = el(j, 2, 3);
    out[4 * j + 3] = el(j, 4, 5);
  }
  return rows;
}
