// C-ABI building-block entry points (declared in include/mi_gp.h, "block-level operations").
#include <mutex>
#include "migp_kernels.h"
#include "../../include/mi_gp.h"

using namespace migp;

static thread_local char g_err[256] = "";
static int fail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return -2;
}
extern "C" const char* mi_gp_last_global_error(void) { return g_err; }
void migp::set_global_error(const char* text) { snprintf(g_err, sizeof(g_err), "%s", text); }

// kernel attributes (dynamic LDS limits) are per device: set them once for each device a block-level call runs on
static std::mutex g_init_mutex;
static unsigned long long g_init_devices = 0;
static int ensure_init() {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(e, "hipGetDevice");
  std::lock_guard<std::mutex> lock(g_init_mutex);
  if (dev < 64 && (g_init_devices >> dev) & 1ull) return 0;
  e = gemm_f64_enable_lds();
  if (e == hipSuccess) e = leaf_enable_lds();
  if (e != hipSuccess) return fail(e, "enable_lds");
  if (dev < 64) g_init_devices |= 1ull << dev;
  return 0;
}

extern "C" int mi_gp_gemm_f64(int transa, int transb, int m, int n, int k, double alpha, const double* A, long lda,
                              const double* B, long ldb, double beta, double* C, long ldc, int tri, int kmode,
                              int batch, long strideA, long strideB, long strideC, void* stream) {
  if (m % 128 || n % 128 || k % 32 || m <= 0 || n <= 0 || k < 0 || (lda & 1) || (ldb & 1)) {
    snprintf(g_err, sizeof(g_err), "mi_gp_gemm_f64: m,n must be multiples of 128, k of 32, lda/ldb even");
    return -1;
  }
  if (int r = ensure_init()) return r;
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.strideA = strideA; p.strideB = strideB; p.strideC = strideC;
  p.mt = m / 128; p.nt = n / 128; p.k = k; p.tri = tri; p.kmode = kmode; p.alpha = alpha; p.beta = beta;
  // op(A) = A (m x k row-major) -> [x][k]; op(A) = A^T with A stored k x m -> [k][x]
  // op(B) = B (k x n row-major) -> [k][x]; op(B) = B^T with B stored n x k -> [x][k]
  hipError_t e = launch_gemm_f64(p, transa ? 1 : 0, transb ? 0 : 1, batch < 1 ? 1 : batch, (hipStream_t)stream);
  if (e != hipSuccess) return fail(e, "launch_gemm_f64");
  return 0;
}

// The same product with the launcher's per-call knobs exposed (a handle carries them as options 7 / 9 / 14): which tile
// size serves the launch, whether a 128x128-tile launch finishes its last partial round on 64x64 tiles, and the band
// height of the trapezoid tile order.  For A/B measurements of single launches (tools/bench_gemm_shapes.py) and tests.
extern "C" int mi_gp_gemm_f64_tuned(int transa, int transb, int m, int n, int k, double alpha, const double* A, long lda,
                                    const double* B, long ldb, double beta, double* C, long ldc, int tri, int kmode,
                                    int small_below, int tail_small, int band, int one_per_cu, void* stream) {
  if (m % 128 || n % 128 || k % 32 || m <= 0 || n <= 0 || k < 0 || (lda & 1) || (ldb & 1)) {
    snprintf(g_err, sizeof(g_err), "mi_gp_gemm_f64_tuned: m,n must be multiples of 128, k of 32, lda/ldb even");
    return -1;
  }
  if (int r = ensure_init()) return r;
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.strideA = p.strideB = p.strideC = 0;
  p.mt = m / 128; p.nt = n / 128; p.k = k; p.tri = tri; p.kmode = kmode; p.alpha = alpha; p.beta = beta;
  p.small_below = small_below; p.tail_small = tail_small ? 1 : 0; p.band = band; p.one_per_cu = one_per_cu ? 1 : 0;
  hipError_t e = launch_gemm_f64(p, transa ? 1 : 0, transb ? 0 : 1, 1, (hipStream_t)stream);
  if (e != hipSuccess) return fail(e, "launch_gemm_f64");
  return 0;
}

// C = beta*C + alpha*A*B^T with A (m rows) and B (n rows) stored as k-segments (GemmParams::kseg): the operand form of the
// sharded driver's piece-major panel buffers
extern "C" int mi_gp_gemm_nt_kseg(int m, int n, int k, double alpha, const double* A, long lda, const double* B, long ldb,
                                  int kseg, long kseg_stride, double beta, double* C, long ldc, int tri, int small_below,
                                  void* stream) {
  if (m % 128 || n % 128 || m <= 0 || n <= 0 || kseg <= 0 || kseg % 128 || k <= 0 || k % kseg || (lda & 1) || (ldb & 1) ||
      (kseg_stride & 1)) {
    snprintf(g_err, sizeof(g_err), "mi_gp_gemm_nt_kseg: m, n, kseg multiples of 128, k a multiple of kseg, even strides");
    return -1;
  }
  if (int r = ensure_init()) return r;
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.strideA = p.strideB = p.strideC = 0;
  p.mt = m / 128; p.nt = n / 128; p.k = k; p.tri = tri; p.kmode = 0; p.alpha = alpha; p.beta = beta;
  p.kseg = kseg; p.kseg_stride = kseg_stride;
  if (small_below >= 0) p.small_below = small_below;
  hipError_t e = launch_gemm_f64(p, 0, 0, 1, (hipStream_t)stream);
  if (e != hipSuccess) return fail(e, "launch_gemm_f64");
  return 0;
}

// ---------------------------------------------------------------- distributed-matrix building blocks
static KernSpec make_spec(int d, int nkern, const int* kernel_ids, const int* ops) {
  KernSpec s;
  s.nkern = nkern;
  s.d = d;
  for (int i = 0; i < MAX_KERN; ++i) {
    s.kid[i] = i < nkern ? kernel_ids[i] : 0;
    s.op[i] = i < nkern ? ops[i] : 0;
  }
  return s;
}

extern "C" int mi_gp_assemble_block(int d, int nkern, const int* kernel_ids, const int* ops, const double* theta_dev,
                                    const double* Xrows_dev, int nrows, const double* Xcols_dev, int ncols,
                                    int row0, int col0, double* K_dev, long ldk, int rows_pad, int cols_pad,
                                    int noise_form, void* stream) {
  if (d <= 0 || nkern <= 0 || nkern > MAX_KERN || rows_pad % 64 || cols_pad % 64 || nrows < 0 || ncols < 0 || (ldk & 1)) {
    snprintf(g_err, sizeof(g_err), "mi_gp_assemble_block: bad argument (pads multiples of 64, ldk even)");
    return -1;
  }
  const KernSpec spec = make_spec(d, nkern, kernel_ids, ops);
  hipError_t e = launch_assemble(spec, theta_dev, Xrows_dev, nrows, Xcols_dev, ncols, K_dev, ldk, rows_pad, cols_pad, 0,
                                 noise_form, (hipStream_t)stream, row0 - col0);
  if (e != hipSuccess) return fail(e, "assemble_block");
  return 0;
}

// factor the w leading tile columns of a (ntr x w)-tile lower trapezoid stored at A (leading dimension
// lda): leaf + strip + in-panel updates, everything on `stream`
static hipError_t panel_rec(double* A, long lda, int ntr, int c0, int w, double* dinv, int* info, int col_base,
                            hipStream_t st) {
  hipError_t e;
  if (w == 1) {
    double* blk = A + (long)c0 * 128 * lda + (long)c0 * 128;
    e = launch_potrf_leaf128(blk, lda, dinv + (size_t)c0 * MINV_ELEMS, col_base + c0 * 128, info, st);
    if (e != hipSuccess) return e;
    return launch_trsm_strip128(dinv + (size_t)c0 * MINV_ELEMS, blk + 128 * lda, lda, (ntr - c0 - 1) * 128, st);
  }
  const int w1 = w / 2, w2 = w - w1;
  e = panel_rec(A, lda, ntr, c0, w1, dinv, info, col_base, st);
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
  return panel_rec(A, lda, ntr, c0 + w1, w2, dinv, info, col_base, st);
}

hipError_t migp::chol_panel_blocks(double* A, long lda, int row_tiles, int w_tiles, double* dinv, int* info, int col_base,
                                   hipStream_t st) {
  return panel_rec(A, lda, row_tiles, 0, w_tiles, dinv, info, col_base, st);
}
int migp::ensure_kernel_attributes() { return ensure_init(); }

extern "C" int mi_gp_chol_panel(double* A_dev, long lda, int row_tiles, int w_tiles, double* dinv_dev, int* info_dev,
                                int col_base, void* stream) {
  if (!A_dev || !dinv_dev || !info_dev || w_tiles <= 0 || row_tiles < w_tiles || (lda & 1)) {
    snprintf(g_err, sizeof(g_err), "mi_gp_chol_panel: bad argument");
    return -1;
  }
  if (int r = ensure_init()) return r;
  hipError_t e = panel_rec(A_dev, lda, row_tiles, 0, w_tiles, dinv_dev, info_dev, col_base, (hipStream_t)stream);
  if (e != hipSuccess) return fail(e, "chol_panel");
  return 0;
}

// out[1] += sum log L_ii over n diagonal entries, out[2] += sum beta_i^2 (out[0] is scratch)
extern "C" int mi_gp_lml_partial(const double* L_dev, long ld, const double* beta_dev, int n, double* out_dev,
                                 void* stream) {
  hipError_t e = launch_lml_reduce(L_dev, ld, beta_dev, n, out_dev, (hipStream_t)stream);
  if (e != hipSuccess) return fail(e, "lml_partial");
  return 0;
}

// ---------------------------------------------------------------- sharded gradient building blocks (third row)
// X L^T = B in place for tile columns [c0, c0 + w) of the factor; B points at the first of those columns
static hipError_t trsm_block_rec(const double* L, long ldl, const double* dinv, double* B, long ldb, int m, int cbase,
                                 int c0, int w, hipStream_t st) {
  if (w == 1)
    return launch_trsm_strip128(dinv + (size_t)c0 * MINV_ELEMS, B + (long)(c0 - cbase) * 128, ldb, m, st);
  const int w1 = w / 2, w2 = w - w1;
  hipError_t e = trsm_block_rec(L, ldl, dinv, B, ldb, m, cbase, c0, w1, st);
  if (e != hipSuccess) return e;
  // B[:, c0+w1 : c0+w) -= X[:, c0 : c0+w1) * L[c0+w1 : c0+w, c0 : c0+w1)^T
  GemmParams p;
  p.A = B + (long)(c0 - cbase) * 128;
  p.B = L + (long)(c0 + w1) * 128 * ldl + (long)c0 * 128;
  p.C = B + (long)(c0 + w1 - cbase) * 128;
  p.lda = ldb; p.ldb = ldl; p.ldc = ldb;
  p.strideA = p.strideB = p.strideC = 0;
  p.mt = m / 128; p.nt = w2; p.k = w1 * 128; p.tri = 0; p.kmode = 0; p.alpha = -1.0; p.beta = 1.0;
  e = launch_gemm_f64(p, 0, 0, 1, st);
  if (e != hipSuccess) return e;
  return trsm_block_rec(L, ldl, dinv, B, ldb, m, cbase, c0 + w1, w2, st);
}

extern "C" int mi_gp_trsm_block(const double* L_dev, long ldl, const double* dinv_dev, int c0_tiles, int w_tiles,
                                double* B_dev, long ldb, int m, void* stream) {
  if (!L_dev || !dinv_dev || !B_dev || c0_tiles < 0 || w_tiles <= 0 || m <= 0 || m % 128 || (ldl & 1) || (ldb & 1)) {
    snprintf(g_err, sizeof(g_err), "mi_gp_trsm_block: bad argument (m multiple of 128, even leading dimensions)");
    return -1;
  }
  if (int r = ensure_init()) return r;
  hipError_t e = trsm_block_rec(L_dev, ldl, dinv_dev, B_dev, ldb, m, c0_tiles, c0_tiles, w_tiles, (hipStream_t)stream);
  if (e != hipSuccess) return fail(e, "trsm_block");
  return 0;
}

extern "C" int mi_gp_trmv_upper(const double* U_dev, long ld, const double* x_dev, int n, double* out_dev, void* stream) {
  if (!U_dev || !x_dev || !out_dev || n <= 0) {
    snprintf(g_err, sizeof(g_err), "mi_gp_trmv_upper: bad argument");
    return -1;
  }
  hipError_t e = launch_trmv_upper(U_dev, ld, x_dev, n, out_dev, (hipStream_t)stream);
  if (e != hipSuccess) return fail(e, "trmv_upper");
  return 0;
}

extern "C" long mi_gp_grad_contract_block_scratch(int n, int col0, int cols, int ntheta) {
  return (long)grad_contract_slab_blocks(n, col0, cols) * ntheta;
}

extern "C" int mi_gp_grad_contract_block(int d, int nkern, const int* kernel_ids, const int* ops, const double* theta_dev,
                                         const double* X_dev, int n, const double* W_dev, long ldw, int row0, int col0,
                                         int cols, const double* alpha_dev, double* part_dev, long part_len,
                                         double* grad_dev, void* stream) {
  if (d <= 0 || nkern <= 0 || nkern > MAX_KERN || !theta_dev || !X_dev || !W_dev || !alpha_dev || !part_dev || !grad_dev ||
      n <= 0 || row0 < 0 || row0 > col0 || col0 % 64 || row0 % 64 || cols <= 0 || cols % 64) {
    snprintf(g_err, sizeof(g_err), "mi_gp_grad_contract_block: bad argument (row0 <= col0, multiples of 64)");
    return -1;
  }
  const int ntheta = nkern * d + 2 * nkern + 2;
  if (part_len < (long)grad_contract_slab_blocks(n, col0, cols) * ntheta) {
    snprintf(g_err, sizeof(g_err), "mi_gp_grad_contract_block: scratch shorter than mi_gp_grad_contract_block_scratch()");
    return -1;
  }
  const KernSpec spec = make_spec(d, nkern, kernel_ids, ops);
  hipError_t e = launch_grad_contract_slab(spec, theta_dev, X_dev, n, W_dev, ldw, row0, col0, cols, alpha_dev, part_dev,
                                           grad_dev, (hipStream_t)stream);
  if (e != hipSuccess) return fail(e, "grad_contract_block");
  return 0;
}
