// Internal kernel-launch interface of libmi_gp.so (not part of the C-ABI; see include/mi_gp.h).
#pragma once
#include <hip/hip_runtime.h>

namespace migp {

// Batched evaluation (mi_gp_lml_batch / mi_gp_lml_grad_batch): nb covariances of the SAME inputs, one theta each, factorised
// in lockstep -- every launch carries blockIdx.z = problem, so the latency-bound panel chain of one problem runs beside the
// chains of the others.  Strides are in elements; nb = 1 with zero strides is the ordinary single evaluation.
struct Batch {
  int nb = 1;
  long sK = 0, sZ = 0, sW = 0;  // between consecutive problems' K / Z (U = L^-T) / W (K^-1) matrices
  long sdinv = 0;               // ... leaf inverses (ntc * MINV_ELEMS doubles)
  long salpha = 0;              // ... alpha vectors (np doubles)
  long spart = 0;               // ... gradient partial sums
  int stheta = 0;               // ... theta vectors (device copy, pinned host source, pinned gradient)
  int sinfo = 0;                // ... bad-pivot words (ints)
  int sout = 0;                 // ... scalar records in pinned host memory (doubles)
};

// ---------------------------------------------------------------- gemm_f64.hip
// C[mt*128 x nt*128] = beta*C + alpha * op(A) * op(B), all fp64 row-major with leading dimensions.
struct GemmParams {
  const double* A;
  const double* B;
  double* C;
  long lda, ldb, ldc;
  long strideA, strideB, strideC;  // batch strides in elements (blockIdx.z)
  // two-level batch (node-batched launches of the triangular inverse for nb problems): blockIdx.z = z1 + batch1 * z2, level-2
  // strides apply to z2; batch1 = 0: one level
  int batch1 = 0;
  long strideA2 = 0, strideB2 = 0, strideC2 = 0;
  int mt, nt;                      // output tiles (128 x 128) in rows / columns
  int k;                           // contraction length, multiple of 32
  int tri;                         // 1: only tiles with tj <= min(ti, nt-1)  (SYRK / trapezoid)
  int kmode;                       // 0 full k; 1 k >= tj*128; 2 k < (ti+1)*128; 3 k >= ti*128; 4 k < (tj+1)*128
  double alpha, beta;
  // launcher options (per call; a handle keeps its own values, mi_gp_set_option 7 / 14)
  int small_below = 1024;           // launches with fewer 128x128 tiles than this run on the 64x64-tile kernel
  int band = 8;                     // band height (tile rows) of the band-column-major order of uniform-k trapezoid launches, 0 = row-major
  int hiprio = 0;                   // 64x64-tile kernel only: raise the waves' issue priority (launches on the panel chain)
  int one_per_cu = 0;               // request > half a CU's LDS so that one workgroup per CU runs (leaves room for
                                    // the panel chain's leaf / strip kernels next to a bulk update)
  int tail_small = 1;               // uniform-k 128x128-tile launches: the tiles beyond the last full round of 512 go to the
                                    // 64x64-tile kernel in a second launch (mi_gp_set_option 9)
  // 128x128-tile launches only: restrict the launch to tiles [tile0, tile0 + tile_cnt) of the enumeration (tile_cnt 0: all).
  // The Cholesky driver splits a bulk update into a part that runs one workgroup per CU beside the panel chain and a part
  // that runs two per CU after it (mi_gp_set_option 18); which kernel computes which tile does not depend on the split.
  int tile0 = 0, tile_cnt = 0;
  // uniform-k trapezoid launches on the 128x128-tile kernel: enumerate the tiles of the first `fc` tile columns first (row by
  // row, like a narrow trapezoid), then the rest in the banded order.  With a sub-range launch over a prefix this lets the
  // driver finish the NEXT super-panel's columns ahead of the rest of a trailing update inside ONE enumeration.
  int fc = 0;
  // k-segmented operands (NT form, kmode 0 only): A and B are stored as k-segments of `kseg` columns (a multiple of 128),
  // segment g of an operand at base + g * kseg_stride doubles, rows lda / ldb apart inside a segment.  The sharded driver's
  // panel buffer is laid out this way (one contiguous piece per tile column, broadcast as soon as it is final).  0: one segment.
  int kseg = 0;
  long kseg_stride = 0;
  // k-segmented UPDATE (kmode 0, beta = 1): every kflush columns of k (a multiple of 16; the driver uses 128) the tile takes the
  // partial sum, C = beta C + alpha acc rounded there, and the accumulators restart -- the bits of one launch per segment in one
  // launch, with C read and written once.  Such launches run on the 64x64-tile kernel whatever their size.  0: off.
  int kflush = 0;
  // set by the launcher for that second launch: 64x64 tile 4 e + quadrant belongs to 128x128 tile sub_base + e of the
  // parent enumeration (sub_mt x sub_nt tiles of 128); -1: off
  int sub_base = -1, sub_mt = 0, sub_nt = 0;
  // The factorisation's trapezoids end in the y^T tile row: rows np .. np + 127 hold y^T in row np and zeros below, so the LOWER
  // 64-row half of that tile row is all zeros before and after every update.  1: the 64x64-tile kernel's workgroups of that half
  // return at once (2-8 % of the workgroups of an update at N = 2048 .. 8192; the 128x128-tile kernel computes whole tiles)
  int dead_last_half = 0;
  // Panel-list mode (pl != nullptr; sharded driver, api_shard.hip): ONE launch updates the lower trapezoids of pl_n column
  // panels of a block-cyclically distributed matrix, C_e -= P[rows >= g_e] P[rows of panel e]^T for e = pl_first ...
  // pl[e] = {cum, g, ccol, w}: 128x128 tiles of the panels before e, global tile row of panel e's diagonal block, tile
  // column of its first column in C (local storage), width in tile columns; pl[count].cum closes the table.  C's rows are
  // global rows; A = B = the row-panel buffer whose row 0 is global tile row pl_abase; pl_rows = global tile rows (with the
  // y block).  NT form, kmode 0, alpha/beta as given.  mt / nt are ignored.
  const int4* pl = nullptr;
  int pl_first = 0, pl_n = 0, pl_abase = 0, pl_rows = 0;
  int pl_tiles = 0;  // pl[pl_first + pl_n].cum - pl[pl_first].cum (the host knows the table)
};
// opX_kmajor = 0: operand stored [x][k] (A row-major m x k / B stored n x k, i.e. "B^T");
// opX_kmajor = 1: operand stored [k][x].
// part: 0 the whole product; 1 / 2 only the first / second launch of a split one (per-launch event timing); see gemm_tail_tiles
hipError_t launch_gemm_f64(const GemmParams& p, int opA_kmajor, int opB_kmajor, int batch, hipStream_t stream, int part = 0);
int gemm_tail_tiles(const GemmParams& p, int batch);  // 128x128 tiles that the second launch of a split product takes (0: not split)
hipError_t gemm_f64_enable_lds();
bool gemm_uses_small_tiles(const GemmParams& p, int batch);  // true: the 64x64-tile kernel will run

// ---------------------------------------------------------------- leaf_f64.hip
hipError_t leaf_enable_lds();
constexpr int MINV_ELEMS = 128 * 128;  // doubles per leaf inverse
// in-place lower Cholesky of one 128x128 diagonal block; minv receives M = L^-1 (128 x 128 lower triangular, 16x16 tiles in the strip kernel's operand order,
// zeros above the diagonal inside the diagonal 16x16 tiles; the tiles above the block diagonal are not written and
// never read); *info gets atomicMin(col0 + j + 1) on a bad pivot.
// yrow (optional): row 0 of the 128-row block right below Ablk, solved in place against the leaf's inverse (beta = y M^T)
// wait_ptr (optional): the launch ends only once *wait_ptr >= wait_val (a cross-stream signal; see leaf_f64.hip)
hipError_t launch_potrf_leaf128(double* Ablk, long lda, double* minv, int col0, int* info, hipStream_t stream,
                                double* yrow = nullptr, const Batch* bt = nullptr, const unsigned* wait_ptr = nullptr,
                                unsigned wait_val = 0, int poll_log2 = 22, unsigned* start_wr = nullptr);
// (start_wr, optional: raised to wait_val by the first workgroup as it starts)
// one lane: *wr = val (if wr), then wait for *wt >= val (if wt); a poll that gives up (after 2^poll_log2 sleeps) puts
// SIGNAL_TIMEOUT_INFO into the nb bad-pivot words info[p * sinfo]
constexpr int SIGNAL_TIMEOUT_INFO = -99;
hipError_t launch_signal_write_wait(unsigned* wr, const unsigned* wt, unsigned val, int* info, hipStream_t stream, int nb = 1,
                                    int sinfo = 0, int poll_log2 = 22);
// element (row, col) of a 128 x 128 block kept as 16x16 tiles in the strip kernel's MFMA operand order (leaf_f64.hip: the leaf's
// inverse, the strips' operand-order copies of their first rows)
__device__ __forceinline__ int minv_index(int row, int col) {
  const int jb = row >> 4, n = row & 15, kb = col >> 4, c = col & 15;
  return (jb * 8 + kb) * 256 + (c & 2) * 64 + ((c >> 2) * 16 + n) * 2 + (c & 1);
}

// ---------------------------------------------------------------- thin_f64.hip
// C[ti, tj] -= P[ti] P[tj]^T over the lower trapezoid of mt x nt tiles (tj <= ti), k = 128 (nt <= 2) or k = 256 (nt = 1):
// 16-row x 64-column workgroups, for the panel chain's short updates.  wr (optional): workgroup 0 raises *wr to val
// lsw: the rows of P that are the B operand (tile rows 0 .. nt - 1) in operand order, as the strip in front of the update wrote
// them (launch_trsm_strip128's lsw); k = 256: the B operand's first 128 k from lsw, the other 128 from lsw2 (two strips' copies)
hipError_t launch_syrk_thin(const double* P, double* C, long ld, int mt, int nt, int k, hipStream_t stream, const Batch* bt,
                            unsigned* wr, unsigned val, const double* lsw, const double* lsw2 = nullptr);

// ---------------------------------------------------------------- leaf_f64.hip (continued)
// X * L^T = B in place on the m x 128 panel B (m multiple of 16, ldb even) as X = B * M^T with the leaf's inverse M.
// lsw (optional): the first lsw_blocks 16-row groups of X are also written there in MFMA operand order (128 rows per 16384
// doubles; problem z of a batch at lsw + z * bt->sdinv) -- the B operand of launch_syrk_thin
hipError_t launch_trsm_strip128(const double* minv, double* B, long ldb, int m, hipStream_t stream, const Batch* bt = nullptr,
                                long sB2 = 0, double* lsw = nullptr, int lsw_blocks = 0);
// batched form: pair b uses minv + b * MINV_ELEMS and B + b * strideB (m rows each)
// bt (optional): a second batch level over problems (blockIdx.z): minv + z * bt->sdinv, B + z * sB2
hipError_t launch_trsm_strip128_batched(const double* minv, double* B, long ldb, long strideB, int m, int batch,
                                        hipStream_t stream, const Batch* bt = nullptr, long sB2 = 0, double* lsw = nullptr,
                                        int lsw_blocks = 0);

// ---------------------------------------------------------------- assemble.hip
enum { KID_RBF = 0, KID_MATERN52 = 1, KID_MATERN32 = 2, KID_EXPONENTIAL = 3, KID_RATQUAD = 4 };
constexpr int MAX_KERN = 8;  // the gradient kernels are specialised for 1..4 components and take 5..8 through one
                             // instantiation with a run-time component count (arrays in scratch: slower, same arithmetic)
struct KernSpec {
  int nkern;
  int d;
  int kid[MAX_KERN];
  int op[MAX_KERN];  // op[i] joins component i and i+1: 0 '+', 1 '*'
};
// sym=1: lower 64x64 tiles of K(X1,X1) + noise on the diagonal, identity in the padding;
// sym=0: full K(X1,X2), zeros in the padding.  noise_form: 0 marginal, 1 conditional, 2 explicit.
hipError_t launch_assemble(const KernSpec& spec, const double* theta, const double* X1, int n1, const double* X2,
                           int n2, double* K, long ldk, int rows_pad, int cols_pad, int sym, int noise_form,
                           hipStream_t stream, int diag_shift = -2147483647 - 1, const double* extra_diag = nullptr,
                           const Batch* bt = nullptr);
// diag_shift (sym=0 only): local element (i, j) is on the global diagonal when i + diag_shift == j
// (rectangular blocks of a distributed covariance); the default means "no diagonal" (cross-covariance).
// info (optional): reset to 0x7f7f7f7f ("no bad pivot") by the same launch
hipError_t launch_set_yrows(double* K, long ldk, int row0, int cols_pad, const double* y, int n, hipStream_t stream,
                            int* info = nullptr, const double* theta_src = nullptr, double* theta_dst = nullptr, int ntheta = 0,
                            const Batch* bt = nullptr);
// info (optional): its first word is forwarded as out[3]
// part / sync (optional, zeroed once by the owner): 2 * LML_REDUCE_BLOCKS doubles and one unsigned PER PROBLEM -- with them the
// reduction runs on LML_REDUCE_BLOCKS workgroups (fixed slice order), without them on one
constexpr int LML_REDUCE_BLOCKS = 16;
hipError_t launch_lml_reduce(const double* L, long ld, const double* beta, int n, double* out, hipStream_t stream,
                             const int* info = nullptr, const Batch* bt = nullptr, double* part = nullptr,
                             unsigned* sync = nullptr, double seq = 0.0);

// ---------------------------------------------------------------- grad_predict.hip
hipError_t launch_set_identity_blocks(double* U, long ld, int nblocks, hipStream_t stream, const Batch* bt = nullptr);
// bt: U + z * sZ, beta + z * sK (beta is a row of the factor's matrix), alpha + z * salpha
hipError_t launch_trmv_upper(const double* U, long ld, const double* beta, int n, double* alpha, hipStream_t stream,
                             const Batch* bt = nullptr);
// out = U^T x (= L^-1 x for U = L^-T)
hipError_t launch_trmv_upper_t(const double* U, long ld, const double* x, int n, double* out, hipStream_t stream);
int grad_contract_blocks(int n);
// part: [grad_contract_blocks(n)][ntheta] scratch; grad: [ntheta] (natural parameters, C-ABI order)
hipError_t launch_grad_contract(const KernSpec& spec, const double* theta, const double* X, int n, const double* W,
                                long ldw, const double* alpha, double* part, double* grad, hipStream_t stream,
                                const Batch* bt = nullptr, unsigned* done = nullptr, double* flag = nullptr, double seq = 0.0);
// done / flag / seq (and launch_lml_reduce's seq): the evaluation's LAST kernel publishes its sequence number in pinned host
// memory -- lml_reduce in out[4], the gradient's final reduction in flag[0] (per problem of a batch: + sout) -- see wait_evaluation()
// column slab [col0, col0+cols) of the lower triangle (distributed K^-1): W points at element (row0, col0), row0 <= col0
int grad_contract_slab_blocks(int n, int col0, int cols);
hipError_t launch_grad_contract_slab(const KernSpec& spec, const double* theta, const double* X, int n, const double* W,
                                     long ldw, int row0, int col0, int cols, const double* alpha, double* part,
                                     double* grad, hipStream_t stream);
// gx: [n][d] dLML/dX from Kinv (lower triangle in W) and alpha; d <= 128
int grad_x_splits(int n, int d);  // column splits; scratch of [splits][n][d] doubles is needed when > 1
hipError_t launch_grad_x(const KernSpec& spec, const double* theta, const double* X, int n, const double* W, long ldw,
                         const double* alpha, double* gx, double* scratch, hipStream_t stream);
// dmean/dvar: [m][d] gradients of the conditional at m points; w: row p = K^-1 k(X, x*_p) (leading dimension ldw)
hipError_t launch_predict_grad(const KernSpec& spec, const double* theta, const double* X, int n, const double* xstar,
                               int m, const double* alpha, const double* w, long ldw, double* dmean, double* dvar,
                               hipStream_t stream);
hipError_t launch_predict_reduce(const double* A, long lda, const double* beta, int n, int m, double kdiag,
                                 double noise, double* mean, double* var, hipStream_t stream);

// ---------------------------------------------------------------- api_blocks.hip
// text behind mi_gp_last_global_error() (calls that have no handle to carry it: mi_gp_create, the block-level entries)
void set_global_error(const char* text);
// leaf + strip + in-panel updates of the w_tiles leading tile columns of a (row_tiles x w_tiles)-tile lower trapezoid
// (the body of mi_gp_chol_panel)
hipError_t chol_panel_blocks(double* A, long lda, int row_tiles, int w_tiles, double* dinv, int* info, int col_base,
                             hipStream_t st);
int ensure_kernel_attributes();  // per-device dynamic-LDS limits of the GEMM / leaf kernels; 0 or a C-ABI error code

}  // namespace migp
