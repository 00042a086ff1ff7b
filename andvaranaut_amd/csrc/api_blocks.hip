// C-ABI building-block entry points (declared in include/mi_gp.h, "block-level operations").
#include "migp_kernels.h"
#include "../../include/mi_gp.h"

using namespace migp;

static thread_local char g_err[256] = "";
static int fail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return -2;
}
extern "C" const char* mi_gp_last_global_error(void) { return g_err; }

static bool g_lds_enabled = false;
static int ensure_init() {
  if (!g_lds_enabled) {
    hipError_t e = gemm_f64_enable_lds();
    if (e != hipSuccess) return fail(e, "gemm_f64_enable_lds");
    g_lds_enabled = true;
  }
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
