// Internal kernel-launch interface of libmi_gp.so (not part of the C-ABI; see include/mi_gp.h).
#pragma once
#include <hip/hip_runtime.h>

namespace migp {

// ---------------------------------------------------------------- gemm_f64.hip
// C[mt*128 x nt*128] = beta*C + alpha * op(A) * op(B), all fp64 row-major with leading dimensions.
struct GemmParams {
  const double* A;
  const double* B;
  double* C;
  long lda, ldb, ldc;
  long strideA, strideB, strideC;  // batch strides in elements (blockIdx.z)
  int mt, nt;                      // output tiles (128 x 128) in rows / columns
  int k;                           // contraction length, multiple of 16
  int tri;                         // 1: only tiles with tj <= min(ti, nt-1)  (SYRK / trapezoid)
  int kmode;                       // 0 full k; 1 k >= tj*128; 2 k < (ti+1)*128; 3 k >= ti*128
  double alpha, beta;
};
// opX_kmajor = 0: operand stored [x][k] (A row-major m x k / B stored n x k, i.e. "B^T");
// opX_kmajor = 1: operand stored [k][x].
hipError_t launch_gemm_f64(const GemmParams& p, int opA_kmajor, int opB_kmajor, int batch, hipStream_t stream);
hipError_t gemm_f64_enable_lds();

}  // namespace migp
