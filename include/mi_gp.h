/* mi_gp.h -- C-ABI of libmi_gp.so, the MI355X (gfx950) GP log-marginal-likelihood backend.
 *
 * The reference (andrew-angus/andvaranaut) has no FFI for this path: the seam is the PyMC model
 * built in GPMCMC.__fit (gpmcmc.py:189-323) and consumed by pm.find_MAP / pm.sample / gp.predict
 * (gpmcmc.py:345,351,593).  Each entry point below names the reference call site it replaces.
 * All pointers marked _dev are device (HBM) addresses, e.g. torch.Tensor.data_ptr(); the library
 * borrows them and never frees them.  Every function returns int:
 *   0  ok;  >0  LAPACK-style info (1-based index of the first non-positive pivot);
 *   <0 bad argument (-1) or HIP/RCCL failure (-2); text via mi_gp_last_global_error().
 */
#ifndef MI_GP_H
#define MI_GP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* mi_gp_last_global_error(void);

/* ---- block-level operations (also used by the multi-GPU driver and the parity tests) ---- */

/* C = beta*C + alpha*op(A)*op(B) in fp64 on v_mfma_f64_4x4x4_4b_f64; row-major, m,n multiples of
 * 128, k multiple of 16.  transa=0: A is m x k; 1: A is k x m.  transb=0: B is k x n; 1: B is n x k.
 * tri=1 computes only tiles on/below the block diagonal; kmode restricts k per tile for triangular
 * operands (0 full, 1 k>=col-tile start, 2 k<row-tile end, 3 k>=row-tile start).
 * Replaces the OpenBLAS dgemm/dsyrk calls inside LAPACK dpotrf/dtrtri/dlauum that PyTensor's
 * Cholesky Op reaches (gpmcmc.py:313; scipy.linalg.cholesky). */
int mi_gp_gemm_f64(int transa, int transb, int m, int n, int k, double alpha, const double* A_dev, long lda,
                   const double* B_dev, long ldb, double beta, double* C_dev, long ldc, int tri, int kmode,
                   int batch, long strideA, long strideB, long strideC, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
