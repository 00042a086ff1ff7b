// Test infrastructure (tests/test_gpu_leaf_protocol.py): the CHECKED flavour of potrf_leaf128_kernel -- the library's own source
// compiled with -DLEAF_CHECKED, in which every meeting point of the leaf's eight waves verifies that the LDS arrival counters hold
// exactly what the protocol allows there (csrc/leaf_f64.hip, namespace roles) and reports a mismatch through the info word as
// LEAF_PROTOCOL_INFO - site.  Built into tools/libleafcheck.so by __graft_entry__.build(); not part of libmi_gp.so.
#define LEAF_CHECKED
#include "../andvaranaut_amd/csrc/leaf_f64.hip"

using namespace migp;

// nb blocks of 128 x 128 (lower triangle used), lda doubles per row, block z at A_host + z * 128 * lda; with_yrow: row 128 of each
// block's storage holds a right-hand side that the leaf solves in place (beta = y M^T).  Outputs: the factor in place of A,
// M = L^-1 row-major (nb x 128 x 128), info per block.  Returns 0 or a HIP error code.
extern "C" __attribute__((visibility("default"))) int leaf_check_run(double* A_host, long lda, int nb, int with_yrow, double* M_host,
                                                                     int* info_host, int reps) {
  const long rows = 128 + (with_yrow ? 128 : 0);
  const size_t abytes = sizeof(double) * (size_t)nb * rows * lda;
  double *dA = nullptr, *dA0 = nullptr, *dM = nullptr;
  int* dinfo = nullptr;
  hipError_t e = hipMalloc(&dA, abytes);
  if (e == hipSuccess) e = hipMalloc(&dA0, abytes);
  if (e == hipSuccess) e = hipMalloc(&dM, sizeof(double) * (size_t)nb * MINV_ELEMS);
  if (e == hipSuccess) e = hipMalloc(&dinfo, sizeof(int) * 4 * nb);
  if (e == hipSuccess) e = hipMemcpy(dA0, A_host, abytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = leaf_enable_lds();
  Batch bt;
  bt.nb = nb;
  bt.sK = rows * lda;
  bt.sdinv = MINV_ELEMS;
  bt.sinfo = 4;
  for (int r = 0; r < reps && e == hipSuccess; ++r) {  // (repeats: the protocol check is on every launch; info accumulates by atomicMin)
    e = hipMemcpy(dA, dA0, abytes, hipMemcpyDeviceToDevice);
    if (e == hipSuccess && r == 0) e = hipMemset(dinfo, 0x7f, sizeof(int) * 4 * nb);
    if (e == hipSuccess && r == 0) e = hipMemset(dM, 0, sizeof(double) * (size_t)nb * MINV_ELEMS);
    if (e == hipSuccess) e = launch_potrf_leaf128(dA, lda, dM, 0, dinfo, 0, with_yrow ? dA + 128 * lda : nullptr, nb > 1 ? &bt : nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(A_host, dA, abytes, hipMemcpyDeviceToHost);
  double* Mt = new double[(size_t)nb * MINV_ELEMS];
  int* inf = new int[4 * nb];
  if (e == hipSuccess) e = hipMemcpy(Mt, dM, sizeof(double) * (size_t)nb * MINV_ELEMS, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(inf, dinfo, sizeof(int) * 4 * nb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) {
    for (int z = 0; z < nb; ++z) {
      info_host[z] = inf[4 * z];
      double* M = M_host + (size_t)z * 128 * 128;
      for (int i = 0; i < 128 * 128; ++i) M[i] = 0.0;
      // the leaf writes M as 16x16 tiles in the strip's operand order (minv_index); tiles above the block diagonal are not written
      for (int row = 0; row < 128; ++row)
        for (int col = 0; col < (row / 16 + 1) * 16; ++col) {
          const int jb = row >> 4, n = row & 15, kb = col >> 4, c = col & 15;
          M[row * 128 + col] = Mt[(size_t)z * MINV_ELEMS + (jb * 8 + kb) * 256 + (c & 2) * 64 + ((c >> 2) * 16 + n) * 2 + (c & 1)];
        }
    }
  }
  delete[] Mt;
  delete[] inf;
  (void)hipFree(dA); (void)hipFree(dA0); (void)hipFree(dM); (void)hipFree(dinfo);
  return (int)e;
}

extern "C" __attribute__((visibility("default"))) int leaf_check_protocol_info(void) { return LEAF_PROTOCOL_INFO; }
