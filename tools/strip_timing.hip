// Dev harness: trsm_strip128 timing vs rows, and correctness against a host solve.
#define LEAF_STAMPS
#include "../andvaranaut_amd/csrc/leaf_f64.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>
using namespace migp;
int main() {
  const int n = 128; const long lda = 16400;
  std::mt19937 rng(1); std::normal_distribution<double> nd;
  std::vector<double> G(n * n), A(n * n, 0.0);
  for (auto& v : G) v = nd(rng);
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k]; A[i * n + j] = s / n + (i == j ? 1.0 : 0.0); }
  const int mmax = 16384 + 128;
  double *dA, *dinv; int* info; hipMalloc(&dA, (size_t)(mmax + 128) * lda * 8); hipMalloc(&dinv, 2048 * 8); hipMalloc(&info, 16);
  hipMemset(dA, 0, (size_t)(mmax + 128) * lda * 8);
  for (int i = 0; i < n; ++i) hipMemcpy(dA + (long)i * lda, A.data() + i * n, n * 8, hipMemcpyHostToDevice);
  std::vector<double> B((size_t)256 * n);
  for (auto& v : B) v = nd(rng);
  leaf_enable_lds();
  hipMemset(info, 0x7f, 16);
  launch_potrf_leaf128(dA, lda, dinv, 0, info, 0);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int m : {128, 1024, 4096, 16384 + 128}) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      for (int i = 0; i < 256; ++i) hipMemcpy(dA + (long)(128 + i) * lda, B.data() + (size_t)i * n, n * 8, hipMemcpyHostToDevice);
      hipDeviceSynchronize();
      hipEventRecord(e0); launch_trsm_strip128(dinv, dA + 128 * lda, lda, m, 0); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
    }
    printf("strip m=%5d: %.1f us\n", m, best * 1e3);
  }
  // correctness on the first 256 rows of the last run
  std::vector<double> L(n * n), X((size_t)256 * n);
  for (int i = 0; i < n; ++i) hipMemcpy(L.data() + i * n, dA + (long)i * lda, n * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < 256; ++i) hipMemcpy(X.data() + (size_t)i * n, dA + (long)(128 + i) * lda, n * 8, hipMemcpyDeviceToHost);
  double err = 0;
  for (int r = 0; r < 256; ++r) for (int c = 0; c < n; ++c) { double s = 0; for (int k = 0; k <= c; ++k) s += X[(size_t)r * n + k] * L[c * n + k]; err = std::max(err, std::fabs(s - B[(size_t)r * n + c])); }
  printf("max |X L^T - B| = %.3e\n", err);
  return 0;
}
