// Dev harness: phase breakdown of potrf_leaf128 (s_memtime stamps) and of trsm_strip128.
#define LEAF_STAMPS
#include "../andvaranaut_amd/csrc/leaf_f64.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>
using namespace migp;
int main() {
  const int n = 128; const long lda = 144;
  std::vector<double> A(n * lda, 0.0), G(n * n);
  std::mt19937 rng(1); std::normal_distribution<double> nd;
  for (auto& v : G) v = nd(rng);
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k]; A[i * lda + j] = s / n + (i == j ? 1.0 : 0.0); }
  std::vector<double> A0 = A;
  double *dA, *dinv; int* info; hipMalloc(&dA, A.size() * 8); hipMalloc(&dinv, 16384 * 8); hipMemset(dinv, 0, 16384 * 8); hipMalloc(&info, 16);
  leaf_enable_lds();
  unsigned long long z[16] = {0};
  float best = 1e9;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep) {
    hipMemcpy(dA, A0.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemset(info, 0x7f, 16);
    hipMemcpyToSymbol(HIP_SYMBOL(g_leaf_stamps), z, sizeof(z));
    hipEventRecord(e0); launch_potrf_leaf128(dA, lda, dinv, 0, info, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
  }
  unsigned long long st[16]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_leaf_stamps), sizeof(st));
  printf("inverse (thread 0): dinv compute %llu | dinv write+sync %llu | stage A (3 levels, incl. previous syncs) %llu | stage B %llu | rest -> slot dinv\n", st[11], st[12], st[13], st[14]);
  printf("wave 1 cycles: barrier wait %llu | solve_row %llu | arrive+stream-out+spin %llu | trailing %llu\n", st[15], st[9], st[8], st[10]);
  printf("leaf kernel %.1f us; cycles: load %llu | A(diag) %llu | B(trsm) %llu | C(update) %llu | store %llu | dinv %llu || w0: tile %llu factor %llu\n", best * 1e3, st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
  hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
  // check L L^T = A0
  double err = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += A[i * lda + k] * A[j * lda + k]; err = std::max(err, std::fabs(s - A0[i * lda + j])); }
  printf("max |L L^T - A| = %.3e\n", err);
  // check M L = I (M = the explicit inverse the leaf streams out; tiles above the block diagonal are never written)
  std::vector<double> M(n * n);
  hipMemcpy(M.data(), dinv, M.size() * 8, hipMemcpyDeviceToHost);
  double ierr = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = j; k <= i; ++k) s += M[i * n + k] * A[k * lda + j]; ierr = std::max(ierr, std::fabs(s - (i == j ? 1.0 : 0.0))); }
  double up = 0;
  for (int i = 0; i < n; ++i) for (int j = i + 1; j < (i / 16 + 1) * 16; ++j) up = std::max(up, std::fabs(M[i * n + j]));
  printf("max |M L - I| = %.3e, max |M| above the diagonal inside diagonal tiles = %.3e\n", ierr, up);
  return 0;
}
