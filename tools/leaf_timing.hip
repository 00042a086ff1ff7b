// Dev harness: phase breakdown of potrf_leaf128 (s_memtime stamps) and of trsm_strip128.
#ifndef NO_STAMPS
#define LEAF_STAMPS
#endif
#include "../andvaranaut_amd/csrc/leaf_f64.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
#include <random>
using namespace migp;
#ifdef LEAF_STAMPS
#define STAMPS_RESET() hipMemcpyToSymbol(HIP_SYMBOL(g_leaf_stamps), z, sizeof(z))
#define STAMPS_READ() hipMemcpyFromSymbol(st, HIP_SYMBOL(g_leaf_stamps), sizeof(st))
#else
#define STAMPS_RESET() (void)z
#define STAMPS_READ() memset(st, 0, sizeof(st))
#endif
int main() {
  const int n = 128; const long lda = 144;
  std::vector<double> A(n * lda, 0.0), G(n * n);
  std::mt19937 rng(1); std::normal_distribution<double> nd;
  for (auto& v : G) v = nd(rng);
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k]; A[i * lda + j] = s / n + (i == j ? 1.0 : 0.0); }
  std::vector<double> A0 = A;
  double *dA, *dinv; int* info; hipMalloc(&dA, A.size() * 8); hipMalloc(&dinv, 16384 * 8); hipMemset(dinv, 0, 16384 * 8); hipMalloc(&info, 16);
  leaf_enable_lds();
  unsigned long long z[32] = {0};
  float best = 1e9;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 20; ++rep) {
    hipMemcpy(dA, A0.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemset(info, 0x7f, 16);
    STAMPS_RESET();
    hipEventRecord(e0); launch_potrf_leaf128(dA, lda, dinv, 0, info, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
  }
  unsigned long long st[32]; STAMPS_READ();
  printf("leaf kernel %.1f us\n", best * 1e3);
  printf("wave 0 cycles (8 blocks): load panel %llu | elimination %llu | scale+write+flag %llu | next diagonal tile %llu | wait urgent %llu | final barrier %llu\n", st[0], st[24]+st[25]+st[26]+st[27]+st[28]+st[29]+st[30]+st[31], st[2], st[3], st[4], st[5]);
  printf("elimination per block:"); for (int i = 24; i < 32; ++i) printf(" %llu", st[i]); printf("\n");
  printf("wave 1 cycles: initial load %llu | wait ready %llu | urgent %llu | lazy %llu | stream out %llu\n", st[8], st[9], st[10], st[11], st[12]);
  printf("inverse (thread 0): dinv compute %llu | dinv write+sync %llu | stage A %llu | stage B %llu | rest %llu\n", st[16], st[17], st[18], st[19], st[20]);
  hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
  // check L L^T = A0
  double err = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += A[i * lda + k] * A[j * lda + k]; err = std::max(err, std::fabs(s - A0[i * lda + j])); }
  printf("max |L L^T - A| = %.3e\n", err);
  {  // per 16x16 block error against a host Cholesky
    std::vector<double> R = A0;
    for (int j = 0; j < n; ++j) {
      double d = R[j * lda + j]; for (int k = 0; k < j; ++k) d -= R[j * lda + k] * R[j * lda + k];
      d = std::sqrt(d); R[j * lda + j] = d;
      for (int i = j + 1; i < n; ++i) { double v = R[i * lda + j]; for (int k = 0; k < j; ++k) v -= R[i * lda + k] * R[j * lda + k]; R[i * lda + j] = v / d; }
    }
    printf("block errors vs host Cholesky (row block down, column block across):\n");
    for (int rb = 0; rb < 8; ++rb) { for (int cb = 0; cb <= rb; ++cb) { double e = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int r = 16 * rb + i, c = 16 * cb + j; if (c <= r) { double dd = std::fabs(A[r * lda + c] - R[r * lda + c]); if (!(dd <= 1e300)) dd = 1e300; e = std::max(e, dd); } } printf(" %8.1e", e); } printf("\n"); }
  }
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
