// Dev harness: correctness of potrf_leaf128 (L L^T = A, M L = I) and its timeline: one s_memtime probe per launch at a
// selected program point (see LEAF_PROBE in leaf_f64.hip), so the probes do not perturb what they measure.
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
int main() {
  const int n = 128; const long lda = 144;
  std::vector<double> A(n * lda, 0.0), G(n * n);
  std::mt19937 rng(1); std::normal_distribution<double> nd;
  for (auto& v : G) v = nd(rng);
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += G[i * n + k] * G[j * n + k]; A[i * lda + j] = s / n + (i == j ? 1.0 : 0.0); }
  std::vector<double> A0 = A;
  double *dA, *dinv; int* info; hipMalloc(&dA, A.size() * 8); hipMalloc(&dinv, 16384 * 8); hipMemset(dinv, 0, 16384 * 8); hipMalloc(&info, 16);
  leaf_enable_lds();
  float best = 1e9;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&]() {
    hipMemcpy(dA, A0.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipMemset(info, 0x7f, 16);
    hipEventRecord(e0); launch_potrf_leaf128(dA, lda, dinv, 0, info, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
  };
  for (int rep = 0; rep < 20; ++rep) run();
  printf("leaf kernel %.1f us (events, best of 20)\n", best * 1e3);
#ifdef LEAF_STAMPS
  auto probe = [&](int code) {
    unsigned long long zero = 0, out = 0, lo = ~0ull;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemcpyToSymbol(HIP_SYMBOL(g_leaf_probe_sel), &code, sizeof(int));
      hipMemcpyToSymbol(HIP_SYMBOL(g_leaf_probe_out), &zero, sizeof(zero));
      run();
      hipMemcpyFromSymbol(&out, HIP_SYMBOL(g_leaf_probe_out), sizeof(out));
      if (out && out < lo) lo = out;
    }
    return lo == ~0ull ? 0ull : lo;
  };
  printf("cycles since the wave's start (s_memtime ticks), one probe per launch, min of 3\n");
  printf("wave 0  jb: iteration start | panel loaded | eliminated | published + own rows written | next diagonal tile | urgent tiles seen\n");
  for (int jb = 0; jb < 8; ++jb) { printf("  %d:", jb); for (int k = 0; k < 6; ++k) printf(" %7llu", probe(8 * jb + k)); printf("\n"); }
  printf("wave 0 end of loop: %llu\n", probe(63));
  printf("wave 1  initial load done: %llu\n", probe(62));
  printf("wave 1 (helper)  jb: ready seen | urgent arrived | lazy done | stream-out issued\n");
  for (int jb = 0; jb < 8; ++jb) { printf("  %d:", jb); for (int k : {0, 1, 2, 4}) printf(" %7llu", probe(64 + 8 * jb + k)); printf("\n"); }
  printf("wave 5 (inverse)  jb: ready seen | block row of M done | T terms done\n");
  for (int jb = 0; jb < 8; ++jb) { printf("  %d:", jb); for (int k = 0; k < 3; ++k) printf(" %7llu", probe(128 + 8 * jb + k)); printf("\n"); }
#endif
  hipMemcpy(A.data(), dA, A.size() * 8, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += A[i * lda + k] * A[j * lda + k]; err = std::max(err, std::fabs(s - A0[i * lda + j])); }
  printf("max |L L^T - A| = %.3e\n", err);
  std::vector<double> M(n * n);
  hipMemcpy(M.data(), dinv, M.size() * 8, hipMemcpyDeviceToHost);
  {  // the leaf writes M in 16x16 tiles in the strip's operand order (leaf_f64.hip: minv_index); back to row-major for the check
    std::vector<double> R(M.size(), 0.0);
    for (int row = 0; row < n; ++row)
      for (int col = 0; col < (row / 16 + 1) * 16; ++col) {
        const int jb = row >> 4, nn = row & 15, kb = col >> 4, c = col & 15;
        R[row * n + col] = M[(jb * 8 + kb) * 256 + (c & 2) * 64 + ((c >> 2) * 16 + nn) * 2 + (c & 1)];
      }
    M = R;
  }
  double ierr = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = j; k <= i; ++k) s += M[i * n + k] * A[k * lda + j]; ierr = std::max(ierr, std::fabs(s - (i == j ? 1.0 : 0.0))); }
  double up = 0;
  for (int i = 0; i < n; ++i) for (int j = i + 1; j < (i / 16 + 1) * 16; ++j) up = std::max(up, std::fabs(M[i * n + j]));
  printf("max |M L - I| = %.3e, max |M| above the diagonal inside diagonal tiles = %.3e\n", ierr, up);
  return 0;
}
