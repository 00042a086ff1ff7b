// Accuracy of the device exp / sqrt sequences of csrc/migp_math.h against libm (device) over their argument ranges.
// hipcc --offload-arch=gfx950 -O3 -I andvaranaut_amd/csrc tools/probe_math.hip -o tools/probe_math && ./tools/probe_math
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "migp_math.h"

__global__ void eval(const double* x, double* e1, double* e2, double* s1, double* s2, double* s3, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  e1[i] = migp::exp_nonpos(x[i]);
  e2[i] = exp(x[i]);
  const double t = -x[i] * 1e-3 + 1e-12;
  s1[i] = migp::sqrt_pos(t);
  s2[i] = sqrt(t);
  s3[i] = migp::sqrt_pos(-x[i] * -x[i] * 1e150 + 1e-300);
}

static double ulps(double a, double b) {
  if (a == b) return 0.0;
  long long ia, ib;
  memcpy(&ia, &a, 8);
  memcpy(&ib, &b, 8);
  return (double)llabs(ia - ib);
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    const double u = (double)(st >> 11) / 9007199254740992.0;
    x[i] = (i & 1) ? -745.0 * u : -40.0 * u * u;  // dense near 0, sparse down to the underflow threshold
  }
  x[0] = 0.0; x[1] = -1e-300; x[2] = -745.2; x[3] = -800.0; x[4] = -INFINITY; x[5] = -708.4;
  double *dx, *d[5];
  hipMalloc(&dx, n * 8);
  for (auto& p : d) hipMalloc(&p, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  eval<<<(n + 255) / 256, 256>>>(dx, d[0], d[1], d[2], d[3], d[4], n);
  std::vector<double> h[5];
  for (int k = 0; k < 5; ++k) { h[k].resize(n); hipMemcpy(h[k].data(), d[k], n * 8, hipMemcpyDeviceToHost); }
  double me = 0, ms = 0, mh = 0; int ie = 0, is = 0;
  for (int i = 0; i < n; ++i) {
    const bool sub = h[1][i] != 0.0 && h[1][i] < 2.3e-308;  // subnormal results: compare in absolute units of the smallest subnormal
    const double ue = sub ? fabs(h[0][i] - h[1][i]) / 4.9406564584124654e-324 : ulps(h[0][i], h[1][i]);
    if (ue > me) { me = ue; ie = i; }
    if (!std::isfinite(x[i])) continue;
    const double us = ulps(h[2][i], h[3][i]);
    if (us > ms) { ms = us; is = i; }
    const double hs = ulps(h[2][i], sqrt(-x[i] * 1e-3 + 1e-12));
    if (hs > mh) mh = hs;
  }
  printf("exp_nonpos vs device libm: max %.0f ulp at x = %.17g (%.17g vs %.17g)\n", me, x[ie], h[0][ie], h[1][ie]);
  printf("sqrt_pos vs device libm: max %.0f ulp at t = %.17g; vs host sqrt: %.0f ulp\n", ms, -x[is] * 1e-3 + 1e-12, mh);
  printf("exp_nonpos(-inf) = %g, exp_nonpos(-800) = %g, exp_nonpos(-745.2) = %g (libm %g), exp_nonpos(0) = %.17g\n", h[0][4], h[0][3], h[0][2], h[1][2], h[0][0]);
  return 0;
}
