// Accuracy of the hardware fp64 reciprocal / rsqrt seeds (decides how many Newton steps the leaf needs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
__global__ void k(const double* x, double* r, double* s, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { r[i] = __builtin_amdgcn_rcp(x[i]); s[i] = __builtin_amdgcn_rsq(x[i]); }
}
int main() {
  const int n = 1 << 20;
  double* hx = (double*)malloc(8 * n); double* hr = (double*)malloc(8 * n); double* hs = (double*)malloc(8 * n);
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = ldexp(1.0 + rand() / (double)RAND_MAX, (rand() % 40) - 20);
  double *x, *r, *s; hipMalloc(&x, 8 * n); hipMalloc(&r, 8 * n); hipMalloc(&s, 8 * n);
  hipMemcpy(x, hx, 8 * n, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(x, r, s, n); hipDeviceSynchronize();
  hipMemcpy(hr, r, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(hs, s, 8 * n, hipMemcpyDeviceToHost);
  double er = 0, es = 0;
  for (int i = 0; i < n; ++i) {
    er = fmax(er, fabs(hr[i] * hx[i] - 1.0));
    es = fmax(es, fabs(hs[i] * hs[i] * hx[i] - 1.0) * 0.5);
  }
  printf("v_rcp_f64 max rel err %.3e (2^%.1f)   v_rsq_f64 max rel err %.3e (2^%.1f)\n", er, log2(er), es, log2(es));
  return 0;
}
