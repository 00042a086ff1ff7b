// Probe: v_mfma_f64_16x16x4_f64 issue rate + operand/result lane maps on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f64_probe.hip -o tools/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template<int NACC>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, double seed) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0,0,0,0};
  double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// layout: one wave. D = A(16x4) * B(4x16). Host supplies A,B row-major; kernel uses the
// guide's claimed maps: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D[row=(l>>4)+4r][col=l&15]
__global__ void layout_kernel(const double* A, const double* B, double* D) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template<int NACC>
int run_rate(int blocks, int threads, int iters) {
  double* out; CK(hipMalloc(&out, sizeof(double) * blocks * threads));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate_kernel<NACC><<<blocks, threads>>>(out, 10, 1.0);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    rate_kernel<NACC><<<blocks, threads>>>(out, iters, 1.0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double waves = (double)blocks * threads / 64;
  double flops = waves * iters * NACC * 2048.0;
  printf("NACC=%2d blocks=%5d threads=%4d iters=%d: %.3f ms  %.2f TFLOP/s\n", NACC, blocks, threads, iters, best, flops / best * 1e-9);
  CK(hipFree(out));
  return 0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  // layout check with asymmetric integer data
  std::vector<double> A(64), B(64), D(256), R(256, 0.0);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i * 7 + k * 3;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 2 + k * 11 + j * 5 + (j * j) % 7;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
  double *dA, *dB, *dD; CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
  CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(dA, dB, dD); CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
  int bad = 0; for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
  printf("layout check: %d mismatches of 256 (0 = guide's f64 maps are right)\n", bad);
  int cus = p.multiProcessorCount;
  run_rate<16>(cus, 256, 4000);      // 1 wave / SIMD
  run_rate<16>(cus * 2, 256, 4000);  // 2 waves / SIMD
  run_rate<4>(cus, 256, 16000);
  run_rate<1>(cus, 256, 64000);      // dependent chain
  run_rate<2>(cus, 256, 32000);
  run_rate<16>(cus * 8, 256, 4000);
  return 0;
}
