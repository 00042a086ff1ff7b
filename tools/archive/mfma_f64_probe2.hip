// Probe 5: v_mfma_f64_16x16x4_f64 with DISTINCT A/B operand registers (as a real GEMM issues it),
// vs the same-operand loop of probe 1, with in-kernel cycle counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template<int MODE>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* clk, int iters, const double* src) {
  d4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0,0,0,0};
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[threadIdx.x + 64 * i]; b[i] = src[threadIdx.x + 64 * (i + 4)]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MODE == 0) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc[i][j], 0, 0, 0);
      }
    if (MODE == 0) {
      // perturb operands so the compiler cannot hoist anything (cheap VALU)
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(a[i])); asm volatile("" : "+v"(b[i])); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template<int MODE> int run(const char* name, int blocks, int iters, const double* src) {
  double* out; unsigned long long* clk;
  CK(hipMalloc(&out, 8 * blocks * 256)); CK(hipMalloc(&clk, 8 * blocks));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; ++w) k<MODE><<<blocks, 256>>>(out, clk, iters, src);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(out, clk, iters, src); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); }
  std::vector<unsigned long long> h(blocks); CK(hipMemcpy(h.data(), clk, 8 * blocks, hipMemcpyDeviceToHost));
  double flops = blocks * 4.0 * iters * 16.0 * 2048.0;
  printf("%-28s blocks=%4d: %.3f ms %.2f TFLOP/s cycles/mfma=%.1f\n", name, blocks, best, flops / best * 1e-9, (double)h[0] / (iters * 16.0));
  return 0;
}
int main() {
  std::vector<double> h(512); for (int i = 0; i < 512; ++i) h[i] = 0.001 * (i % 97) - 0.03;
  double* src; CK(hipMalloc(&src, 4096)); CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
  run<0>("16x16x4 distinct operands", 256, 4000, src);
  run<0>("16x16x4 distinct operands", 512, 4000, src);
  run<1>("16x16x4 same operands", 256, 4000, src);
  return 0;
}
