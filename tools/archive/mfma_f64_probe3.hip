// Probe 6: bare issue rate of both fp64 MFMA shapes with inline asm (accumulators pinned in AGPRs,
// so hipcc cannot wrap the loop in v_accvgpr copies as it did in probe 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
template<int MODE>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* clk, int iters, const double* src) {
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[threadIdx.x + 64 * i]; b[i] = src[threadIdx.x + 64 * (i + 4)]; }
  double s = 0;
  unsigned long long t0, t1;
  if (MODE == 0) {
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0,0,0,0};
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[i >> 2]));
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    double acc[64];
    for (int i = 0; i < 64; ++i) acc[i] = 0;
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 64; ++i)
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 3]));
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 64; ++i) s += acc[i];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template<int MODE> int run(const char* name, int blocks, int iters, const double* src, double flop_per_iter_wave, double inst_per_iter) {
  double* out; unsigned long long* clk;
  CK(hipMalloc(&out, 8 * blocks * 256)); CK(hipMalloc(&clk, 8 * blocks));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; ++w) k<MODE><<<blocks, 256>>>(out, clk, iters, src);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(out, clk, iters, src); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); }
  std::vector<unsigned long long> h(blocks); CK(hipMemcpy(h.data(), clk, 8 * blocks, hipMemcpyDeviceToHost));
  double flops = blocks * 4.0 * iters * flop_per_iter_wave;
  printf("%-22s blocks=%4d: %.3f ms %.2f TFLOP/s cycles/mfma=%.2f\n", name, blocks, best, flops / best * 1e-9, (double)h[0] / (iters * inst_per_iter));
  return 0;
}
int main() {
  std::vector<double> h(512); for (int i = 0; i < 512; ++i) h[i] = 0.001 * (i % 97) - 0.03;
  double* src; CK(hipMalloc(&src, 4096)); CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
  run<0>("asm 16x16x4", 256, 4000, src, 16 * 2048.0, 16);
  run<0>("asm 16x16x4", 512, 4000, src, 16 * 2048.0, 16);
  run<1>("asm 4x4x4_4b", 256, 4000, src, 64 * 512.0, 64);
  run<1>("asm 4x4x4_4b", 512, 4000, src, 64 * 512.0, 64);
  return 0;
}
