// Probe 2: fp64 VALU FMA rate vs fp64 MFMA (16x16x4 and 4x4x4) with in-kernel clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template<int MODE>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* clk, int iters, double seed) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  if (MODE == 0) {  // VALU fma f64, 16 independent chains
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = seed * i;
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    for (int i = 0; i < 16; ++i) s += acc[i];
  } else if (MODE == 1) {  // mfma 16x16x4
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0,0,0,0};
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {  // mfma 4x4x4 (4 blocks)
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    double a = seed + threadIdx.x * 1e-3, b = seed - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template<int MODE>
int run(const char* name, int blocks, int iters, double flop_per_inst_wave) {
  double* out; unsigned long long* clk;
  CK(hipMalloc(&out, sizeof(double) * blocks * 256)); CK(hipMalloc(&clk, 16 * blocks));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // warm for ~1 s to settle clocks
  for (int w = 0; w < 50; ++w) k<MODE><<<blocks, 256>>>(out, clk, iters, 1.0);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0)); k<MODE><<<blocks, 256>>>(out, clk, iters, 1.0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  std::vector<unsigned long long> h(2 * blocks); CK(hipMemcpy(h.data(), clk, 16 * blocks, hipMemcpyDeviceToHost));
  std::vector<double> mhz; for (int b = 0; b < blocks; ++b) mhz.push_back((double)h[2*b] / (double)h[2*b+1] * 100.0);
  std::sort(mhz.begin(), mhz.end());
  double waves = blocks * 4.0, flops = waves * iters * 16.0 * flop_per_inst_wave;
  double cyc_per_inst = (double)h[0] / (iters * 16.0);
  printf("%-12s blocks=%4d: %.3f ms  %.2f TFLOP/s  clock(median)=%.0f MHz  cycles/inst(wave0)=%.1f\n", name, blocks, best, flops / best * 1e-9, mhz[mhz.size()/2], cyc_per_inst);
  return 0;
}
int main() {
  run<0>("valu_fma64", 256, 20000, 128.0);
  run<0>("valu_fma64", 512, 20000, 128.0);
  run<0>("valu_fma64", 1024, 20000, 128.0);
  run<1>("mfma16x16x4", 256, 4000, 2048.0);
  run<1>("mfma16x16x4", 512, 4000, 2048.0);
  run<2>("mfma4x4x4", 256, 8000, 512.0);
  run<2>("mfma4x4x4", 512, 8000, 512.0);
  return 0;
}
