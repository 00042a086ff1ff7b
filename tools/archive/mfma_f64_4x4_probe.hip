// Probe 3: lane maps of v_mfma_f64_4x4x4_4b_f64 incl. cbsz/abid A-broadcast and blgp negate bits,
// and its issue rate with broadcast on.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

// raw dump: every lane supplies a(l), b(l), c(l); returns d(l) for a given (cbsz,abid,blgp)
template<int CBSZ, int ABID, int BLGP>
__global__ void raw(const double* a, const double* b, const double* c, double* d) {
  int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], CBSZ, ABID, BLGP);
}

template<int BCAST>
__global__ __launch_bounds__(256) void rate(double* out, unsigned long long* clk, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = 0;
  double a = 1.0 + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
    if (BCAST) {
#pragma unroll
      for (int i = 0; i < 16; i += 4) {
        acc[i+0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i+0], 2, 0, 0);
        acc[i+1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i+1], 2, 1, 0);
        acc[i+2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i+2], 2, 2, 0);
        acc[i+3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i+3], 2, 3, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

static std::vector<double> A(64), B(64), C(64), D(64);
static double *dA, *dB, *dC, *dD;
template<int CBSZ, int ABID, int BLGP> int go() {
  raw<CBSZ, ABID, BLGP><<<1, 64>>>(dA, dB, dC, dD);
  CK(hipMemcpy(D.data(), dD, 512, hipMemcpyDeviceToHost));
  return 0;
}
// hypothesis H: lane l: blk=l>>4, A_blk[i=l&3][k=(l>>2)&3], B_blk[k=(l>>2)&3][j=l&3], D_blk[i=(l>>2)&3][j=l&3]
static int check(const char* name, int cbsz, int abid, int nega, int negb, int negc) {
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int blk = l >> 4, i = (l >> 2) & 3, j = l & 3;
    int ablk = blk;
    if (cbsz) { int grp = 1 << cbsz; ablk = (blk / grp) * grp + abid % grp; }
    double s = negc ? -C[l] : C[l];
    for (int k = 0; k < 4; ++k) {
      double av = A[ablk * 16 + k * 4 + i];   // lane with (l&3)=i, (l>>2)&3=k
      double bv = B[blk * 16 + k * 4 + j];
      s += (nega ? -av : av) * (negb ? -bv : bv);
    }
    if (s != D[l]) ++bad;
  }
  printf("%-28s mismatches=%d\n", name, bad);
  return bad;
}
int main() {
  for (int l = 0; l < 64; ++l) { A[l] = 1 + (l * 7) % 23 + (l / 16) * 31; B[l] = 2 + (l * 5) % 19 + (l / 16) * 13 + (l % 4) * 3; C[l] = 1000 + l; }
  CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 512)); CK(hipMalloc(&dD, 512));
  CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), 512, hipMemcpyHostToDevice));
  go<0,0,0>(); check("cbsz0", 0, 0, 0, 0, 0);
  go<2,0,0>(); check("cbsz2 abid0", 2, 0, 0, 0, 0);
  go<2,1,0>(); check("cbsz2 abid1", 2, 1, 0, 0, 0);
  go<2,3,0>(); check("cbsz2 abid3", 2, 3, 0, 0, 0);
  go<1,1,0>(); check("cbsz1 abid1", 1, 1, 0, 0, 0);
  go<0,0,1>(); check("blgp1 (negA?)", 0, 0, 1, 0, 0);
  go<0,0,2>(); check("blgp2 (negB?)", 0, 0, 0, 1, 0);
  go<0,0,4>(); check("blgp4 (negC?)", 0, 0, 0, 0, 1);
  go<2,2,1>(); check("cbsz2 abid2 negA", 2, 2, 1, 0, 0);
  // dump a few raw values for manual decode in case hypothesis fails
  go<0,0,0>(); printf("raw D[0..7]:"); for (int l = 0; l < 8; ++l) printf(" %.0f", D[l]); printf("\n");
  // rate
  double* out; unsigned long long* clk; int blocks = 512, iters = 8000;
  CK(hipMalloc(&out, 8 * blocks * 256)); CK(hipMalloc(&clk, 8 * blocks));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    for (int w = 0; w < 20; ++w) { if (mode) rate<1><<<blocks,256>>>(out, clk, iters); else rate<0><<<blocks,256>>>(out, clk, iters); }
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      if (mode) rate<1><<<blocks,256>>>(out, clk, iters); else rate<0><<<blocks,256>>>(out, clk, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    double flops = blocks * 4.0 * iters * 16.0 * 512.0;
    printf("rate bcast=%d: %.3f ms  %.2f TFLOP/s\n", mode, best, flops / best * 1e-9);
  }
  return 0;
}
