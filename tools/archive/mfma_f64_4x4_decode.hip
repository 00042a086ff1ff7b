// Decode lane maps of v_mfma_f64_4x4x4_4b_f64 by one-hot probing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
template<int CBSZ, int ABID, int BLGP>
__global__ void onehot(double* d) {  // block = pa*64+pb
  int l = threadIdx.x, pa = blockIdx.x >> 6, pb = blockIdx.x & 63;
  double a = (l == pa) ? 1.0 : 0.0, b = (l == pb) ? 1.0 : 0.0;
  d[blockIdx.x * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, BLGP);
}
static double* dD; static std::vector<double> D(4096 * 64);
template<int CBSZ, int ABID, int BLGP> int go(const char* name, int verbose_lanes) {
  onehot<CBSZ, ABID, BLGP><<<4096, 64>>>(dD);
  CK(hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost));
  printf("== %s\n", name);
  for (int l = 0; l < 64; ++l) {
    if (l >= verbose_lanes && l % 16 != 0 && l != 63) continue;
    printf("  D[lane %2d] = sum of", l);
    for (int pa = 0; pa < 64; ++pa) for (int pb = 0; pb < 64; ++pb) { double v = D[(pa * 64 + pb) * 64 + l]; if (v != 0.0) printf(" %sA%d*B%d", v < 0 ? "-" : "", pa, pb); }
    printf("\n");
  }
  return 0;
}
int main() {
  CK(hipMalloc(&dD, D.size() * 8));
  go<0,0,0>("cbsz0 abid0 blgp0", 20);
  go<2,0,0>("cbsz2 abid0", 4);
  go<2,1,0>("cbsz2 abid1", 4);
  go<1,1,0>("cbsz1 abid1", 0);
  go<0,0,1>("blgp1", 2);
  go<0,0,2>("blgp2", 2);
  go<0,0,4>("blgp4", 2);
  return 0;
}
