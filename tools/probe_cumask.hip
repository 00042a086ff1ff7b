// Probe: hipExtStreamCreateWithCUMask on gfx950 -- which (XCC, SE, CU) a mask bit selects, and where
// single-workgroup kernels land.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <map>
#include <vector>
__global__ void where(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(10);
}
static int key(unsigned hw, unsigned xcc) { return ((xcc & 0xf) << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf); }
int main() {
  unsigned* out; hipMalloc(&out, 8 * 4096);
  std::vector<unsigned> h(2 * 4096);
  auto run = [&](hipStream_t s, int nwg, const char* tag) {
    hipMemsetAsync(out, 0xff, 8 * 4096, s);
    where<<<nwg, 64, 0, s>>>(out, 200);
    hipError_t e = hipStreamSynchronize(s);
    hipMemcpy(h.data(), out, 8 * nwg, hipMemcpyDeviceToHost);
    std::map<int, int> perxcc; std::set<int> cus;
    for (int i = 0; i < nwg; ++i) { cus.insert(key(h[2 * i], h[2 * i + 1])); perxcc[h[2 * i + 1] & 0xf]++; }
    printf("%-28s err=%d distinct CUs=%zu ; WGs per XCC:", tag, (int)e, cus.size());
    for (auto& kv : perxcc) printf(" %d:%d", kv.first, kv.second);
    printf("\n");
    return cus;
  };
  hipStream_t s0; hipStreamCreate(&s0);
  run(s0, 4096, "unmasked 4096 WGs");
  for (int w = 0; w < 8; ++w) {
    unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    mask[w] = 0xffffffffu;
    hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
    if (e != hipSuccess) { printf("create word %d failed: %s\n", w, hipGetErrorString(e)); continue; }
    char tag[64]; snprintf(tag, 64, "mask word %d = ffffffff", w);
    auto cus = run(s, 2048, tag);
    if (w < 2) { printf("   keys(xcc<<12|se<<8|sh<<4|cu):"); for (int k : cus) printf(" %04x", k); printf("\n"); }
    hipStreamDestroy(s);
  }
  {  // low byte of every word
    unsigned mask[8]; for (int w = 0; w < 8; ++w) mask[w] = 0x000000ffu;
    hipStream_t s; hipExtStreamCreateWithCUMask(&s, 8, mask);
    auto cus = run(s, 2048, "mask 0xff in every word");
    printf("   keys:"); for (int k : cus) printf(" %04x", k); printf("\n");
    hipStreamDestroy(s);
  }
  {  // all but the lowest 8 bits of word 0
    unsigned mask[8]; for (int w = 0; w < 8; ++w) mask[w] = 0xffffffffu; mask[0] = 0xffffff00u;
    hipStream_t s; hipExtStreamCreateWithCUMask(&s, 8, mask);
    run(s, 4096, "all but bits 0..7");
    hipStreamDestroy(s);
  }
  // where do single-WG kernels land?
  printf("single-WG kernels land on (xcc):");
  for (int i = 0; i < 12; ++i) {
    where<<<1, 64, 0, s0>>>(out, 0); hipStreamSynchronize(s0);
    hipMemcpy(h.data(), out, 8, hipMemcpyDeviceToHost);
    printf(" %u", h[1] & 0xf);
  }
  printf("\n");
  return 0;
}
