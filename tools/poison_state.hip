// Stale-state poison for gfx950 (round 4; DESIGN.md section 5.6, tests/test_gpu_register_poison.py).
//
// A wave starts with whatever earlier waves left in its registers, its LDS and its private segment.  A kernel that reads
// any of that before writing it returns results that depend on the process's history -- round 3's grad_x_kernel<4,1>,
// whose accumulators were "spilled" to AGPRs with EXEC == 0 by the compiler, was bit-stable in a fresh process and wrong
// after other kernels had run.  poison_state() fills ONE class of that state chip-wide with a chosen pattern, so that such
// a read shows up as NaN (pattern 0x7ff80000), as a wrong finite value (0x40590000), or disappears (0):
//   kind 0  LDS                       64 KB workgroups, enough of them to cover every CU's 160 KB several times
//   kind 1  private segment (scratch) 8 KB per lane
//   kind 2  VGPRs v1..v255 + AGPRs a0..a255 (the whole 512-entry file of a SIMD lane)
//   kind 3  SGPRs s8..s99
// The kernels sleep a little so that the waves of one launch are co-resident and cover different slots.
// Build: hipcc -O2 -shared -fPIC --offload-arch=gfx950 tools/poison_state.hip -o tools/libpoison.so  (__graft_entry__.build() does it)
#include <hip/hip_runtime.h>

#define PV(n) "v" #n
#define PA(n) "a" #n
#define PS(n) "s" #n
#define D10(M, t) M(t##0), M(t##1), M(t##2), M(t##3), M(t##4), M(t##5), M(t##6), M(t##7), M(t##8), M(t##9)
#define FROM10(M)                                                                                                          \
  D10(M, 1), D10(M, 2), D10(M, 3), D10(M, 4), D10(M, 5), D10(M, 6), D10(M, 7), D10(M, 8), D10(M, 9), D10(M, 10), D10(M, 11), \
      D10(M, 12), D10(M, 13), D10(M, 14), D10(M, 15), D10(M, 16), D10(M, 17), D10(M, 18), D10(M, 19), D10(M, 20), D10(M, 21),   \
      D10(M, 22), D10(M, 23), D10(M, 24), M(250), M(251), M(252), M(253), M(254), M(255)
#define V1_255 PV(1), PV(2), PV(3), PV(4), PV(5), PV(6), PV(7), PV(8), PV(9), FROM10(PV)
#define A0_255 PA(0), PA(1), PA(2), PA(3), PA(4), PA(5), PA(6), PA(7), PA(8), PA(9), FROM10(PA)
#define S8_99 PS(8), PS(9), D10(PS, 1), D10(PS, 2), D10(PS, 3), D10(PS, 4), D10(PS, 5), D10(PS, 6), D10(PS, 7), D10(PS, 8), D10(PS, 9)

__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned hi) {
  extern __shared__ unsigned lds[];
  for (int e = threadIdx.x; e < 65536 / 4; e += 256) lds[e] = (e & 1) ? hi : 0xdeadbeefu;
  __syncthreads();
  for (int s = 0; s < 20; ++s) __builtin_amdgcn_s_sleep(100);
}

__global__ __launch_bounds__(256) void poison_scratch_kernel(unsigned hi, unsigned* sink) {
  volatile unsigned buf[2048];  // 8 KB per lane
  for (int e = 0; e < 2048; ++e) buf[e] = (e & 1) ? hi : 0xdeadbeefu;
  unsigned s = 0;
  for (int e = threadIdx.x & 7; e < 2048; e += 97) s += buf[e];
  if (s == 12345u) sink[0] = s;
  for (int t = 0; t < 10; ++t) __builtin_amdgcn_s_sleep(100);
}

__global__ __launch_bounds__(64) void poison_vgpr_kernel(unsigned hi) {
  asm volatile(
      ".set migp_i, 1\n.rept 255\nv_mov_b32 v[migp_i], %0\n.set migp_i, migp_i + 1\n.endr\n"
      ".set migp_i, 0\n.rept 256\nv_accvgpr_write_b32 a[migp_i], %0\n.set migp_i, migp_i + 1\n.endr\ns_nop 4" ::"v"(hi)
      : V1_255, A0_255);
  for (int t = 0; t < 10; ++t) __builtin_amdgcn_s_sleep(100);
}

__global__ __launch_bounds__(64) void poison_sgpr_kernel(unsigned hi) {
  asm volatile(".set migp_i, 8\n.rept 92\ns_mov_b32 s[migp_i], %0\n.set migp_i, migp_i + 1\n.endr\ns_nop 4" ::"s"(hi) : S8_99);
  for (int t = 0; t < 10; ++t) __builtin_amdgcn_s_sleep(100);
}

// returns a hipError_t value (0 = success); synchronises the device
extern "C" int poison_state(int kind, unsigned hi) {
  hipError_t e = hipSuccess;
  unsigned* sink = nullptr;
  if (kind == 0) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    if (e != hipSuccess) return (int)e;
    poison_lds_kernel<<<256 * 8, 256, 65536>>>(hi);
  } else if (kind == 1) {
    e = hipMalloc(&sink, 64);
    if (e != hipSuccess) return (int)e;
    poison_scratch_kernel<<<256 * 16, 256>>>(hi, sink);
  } else if (kind == 2) {
    poison_vgpr_kernel<<<256 * 4 * 8, 64>>>(hi);
  } else if (kind == 3) {
    poison_sgpr_kernel<<<256 * 4 * 16, 64>>>(hi);
  } else {
    return -1;
  }
  e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (sink) (void)hipFree(sink);
  return (int)e;
}
