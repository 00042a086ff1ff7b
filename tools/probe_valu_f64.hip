// Latency / issue-rate probe for the fp64 VALU instructions on the leaf kernel's serial chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#define T0() asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1(i) do { unsigned long long t1; asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); if (threadIdx.x == 0) out[i] = t1 - t0; } while (0)
#define REP10(x) x x x x x x x x x x
#define REP100(x) REP10(REP10(x))
__global__ void probe(unsigned long long* out, double* sink, double seed) {
  unsigned long long t0;
  double a = seed + threadIdx.x, b = 1.0000001, c = 0.5, d0 = a, d1 = a + 1, d2 = a + 2, d3 = a + 3;
  T0(); T1(0);  // empty
  T0(); REP100(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) T1(1);  // dependent fma
  T0(); REP100(asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d0) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d1) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d2) : "v"(b), "v"(c)); asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d3) : "v"(b), "v"(c));) T1(2);  // 4 independent chains (400 instr)
  T0(); REP100(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) T1(3);  // dependent rcp
  T0(); REP100(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d0) : "v"(b), "v"(c)); asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(d1) : "v"(b), "v"(c)); asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(d2) : "v"(b), "v"(c)); asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(d3) : "v"(b), "v"(c));) T1(4);  // 400 independent-ish dpp fmac
  T0(); REP100(asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a));) T1(5);  // dependent dpp mov
  T0(); REP100(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) T1(6);  // dependent mul
  T0(); REP100(asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(d0) : "v"(b), "v"(c));) T1(7);  // dependent dpp fmac (acc chain)
  T0(); REP100(asm volatile("v_rsq_f64 %0, %0" : "+v"(a));) T1(8);
  T0(); REP100(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(*(int*)&a));) T1(9);
  sink[threadIdx.x] = a + d0 + d1 + d2 + d3;
}
int main() {
  unsigned long long* out; double* sink;
  hipMalloc(&out, 128); hipMalloc(&sink, 64 * 8);
  probe<<<1, 64>>>(out, sink, 1.0); hipDeviceSynchronize();
  probe<<<1, 64>>>(out, sink, 1.0); hipDeviceSynchronize();
  unsigned long long h[16]; hipMemcpy(h, out, 128, hipMemcpyDeviceToHost);
  const char* names[] = {"empty", "100 dep fma", "400 indep fma", "100 dep rcp", "400 indep fmac_dpp", "100 dep mov_dpp(+nop1)", "100 dep mul", "100 dep fmac_dpp", "100 dep rsq", "100 dep mov_b32 quad dpp"};
  for (int i = 0; i < 10; ++i) printf("%-28s %6llu ticks (s_memtime @100MHz?)\n", names[i], h[i]);
  return 0;
}
