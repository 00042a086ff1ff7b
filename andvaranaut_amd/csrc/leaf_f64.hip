// Leaf kernels of the blocked fp64 Cholesky:
//   potrf_leaf128 : in-place lower Cholesky of one 128x128 diagonal block, LDS resident, plus its explicit
//                   inverse M = L^-1 (128x128 lower triangular, row-major, consumed by trsm_strip128).
//   trsm_strip128 : X * L^T = B for row strips of a 128-column panel (LAPACK dtrsm R,L,T,N) as the product
//                   X = B * M^T: no dependency chain along the columns, so the 128 output columns of 16 rows are
//                   split over the four waves of a workgroup and every operand is ONE global round trip
//                   (round 1 solved by substitution over eight 16-column blocks: 144 dependent MFMAs per wave
//                   and two dependent round trips, 17 us alone and 50+ us next to a trailing update; rocBLAS'
//                   dtrsm inverts 128x128 diagonal blocks the same way).
// The reference reaches the same arithmetic through scipy.linalg.cholesky / LAPACK dpotrf
// (gpmcmc.py:313 and pm.gp.Marginal at gpmcmc.py:321-323).
#include "migp_kernels.h"

namespace migp {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int LEAF = 128;
constexpr int SB = 16;        // sub-block width
// LDS image of the leaf: only the lower block-trapezoid is kept.  Row block b (rows 16b..16b+15)
// stores columns 0..16(b+1)-1 with row stride 16b+18 doubles (16-byte aligned, and 18 or 2 mod 32 so
// that 16 consecutive rows read at one column hit 16 distinct bank pairs).  75.8 KB instead of 133 KB:
// the leaf can then share a CU with one 72 KB GEMM workgroup instead of waiting for an empty CU.
constexpr int LEAF_ELEMS = 9472;  // sum_b 16 * (16 (b+1) + 2)
__device__ __forceinline__ int soff(int row) {
  const int b = row >> 4;
  return 128 * b * (b + 1) + 32 * b + (row & 15) * (16 * b + 18);
}

// 1/sqrt(x) for normal positive x: hardware seed (v_rsq_f64) + two Goldschmidt steps + one
// Newton correction; ~1 ulp, about 15 dependent FMAs instead of the ~80-instruction
// correctly-rounded sqrt + divide sequence (which dominated the per-pivot latency of the leaf).
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  h = __builtin_fma(h, r, h);
  double inv = h + h;
  const double t = __builtin_fma(-x * inv, inv, 1.0);
  return __builtin_fma(0.5 * inv, t, inv);
}

#ifdef LEAF_STAMPS
// Dev probe (tools/leaf_timing.hip): ONE time stamp per launch, at the program point the host selects, relative to the
// wave's start -- s_memtime is a scalar memory round trip (~250 cycles), so a kernel with a stamp at every phase boundary
// measures mostly its stamps.  Wave 0 sites: 8 jb + k; wave 1 sites: 64 + 8 jb + k.
__device__ unsigned long long g_leaf_probe_out;
__device__ int g_leaf_probe_sel;
#define LEAF_PROBE(code)                                                                    \
  do {                                                                                      \
    if (probe_sel == (code) && (threadIdx.x == 0 || threadIdx.x == 64 || threadIdx.x == 320)) { \
      unsigned long long t_;                                                                \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
      g_leaf_probe_out = t_ - t_start;                                                      \
    }                                                                                       \
  } while (0)
#else
#define LEAF_PROBE(code)
#endif

// LDS words the waves of the leaf meet through, as LDS instructions: through a generic `volatile int*` the compiler emits
// FLAT loads / stores with system-scope cache bits and vmcnt waits (round 4's listing), and a workgroup-scope fence also
// waits for the wave's outstanding GLOBAL stores (the column stream-out).
typedef __attribute__((address_space(3))) int lds_int_t;
typedef __attribute__((address_space(3))) double lds_double_t;
__device__ __forceinline__ int lds_load(volatile lds_int_t* p) { return *p; }
__device__ __forceinline__ void lds_store(volatile lds_int_t* p, int v) { *p = v; }
__device__ __forceinline__ void lds_add(lds_int_t* p, int v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wave_lds_fence() {
  // this wave's LDS operations have completed (they execute in order; the CU's LDS is coherent for the workgroup), and
  // the compiler moves no memory access across the point
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// ---- Who does what in the leaf's eight waves.  The waves meet through LDS counters, not barriers: a phase that waits for "every
// helper" waits for a COUNT, and that count has to be the number of waves that execute the arriving side.  Round 5 got it wrong
// once (waves 4..7 also ran the helpers' image load and counted as load arrivals: the helpers started on an incomplete image,
// wrong factors now and then -- profiles/NOTES_r05.md item 6).  Every expected count below is derived from this ONE table, and
// the checked flavour of the kernel (-DLEAF_CHECKED: tools/leaf_check.hip, tests/test_gpu_leaf_protocol.py) verifies at every
// meeting point that the counters hold exactly what the protocol allows there and reports a mismatch through the info word.
namespace roles {
enum Role { CHAIN, HELPER, IDLE, INVERSE };
constexpr int NWAVES = 8;
constexpr Role ROLE[NWAVES] = {CHAIN, HELPER, HELPER, HELPER, IDLE, INVERSE, INVERSE, INVERSE};
constexpr int count(Role r) {
  int n = 0;
  for (int w = 0; w < NWAVES; ++w) n += ROLE[w] == r ? 1 : 0;
  return n;
}
constexpr int first(Role r) {
  for (int w = 0; w < NWAVES; ++w)
    if (ROLE[w] == r) return w;
  return -1;
}
constexpr bool contiguous(Role r) {
  for (int w = 0; w < NWAVES; ++w)
    if ((ROLE[w] == r) != (w >= first(r) && w < first(r) + count(r))) return false;
  return true;
}
constexpr int N_HELPERS = count(HELPER), N_INVERSE = count(INVERSE);
constexpr int FIRST_HELPER = first(HELPER), FIRST_INVERSE = first(INVERSE);
static_assert(count(CHAIN) == 1 && first(CHAIN) == 0, "wave 0 is the chain");
static_assert(contiguous(HELPER) && contiguous(INVERSE), "a role's waves are consecutive (the dispatch below indexes them)");
static_assert(N_HELPERS == 3 && N_INVERSE == 3, "the tile dealing (lt::, three waves) and the inverse's column dealing assume three each");
// wave 4 * k + s runs on SIMD s: the chain has SIMD 0 to itself, inverse wave i shares a SIMD with helper i
static_assert((FIRST_INVERSE - FIRST_HELPER) % 4 == 0 && FIRST_HELPER % 4 != 0, "helper i and inverse wave i share a SIMD, none shares the chain's");
__device__ __forceinline__ bool is_helper(int wave) { return wave >= FIRST_HELPER && wave < FIRST_HELPER + N_HELPERS; }
__device__ __forceinline__ bool is_inverse(int wave) { return wave >= FIRST_INVERSE && wave < FIRST_INVERSE + N_INVERSE; }
constexpr int URGENT_ITERS = 6;  // iterations jb = 0 .. 5 have urgent tiles (the next block column's)
}  // namespace roles
[[maybe_unused]] constexpr int LEAF_PROTOCOL_INFO = -1000;  // checked flavour: info = LEAF_PROTOCOL_INFO - site on a protocol mismatch
#ifdef LEAF_CHECKED
#define LEAF_EXPECT(cond, site)                                                              \
  do {                                                                                       \
    if (!(cond) && (threadIdx.x & 63) == 0) atomicMin(info, LEAF_PROTOCOL_INFO - (site));    \
  } while (0)
#else
#define LEAF_EXPECT(cond, site)
#endif

// 1/x for normal x: hardware seed (v_rcp_f64) + two Newton steps (~1 ulp).
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
}

// broadcast lane `src`'s double to the whole wave through scalar registers (v_readlane_b32 x 2)
__device__ __forceinline__ double bcast_lane(double v, int src) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), src);
  const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// acc += bcast(src from lane C of each 16-lane row) * mul   -- one instruction: v_fmac_f64 with the
// DPP row_newbcast control (the only DPP form fp64 VALU ops have on gfx90a+).  hipcc cannot see the
// "VALU write -> DPP read of the same VGPR" hazard (2 wait states) inside inline asm: callers must
// not pass a `src` written by the immediately preceding VALU instruction (mov_rowbcast pads itself).
// Two wait states tied to the VALUE: the asm takes `v` in and out, so its producer is scheduled before the s_nop and every
// DPP consumer after it (a bare `asm volatile("s_nop 1")` orders nothing -- the compiler moved rs's last Newton step behind
// it in round 2's build; tests/test_dpp_hazard.py disassembles the library and checks every DPP read).
__device__ __forceinline__ void dpp_settle(double& v) { asm volatile("s_nop 1" : "+v"(v)); }
template <int C>
__device__ __forceinline__ void fmac_rowbcast(double& acc, double src, double mul) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
               : "+v"(acc)
               : "v"(src), "v"(mul), "n"(C));
}
template <int C>
__device__ __forceinline__ double mov_rowbcast(double src) {
  double out;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf"
               : "=v"(out)
               : "v"(src), "n"(C));
  return out;
}

// broadcast lane G of every quad (4 consecutive lanes) to the quad: two 32-bit DPP moves
template <int G>
__device__ __forceinline__ double quad_bcast(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(u & 0xffffffffu), G * 0x55, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(u >> 32), G * 0x55, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// ---- The chain wave's elimination (round 5).
// Wave 0 holds NA register panels of 16 columns: a[0] = the 16x16 diagonal block, row (lane & 15) per lane (the four 16-lane
// DPP rows mirror each other), a[1..] = rows BELOW the block, one per lane.  One square-root-free elimination step J is
//   w_i = -a_i[J] / p_J ;  a_i[C] += w_i * L~[C][J]  (C > J),  L~[C][J] = a_0[J] of lane C, fetched by the row_newbcast:C
// of v_fmac_f64_dpp -- the SAME instruction updates the diagonal block and the rows below it, so the triangular solve of
// the rows below (rounds 1-4: a separate substitution after the factor, 2.1k cycles per 16 columns on the chain, and a
// 2.4k-cycle one-row-per-thread substitution on the helper waves) rides in the issue slots of the factorisation.
// Rounds 1-4 also left the schedule to the compiler, which emitted, per pivot, all fillers, THEN the pivot broadcast and
// the whole reciprocal chain (disassembly of round 4: ~245 cycles per pivot against ~70 of dependent latency).  Here the
// fillers of step J - 1 are issued inside the latency gaps of step J's chain (rcp 20 cycles, every dependent fp64 op 8,
// the DPP broadcast 17; an independent fp64 instruction issues every ~5.3: tools/probe_valu_f64.hip), pinned by
// sched_barriers.  The reciprocal is the hardware seed y0 (2^-24) times (1 + e + e^2), e = 1 - p y0: relative error e^3 =
// 2^-73, three dependent operations behind the seed instead of five.
#define MIGP_SB0() __builtin_amdgcn_sched_barrier(0)
template <int I, int N, class F>
__device__ __forceinline__ void lt_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    lt_static_for<I + 1, N>(f);
  }
}
namespace ch {
// The chain in `asm volatile`: volatile statements keep their source order, so the schedule below IS the instruction order
// (arithmetic left to the compiler floats: instruction selection puts an unchained multiply next to its first user, whatever
// sched_barriers stand in between -- the first form of this round had the three multiplications in front of e0).  Wait
// states the hardware does not interlock and the compiler cannot see inside asm: a transcendental result needs one
// instruction before its first VALU reader (s_nop 0 behind v_rcp_f64), a DPP read two behind the VALU write of its
// operand (callers; tests/test_dpp_hazard.py checks the library).
template <int C>
__device__ __forceinline__ void fmac_bc(double& acc, double src, double mul) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(C));
}
template <int C>
__device__ __forceinline__ double mov_bc(double src) {  // two instructions since src was written: the caller's job
  double out;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "=v"(out) : "v"(src), "n"(C));
  return out;
}
template <int C>
__device__ __forceinline__ double mov_bc_padded(double src) {
  double out;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "=v"(out) : "v"(src), "n"(C));
  return out;
}
__device__ __forceinline__ double a_rcp(double p) {
  double y;
  asm volatile("v_rcp_f64 %0, %1\n\ts_nop 0" : "=v"(y) : "v"(p));
  return y;
}
__device__ __forceinline__ double a_one_minus(double p, double y) {  // 1 - p y
  double e;
  asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e) : "v"(p), "v"(y));
  return e;
}
__device__ __forceinline__ double a_mul(double x, double y) {
  double t;
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(t) : "v"(x), "v"(y));
  return t;
}
__device__ __forceinline__ double a_nq(double e) {  // -(e + e^2)
  double q;
  asm volatile("v_fma_f64 %0, -%1, %1, -%1" : "=v"(q) : "v"(e));
  return q;
}
__device__ __forceinline__ double a_fma_negc(double t, double q) {  // t q - t
  double w;
  asm volatile("v_fma_f64 %0, %1, %2, -%1" : "=v"(w) : "v"(t), "v"(q));
  return w;
}
// fillers [LO, HI) of elimination step JP: a[i][C] += bcast_C(a[0][JP]) * w[i] for C = JP + 2 .. 15, enumerated C-major
template <int JP, int NA, int LO, int HI>
__device__ __forceinline__ void fillers(double (&a)[NA][16], const double (&w)[NA]) {
  lt_static_for<LO, HI>([&](auto Ic) {
    constexpr int idx = decltype(Ic)::value, C = JP + 2 + idx / NA, i = idx % NA;
    fmac_bc<C>(a[i][C], a[0][JP], w[i]);
  });
}
template <int J, int NA>
struct Step {
  // p: pivot J, broadcast; wp: the multipliers of step J - 1, whose fillers (columns J + 1 .. 15) are issued here, behind
  // the reciprocal seed.  The columns stay UNNORMALISED (X~ = a): the rank-16 updates of the trailing tiles are
  // tile -= (X~ D^-1) X~^T, so nothing on the chain waits for the reciprocal square roots of the pivots (rounds 1-4 and the
  // first form of this round normalised on the chain: 1.7k cycles per 16 columns) -- the helpers normalise a column block
  // an iteration later.
  static __device__ __forceinline__ void run(double (&a)[NA][16], double p, const double (&wp)[NA]) {
    constexpr int NF = J >= 1 ? NA * (15 - J) : 0;              // fillers of step J - 1
    constexpr int want_fm = NA >= 3 ? 0 : 3 - NA;              // two instructions between the critical fmac and the DPP read of its result
    constexpr int n_fm = NF - 1 >= want_fm ? want_fm : (NF - 1 > 0 ? NF - 1 : 0);
    constexpr int n1 = NF - n_fm;                              // behind the reciprocal seed (20 cycles); filler 0 (the next pivot's column) is here
    if constexpr (J < 15) {
      const double y0 = a_rcp(p);
      fillers<J - 1, NA, 0, n1>(a, wp);
      const double e0 = a_one_minus(p, y0);
      double t[NA], w[NA];
      t[0] = a_mul(a[0][J], y0);
      const double nq = a_nq(e0);
#pragma unroll
      for (int i = 1; i < NA; ++i) t[i] = a_mul(a[i][J], y0);
      w[0] = a_fma_negc(t[0], nq);
#pragma unroll
      for (int i = 1; i < NA; ++i) w[i] = a_fma_negc(t[i], nq);
      fmac_bc<J + 1>(a[0][J + 1], a[0][J], w[0]);
      lt_static_for<1, NA>([&](auto Ic) { fmac_bc<J + 1>(a[decltype(Ic)::value][J + 1], a[0][J], w[decltype(Ic)::value]); });
      fillers<J - 1, NA, n1, NF>(a, wp);
      double pn;
      if constexpr (NA - 1 + n_fm >= 2) pn = mov_bc<J + 1>(a[0][J + 1]);
      else pn = mov_bc_padded<J + 1>(a[0][J + 1]);
      Step<J + 1, NA>::run(a, pn, w);
    }
  }
};
}  // namespace ch

// 16x16 triangular inverse by substitution with the matrix held one ROW per lane (a[k] = L16[lane][k]): column c of the
// inverse is solved by lane c, and L16[r][k] reaches it through the row_newbcast:R of v_fmac_f64_dpp -- no LDS reads on the
// chain (the LDS-read form took 4.2k cycles per block, this one ~1k).  nz[k] = -z[k].
template <int R, int K>
struct DinvRow {
  static __device__ __forceinline__ void run(double& s0, double& s1, const double (&a)[16], const double (&nz)[16]) {
    if (K & 1) fmac_rowbcast<R>(s1, a[K], nz[K]);
    else fmac_rowbcast<R>(s0, a[K], nz[K]);
    DinvRow<R, K + 1>::run(s0, s1, a, nz);
  }
};
template <int R>
struct DinvRow<R, R> {
  static __device__ __forceinline__ void run(double&, double&, const double (&)[16], const double (&)[16]) {}
};
template <int R>
struct DinvStep {
  static __device__ __forceinline__ void run(double (&z)[16], double (&nz)[16], const double (&a)[16], const double (&iv)[16], int c) {
    double s0 = (R == c) ? 1.0 : 0.0, s1 = 0.0;
    DinvRow<R, 0>::run(s0, s1, a, nz);
    z[R] = (R >= c) ? (s0 + s1) * iv[R] : 0.0;
    nz[R] = -z[R];
    DinvStep<R + 1>::run(z, nz, a, iv, c);
  }
};
template <>
struct DinvStep<16> {
  static __device__ __forceinline__ void run(double (&)[16], double (&)[16], const double (&)[16], const double (&)[16], int) {}
};

// Trailing 16x16 tiles of the leaf in block coordinates (r >= c >= 1; tile (1,1) is wave 0's from the start), in dealing
// order: tile i belongs to helper wave i % 3 (waves 1..3), register slot i / 3 -- nine tiles per wave, and every
// iteration's live tiles are split within one tile of evenly.  A tile lives in its wave's REGISTERS from the start of the
// loop until its last rank-16 update (left-looking accumulation: tile (r,c) -= X(r,k) X(c,k)^T for k = 0 .. c-1, the
// diagonal tiles up to k = c-2, after which wave 0 takes them over) and is written to the LDS image exactly once.
namespace lt {
constexpr int NTT = 27;
constexpr int TR[NTT] = {2, 3, 4, 5, 6, 7, 2, 3, 4, 5, 6, 7, 3, 4, 5, 6, 7, 4, 5, 6, 7, 5, 6, 7, 6, 7, 7};
constexpr int TC[NTT] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7};
constexpr bool live(int i, int jb) { return TR[i] == TC[i] ? jb <= TC[i] - 2 : jb <= TC[i] - 1; }
constexpr bool last(int i, int jb) { return TR[i] == TC[i] ? jb == TC[i] - 2 : jb == TC[i] - 1; }
// phase 0: the tiles that get their LAST update in iteration jb (block column jb + 1 and the diagonal tile jb + 2 -- what
// wave 0 needs next); phase 1: the other live tiles
constexpr bool sel(int i, int jb, int ph) { return live(i, jb) && (last(i, jb) == (ph == 0)); }
// does wave w read block row b of column block jb in phase ph -- as an A operand (multipliers W) / as a B operand (X~)?
constexpr bool needs_a(int w, int jb, int b, int ph) {
  for (int s = 0; s < 9; ++s)
    if (sel(3 * s + w, jb, ph) && TR[3 * s + w] == b) return true;
  return false;
}
constexpr bool needs_b(int w, int jb, int b, int ph) {
  for (int s = 0; s < 9; ++s)
    if (sel(3 * s + w, jb, ph) && TC[3 * s + w] == b) return true;
  return false;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}
}  // namespace lt

// info: 0 = ok, else 1-based global index of the first non-positive (or NaN) pivot (atomicMin'd).
//
// Structure (round 5) per 16-column block jb of the 128x128 leaf:
//  wave 0 (the chain): holds the diagonal 16x16 block AND every row below it in registers (ch::Step above: two or three
//      16-column panels per lane), eliminates the 16 columns, writes the unnormalised columns X~ of the rows below to the
//      LDS image with the pivots d and -1/d, raises `ready`, applies the rank-16 update  -= (X~ D^-1) X~^T  to the NEXT
//      diagonal tile on MFMA and waits for the helpers' urgent tiles -- the rest of the next panel.  Sixteen otherwise idle
//      lanes carry the rows of the identity through the same elimination: Y_jb = L~_jb^-T, the block's triangular inverse
//      up to scalings, for free.
//  waves 1..3 (helpers): keep the 27 trailing 16x16 tiles in registers (namespace lt), left-looking.  Behind `ready` they
//      first finish the tiles of the NEXT block column (their last update: results go to the LDS image, `urgent` is
//      raised), then update the other live tiles while wave 0 already eliminates the next block, and normalise column
//      block jb - 1 on its way out to memory (L = X~ D^-1/2; the LDS image stays unnormalised).
//  waves 5..7 (the inverse; wave 4 idles): beside the helpers on SIMDs 1..3, they form block row jb of M = L^-1 from Y_jb
//      and the rows above (see `inverse`).
// No workgroup barrier inside the loop: the sides meet through LDS counters.  The explicit inverse used to be a phase of its
// own behind the factorisation (8 substitutions + three block-doubling levels, ~6 us of the leaf's 30); what is left behind
// the last pivot now is one 16x16x16 product per tile of the last block row.
constexpr int YB_LD = 18;                       // row stride of an identity-row buffer (16 rows at one column: 16 bank pairs)
constexpr int MT_ELEMS = 256;                   // one 16x16 tile of M^ in MFMA operand order: [r][lane] <-> element [kq + 4 r][n]
__device__ __forceinline__ constexpr int mt_off(int b, int c) { return (b * (b + 1) / 2 + c) * MT_ELEMS; }
// Where element (row, col) of M = L^-1 lives in the leaf's 16384-double output: 16x16 tiles (row tile jb, column tile kb; only
// kb <= jb are written or read), each tile in the order the strip kernel's MFMA B-operand loads want it -- lane l = 16 q + n
// of a wave holds M[16 jb + n][16 kb + 4 q + s], s = 0..3, so a tile is two runs of 64 lanes x 2 doubles: s < 2, then s >= 2.
// A wave's operand load is then 1 KB of consecutive addresses (round 5; row-major until then: every quarter-wave touched 16
// rows, and the strip was bound by the texture addresser, not by memory or MFMA).
// (minv_index itself lives in migp_kernels.h)

__device__ __forceinline__ void potrf_leaf128_body(double* __restrict__ Ablk, long lda, double* __restrict__ minv,
                                                    int col0, int* __restrict__ info, double* smem, double* yrow) {
  double* S = smem;                     // packed lower block-trapezoid, see soff(): X~ (unnormalised columns of L)
  double* pvt = smem + LEAF_ELEMS;      // [128] the pivots d_c (L[c][c]^2)
  double* nra = pvt + LEAF;             // [128] -1 / d_c
  double* Ybuf = nra + LEAF;            // [4][16][YB_LD] identity rows of block jb (buffer jb & 3)
  double* Mh = Ybuf + 4 * SB * YB_LD;   // [36][MT_ELEMS] M^ = D^-1/2 M (block row b, block column c <= b at mt_off(b, c))
  lds_int_t* sync_a = (lds_int_t*)(Mh + 36 * MT_ELEMS);
  volatile lds_int_t* sync_w = sync_a;  // [0] column blocks published by wave 0, [1] urgent
  // arrivals of waves 1..3, [2] their load arrivals, [3] the arrivals of the inverse's waves (5..7) with a block row of M^
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: tile coordinates and soff() bases derived
                                                              // from it are computed on the scalar unit (v_mul_lo_u32 is quarter rate)
#ifdef LEAF_STAMPS
  const int probe_sel = __builtin_amdgcn_readfirstlane(g_leaf_probe_sel);
  unsigned long long t_start;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start)::"memory");
#endif
  typedef double double2_t __attribute__((ext_vector_type(2)));
  if (tid == 64) { sync_w[0] = 0; sync_w[1] = 0; sync_w[2] = 0; sync_w[3] = 0; }
  __syncthreads();

  // tile (rb, cb) of the LDS image as MFMA operands / result:
  //   A operand: lane holds [16 rb + n][16 cb + 4 s + kq];  B operand: [16 rb + 4 s + kq][16 cb + n];  D: [16 rb + kq + 4 r][16 cb + n]
  const int nn = lane & 15, kq = lane >> 4;
  auto put = [&](const double4_t& v, int rb, int cb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) S[soff(16 * rb + kq + 4 * r) + 16 * cb + nn] = v[r];
  };

  if (wave == 0) {
    // ---------------------------------------------------------------- the chain
    auto iteration = [&](auto NAc, int jb) {
      constexpr int NA = decltype(NAc)::value;
      const int j0 = jb * SB;
      const int r16 = lane & 15;
      LEAF_PROBE(8 * jb + 0);
      // panel 0: the diagonal block's row r16 (mirrored); panel i >= 1, lane l: row g = j0 + 16 + 64 (i - 1) + l of the leaf,
      // or, for 128 <= g < 144, row g - 128 of the identity, or nothing
      double a[NA][16];
      lds_double_t* dst[NA];  // (explicit LDS pointers: a pointer chosen between two LDS arrays per lane becomes a FLAT access otherwise)
      bool writes[NA];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int g = i == 0 ? j0 + r16 : j0 + SB + 64 * (i - 1) + lane;
        const bool real = g < LEAF, ident = !real && g < LEAF + SB;
        writes[i] = i == 0 ? lane < SB : (real || ident);
        const int off = real ? soff(g) + j0 : (int)(Ybuf - smem) + (jb & 3) * (SB * YB_LD) + (ident ? g - LEAF : 0) * YB_LD;
        dst[i] = (lds_double_t*)smem + off;
        // block 0 straight from memory: the chain starts one global round trip after the launch.  Lanes without a row of
        // the leaf read the block's own row and are overwritten below.
        if (jb == 0) {
          const double* src = Ablk + (long)(real ? g : r16) * lda;
#pragma unroll
          for (int c = 0; c < SB; c += 2) {
            const double2_t v = *reinterpret_cast<const double2_t*>(src + c);
            a[i][c] = v.x;
            a[i][c + 1] = v.y;
          }
        } else {
          const lds_double_t* src = (const lds_double_t*)smem + (real ? off : soff(j0 + r16) + j0);
#pragma unroll
          for (int c = 0; c < SB; c += 2) {
            const double2_t v = *reinterpret_cast<const __attribute__((address_space(3))) double2_t*>(src + c);
            a[i][c] = v.x;
            a[i][c + 1] = v.y;
          }
        }
        if (i == NA - 1) {  // only the last panel can hold identity rows / idle lanes
#pragma unroll
          for (int c = 0; c < SB; ++c) a[i][c] = real ? a[i][c] : ((ident && g - LEAF == c) ? 1.0 : 0.0);
        }
      }
      LEAF_PROBE(8 * jb + 1);
      __builtin_amdgcn_sched_barrier(0);
      {
        double w0[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) w0[i] = 0.0;
        ch::Step<0, NA>::run(a, ch::mov_bc_padded<0>(a[0][0]), w0);
      }
      LEAF_PROBE(8 * jb + 2);
      // lane c keeps the pivot d_c = a[0][c] (a run-time register index would put the panels into scratch) and -1 / d_c
      double my_p = a[0][0];
#pragma unroll
      for (int c = 1; c < SB; ++c) my_p = (r16 == c) ? a[0][c] : my_p;
      double my_nr;
      {
        const double y0 = __builtin_amdgcn_rcp(my_p);
        const double e0 = __builtin_fma(-my_p, y0, 1.0);
        const double nq = __builtin_fma(-e0, e0, -e0);
        my_nr = __builtin_fma(y0, nq, -y0);
      }
      // (the identity rows go into buffer jb & 3: the inverse's waves run on their own and have to be through with block row
      // jb - 4, the buffer's previous user -- they always are; the check is one LDS read)
      if (jb >= 4) {
        while (sync_w[3] < roles::N_INVERSE * (jb - 3)) __builtin_amdgcn_s_sleep(1);
        // (the inverse's waves have seen blocks 0 .. jb - 1 published: at most that many block rows each)
        LEAF_EXPECT(sync_w[3] <= roles::N_INVERSE * jb, 1);
      }
      // X~ of the rows below the block (and the identity rows), the pivots and their reciprocals first: the updates wait for
      // them; the block's own rows are read by nobody before the normalisation
#pragma unroll
      for (int i = 1; i < NA; ++i) {
        if (writes[i]) {
#pragma unroll
          for (int c = 0; c < SB; c += 2) {
            double2_t v;
            v.x = a[i][c];
            v.y = a[i][c + 1];
            *reinterpret_cast<__attribute__((address_space(3))) double2_t*>(dst[i] + c) = v;
          }
        }
      }
      pvt[j0 + r16] = my_p;  // (the four 16-lane rows write the same values)
      nra[j0 + r16] = my_nr;
      wave_lds_fence();
      if (lane == 0) lds_store(sync_w + 0, jb + 1);
      if (writes[0]) {
#pragma unroll
        for (int c = 0; c < SB; c += 2) {
          double2_t v;
          v.x = a[0][c];
          v.y = a[0][c + 1];
          *reinterpret_cast<__attribute__((address_space(3))) double2_t*>(dst[0] + c) = v;
        }
      }
      if constexpr (NA == 2) {
        if (jb == LEAF / SB - 1) {
          // The last block: its L (lane r holds row r of X~) and its tile of M (lane i holds row i of Y = column i of the
          // block's inverse up to scalings) go to memory straight from the registers, while the helpers form the rest of the
          // last block row of M:  L[r][c] = X~[r][c] d_c^-1/2,  M[J][i] = d_J^-1/2 Y[i][J]
          const double rs = fast_rsqrt(my_p);
          const unsigned long long neg = __ballot(!(my_p > 0.0)) & 0xffffull;
          if (neg != 0 && lane == 0) atomicMin(info, col0 + j0 + __ffsll((long long)neg));
#pragma unroll
          for (int c = 0; c < SB; ++c) {
            const double rc = bcast_lane(rs, c);
            if (lane < SB) {
              if (c <= r16) Ablk[(long)(j0 + r16) * lda + j0 + c] = a[0][c] * rc;
              const double m = a[1][c] * rc;  // M[j0 + c][j0 + lane]
              minv[minv_index(j0 + c, j0 + lane)] = m;
              Mh[mt_off(LEAF / SB - 1, LEAF / SB - 1) + 64 * (c >> 2) + 16 * (c & 3) + lane] = m;
            }
          }
        }
      }
      LEAF_PROBE(8 * jb + 3);
    };
    // rank-16 update of the next diagonal tile: S[r0.., r0..] -= (X~ D^-1) X~^T with the rows r0.. of columns j0..j0+15;
    // four independent accumulators (a dependent fp64 MFMA follows its predecessor after ~250 cycles)
    auto update_diag_tile = [&](int jb) {
      const int j0 = jb * SB, r0 = j0 + SB;
      const double4_t z4 = {0.0, 0.0, 0.0, 0.0};
      double4_t acc, p1 = z4, p2 = z4, p3 = z4;
      double av[4], bv[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        bv[s4] = S[soff(r0 + nn) + j0 + 4 * s4 + kq];
        av[s4] = bv[s4] * nra[j0 + 4 * s4 + kq];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = S[soff(r0 + kq + 4 * r) + r0 + nn];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], p3, 0, 0, 0);
      acc = (acc + p1) + (p2 + p3);
#pragma unroll
      for (int r = 0; r < 4; ++r) S[soff(r0 + kq + 4 * r) + r0 + nn] = acc[r];
    };
    for (int jb = 0; jb < LEAF / SB; ++jb) {
      // (the identity rows ride in the last panel's idle lanes; blocks 3 and 7 have none and take a panel for them)
      if (jb <= 3) iteration(std::integral_constant<int, 3>(), jb);
      else iteration(std::integral_constant<int, 2>(), jb);
      if (jb + 1 < LEAF / SB) {
        if (jb == 0) {  // the helpers' part of the LDS image (every column right of block 0)
          while (sync_w[2] < roles::N_HELPERS) __builtin_amdgcn_s_sleep(1);
          wave_lds_fence();
          LEAF_EXPECT(sync_w[2] == roles::N_HELPERS, 2);
        }
        update_diag_tile(jb);
        wave_lds_fence();
        LEAF_PROBE(8 * jb + 4);
        if (jb < roles::URGENT_ITERS) {
          while (sync_w[1] < roles::N_HELPERS * (jb + 1)) {}
          wave_lds_fence();
          // (no helper passes its wait for block jb + 1 before this wave publishes it: exactly the arrivals of blocks 0 .. jb)
          LEAF_EXPECT(sync_w[1] == roles::N_HELPERS * (jb + 1), 3);
        }
        LEAF_PROBE(8 * jb + 5);
      }
    }
  } else {
    // ---------------------------------------------------------------- the helpers
    const int t = tid - 64 * roles::FIRST_HELPER;
    constexpr int NHT = 64 * roles::N_HELPERS;  // helper threads
    if (roles::is_helper(wave)) {  // (the helpers; the inverse's waves start at the first published block)
      // the LDS image right of column block 0: row block b holds 16 rows x 8 b pieces of 16 bytes there
      double2_t v[21];  // sum over row blocks of ceil(128 b / NHT)
      static_assert(NHT == 192, "v[] is sized for 192 loading threads");
      int u = 0;
#pragma unroll
      for (int bb = 1; bb < 8; ++bb) {
        const int per = 8 * bb, cnt = 16 * per;
#pragma unroll
        for (int idx0 = 0; idx0 < cnt; idx0 += NHT) {
          const int idx = idx0 + t;
          if (idx < cnt) v[u] = *reinterpret_cast<const double2_t*>(Ablk + (long)(16 * bb + idx / per) * lda + 16 + 2 * (idx % per));
          ++u;
        }
      }
      u = 0;
#pragma unroll
      for (int bb = 1; bb < 8; ++bb) {
        const int per = 8 * bb, cnt = 16 * per;
#pragma unroll
        for (int idx0 = 0; idx0 < cnt; idx0 += NHT) {
          const int idx = idx0 + t;
          if (idx < cnt) *reinterpret_cast<double2_t*>(S + soff(16 * bb + idx / per) + 16 + 2 * (idx % per)) = v[u];
          ++u;
        }
      }
      wave_lds_fence();
      if (lane == 0) lds_add(sync_a + 2, 1);
      while (sync_w[2] < roles::N_HELPERS) __builtin_amdgcn_s_sleep(1);
      wave_lds_fence();
      LEAF_EXPECT(sync_w[2] == roles::N_HELPERS, 4);
    }
    LEAF_PROBE(62);
    // Register-resident trailing tiles (see namespace lt).  Per phase a wave reads one operand set (4 doubles per lane) per
    // block row it touches -- X~ rows: as they are the MFMA B operand of the tiles in that block column, times -1 / d_k
    // the A operand of the tiles in that block row -- and issues its tiles' MFMAs interleaved.
    double4_t tacc[9];
    auto trailing = [&](auto JBc, auto Wc, auto PHc) {
      constexpr int JB = decltype(JBc)::value, W = decltype(Wc)::value, PH = decltype(PHc)::value;
      double opa[8][4], opb[8][4], nr[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) nr[s4] = nra[16 * JB + 4 * s4 + kq];
      lt::static_for<1, 8>([&](auto Bc) {
        constexpr int B = decltype(Bc)::value;
        if constexpr (lt::needs_a(W, JB, B, PH) || lt::needs_b(W, JB, B, PH)) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) opb[B][s4] = S[soff(16 * B + nn) + 16 * JB + 4 * s4 + kq];
        }
        if constexpr (lt::needs_a(W, JB, B, PH)) {
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) opa[B][s4] = opb[B][s4] * nr[s4];
        }
      });
      if constexpr (PH == 0) {
        // the tiles wave 0 waits for: two accumulator chains of two MFMAs per tile, results to the LDS image
        double4_t t2[9];
        lt::static_for<0, 9>([&](auto Sc) {
          constexpr int Sl = decltype(Sc)::value, I = 3 * Sl + W;
          if constexpr (lt::sel(I, JB, PH)) t2[Sl] = (double4_t){0.0, 0.0, 0.0, 0.0};
        });
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4)
          lt::static_for<0, 9>([&](auto Sc) {
            constexpr int Sl = decltype(Sc)::value, I = 3 * Sl + W;
            if constexpr (lt::sel(I, JB, PH)) {
              tacc[Sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(opa[lt::TR[I]][s4], opb[lt::TC[I]][s4], tacc[Sl], 0, 0, 0);
              t2[Sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(opa[lt::TR[I]][s4 + 2], opb[lt::TC[I]][s4 + 2], t2[Sl], 0, 0, 0);
            }
          });
        lt::static_for<0, 9>([&](auto Sc) {
          constexpr int Sl = decltype(Sc)::value, I = 3 * Sl + W;
          if constexpr (lt::sel(I, JB, PH)) put(tacc[Sl] + t2[Sl], lt::TR[I], lt::TC[I]);
        });
      } else {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
          lt::static_for<0, 9>([&](auto Sc) {
            constexpr int Sl = decltype(Sc)::value, I = 3 * Sl + W;
            if constexpr (lt::sel(I, JB, PH))
              tacc[Sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(opa[lt::TR[I]][s4], opb[lt::TC[I]][s4], tacc[Sl], 0, 0, 0);
          });
      }
    };
    auto load_tiles = [&](auto Wc) {
      constexpr int W = decltype(Wc)::value;
      lt::static_for<0, 9>([&](auto Sc) {
        constexpr int I = 3 * decltype(Sc)::value + W;
#pragma unroll
        for (int r = 0; r < 4; ++r) tacc[decltype(Sc)::value][r] = S[soff(16 * lt::TR[I] + kq + 4 * r) + 16 * lt::TC[I] + nn];
      });
    };
    // Column block cb is final for rows >= 16 cb once wave 0 has published it: L = X~ D^-1/2 goes out to memory, 8 pieces
    // of 16 B per row (the LDS image keeps X~).  A thread's pieces all lie in ONE column pair (192 is a multiple of 8): two
    // reciprocal square roots per thread.
    auto stream_out = [&](int cb) {  // at most 6 pieces per thread: all LDS reads first, then the stores (one round trip)
      const int c0 = cb * SB, npiece = (LEAF - c0) * 8;
      const int cp = c0 + 2 * (t & 7);
      const double d0 = pvt[cp], d1 = pvt[cp + 1];
      const double rs0 = fast_rsqrt(d0), rs1 = fast_rsqrt(d1);
      if (t < 8) {
        int bad = 0;
        if (!(d1 > 0.0)) bad = cp + 2;
        if (!(d0 > 0.0)) bad = cp + 1;
        if (bad != 0) atomicMin(info, col0 + bad);
      }
      double2_t v[6];
      bool full[6], half[6];
      double* dst[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int it = t + NHT * u;
        const int r = c0 + (it >> 3), c2 = c0 + 2 * (it & 7);
        const bool in = it < npiece && c2 <= r;
        full[u] = in && c2 + 1 <= r;
        half[u] = in && c2 + 1 > r;
        dst[u] = Ablk + (long)r * lda + c2;
        if (in) v[u] = *reinterpret_cast<const double2_t*>(S + soff(r) + c2);
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        v[u].x *= rs0;
        v[u].y *= rs1;
        if (full[u]) *reinterpret_cast<double2_t*>(dst[u]) = v[u];
        else if (half[u]) dst[u][0] = v[u].x;
      }
    };
    // ---- M = L^-1 (16x16 tiles in operand order in `minv`, see minv_index), block row by block row, in the helpers' idle time.
    // With L = X~ D^-1/2 (X~ the unnormalised columns, D = diag(d)) and Y_b = the identity rows wave 0 carried through block b
    // (Y_b[i][J] = (L~_bb^-T)[i][J] with L~_bb = X~_bb D_b^-1 unit lower triangular), M^ = D^-1/2 M obeys
    //   M^[b][b] = D_b^-1 Y_b^T,     M^[b][c] = -D_b^-1 Y_b^T  T[b][c],   T[b][c] = sum_{k = c}^{b - 1} X~[b][k] M^[k][c]   (c < b)
    // -- products of tiles that are in the LDS image unnormalised, so a block row never waits for a reciprocal square root;
    // M = D^1/2 M^ is scaled on its way to memory.  Wave w takes the block columns c = w, w + 3, w + 6.  T[b][.] is formed
    // one iteration ahead (everything but Y_b is there), so behind the last pivot only the 16x16x16 products with Y_7 remain.
    // An MFMA result (lane: [kq + 4 r][n]) IS the B operand of the next product (lane: [4 s + kq][n]): T goes from the
    // accumulators of the first product into the second, and M^ tiles are stored in that order ([r][lane]: no bank conflicts).
    double4_t tsum[3], tbulk[3];  // T[b][c] of this wave's columns for the next block row (complete) / the one after (terms k <= b - 2)
    // acc[q] += sum_{k = KLO}^{KHI - 1} X~[B][k] M^[k][c] for this wave's columns c = W + 3 q <= k: k-outer, the columns
    // interleaved, four partial accumulators per tile -- up to twelve independent MFMA chains (a dependent fp64 MFMA follows
    // its predecessor after ~250 cycles) and one read of the X~[B][k] operand per k
    auto inv_acc = [&](auto Bc, auto Wc, auto KLOc, auto KHIc, double4_t (&acc)[3]) {
      constexpr int B = decltype(Bc)::value, W = decltype(Wc)::value, KLO = decltype(KLOc)::value, KHI = decltype(KHIc)::value;
      if constexpr (W < KHI && KLO < KHI) {
        const double4_t z4 = {0.0, 0.0, 0.0, 0.0};
        double4_t p[3][4];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) p[q][s4] = z4;
        lt::static_for<(KLO > W ? KLO : W), KHI>([&](auto Kc) {
          constexpr int K = decltype(Kc)::value;
          double xa[4];
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) xa[s4] = S[soff(16 * B + nn) + 16 * K + 4 * s4 + kq];
          lt::static_for<0, 3>([&](auto Qc) {
            constexpr int Q = decltype(Qc)::value, C = W + 3 * Q;
            if constexpr (C <= K) {
#pragma unroll
              for (int s4 = 0; s4 < 4; ++s4)
                p[Q][s4] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[s4], Mh[mt_off(K, C) + 64 * s4 + lane], p[Q][s4], 0, 0, 0);
            }
          });
        });
        lt::static_for<0, 3>([&](auto Qc) {
          constexpr int Q = decltype(Qc)::value, C = W + 3 * Q;
          if constexpr (C < KHI) acc[Q] += (p[Q][0] + p[Q][1]) + (p[Q][2] + p[Q][3]);
        });
      }
    };
    auto inv_final = [&](auto Bc, auto Wc) {  // block row B of M^ and of M for this wave's columns c < B (and the diagonal tile)
      constexpr int B = decltype(Bc)::value, W = decltype(Wc)::value;
      constexpr bool LAST = B == LEAF / SB - 1;  // nothing reads M^ of the last block row: M itself, and wave 0 takes its diagonal tile
      if constexpr (W < B || (W == B % 3 && !LAST)) {
        const double* Y = Ybuf + (B & 3) * (SB * YB_LD);
        // A operand of the second product: (-D_b^-1 Y_b^T)[n][4 s + kq] = nra[16 B + n] * Y[4 s + kq][n]; for the last block row
        // the scaling of M's rows by sqrt(d) is folded in: -d^-1/2 instead of -1 / d (one reciprocal square root per lane)
        const double nrn = LAST ? -fast_rsqrt(pvt[16 * B + nn]) : nra[16 * B + nn];
        double ya[4], sq[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) ya[s4] = nrn * Y[(4 * s4 + kq) * YB_LD + nn];
        if constexpr (!LAST) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {  // rows kq + 4 r of block row B are scaled by sqrt(d) on their way out
            const double d = pvt[16 * B + kq + 4 * r];
            sq[r] = d * fast_rsqrt(d);
          }
        }
        lt::static_for<0, 3>([&](auto Qc) {
          constexpr int Q = decltype(Qc)::value, C = W + 3 * Q;
          if constexpr (C < B) {
            const double4_t z4 = {0.0, 0.0, 0.0, 0.0};
            double4_t m0 = z4, m1 = z4;
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[0], tsum[Q][0], m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[1], tsum[Q][1], m1, 0, 0, 0);
            m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[2], tsum[Q][2], m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[3], tsum[Q][3], m1, 0, 0, 0);
            m0 += m1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              Mh[mt_off(B, C) + 64 * r + lane] = m0[r];
              minv[minv_index(16 * B + kq + 4 * r, 16 * C + nn)] = LAST ? m0[r] : m0[r] * sq[r];
            }
          } else if constexpr (C == B && !LAST) {  // the diagonal tile: D_b^-1 Y_b^T, element [J][i] = -nra[J] Y[i][J]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int J = kq + 4 * r;
              const double v = -nra[16 * B + J] * Y[nn * YB_LD + J];
              Mh[mt_off(B, B) + 64 * r + lane] = v;
              minv[minv_index(16 * B + J, 16 * B + nn)] = v * sq[r];
            }
          }
        });
      }
      wave_lds_fence();
      if (lane == 0) lds_add(sync_a + 3, 1);
    };
    // T of the next block row gets its last term (k = JB: block row JB of M^ is the newest, every helper has to be through
    // with it), T of the one after its terms k <= JB
    auto inv_t = [&](auto JBc, auto Wc) {
      constexpr int JB = decltype(JBc)::value, NB = LEAF / SB;
      const double4_t z4 = {0.0, 0.0, 0.0, 0.0};
      if constexpr (JB + 1 < NB) {
        while (sync_w[3] < roles::N_INVERSE * (JB + 1)) __builtin_amdgcn_s_sleep(1);
        wave_lds_fence();
        // (a faster wave may have delivered block row JB + 1 already -- not JB + 2: that needs this wave's row JB + 1)
        LEAF_EXPECT(sync_w[3] < roles::N_INVERSE * (JB + 2), 5);
#pragma unroll
        for (int q = 0; q < 3; ++q) tsum[q] = tbulk[q];
        inv_acc(std::integral_constant<int, JB + 1>(), Wc, JBc, std::integral_constant<int, JB + 1>(), tsum);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) tbulk[q] = z4;
      if constexpr (JB + 2 < NB)
        inv_acc(std::integral_constant<int, JB + 2>(), Wc, std::integral_constant<int, 0>(), std::integral_constant<int, JB + 1>(), tbulk);
    };
    auto helper = [&](auto Wc) {
      load_tiles(Wc);
      // fully unrolled: every copy sees a constant jb, the tile accumulators have plain live ranges (no loop-carried phis
      // through a switch: that form cost 200 registers of copies and moved the accumulators to AGPRs)
      lt::static_for<0, LEAF / SB>([&](auto JBc) {
        constexpr int JB = decltype(JBc)::value;
        while (sync_w[0] < JB + 1) __builtin_amdgcn_s_sleep(1);  // (a tight spin takes issue slots from the inverse wave on this SIMD)
        wave_lds_fence();
        // (the chain publishes block JB + 1 only behind this wave's urgent arrival for block JB, while there are urgent tiles)
        LEAF_EXPECT(JB < roles::URGENT_ITERS ? sync_w[0] == JB + 1 : sync_w[0] <= LEAF / SB, 6);
        LEAF_PROBE(64 + 8 * JB + 0);
        if constexpr (JB < roles::URGENT_ITERS) {
          trailing(JBc, Wc, std::integral_constant<int, 0>());
          wave_lds_fence();
          if (lane == 0) lds_add(sync_a + 1, 1);
          LEAF_PROBE(64 + 8 * JB + 1);
          trailing(JBc, Wc, std::integral_constant<int, 1>());
          LEAF_PROBE(64 + 8 * JB + 2);
        }
        if constexpr (JB >= 1) stream_out(JB - 1);
        LEAF_PROBE(64 + 8 * JB + 4);
      });
      // (wave 0 writes the last diagonal block of L and of M from its registers)
    };
    // The inverse's waves: the workgroup has eight waves, and wave 5 + w shares SIMD 1 + w with helper w (the waves of a
    // workgroup are dealt to the SIMDs round robin) -- the hardware interleaves the two instruction streams, so the inverse
    // fills the issue slots the helper leaves and never delays an urgent tile.  (As ONE instruction stream the helpers
    // reached the urgent phases of iterations 4 and 5 ~2k cycles late: N = 2048 0.746 -> 0.735 ms, N = 4096 1.660 -> 1.633,
    // N = 8192 4.93 -> 4.89, same bits.)
    auto inverse_worker = [&](auto Wc) {
#pragma unroll
      for (int q = 0; q < 3; ++q) tbulk[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
      lt::static_for<0, LEAF / SB>([&](auto JBc) {
        constexpr int JB = decltype(JBc)::value;
        while (sync_w[0] < JB + 1) __builtin_amdgcn_s_sleep(1);
        wave_lds_fence();
        // (the chain re-uses identity-row buffer JB & 3 for block JB + 4 only behind this wave's block row JB)
        LEAF_EXPECT(sync_w[0] <= JB + 4 && sync_w[0] <= LEAF / SB, 7);
        LEAF_PROBE(128 + 8 * JB + 0);
        inv_final(JBc, Wc);
        LEAF_PROBE(128 + 8 * JB + 1);
        inv_t(JBc, Wc);
        LEAF_PROBE(128 + 8 * JB + 2);
      });
    };
    if (wave == roles::FIRST_HELPER) helper(std::integral_constant<int, 0>());
    else if (wave == roles::FIRST_HELPER + 1) helper(std::integral_constant<int, 1>());
    else if (wave == roles::FIRST_HELPER + 2) helper(std::integral_constant<int, 2>());
    else if (wave == roles::FIRST_INVERSE) inverse_worker(std::integral_constant<int, 0>());
    else if (wave == roles::FIRST_INVERSE + 1) inverse_worker(std::integral_constant<int, 1>());
    else if (wave == roles::FIRST_INVERSE + 2) inverse_worker(std::integral_constant<int, 2>());
    // (the idle wave would share SIMD 0 with the chain: it does nothing)
  }
  LEAF_PROBE(63);
#ifdef LEAF_CHECKED
  // every wave is through: the counters hold exactly one arrival per executing wave and phase
  __syncthreads();
  LEAF_EXPECT(sync_w[0] == LEAF / SB, 8);
  LEAF_EXPECT(sync_w[1] == roles::N_HELPERS * roles::URGENT_ITERS, 9);
  LEAF_EXPECT(sync_w[2] == roles::N_HELPERS, 10);
  LEAF_EXPECT(sync_w[3] == roles::N_INVERSE * (LEAF / SB), 11);
#endif
  // Last tile column of an evaluation: the only rows below are the y^T row block (one non-zero row), so the forward
  // solve of these 128 columns, beta = y M^T, is done here against the inverse that is still in LDS -- the strip launch
  // for that block (6 us of launch and round trips for 16k flops) is skipped.
  if (yrow != nullptr) {
    __syncthreads();
    double* ys = Ybuf;  // scratch
    if (tid < LEAF) ys[tid] = yrow[tid];
    __syncthreads();
    const int c = (tid & 255) >> 1, half = tid & 1;  // (waves 4..7 repeat waves 0..3: same values, same addresses)
    // M[c][k] = sqrt(d_c) M^[c][k]; M^ element [i][j] of tile (b, cb) sits at mt_off(b, cb) + 64 (i >> 2) + 16 (i & 3) + j
    const int b = c >> 4, i = c & 15;
    const double* mrow = Mh + mt_off(b, 0) + 64 * (i >> 2) + 16 * (i & 3);
    double acc = 0.0;
    for (int k = half; k <= c; k += 2) acc = __builtin_fma(mrow[(k >> 4) * MT_ELEMS + (k & 15)], ys[k], acc);
    acc += __shfl_xor(acc, 1);
    const double d = pvt[c];
    if (half == 0) yrow[c] = b == LEAF / SB - 1 ? acc : acc * (d * fast_rsqrt(d));  // (the last block row is stored scaled)
  }
}

// wait_ptr (optional): a cross-stream signal this launch has to see at wait_val or above before it ends (the driver folds
// the panel stream's wait for the main stream's next-panel update into the leaf that precedes the first reader of those
// columns: one polling lane at the end of a kernel that is a single workgroup anyway, instead of a runtime wait kernel of
// 5-9 us on the chain).  The poll gives up after ~2^22 sleeps (seconds) and reports through the bad-pivot word -- and where that
// word already says that an earlier poll of this evaluation gave up, it does not wait at all: the evaluation is lost, the
// library re-runs it with event edges (api_gp.hip), and ONE limit is all the time that costs.
__device__ __forceinline__ void poll_signal(const unsigned* ptr, unsigned val, int* info, int limit_log2) {
  long spins = 0;
  const long limit = 1L << limit_log2;
  if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == SIGNAL_TIMEOUT_INFO) return;
  while (__hip_atomic_load(ptr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < val) {
    __builtin_amdgcn_s_sleep(4);
    if (++spins > limit) {
      atomicMin(info, SIGNAL_TIMEOUT_INFO);
      break;
    }
  }
}

__global__ __launch_bounds__(512, 1) void potrf_leaf128_kernel(double* __restrict__ Ablk, long lda,
                                                                double* __restrict__ minv, int col0,
                                                                int* __restrict__ info, double* yrow, long sA, long sminv,
                                                                int sinfo, const unsigned* wait_ptr, unsigned wait_val,
                                                                int poll_log2, unsigned* start_wr) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __builtin_amdgcn_s_setprio(3);  // the leaf is the panel chain: win the issue arbitration against bulk GEMM waves on its CU
  const long z = blockIdx.x;  // batched evaluation: one workgroup per problem
  // start_wr (optional): "everything queued on this stream before me is done, and I have my CU" -- the main stream's next
  // update waits for it, so that it does not fill the chip in front of this workgroup (column mode, api_gp.hip)
  if (start_wr != nullptr && z == 0 && threadIdx.x == 0) __hip_atomic_store(start_wr, wait_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  potrf_leaf128_body(Ablk + z * sA, lda, minv + z * sminv, col0, info + z * sinfo, smem, yrow ? yrow + z * sA : nullptr);
  if (wait_ptr != nullptr && threadIdx.x == 0) poll_signal(wait_ptr, wait_val, info + z * sinfo, poll_log2);
}

// One lane: raise *wr to `val` (if wr) and then wait for *wt >= val (if wt).  The panel stream's two edges at a
// super-panel boundary -- tell the main stream the panel is done, wait for the main stream's previous bulk update -- in
// ONE launch instead of two runtime kernels (hipStreamWriteValue32 + hipStreamWaitValue32, ~5 us each on the chain).
// A poll that gives up marks the bad-pivot word of EVERY problem of a batch (nb words, sinfo apart).
__global__ void signal_write_wait_kernel(unsigned* wr, const unsigned* wt, unsigned val, int* info, int nb, int sinfo, int poll_log2) {
  if (threadIdx.x == 0) {
    if (wr != nullptr) __hip_atomic_store(wr, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (wt != nullptr) {
      int local = __hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      poll_signal(wt, val, &local, poll_log2);
      if (local == SIGNAL_TIMEOUT_INFO)
        for (int p = 0; p < nb; ++p) atomicMin(info + (long)p * sinfo, SIGNAL_TIMEOUT_INFO);
    }
  }
}

hipError_t launch_signal_write_wait(unsigned* wr, const unsigned* wt, unsigned val, int* info, hipStream_t stream, int nb, int sinfo,
                                    int poll_log2) {
  signal_write_wait_kernel<<<1, 64, 0, stream>>>(wr, wt, val, info, nb, sinfo, poll_log2);
  return hipGetLastError();
}

// X * L^T = B in place on B (m x 128, leading dimension ldb even, m multiple of 16 RG) as X = B * M^T with M = L^-1
// (128 x 128 lower triangular in the leaf's tile order -- minv_index --, zeros above the diagonal of its diagonal 16x16 tiles).
// One workgroup per 16 RG rows; wave w owns the output column blocks {w, 7 - w} (9 of the 36 lower k-blocks each: the
// triangle is split evenly), 36 MFMAs per wave and row group.  Within a 16-wide k-block the lane quarter q = lane >> 4
// covers k = 16 kb + 4 q + s (s = the MFMA step), so every lane fetches 4 CONTIGUOUS doubles of its row of B and of its
// row of M per k-block: 128-byte row segments, whole cache lines, straight into MFMA operand registers -- no LDS.
// RG row groups per workgroup reuse the wave's 9 tiles of M (RG = 1: lowest latency, 16 rows per workgroup; RG = 4: a
// quarter of the operand traffic for tall panels).  A row group's rows are in registers in every wave (s_waitcnt +
// barrier) before any wave stores to them: in place is safe.
// (Round 1 also had a fused leaf + strip launch whose strip workgroups spun on a flag of the leaf workgroup: with the
// strip down to one round trip the in-launch release / acquire hand-off costs more than the launch boundary it saved,
// and it was the only inter-workgroup wait in the library -- removed in round 2.)
// lsw (optional): the first lsw_blocks 16-row groups of the result are ALSO written there, 128 rows per 16384 doubles in the
// operand order of minv_index -- the B operand of the thin update that follows (thin_f64.hip) as coalesced loads.
template <int RG>
__device__ __forceinline__ void trsm_strip128_body(const double* __restrict__ minv, double* __restrict__ B, long ldb, int blk,
                                                   double* __restrict__ lsw, int lsw_blocks) {
  typedef double double2_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int jb0 = wave, jb1 = 7 - wave;  // jb0 < jb1; k-blocks 0 .. jb1 are needed
  double2_t a[8][2];  // one row group's operands
  // The workgroup's 16 RG x 128 rows come in as 1 KB row loads (four per wave and row group) and reach operand order through
  // LDS (rows 130 doubles apart; PMC: ~40 % of the few LDS cycles are still bank conflicts, ~160 cycles per workgroup -- profiles/r05_pmc_lds_conflicts.txt).  Loaded straight into operand
  // registers -- every quarter-wave touching 16 rows -- the kernel was bound by the texture addresser (round 5; RG > 1 until
  // then also refilled a group's registers behind their last MFMA and paid a barrier per group for storing in place).
  __shared__ __attribute__((aligned(16))) double As[RG * 16 * 130];
  double2_t stage[4 * RG];
#pragma unroll
  for (int i = 0; i < 4 * RG; ++i)
    stage[i] = *reinterpret_cast<const double2_t*>(B + ((long)blk * (16 * RG) + 4 * RG * wave + i) * ldb + 2 * lane);
  // M tiles (jb, kb), kb <= jb, in operand order (minv_index): (jb0 + 1) + (jb1 + 1) = 9 tiles, two coalesced 1 KB loads each
  const double* m0 = minv + (long)(jb0 * 8) * 256 + 2 * lane;
  const double* m1 = minv + (long)(jb1 * 8) * 256 + 2 * lane;
  double2_t b0[4][2], b1[8][2];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (kb <= jb0) {
      b0[kb][0] = *reinterpret_cast<const double2_t*>(m0 + 256 * kb);
      b0[kb][1] = *reinterpret_cast<const double2_t*>(m0 + 256 * kb + 128);
    }
  }
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    if (kb <= jb1) {
      b1[kb][0] = *reinterpret_cast<const double2_t*>(m1 + 256 * kb);
      b1[kb][1] = *reinterpret_cast<const double2_t*>(m1 + 256 * kb + 128);
    }
  }
#pragma unroll
  for (int i = 0; i < 4 * RG; ++i) *reinterpret_cast<double2_t*>(As + (4 * RG * wave + i) * 130 + 2 * lane) = stage[i];
  __syncthreads();  // (every wave's rows have left memory: storing to them in place is safe from here on)
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    // Four partial accumulators per output tile, one per MFMA step of a k-block (round 4): a dependent fp64 MFMA follows
    // its predecessor after ~250 cycles but an independent one after 64, and with ONE accumulator per tile the wave that
    // owns column block 7 ran a chain of 32 per row group (8000 of the ~16000 cycles of a one-group strip, and nearly all
    // of a four-group strip's 16 us).  Now the longest chain is 8 deep and the partial sums are added pairwise at the end.
    const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
    double4_t p1[4] = {zero4, zero4, zero4, zero4}, p0[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      if (kb <= jb1) {
        a[kb][0] = *reinterpret_cast<const double2_t*>(As + (16 * rg + n) * 130 + 16 * kb + 4 * q);
        a[kb][1] = *reinterpret_cast<const double2_t*>(As + (16 * rg + n) * 130 + 16 * kb + 4 * q + 2);
      }
    }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      if (kb <= jb1) {
        p1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb][0].x, b1[kb][0].x, p1[0], 0, 0, 0);
        p1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb][0].y, b1[kb][0].y, p1[1], 0, 0, 0);
        p1[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb][1].x, b1[kb][1].x, p1[2], 0, 0, 0);
        p1[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb][1].y, b1[kb][1].y, p1[3], 0, 0, 0);
      }
      if (kb < 4 && kb <= jb0) {
        p0[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb][0].x, b0[kb][0].x, p0[0], 0, 0, 0);
        p0[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb][0].y, b0[kb][0].y, p0[1], 0, 0, 0);
        p0[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb][1].x, b0[kb][1].x, p0[2], 0, 0, 0);
        p0[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb][1].y, b0[kb][1].y, p0[3], 0, 0, 0);
      }
    }
    const double4_t x0 = (p0[0] + p0[1]) + (p0[2] + p0[3]), x1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
    // D layout: lane holds X[16 (blk RG + rg) + q + 4 r][16 jb + n]
    double* out = B + ((long)blk * (16 * RG) + 16 * rg + q) * ldb + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      out[(long)(4 * r) * ldb + 16 * jb0] = x0[r];
      out[(long)(4 * r) * ldb + 16 * jb1] = x1[r];
    }
    const int g = blk * RG + rg;  // 16-row group of the strip
    if (lsw != nullptr && g < lsw_blocks) {
      double* sw = lsw + (long)(g >> 3) * (LEAF * LEAF);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sw[minv_index(16 * (g & 7) + q + 4 * r, 16 * jb0 + n)] = x0[r];
        sw[minv_index(16 * (g & 7) + q + 4 * r, 16 * jb1 + n)] = x1[r];
      }
    }
  }
}

#ifndef MIGP_XCD_MAP_STRIP
#define MIGP_XCD_MAP_STRIP 1
#endif
// one launch serves `gridDim.y` independent (M, B) pairs: M at minv + y * 16384, B at B + y * strideB
template <int RG>
__global__ __launch_bounds__(256) void trsm_strip128_kernel(const double* __restrict__ minv, double* __restrict__ B, long ldb,
                                                             long strideB, long sminv2, long sB2, double* __restrict__ lsw,
                                                             int lsw_blocks) {
  __builtin_amdgcn_s_setprio(3);
  // Row groups -> XCDs in contiguous ranges (workgroup b runs on XCD b % 8), the same way the GEMM kernels map their tile
  // rows: the update that follows reads this strip's rows, and the next strip reads what that update wrote, out of the L2
  // that already holds them.
  int blk = (int)blockIdx.x;
  if (MIGP_XCD_MAP_STRIP) {
    const int nblk = (int)gridDim.x, x = blk & 7, q = nblk >> 3, r = nblk & 7;
    blk = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (blk >> 3);
  }
  // blockIdx.z: problem of a batched evaluation (second batch level)
  trsm_strip128_body<RG>(minv + (long)blockIdx.y * (LEAF * LEAF) + (long)blockIdx.z * sminv2,
                         B + (long)blockIdx.y * strideB + (long)blockIdx.z * sB2, ldb, blk,
                         lsw ? lsw + (long)blockIdx.z * sminv2 : nullptr, lsw_blocks);
}

// The leaf asks for more LDS than it uses, so that no 72 KB GEMM workgroup fits beside it on a CU: since round 5 its
// registers (152) would fit beside a bulk wave on a SIMD, and a chain wave that shares the SIMD's fp64 pipe with MFMA-saturated
// waves runs about half as fast (N = 16384: 25.8 -> 26.2 ms when the two were allowed to share).
constexpr size_t LEAF_LDS_BYTES = sizeof(double) * (LEAF_ELEMS + 2 * LEAF + 4 * SB * YB_LD + 36 * MT_ELEMS + 4);
static_assert(LEAF_LDS_BYTES > 96 * 1024 && LEAF_LDS_BYTES <= 160 * 1024, "the leaf takes a CU's LDS to itself");

hipError_t leaf_enable_lds() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_leaf128_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS_BYTES);
  return e;
}

hipError_t launch_potrf_leaf128(double* Ablk, long lda, double* minv, int col0, int* info, hipStream_t stream, double* yrow,
                                const Batch* bt, const unsigned* wait_ptr, unsigned wait_val, int poll_log2, unsigned* start_wr) {
  potrf_leaf128_kernel<<<bt ? bt->nb : 1, 512, LEAF_LDS_BYTES, stream>>>(Ablk, lda, minv, col0, info, yrow, bt ? bt->sK : 0,
                                                                        bt ? bt->sdinv : 0, bt ? bt->sinfo : 0, wait_ptr, wait_val,
                                                                        poll_log2, start_wr);
  return hipGetLastError();
}

// rows per workgroup by panel height: 16 while one round of workgroups covers the panel (lowest latency), 32 / 64 for
// tall panels (the wave's tiles of M are reused, 1/2 and 1/4 of the operand traffic); m is a multiple of 64 or of 16
hipError_t launch_trsm_strip128_batched(const double* minv, double* B, long ldb, long strideB, int m, int batch,
                                        hipStream_t stream, const Batch* bt, long sB2, double* lsw, int lsw_blocks) {
  if (m <= 0 || batch <= 0) return hipSuccess;
  const int nb = bt ? bt->nb : 1;
  const long sm2 = bt ? bt->sdinv : 0;
  const long rows = (long)m * batch * nb;
  if (rows > 8192 && m % 64 == 0) trsm_strip128_kernel<4><<<dim3(m / 64, batch, nb), 256, 0, stream>>>(minv, B, ldb, strideB, sm2, sB2, lsw, lsw_blocks);
  else if (rows > 4096 && m % 32 == 0) trsm_strip128_kernel<2><<<dim3(m / 32, batch, nb), 256, 0, stream>>>(minv, B, ldb, strideB, sm2, sB2, lsw, lsw_blocks);
  else trsm_strip128_kernel<1><<<dim3(m / 16, batch, nb), 256, 0, stream>>>(minv, B, ldb, strideB, sm2, sB2, lsw, lsw_blocks);
  return hipGetLastError();
}

hipError_t launch_trsm_strip128(const double* minv, double* B, long ldb, int m, hipStream_t stream, const Batch* bt, long sB2,
                                double* lsw, int lsw_blocks) {
  return launch_trsm_strip128_batched(minv, B, ldb, 0, m, 1, stream, bt, sB2, lsw, lsw_blocks);
}

}  // namespace migp
