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
__device__ unsigned long long g_leaf_stamps[32];
#define LEAF_STAMP(i)                                                                  \
  do {                                                                                 \
    if (threadIdx.x == 0) {                                                            \
      unsigned long long t_;                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
      g_leaf_stamps[i] += t_ - t_prev;                                                 \
      t_prev = t_;                                                                     \
    }                                                                                  \
  } while (0)
// the same for the first thread of wave 1 (slots 8..15)
#define LEAF_STAMP1(i)                                                                 \
  do {                                                                                 \
    if (threadIdx.x == 64) {                                                           \
      unsigned long long t_;                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
      g_leaf_stamps[i] += t_ - t_prev1;                                                \
      t_prev1 = t_;                                                                    \
    }                                                                                  \
  } while (0)
#else
#define LEAF_STAMP(i)
#define LEAF_STAMP1(i)
#endif

// LDS words the waves of the leaf meet through, as LDS instructions: through a generic `volatile int*` the compiler emits
// FLAT loads / stores with system-scope cache bits and vmcnt waits (round 4's listing), and a workgroup-scope fence also
// waits for the wave's outstanding GLOBAL stores (the column stream-out).
typedef __attribute__((address_space(3))) int lds_int_t;
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
//      16-column panels per lane), eliminates the 16 columns, writes the unnormalised columns X~ to the LDS image and the
//      multipliers W = -X~ D^-1 to a side buffer, raises `ready`, applies the rank-16 update  += W X~^T  to the NEXT
//      diagonal tile on MFMA and waits for the helpers' urgent tiles -- the rest of the next panel.
//  waves 1..3 (helpers): keep the 27 trailing 16x16 tiles in registers (namespace lt), left-looking.  Behind `ready` they
//      first finish the tiles of the NEXT block column (their last update: results go to the LDS image, `urgent` is
//      raised), then update the other live tiles while wave 0 already eliminates the next block, then normalise the
//      PREVIOUS column block (L = X~ D^-1/2, in the LDS image and out to memory).  No workgroup barrier inside the loop:
//      the sides meet through LDS counters.
__device__ __forceinline__ void potrf_leaf128_body(double* __restrict__ Ablk, long lda, double* __restrict__ minv,
                                                    int col0, int* __restrict__ info, double* smem, double* yrow) {
  double* S = smem;                     // packed lower block-trapezoid, see soff()
  double* invd = smem + LEAF_ELEMS;     // [128] 1 / L[c][c]
  double* pvt = invd + LEAF;            // [128] the pivots d_c (L[c][c]^2)
  double* nrv = pvt + LEAF;             // [2][16] -1 / d_c of the current column block (buffer jb & 1)
  lds_int_t* sync_a = (lds_int_t*)(nrv + 2 * SB);
  volatile lds_int_t* sync_w = sync_a;  // [0] column blocks published by wave 0, [1] urgent
  // arrivals of waves 1..3, [2] their load arrivals, [3] their lazy-phase arrivals
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: tile coordinates and soff() bases derived
                                                              // from it are computed on the scalar unit (v_mul_lo_u32 is quarter rate)
#ifdef LEAF_STAMPS
  unsigned long long t_prev = 0, t_prev1 = 0;
  if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  if (threadIdx.x == 64) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev1)::"memory");
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
      int row[3];
      bool valid[3];
      row[0] = j0 + r16;
      row[1] = j0 + SB + lane;
      row[2] = j0 + SB + 64 + lane;
      valid[0] = lane < SB;
      valid[1] = row[1] < LEAF;
      valid[2] = row[2] < LEAF;
      double a[NA][16];
      if (jb == 0) {  // straight from memory: the chain starts one global round trip after the launch
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const double* src = Ablk + (long)(valid[i] ? row[i] : row[0]) * lda;
#pragma unroll
          for (int c = 0; c < SB; c += 2) {
            const double2_t v = *reinterpret_cast<const double2_t*>(src + c);
            a[i][c] = v.x;
            a[i][c + 1] = v.y;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const double* src = S + soff(valid[i] ? row[i] : row[0]) + j0;
#pragma unroll
          for (int c = 0; c < SB; c += 2) {
            const double2_t v = *reinterpret_cast<const double2_t*>(src + c);
            a[i][c] = v.x;
            a[i][c + 1] = v.y;
          }
        }
      }
      LEAF_STAMP(0);
      __builtin_amdgcn_sched_barrier(0);
      {
        double w0[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) w0[i] = 0.0;
        ch::Step<0, NA>::run(a, ch::mov_bc_padded<0>(a[0][0]), w0);
      }
      LEAF_STAMP(24 + jb);
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
      // X~ of the rows below the block, the pivots and their reciprocals first: the updates wait for them; the block's own
      // rows are read by nobody before the normalisation
#pragma unroll
      for (int i = 1; i < NA; ++i) {
        if (valid[i]) {
          double* dst = S + soff(row[i]) + j0;
#pragma unroll
          for (int c = 0; c < SB; c += 2) {
            double2_t v;
            v.x = a[i][c];
            v.y = a[i][c + 1];
            *reinterpret_cast<double2_t*>(dst + c) = v;
          }
        }
      }
      pvt[j0 + r16] = my_p;  // (the four 16-lane rows write the same values)
      nrv[(jb & 1) * SB + r16] = my_nr;
      wave_lds_fence();
      if (lane == 0) lds_store(sync_w + 0, jb + 1);
      if (lane < SB) {
        double* dst = S + soff(row[0]) + j0;
#pragma unroll
        for (int c = 0; c < SB; c += 2) {
          double2_t v;
          v.x = a[0][c];
          v.y = a[0][c + 1];
          *reinterpret_cast<double2_t*>(dst + c) = v;
        }
      }
      LEAF_STAMP(2);
    };
    // rank-16 update of the next diagonal tile: S[r0.., r0..] -= (X~ D^-1) X~^T with the rows r0.. of columns j0..j0+15;
    // four independent accumulators (a dependent fp64 MFMA follows its predecessor after ~250 cycles)
    auto update_diag_tile = [&](int jb) {
      const int j0 = jb * SB, r0 = j0 + SB;
      const double* nr = nrv + (jb & 1) * SB;
      const double4_t z4 = {0.0, 0.0, 0.0, 0.0};
      double4_t acc, p1 = z4, p2 = z4, p3 = z4;
      double av[4], bv[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        bv[s4] = S[soff(r0 + nn) + j0 + 4 * s4 + kq];
        av[s4] = bv[s4] * nr[4 * s4 + kq];
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
      if (jb <= 2) iteration(std::integral_constant<int, 3>(), jb);
      else if (jb <= 6) iteration(std::integral_constant<int, 2>(), jb);
      else iteration(std::integral_constant<int, 1>(), jb);
      if (jb + 1 < LEAF / SB) {
        if (jb == 0) {  // the helpers' part of the LDS image (every column right of block 0)
          while (sync_w[2] < 3) __builtin_amdgcn_s_sleep(1);
          wave_lds_fence();
        }
        update_diag_tile(jb);
        wave_lds_fence();
        LEAF_STAMP(3);
        if (jb <= 5) {
          while (sync_w[1] < 3 * (jb + 1)) {}
          wave_lds_fence();
        }
        LEAF_STAMP(4);
      }
    }
    wave_lds_fence();  // the last block's own rows (written behind its flag)
    if (lane == 0) lds_store(sync_w + 0, LEAF / SB + 1);
  } else {
    // ---------------------------------------------------------------- the helpers
    const int t = tid - 64;
    {  // the LDS image right of column block 0: row block b holds 16 rows x 8 b pieces of 16 bytes there
      double2_t v[21];  // sum over row blocks of ceil(128 b / 192)
      int u = 0;
#pragma unroll
      for (int bb = 1; bb < 8; ++bb) {
        const int per = 8 * bb, cnt = 16 * per;
#pragma unroll
        for (int idx0 = 0; idx0 < cnt; idx0 += 192) {
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
        for (int idx0 = 0; idx0 < cnt; idx0 += 192) {
          const int idx = idx0 + t;
          if (idx < cnt) *reinterpret_cast<double2_t*>(S + soff(16 * bb + idx / per) + 16 + 2 * (idx % per)) = v[u];
          ++u;
        }
      }
      wave_lds_fence();
      if (lane == 0) lds_add(sync_a + 2, 1);
      while (sync_w[2] < 3) __builtin_amdgcn_s_sleep(1);
      wave_lds_fence();
    }
    LEAF_STAMP1(8);
    // Register-resident trailing tiles (see namespace lt).  Per phase a wave reads one operand set (4 doubles per lane) per
    // block row it touches -- X~ rows: as they are the MFMA B operand of the tiles in that block column, times -1 / d_k
    // the A operand of the tiles in that block row -- and issues its tiles' MFMAs interleaved.
    double4_t tacc[9];
    auto trailing = [&](auto JBc, auto Wc, auto PHc) {
      constexpr int JB = decltype(JBc)::value, W = decltype(Wc)::value, PH = decltype(PHc)::value;
      double opa[8][4], opb[8][4], nr[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) nr[s4] = nrv[(JB & 1) * SB + 4 * s4 + kq];
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
    // Column block cb is final for rows >= 16 cb once wave 0 has published it and nobody reads its unnormalised form any
    // more: L = X~ D^-1/2 goes back to the LDS image (the inverse phase reads it) and out to memory, 8 pieces of 16 B per
    // row.  A thread's pieces all lie in ONE column pair (192 is a multiple of 8): two reciprocal square roots per thread.
    auto stream_out = [&](int cb) {  // at most 6 pieces per thread: all LDS reads first, then the stores (one round trip)
      const int c0 = cb * SB, npiece = (LEAF - c0) * 8;
      const int cp = c0 + 2 * (t & 7);
      const double d0 = pvt[cp], d1 = pvt[cp + 1];
      const double rs0 = fast_rsqrt(d0), rs1 = fast_rsqrt(d1);
      if (t < 8) {
        invd[cp] = rs0;
        invd[cp + 1] = rs1;
        int bad = 0;
        if (!(d1 > 0.0)) bad = cp + 2;
        if (!(d0 > 0.0)) bad = cp + 1;
        if (bad != 0) atomicMin(info, col0 + bad);
      }
      double2_t v[6];
      bool full[6], half[6];
      double* dst[6];
      double* sdst[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int it = t + 192 * u;
        const int r = c0 + (it >> 3), c2 = c0 + 2 * (it & 7);
        const bool in = it < npiece && c2 <= r;
        full[u] = in && c2 + 1 <= r;
        half[u] = in && c2 + 1 > r;
        dst[u] = Ablk + (long)r * lda + c2;
        sdst[u] = S + soff(r) + c2;
        if (in) v[u] = *reinterpret_cast<const double2_t*>(sdst[u]);
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        v[u].x *= rs0;
        v[u].y *= rs1;
        if (full[u]) {
          *reinterpret_cast<double2_t*>(dst[u]) = v[u];
          *reinterpret_cast<double2_t*>(sdst[u]) = v[u];
        } else if (half[u]) {
          dst[u][0] = v[u].x;
          sdst[u][0] = v[u].x;
        }
      }
    };
    auto helper = [&](auto Wc) {
      load_tiles(Wc);
      // fully unrolled: every copy sees a constant jb, the tile accumulators have plain live ranges (no loop-carried phis
      // through a switch: that form cost 200 registers of copies and moved the accumulators to AGPRs)
      lt::static_for<0, LEAF / SB>([&](auto JBc) {
        constexpr int JB = decltype(JBc)::value;
        while (sync_w[0] < JB + 1) {}
        wave_lds_fence();
        LEAF_STAMP1(9);
        if constexpr (JB <= 5) {
          trailing(JBc, Wc, std::integral_constant<int, 0>());
          wave_lds_fence();
          if (lane == 0) lds_add(sync_a + 1, 1);
          LEAF_STAMP1(10);
          trailing(JBc, Wc, std::integral_constant<int, 1>());
          LEAF_STAMP1(11);
        }
        // every helper is through with the unnormalised column block JB - 1 (its lazy phase of iteration JB - 1), and so is
        // wave 0 (it has published block JB since)
        if constexpr (JB >= 1) {
          while (sync_w[3] < 3 * JB) __builtin_amdgcn_s_sleep(1);
          stream_out(JB - 1);
        }
        wave_lds_fence();
        if (lane == 0) lds_add(sync_a + 3, 1);
        LEAF_STAMP1(12);
      });
      while (sync_w[3] < 3 * (LEAF / SB) || sync_w[0] < LEAF / SB + 1) __builtin_amdgcn_s_sleep(1);
      stream_out(LEAF / SB - 1);
    };
    if (wave == 1) helper(std::integral_constant<int, 0>());
    else if (wave == 2) helper(std::integral_constant<int, 1>());
    else helper(std::integral_constant<int, 2>());
  }
  __syncthreads();
  LEAF_STAMP(5);
  // ---- M = L^-1 (128x128, lower triangular) in place of L in LDS, streamed to `minv` (row-major, ld 128).
  // (1) the eight 16x16 diagonal blocks by substitution: thread (b, c) solves column c of block b;
  // (2) three levels of block doubling [[L11,0],[L21,L22]]^-1 = [[M11,0],[-M22 L21 M11, M22]] on MFMA: at block size
  //     t = 1, 2, 4 tiles stage A forms T = L21 M11 (wave = tile row), stage B forms -M22 T (wave = tile column);
  //     28 tile products per wave, the wave's up to four tiles of a stage accumulate interleaved (a dependent fp64 MFMA
  //     chain issues at a quarter of the independent rate), results replace L21 in LDS between barriers.
  // (An interleaved variant -- one block row of M per iteration of the loop above, in the shadow of wave 0's chain --
  // was built and measured in round 2: 69 us instead of 36: the 16x16 inverse by substitution (4.2k cycles per block on
  // one wave) and five LDS-word syncs per iteration outweigh the 4 us this block costs at the end.)
  {
    double z[SB], nz[SB], a[SB], iv[SB];
    const int b = tid >> 4, c = tid & 15, j0 = b * SB;
    if (tid < LEAF) {  // waves 0 and 1: a 16-lane DPP row = one diagonal block, lane c holds row c of L_bb
      const double* lrow = S + soff(j0 + c) + j0;
#pragma unroll
      for (int k = 0; k < SB; ++k) {
        a[k] = lrow[k];
        iv[k] = invd[j0 + k];
        nz[k] = 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);  // the DPP reads below must not follow the VALU writes of a[] back to back
      DinvStep<0>::run(z, nz, a, iv, c);
    }
    LEAF_STAMP(16);
    __syncthreads();  // every thread has read its diagonal block before it is overwritten
    if (tid < LEAF) {
#pragma unroll
      for (int r = 0; r < SB; ++r) {
        S[soff(j0 + r) + j0 + c] = z[r];  // zeros above the diagonal: the tiles are read whole as MFMA operands
        minv[(long)(j0 + r) * LEAF + j0 + c] = z[r];
      }
    }
    __syncthreads();
    LEAF_STAMP(17);
  }
#pragma unroll
  for (int t = 1; t <= 4; t *= 2) {
    const int node = wave / t, p0 = node * 2 * t, idx = wave % t;
    double4_t res[4], rs2[4][2];  // two partial accumulators per tile (MFMA steps s4 even / odd): chains half as long
#pragma unroll
    for (int c = 0; c < 4; ++c) rs2[c][0] = rs2[c][1] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // stage A: T[a][c] = sum_{k >= c} L[p0+t+a][p0+k] M[p0+k][p0+c], this wave owns tile row a = idx; the k-th
    // products of its t tiles are issued together (independent accumulators)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < t) {
        double av[4];
        const double* ap = S + soff(16 * (p0 + t + idx) + nn) + 16 * (p0 + k) + kq;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) av[s4] = ap[4 * s4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c <= k && c < t)
              rs2[c][s4 & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], S[soff(16 * (p0 + k) + 4 * s4 + kq) + 16 * (p0 + c) + nn], rs2[c][s4 & 1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) res[c] = rs2[c][0] + rs2[c][1];
    LEAF_STAMP(18);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < t) put(res[c], p0 + t + idx, p0 + c);
    __syncthreads();
    // stage B: M21[a][c] = - sum_{k <= a} M[p0+t+a][p0+t+k] T[k][c], this wave owns tile column c = idx
#pragma unroll
    for (int a = 0; a < 4; ++a) rs2[a][0] = rs2[a][1] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < t) {
        double bv[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) bv[s4] = S[soff(16 * (p0 + t + k) + 4 * s4 + kq) + 16 * (p0 + idx) + nn];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int a = 0; a < 4; ++a)
            if (a >= k && a < t)
              rs2[a][s4 & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(S[soff(16 * (p0 + t + a) + nn) + 16 * (p0 + t + k) + 4 * s4 + kq], bv[s4], rs2[a][s4 & 1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) res[a] = rs2[a][0] + rs2[a][1];
    LEAF_STAMP(19);
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (a < t) {
        const double4_t v = -res[a];
        put(v, p0 + t + a, p0 + idx);
#pragma unroll
        for (int r = 0; r < 4; ++r) minv[(long)(16 * (p0 + t + a) + kq + 4 * r) * LEAF + 16 * (p0 + idx) + nn] = v[r];
      }
    }
    __syncthreads();
  }
  LEAF_STAMP(20);
  // Last tile column of an evaluation: the only rows below are the y^T row block (one non-zero row), so the forward
  // solve of these 128 columns, beta = y M^T, is done here against the inverse that is still in LDS -- the strip launch
  // for that block (6 us of launch and round trips for 16k flops) is skipped.
  if (yrow != nullptr) {
    if (tid < LEAF) invd[tid] = yrow[tid];
    __syncthreads();
    const int c = tid >> 1, half = tid & 1;
    const double* mrow = S + soff(c);
    double acc = 0.0;
    for (int k = half; k <= c; k += 2) acc = __builtin_fma(mrow[k], invd[k], acc);
    acc += __shfl_xor(acc, 1);
    if (half == 0) yrow[c] = acc;
  }
}

// wait_ptr (optional): a cross-stream signal this launch has to see at wait_val or above before it ends (the driver folds
// the panel stream's wait for the main stream's next-panel update into the leaf that precedes the first reader of those
// columns: one polling lane at the end of a kernel that is a single workgroup anyway, instead of a runtime wait kernel of
// 5-9 us on the chain).  The poll gives up after ~2^22 sleeps (seconds) and reports through the bad-pivot word.
__device__ __forceinline__ void poll_signal(const unsigned* ptr, unsigned val, int* info) {
  long spins = 0;
  while (__hip_atomic_load(ptr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < val) {
    __builtin_amdgcn_s_sleep(4);
    if (++spins > (1L << 22)) {
      atomicMin(info, SIGNAL_TIMEOUT_INFO);
      break;
    }
  }
}

__global__ __launch_bounds__(256, 1) void potrf_leaf128_kernel(double* __restrict__ Ablk, long lda,
                                                                double* __restrict__ minv, int col0,
                                                                int* __restrict__ info, double* yrow, long sA, long sminv,
                                                                int sinfo, const unsigned* wait_ptr, unsigned wait_val) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  __builtin_amdgcn_s_setprio(3);  // the leaf is the panel chain: win the issue arbitration against bulk GEMM waves on its CU
  const long z = blockIdx.x;  // batched evaluation: one workgroup per problem
  potrf_leaf128_body(Ablk + z * sA, lda, minv + z * sminv, col0, info + z * sinfo, smem, yrow ? yrow + z * sA : nullptr);
  if (wait_ptr != nullptr && threadIdx.x == 0) poll_signal(wait_ptr, wait_val, info + z * sinfo);
}

// One lane: raise *wr to `val` (if wr) and then wait for *wt >= val (if wt).  The panel stream's two edges at a
// super-panel boundary -- tell the main stream the panel is done, wait for the main stream's previous bulk update -- in
// ONE launch instead of two runtime kernels (hipStreamWriteValue32 + hipStreamWaitValue32, ~5 us each on the chain).
__global__ void signal_write_wait_kernel(unsigned* wr, const unsigned* wt, unsigned val, int* info) {
  if (threadIdx.x == 0) {
    if (wr != nullptr) __hip_atomic_store(wr, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (wt != nullptr) poll_signal(wt, val, info);
  }
}

hipError_t launch_signal_write_wait(unsigned* wr, const unsigned* wt, unsigned val, int* info, hipStream_t stream) {
  signal_write_wait_kernel<<<1, 64, 0, stream>>>(wr, wt, val, info);
  return hipGetLastError();
}

// X * L^T = B in place on B (m x 128, leading dimension ldb even, m multiple of 16 RG) as X = B * M^T with M = L^-1
// (row-major 128 x 128, lower triangular, zeros above the diagonal of its diagonal 16x16 tiles).
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
template <int RG>
__device__ __forceinline__ void trsm_strip128_body(const double* __restrict__ minv, double* __restrict__ B, long ldb, int blk) {
  typedef double double2_t __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int jb0 = wave, jb1 = 7 - wave;  // jb0 < jb1; k-blocks 0 .. jb1 are needed
  double* Brow = B + ((long)blk * (16 * RG) + n) * ldb + 4 * q;
  double2_t a[8][2];  // ONE row group's operands; a k-block's pair is refilled with the next group's as soon as it is consumed
  auto load_kb = [&](int rg, int kb) {
    const double* src = Brow + (long)rg * 16 * ldb;
    a[kb][0] = *reinterpret_cast<const double2_t*>(src + 16 * kb);
    a[kb][1] = *reinterpret_cast<const double2_t*>(src + 16 * kb + 2);
  };
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
    if (kb <= jb1) load_kb(0, kb);
  // M rows 16 jb + n, k-blocks 0 .. jb: (jb0 + 1) + (jb1 + 1) = 9 fetches of 4 doubles per lane
  const double* m0 = minv + (long)(16 * jb0 + n) * LEAF + 4 * q;
  const double* m1 = minv + (long)(16 * jb1 + n) * LEAF + 4 * q;
  double2_t b0[4][2], b1[8][2];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (kb <= jb0) {
      b0[kb][0] = *reinterpret_cast<const double2_t*>(m0 + 16 * kb);
      b0[kb][1] = *reinterpret_cast<const double2_t*>(m0 + 16 * kb + 2);
    }
  }
#pragma unroll
  for (int kb = 0; kb < 8; ++kb) {
    if (kb <= jb1) {
      b1[kb][0] = *reinterpret_cast<const double2_t*>(m1 + 16 * kb);
      b1[kb][1] = *reinterpret_cast<const double2_t*>(m1 + 16 * kb + 2);
    }
  }
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    // Four partial accumulators per output tile, one per MFMA step of a k-block (round 4): a dependent fp64 MFMA follows
    // its predecessor after ~250 cycles but an independent one after 64, and with ONE accumulator per tile the wave that
    // owns column block 7 ran a chain of 32 per row group (8000 of the ~16000 cycles of a one-group strip, and nearly all
    // of a four-group strip's 16 us).  Now the longest chain is 8 deep and the partial sums are added pairwise at the end.
    // The registers for the partials come from the second operand set of rounds 2-3 (two row groups in flight): a k-block's
    // operands are refilled with the next group's right behind their last MFMA instead, still a whole MFMA phase ahead.
    const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
    double4_t p1[4] = {zero4, zero4, zero4, zero4}, p0[4] = {zero4, zero4, zero4, zero4};
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
      if (rg + 1 < RG && kb <= jb1) load_kb(rg + 1, kb);
    }
    const double4_t x0 = (p0[0] + p0[1]) + (p0[2] + p0[3]), x1 = (p1[0] + p1[1]) + (p1[2] + p1[3]);
    // every wave's copy of this row group is in registers (its MFMAs consumed it; the next group's loads may still be
    // in flight, they touch other rows) before anybody overwrites the group
    if (rg + 1 < RG) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // D layout: lane holds X[16 (blk RG + rg) + q + 4 r][16 jb + n]
    double* out = B + ((long)blk * (16 * RG) + 16 * rg + q) * ldb + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      out[(long)(4 * r) * ldb + 16 * jb0] = x0[r];
      out[(long)(4 * r) * ldb + 16 * jb1] = x1[r];
    }
  }
}

#ifndef MIGP_XCD_MAP_STRIP
#define MIGP_XCD_MAP_STRIP 1
#endif
// one launch serves `gridDim.y` independent (M, B) pairs: M at minv + y * 16384, B at B + y * strideB
template <int RG>
__global__ __launch_bounds__(256) void trsm_strip128_kernel(const double* __restrict__ minv, double* __restrict__ B, long ldb,
                                                             long strideB, long sminv2, long sB2) {
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
                         B + (long)blockIdx.y * strideB + (long)blockIdx.z * sB2, ldb, blk);
}

// The leaf asks for more LDS than it uses, so that no 72 KB GEMM workgroup fits beside it on a CU: since round 5 its
// registers (152) would fit beside a bulk wave on a SIMD, and a chain wave that shares the SIMD's fp64 pipe with MFMA-saturated
// waves runs about half as fast (N = 16384: 25.8 -> 26.2 ms when the two were allowed to share).
constexpr size_t LEAF_LDS_USED = sizeof(double) * (LEAF_ELEMS + 2 * LEAF + 2 * SB + 4);
constexpr size_t LEAF_LDS_BYTES = LEAF_LDS_USED > 96 * 1024 ? LEAF_LDS_USED : 96 * 1024;

hipError_t leaf_enable_lds() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_leaf128_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS_BYTES);
  return e;
}

hipError_t launch_potrf_leaf128(double* Ablk, long lda, double* minv, int col0, int* info, hipStream_t stream, double* yrow,
                                const Batch* bt, const unsigned* wait_ptr, unsigned wait_val) {
  potrf_leaf128_kernel<<<bt ? bt->nb : 1, 256, LEAF_LDS_BYTES, stream>>>(Ablk, lda, minv, col0, info, yrow, bt ? bt->sK : 0,
                                                                        bt ? bt->sdinv : 0, bt ? bt->sinfo : 0, wait_ptr, wait_val);
  return hipGetLastError();
}

// rows per workgroup by panel height: 16 while one round of workgroups covers the panel (lowest latency), 32 / 64 for
// tall panels (the wave's tiles of M are reused, 1/2 and 1/4 of the operand traffic); m is a multiple of 64 or of 16
hipError_t launch_trsm_strip128_batched(const double* minv, double* B, long ldb, long strideB, int m, int batch,
                                        hipStream_t stream, const Batch* bt, long sB2) {
  if (m <= 0 || batch <= 0) return hipSuccess;
  const int nb = bt ? bt->nb : 1;
  const long sm2 = bt ? bt->sdinv : 0;
  const long rows = (long)m * batch * nb;
  if (rows > 8192 && m % 64 == 0) trsm_strip128_kernel<4><<<dim3(m / 64, batch, nb), 256, 0, stream>>>(minv, B, ldb, strideB, sm2, sB2);
  else if (rows > 4096 && m % 32 == 0) trsm_strip128_kernel<2><<<dim3(m / 32, batch, nb), 256, 0, stream>>>(minv, B, ldb, strideB, sm2, sB2);
  else trsm_strip128_kernel<1><<<dim3(m / 16, batch, nb), 256, 0, stream>>>(minv, B, ldb, strideB, sm2, sB2);
  return hipGetLastError();
}

hipError_t launch_trsm_strip128(const double* minv, double* B, long ldb, int m, hipStream_t stream, const Batch* bt, long sB2) {
  return launch_trsm_strip128_batched(minv, B, ldb, 0, m, 1, stream, bt, sB2);
}

}  // namespace migp
