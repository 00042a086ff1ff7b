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
__device__ unsigned long long g_leaf_stamps[16];
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

__device__ __forceinline__ void wave_lds_fence() {
  // order this wave's LDS writes before its later LDS reads (DS ops execute in order per wave;
  // this only stops the compiler from reordering / caching across the point)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
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
// Forward substitution X = B L16^-T for 16 rows on one wave: lane 4*row + g owns columns g, g+4, g+8, g+12.
template <int K>
struct QuadSolve {
  // lt[K][i] = L16[4i+g][K] and inv[i] = 1/L16[4i+g][4i+g] are preloaded so that no LDS latency sits on the chain
  static __device__ __forceinline__ void run(double (&x)[4], const double (&lt)[16][4], const double (&inv)[4], int g) {
    const double xk = quad_bcast<(K & 3)>(x[K >> 2] * inv[K >> 2]);
    if (g == (K & 3)) x[K >> 2] = xk;
#pragma unroll
    for (int i = K >> 2; i < 4; ++i) {
      // column 4i+g is updated only if it lies right of K (lanes of slot K>>2 with g <= K&3 are done)
      if (i > (K >> 2) || g > (K & 3)) x[i] = __builtin_fma(-xk, lt[K][i], x[i]);
    }
    QuadSolve<K + 1>::run(x, lt, inv, g);
  }
};
template <>
struct QuadSolve<16> {
  static __device__ __forceinline__ void run(double (&)[4], const double (&)[16][4], const double (&)[4], int) {}
};

// One elimination step of the 16x16 diagonal sub-block held one row per lane (a[c], c = 0..15):
// square-root-free form, a_rc -= (a_rj / p_j) * a_cj for c > j, column j+1 first so that the next
// pivot's reciprocal chain starts as early as possible.
template <int J, int C>
struct ElimCols {
  static __device__ __forceinline__ void run(double (&a)[16], double negw) {
    fmac_rowbcast<C>(a[C], a[J], negw);
    ElimCols<J, C + 1>::run(a, negw);
  }
};
template <int J>
struct ElimCols<J, 16> {
  static __device__ __forceinline__ void run(double (&)[16], double) {}
};
template <int J>
struct ElimStep {
  static __device__ __forceinline__ void run(double (&a)[16], double p, int& bad) {
    if (!(p > 0.0) && bad == 0) bad = J + 1;
    const double negw = -a[J] * fast_rcp(p);
    fmac_rowbcast<J + 1>(a[J + 1], a[J], negw);
    const double pn = mov_rowbcast<J + 1>(a[J + 1]);
    ElimCols<J, J + 2>::run(a, negw);
    ElimStep<J + 1>::run(a, pn, bad);
  }
};
template <>
struct ElimStep<15> {
  static __device__ __forceinline__ void run(double (&)[16], double p, int& bad) {
    if (!(p > 0.0) && bad == 0) bad = 16;
  }
};
template <int C>
struct ScaleCols {
  // l[C] = a[C] * rs of lane C: ONE dpp fmac into a zeroed accumulator (round 1: broadcast move + multiply, 25 cycles)
  static __device__ __forceinline__ void run(const double (&a)[16], double rs, double (&l)[16]) {
    l[C] = 0.0;
    fmac_rowbcast<C>(l[C], rs, a[C]);
    ScaleCols<C + 1>::run(a, rs, l);
  }
};
template <>
struct ScaleCols<16> {
  static __device__ __forceinline__ void run(const double (&)[16], double, double (&)[16]) {}
};

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
constexpr bool needs(int w, int jb, int b) {  // does wave w read block row b of column block jb as an operand?
  for (int s = 0; s < 9; ++s)
    if (live(3 * s + w, jb) && (TR[3 * s + w] == b || TC[3 * s + w] == b)) return true;
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
// Structure per 16-column block jb of the 128x128 leaf (all of it LDS resident):
//  (A) wave 0 factors the 16x16 diagonal sub-block in REGISTERS: lane r owns row r, columns are
//      eliminated in square-root-free (LDL^T) form so the per-pivot critical path is
//      row-broadcast -> rcp -> mul -> fma (the rsqrt of all 16 pivots is taken once, in parallel, at
//      the end); column values are broadcast inside v_fmac_f64_dpp row_newbcast, so one elimination
//      is ONE instruction and there is no LDS round trip per pivot;
//  (B) rows below: X = B L16^-T, one thread per row, column-oriented substitution in registers;
//  (C) trailing update of the remaining lower tiles on fp64 MFMA (rank 16).
__device__ __forceinline__ void potrf_leaf128_body(double* __restrict__ Ablk, long lda, double* __restrict__ minv,
                                                    int col0, int* __restrict__ info, double* smem, double* yrow) {
  double* S = smem;                     // packed lower block-trapezoid, see soff()
  double* LdT2 = smem + LEAF_ELEMS;     // [2][16][16]  LdT[k][c] = L16[c][k] of diagonal sub-block jb (buffer jb & 1)
  double* invd = LdT2 + 2 * SB * SB;    // [128] 1 / L[c][c]
  volatile int* sync_w = reinterpret_cast<volatile int*>(invd + LEAF);  // [0] rows published by wave 0, [1] arrivals of waves 1..3,
  // [2] arrivals of waves 1..3 inside the inverse phases, [3] Dinv_b ready (wave 1), [4] wave 0 done with column block jb
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: tile coordinates and soff() bases derived
                                                              // from it are computed on the scalar unit (v_mul_lo_u32 is quarter rate)
#ifdef LEAF_STAMPS
  unsigned long long t_prev = 0, t_prev1 = 0;
  if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  if (threadIdx.x == 64) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev1)::"memory");
#endif

  // load the lower block-trapezoid: wave 0 takes the first 16x16 block and starts factoring it while
  // waves 1..3 stream in the other 4480 16-byte pieces (24 loads in flight per lane, one round trip)
  typedef double double2_t __attribute__((ext_vector_type(2)));
  if (wave == 0) {
    double2_t v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u;  // 128 pieces of block row 0
      v[u] = *reinterpret_cast<const double2_t*>(Ablk + (long)(e >> 3) * lda + 2 * (e & 7));
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u;
      *reinterpret_cast<double2_t*>(S + soff(e >> 3) + 2 * (e & 7)) = v[u];
    }
    wave_lds_fence();
  } else {
    // row block b holds 16 rows x 8(b+1) pieces; compile-time b makes the div/mod cheap
    const int t = tid - 64;
    double2_t v[27];
    int u = 0;
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
      const int per = 8 * (bb + 1), cnt = 16 * per, skip = (bb == 0) ? 128 : 0;  // block 0 belongs to wave 0
#pragma unroll
      for (int idx0 = skip; idx0 < cnt; idx0 += 192) {
        const int idx = idx0 + t;
        if (idx < cnt) v[u] = *reinterpret_cast<const double2_t*>(Ablk + (long)(16 * bb + idx / per) * lda + 2 * (idx % per));
        ++u;
      }
    }
    u = 0;
#pragma unroll
    for (int bb = 0; bb < 8; ++bb) {
      const int per = 8 * (bb + 1), cnt = 16 * per, skip = (bb == 0) ? 128 : 0;
#pragma unroll
      for (int idx0 = skip; idx0 < cnt; idx0 += 192) {
        const int idx = idx0 + t;
        if (idx < cnt) *reinterpret_cast<double2_t*>(S + soff(16 * bb + idx / per) + 2 * (idx % per)) = v[u];
        ++u;
      }
    }
  }
  // (A): factor the 16x16 diagonal sub-block jb in registers (wave 0; lanes 16..63 mirror lanes 0..15)
  auto factor_diag = [&](int jb) {
    const int j0 = jb * SB;
    const int r = lane & 15;
    double a[SB];
    {
      const double* row = S + soff(j0 + r) + j0;
#pragma unroll
      for (int c = 0; c < SB; ++c) a[c] = row[c];
    }
    int bad = 0;
    ElimStep<0>::run(a, mov_rowbcast<0>(a[0]), bad);
    if (bad != 0 && lane == 0) atomicMin(info, col0 + j0 + bad);
    // normalise: L[r][c] = a[c] * rsqrt(p_c); lane c holds p_c = a[c]
    double rs = fast_rsqrt(a[r]);
    double l[SB];
    dpp_settle(rs);  // rs was written by VALU just now: two wait states before the DPP reads below
    ScaleCols<0>::run(a, rs, l);
    double* row = S + soff(j0 + r) + j0;
    double* LdT = LdT2 + (jb & 1) * SB * SB;
#pragma unroll
    for (int c = 0; c < SB; ++c) {
      if (lane < SB && c <= r) {
        row[c] = l[c];
        LdT[c * SB + r] = l[c];
      }
    }
    if (lane < SB) invd[j0 + r] = rs;
  };
  // (B) for one row: X = B * L16^-T by column-oriented forward substitution in registers
  auto solve_row = [&](int jb, int rowidx) {
    const int j0 = jb * SB;
    const double* LdT = LdT2 + (jb & 1) * SB * SB;
    double* row = S + soff(rowidx) + j0;
    double x[SB];
#pragma unroll
    for (int c = 0; c < SB; ++c) x[c] = row[c];
#pragma unroll
    for (int k = 0; k < SB; ++k) {
      x[k] *= invd[j0 + k];
#pragma unroll
      for (int c = k + 1; c < SB; ++c) x[c] = __builtin_fma(-x[k], LdT[k * SB + c], x[c]);
    }
#pragma unroll
    for (int c = 0; c < SB; ++c) row[c] = x[c];
  };
  // (C) one 16x16 tile of the trailing update: S[r0.., c0..] -= X[r0..] X[c0..]^T with X = columns j0..j0+15
  auto update_tile = [&](int j0, int r0, int c0) {
    const int n = lane & 15, kq = lane >> 4;
    double4_t acc;
    double av[4], bv[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      av[s4] = S[soff(r0 + n) + j0 + 4 * s4 + kq];  // X[r0 + (l&15)][k = 4s + (l>>4)]
      bv[s4] = S[soff(c0 + n) + j0 + 4 * s4 + kq];  // X[c0 + (l&15)][k]
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = S[soff(r0 + kq + 4 * r) + c0 + n];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) S[soff(r0 + kq + 4 * r) + c0 + n] = acc[r];
  };

  // tile (rb, cb) of the LDS image as MFMA operands / result:
  //   A operand: lane holds [16 rb + n][16 cb + 4 s + kq];  B operand: [16 rb + 4 s + kq][16 cb + n];  D: [16 rb + kq + 4 r][16 cb + n]
  const int nn = lane & 15, kq = lane >> 4;
  auto put = [&](const double4_t& v, int rb, int cb) {
#pragma unroll
    for (int r = 0; r < 4; ++r) S[soff(16 * rb + kq + 4 * r) + 16 * cb + nn] = v[r];
  };
  LEAF_STAMP(0);
  if (tid == 64) { sync_w[0] = 0; sync_w[1] = 0; sync_w[2] = 0; sync_w[3] = 0; sync_w[4] = 0; }
  if (wave == 0) factor_diag(0);
  __syncthreads();
  LEAF_STAMP(1);
  // Register-resident trailing tiles of waves 1..3 (see namespace lt).  Per iteration a wave reads one operand set
  // (4 doubles per lane) per block row it touches -- the same registers serve as the MFMA A operand of the tiles in that
  // block row and as the B operand of the tiles in that block column -- and issues its live tiles' MFMAs interleaved
  // (up to nine independent accumulators: full issue rate).  Round 2 until here: every tile went LDS -> registers -> LDS
  // in every iteration, three or four at a time (0.7k cycles per tile; 12 LDS accesses per lane and tile instead of ~3).
  double4_t tacc[9];
  auto trailing = [&](auto JBc, auto Wc) {
    constexpr int JB = decltype(JBc)::value, W = decltype(Wc)::value;
    double op[8][4];
    lt::static_for<1, 8>([&](auto Bc) {
      constexpr int B = decltype(Bc)::value;
      if constexpr (lt::needs(W, JB, B)) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) op[B][s4] = S[soff(16 * B + nn) + 16 * JB + 4 * s4 + kq];
      }
    });
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      lt::static_for<0, 9>([&](auto Sc) {
        constexpr int I = 3 * decltype(Sc)::value + W;
        if constexpr (lt::live(I, JB))
          tacc[decltype(Sc)::value] = __builtin_amdgcn_mfma_f64_16x16x4f64(-op[lt::TR[I]][s4], op[lt::TC[I]][s4], tacc[decltype(Sc)::value], 0, 0, 0);
      });
    lt::static_for<0, 9>([&](auto Sc) {
      constexpr int I = 3 * decltype(Sc)::value + W;
      if constexpr (lt::last(I, JB)) put(tacc[decltype(Sc)::value], lt::TR[I], lt::TC[I]);
    });
  };
  auto trailing_all = [&](auto Wc, int jb) {
    switch (jb) {
      case 0: trailing(std::integral_constant<int, 0>(), Wc); break;
      case 1: trailing(std::integral_constant<int, 1>(), Wc); break;
      case 2: trailing(std::integral_constant<int, 2>(), Wc); break;
      case 3: trailing(std::integral_constant<int, 3>(), Wc); break;
      case 4: trailing(std::integral_constant<int, 4>(), Wc); break;
      case 5: trailing(std::integral_constant<int, 5>(), Wc); break;
      default: break;
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
  // Per 16-column block jb, after the diagonal sub-block jb has been factored:
  //   wave 0     : solves the 16 rows of the NEXT diagonal block, publishes them, updates the next diagonal
  //                tile and factors it (the serial chain of the leaf);
  //   waves 1..3 : solve the remaining rows, meet each other and wave 0's rows through two LDS words, apply
  //                the rank-16 MFMA update to every other trailing tile and stream column block jb out.
  // Two loops, one per wave role, with the same number of workgroup barriers (s_barrier counts waves, not code
  // locations): the register-resident tiles of waves 1..3 and the solve / factor state of wave 0 then never share a
  // live range (in one loop body the kernel needed 256 VGPRs + 204 AGPRs of spill space).
  if (wave == 0) {
    for (int jb = 0; jb < LEAF / SB; ++jb) {
      const int j0 = jb * SB;
      if (jb + 1 < LEAF / SB) {
        {  // the 16 rows of the next diagonal block, four lanes per row
          const int g = lane & 3;
          double* row = S + soff(j0 + SB + (lane >> 2)) + j0;
          double x[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] = row[4 * i + g];
          const double* LdT = LdT2 + (jb & 1) * SB * SB;
          double lt[16][4], inv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) inv[i] = invd[j0 + 4 * i + g];
#pragma unroll
          for (int k = 0; k < 15; ++k)
#pragma unroll
            for (int i = k >> 2; i < 4; ++i) lt[k][i] = LdT[k * SB + 4 * i + g];
          __builtin_amdgcn_sched_barrier(0);
          QuadSolve<0>::run(x, lt, inv, g);
#pragma unroll
          for (int i = 0; i < 4; ++i) row[4 * i + g] = x[i];
        }
        wave_lds_fence();
        if (lane == 0) sync_w[0] = jb + 1;
        LEAF_STAMP(2);
        update_tile(j0, j0 + SB, j0 + SB);
        wave_lds_fence();
        LEAF_STAMP(6);
        factor_diag(jb + 1);
        LEAF_STAMP(7);
      }
      __syncthreads();
      LEAF_STAMP(3);
    }
  } else {
    if (wave == 1) load_tiles(std::integral_constant<int, 0>());
    else if (wave == 2) load_tiles(std::integral_constant<int, 1>());
    else load_tiles(std::integral_constant<int, 2>());
#pragma unroll  // fully: every copy sees a constant jb, the tile accumulators have plain live ranges (no loop-carried phis
                // through a switch: that form cost 200 registers of copies and moved the accumulators to AGPRs)
    for (int jb = 0; jb < LEAF / SB; ++jb) {
      const int j0 = jb * SB;
      const int t = tid - 64;
      const int nrest = LEAF - j0 - 2 * SB;  // rows j0+32 .. 127
      LEAF_STAMP1(15);  // wait at the previous iteration's barrier
      if (t < nrest) solve_row(jb, j0 + 2 * SB + t);
      LEAF_STAMP1(9);   // solve_row
      // column block cb is final for rows >= 16 cb once iteration cb's rows are solved: 8 pieces of 16 B per row go to
      // memory.  Block jb - 1 is streamed here, in the time these waves would otherwise spin waiting for wave 0's rows.
      auto stream_out = [&](int cb) {  // at most 6 pieces per thread: all LDS reads first, then the stores (one round trip)
        const int c0 = cb * SB, npiece = (LEAF - c0) * 8;
        double2_t v[6];
        bool full[6], half[6];
        double* dst[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int it = t + 192 * u;
          const int r = c0 + (it >> 3), c2 = c0 + 2 * (it & 7);
          const bool in = it < npiece && c2 <= r;
          full[u] = in && c2 + 1 <= r;
          half[u] = in && c2 + 1 > r;
          dst[u] = Ablk + (long)r * lda + c2;
          if (in) v[u] = *reinterpret_cast<const double2_t*>(S + soff(r) + c2);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          if (full[u]) *reinterpret_cast<double2_t*>(dst[u]) = v[u];
          else if (half[u]) dst[u][0] = v[u].x;
        }
      };
      if (jb + 1 < LEAF / SB) {
        wave_lds_fence();
        if (lane == 0) atomicAdd(const_cast<int*>(sync_w + 1), 1);
        if (jb > 0) stream_out(jb - 1);
        while (sync_w[1] < 3 * (jb + 1) || sync_w[0] < jb + 1) __builtin_amdgcn_s_sleep(1);
        wave_lds_fence();
      } else {
        stream_out(jb - 1);
        stream_out(jb);
      }
      LEAF_STAMP1(8);   // arrive, stream-out, spin for wave 0's rows and the other helpers
      if (wave == 1) trailing_all(std::integral_constant<int, 0>(), jb);
      else if (wave == 2) trailing_all(std::integral_constant<int, 1>(), jb);
      else trailing_all(std::integral_constant<int, 2>(), jb);
      LEAF_STAMP1(10);
      __syncthreads();
    }
  }
  LEAF_STAMP(4);
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
    LEAF_STAMP(11);
    __syncthreads();  // every thread has read its diagonal block before it is overwritten
    if (tid < LEAF) {
#pragma unroll
      for (int r = 0; r < SB; ++r) {
        S[soff(j0 + r) + j0 + c] = z[r];  // zeros above the diagonal: the tiles are read whole as MFMA operands
        minv[(long)(j0 + r) * LEAF + j0 + c] = z[r];
      }
    }
    __syncthreads();
    LEAF_STAMP(12);
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
    LEAF_STAMP(13);
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
    LEAF_STAMP(14);
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
  LEAF_STAMP(5);
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

constexpr size_t LEAF_LDS_BYTES = sizeof(double) * (LEAF_ELEMS + 2 * SB * SB + LEAF + 3);

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
