// Leaf kernels of the blocked fp64 Cholesky:
//   potrf_leaf128 : in-place lower Cholesky of one 128x128 diagonal block, LDS resident, plus the
//                   inverses of its eight 16x16 diagonal sub-blocks (consumed by trsm_strip128).
//   trsm_strip128 : X * L^T = B for row strips of a 128-column panel (LAPACK dtrsm R,L,T,N), done in
//                   transposed space so that every fp64 MFMA result tile is directly the B operand
//                   of the next MFMA (v_mfma_f64_16x16x4_f64: D[row=(l>>4)+4r][col=l&15] is exactly
//                   the B[k=(l>>4)+4s][col=l&15] operand layout).
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
__device__ unsigned long long g_leaf_stamps[8];
#define LEAF_STAMP(i)                                                                  \
  do {                                                                                 \
    if (threadIdx.x == 0) {                                                            \
      unsigned long long t_;                                                           \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
      g_leaf_stamps[i] += t_ - t_prev;                                                 \
      t_prev = t_;                                                                     \
    }                                                                                  \
  } while (0)
#else
#define LEAF_STAMP(i)
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
  static __device__ __forceinline__ void run(const double (&a)[16], double rs, double (&l)[16]) {
    l[C] = a[C] * mov_rowbcast<C>(rs);
    ScaleCols<C + 1>::run(a, rs, l);
  }
};
template <>
struct ScaleCols<16> {
  static __device__ __forceinline__ void run(const double (&)[16], double, double (&)[16]) {}
};

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
__device__ __forceinline__ void potrf_leaf128_body(double* __restrict__ Ablk, long lda, double* __restrict__ dinv,
                                                    int col0, int* __restrict__ info, double* smem) {
  double* S = smem;                     // packed lower block-trapezoid, see soff()
  double* LdT2 = smem + LEAF_ELEMS;     // [2][16][16]  LdT[k][c] = L16[c][k] of diagonal sub-block jb (buffer jb & 1)
  double* invd = LdT2 + 2 * SB * SB;    // [128] 1 / L[c][c]
  volatile int* sync_w = reinterpret_cast<volatile int*>(invd + LEAF);  // [0] rows published by wave 0, [1] arrivals of waves 1..3
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
#ifdef LEAF_STAMPS
  unsigned long long t_prev = 0;
  if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
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
    const double rs = fast_rsqrt(a[r]);
    double l[SB];
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

  LEAF_STAMP(0);
  if (tid == 64) { sync_w[0] = 0; sync_w[1] = 0; }
  if (wave == 0) factor_diag(0);
  __syncthreads();
  LEAF_STAMP(1);
  // Per 16-column block jb, after the diagonal sub-block jb has been factored:
  //   wave 0     : solves the 16 rows of the NEXT diagonal block, publishes them, updates the next diagonal
  //                tile and factors it (the serial chain of the leaf);
  //   waves 1..3 : solve the remaining rows, meet each other and wave 0's rows through two LDS words, apply
  //                the rank-16 MFMA update to every other trailing tile and stream column block jb out.
  for (int jb = 0; jb < LEAF / SB; ++jb) {
    const int j0 = jb * SB;
    if (wave == 0) {
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
    } else {
      const int t = tid - 64;
      const int nrest = LEAF - j0 - 2 * SB;  // rows j0+32 .. 127
      if (t < nrest) solve_row(jb, j0 + 2 * SB + t);
      if (jb + 1 < LEAF / SB) {
        wave_lds_fence();
        if (lane == 0) atomicAdd(const_cast<int*>(sync_w + 1), 1);
        while (sync_w[1] < 3 * (jb + 1) || sync_w[0] < jb + 1) __builtin_amdgcn_s_sleep(1);
        wave_lds_fence();
      }
      // column block jb is final for rows >= j0: 8 pieces of 16 B per row
      for (int it = t; it < (LEAF - j0) * 8; it += 192) {
        const int r = j0 + (it >> 3), c2 = j0 + 2 * (it & 7);
        if (c2 <= r) {
          const double2_t v = *reinterpret_cast<const double2_t*>(S + soff(r) + c2);
          double* dst = Ablk + (long)r * lda + c2;
          if (c2 + 1 <= r) *reinterpret_cast<double2_t*>(dst) = v;
          else dst[0] = v.x;
        }
      }
      const int q = LEAF / SB - 1 - jb;  // trailing tiles per dimension
      int e = 0;
      for (int tr = 1; tr < q; ++tr) {  // tile row 0 = tile (0,0) belongs to wave 0
        for (int tc = 0; tc <= tr; ++tc) {
          if ((e++ % 3) != wave - 1) continue;
          update_tile(j0, j0 + SB + 16 * tr, j0 + SB + 16 * tc);
        }
      }
    }
    __syncthreads();
    LEAF_STAMP(3);
  }
  LEAF_STAMP(4);
  // ---- inverses of the eight 16x16 diagonal sub-blocks: thread (b, c) solves column c of block b
  if (tid < LEAF) {
    const int b = tid >> 4, c = tid & 15, j0 = b * SB;
    double z[SB];
#pragma unroll
    for (int r = 0; r < SB; ++r) z[r] = 0.0;
#pragma unroll
    for (int r = 0; r < SB; ++r) {
      const double* lrow = S + soff(j0 + r) + j0;
      double s0 = (r == c) ? 1.0 : 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < r; ++k) {  // z[k] = 0 for k < c, so no predicate is needed
        if (k & 1) s1 = __builtin_fma(-lrow[k], z[k], s1);
        else s0 = __builtin_fma(-lrow[k], z[k], s0);
      }
      z[r] = (r >= c) ? (s0 + s1) * invd[j0 + r] : 0.0;
    }
    double* out = dinv + (long)b * SB * SB;
#pragma unroll
    for (int r = 0; r < SB; ++r) out[r * SB + c] = z[r];
  }
  LEAF_STAMP(5);
}

__global__ __launch_bounds__(256, 1) void potrf_leaf128_kernel(double* __restrict__ Ablk, long lda,
                                                                double* __restrict__ dinv, int col0,
                                                                int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  potrf_leaf128_body(Ablk, lda, dinv, col0, info, smem);
}

// X * L^T = B, in place on B (m x 128, leading dimension ldb, m multiple of 64).
// One wave per 16 rows; tiles kept transposed: T_j[r] = B[row0 + (l&15)][16j + 4r + (l>>4)].
// The 28 sub-diagonal 16x16 tiles of L and the 8 inverse diagonal blocks are staged once per
// workgroup in LDS in MFMA-fragment order (tile, k4-step, lane), so every A-operand read is one
// conflict-free ds_read_b64 of 64 consecutive doubles.  72 KB: a strip workgroup can share a CU with
// one GEMM workgroup of the concurrent trailing update.
constexpr int STRIP_TILES = 36;  // 28 sub-diagonal tiles of L + 8 diagonal inverses
__device__ __forceinline__ int strip_tile(int i, int j) { return i * (i - 1) / 2 + j; }  // i > j

// `ready` (fused leaf + strip launch only): word the leaf workgroup of the same launch sets once L and dinv are in
// memory; the right-hand side rows are loaded first, so they travel while the leaf is still factoring.
__device__ __forceinline__ void trsm_strip128_body(const double* __restrict__ Lblk, long lda,
                                                    const double* __restrict__ dinv, double* __restrict__ B, long ldb,
                                                    int blk, double* smem, int* ready, int* __restrict__ info) {
  double* Lt = smem;             // [28][4][64]
  double* Dt = smem + 28 * 256;  // [8][4][64]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
#ifdef LEAF_STAMPS
  unsigned long long t_prev = 0;
  if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#endif
  double* Brow = B + ((long)blk * 64 + wave * 16 + n) * ldb;
  double4_t T[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[j][r] = Brow[16 * j + 4 * r + q];
  if (ready) {
    if (tid == 0) {
      // bounded wait (about a second of the 100 MHz wall clock): a lost leaf must not hang the device
      const unsigned long long t0 = wall_clock64();
      bool ok = true;
      while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 100000000ull) { ok = false; break; }
      }
      if (!ok) atomicMin(info, -3);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  // stage L tiles: item = (tile, row, column quad) -> 4 doubles
  {
    typedef double double2_t __attribute__((ext_vector_type(2)));
    // tile row ti (1..7) holds tiles ti(ti-1)/2 .. ; 7 items per thread, all loads issued before the stores
    double2_t v0[7], v1[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int it = tid + 256 * u;
      const int tile = it >> 6, rr = (it >> 2) & 15, cq = it & 3;
      const int ti = (tile >= 21) ? 7 : (tile >= 15) ? 6 : (tile >= 10) ? 5 : (tile >= 6) ? 4 : (tile >= 3) ? 3 : (tile >= 1) ? 2 : 1;
      const int tj = tile - ti * (ti - 1) / 2;
      const double* src = Lblk + (long)(16 * ti + rr) * lda + 16 * tj + 4 * cq;
      v0[u] = *reinterpret_cast<const double2_t*>(src);
      v1[u] = *reinterpret_cast<const double2_t*>(src + 2);
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      const int it = tid + 256 * u;
      const int tile = it >> 6, rr = (it >> 2) & 15, cq = it & 3;
      double* dst = Lt + tile * 256 + cq * 64 + rr;  // + 16*q
      dst[0] = v0[u].x; dst[16] = v0[u].y; dst[32] = v1[u].x; dst[48] = v1[u].y;
    }
    for (int it = tid; it < 8 * 64; it += 256) {
      const int tile = it >> 6, rr = (it >> 2) & 15, cq = it & 3;
      const double* src = dinv + tile * 256 + rr * 16 + 4 * cq;
      const double2_t v0 = *reinterpret_cast<const double2_t*>(src);
      const double2_t v1 = *reinterpret_cast<const double2_t*>(src + 2);
      double* dst = Dt + tile * 256 + cq * 64 + rr;
      dst[0] = v0.x; dst[16] = v0.y; dst[32] = v1.x; dst[48] = v1.y;
    }
  }
  __syncthreads();
#ifdef LEAF_STAMPS
  if (blk == 0) LEAF_STAMP(6);
#endif
  double4_t La[2][7];  // double buffer over block columns j: La[j&1][i-j-1][s]
  double4_t Dj[2];
  auto load_col = [&](int set, int j) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) Dj[set][s4] = Dt[j * 256 + s4 * 64 + lane];
#pragma unroll
    for (int i = j + 1; i < 8; ++i)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) La[set][i - j - 1][s4] = Lt[strip_tile(i, j) * 256 + s4 * 64 + lane];
  };
  load_col(0, 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int cur = j & 1;
    if (j + 1 < 8) load_col(cur ^ 1, j + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the next column's reads ahead of this column's MFMAs
    // X_j = Dinv_j * T_j
    double4_t X = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) X = __builtin_amdgcn_mfma_f64_16x16x4f64(Dj[cur][s4], T[j][s4], X, 0, 0, 0);
    T[j] = X;
    const double4_t Xn = -X;
#pragma unroll
    for (int i = j + 1; i < 8; ++i)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        T[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(La[cur][i - j - 1][s4], Xn[s4], T[i], 0, 0, 0);
  }
#ifdef LEAF_STAMPS
  if (blk == 0) LEAF_STAMP(7);
#endif
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) Brow[16 * j + 4 * r + q] = T[j][r];
}

__global__ __launch_bounds__(256) void trsm_strip128_kernel(const double* __restrict__ Lblk, long lda,
                                                             const double* __restrict__ dinv,
                                                             double* __restrict__ B, long ldb, long strideL,
                                                             long strideB) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  trsm_strip128_body(Lblk + (long)blockIdx.y * strideL, lda, dinv + (long)blockIdx.y * 2048, B + (long)blockIdx.y * strideB,
                     ldb, blockIdx.x, smem, nullptr, nullptr);
}

// Leaf and the strip below it in ONE launch: workgroup 0 factors the diagonal block and raises `ready`; the other
// workgroups have their rows in registers by then and start the solve without a launch boundary in between.
__global__ __launch_bounds__(256, 1) void potrf_leaf_strip128_kernel(double* __restrict__ Ablk, long lda,
                                                                      double* __restrict__ dinv, int col0,
                                                                      int* __restrict__ info, double* __restrict__ B,
                                                                      long ldb, int* __restrict__ ready) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  if (blockIdx.x == 0) {
    potrf_leaf128_body(Ablk, lda, dinv, col0, info, smem);
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    trsm_strip128_body(Ablk, lda, dinv, B, ldb, blockIdx.x - 1, smem, ready, info);
  }
}

constexpr size_t LEAF_LDS_BYTES = sizeof(double) * (LEAF_ELEMS + 2 * SB * SB + LEAF + 2);
constexpr size_t STRIP_LDS_BYTES = sizeof(double) * STRIP_TILES * 256;

// With a whole CU's LDS requested the leaf only starts on a CU that holds nothing else: next to a persistent trailing
// update that leaves a few CUs free (mi_gp_set_option 9) it then runs at its stand-alone speed.
static int g_leaf_exclusive = 0;
void set_leaf_exclusive(int on) { g_leaf_exclusive = on ? 1 : 0; }
constexpr size_t LEAF_LDS_WHOLE_CU = 160 * 1024;

hipError_t leaf_enable_lds() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_leaf128_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS_WHOLE_CU);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_leaf_strip128_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS_WHOLE_CU);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_strip128_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)STRIP_LDS_BYTES);
}

hipError_t launch_potrf_leaf128(double* Ablk, long lda, double* dinv, int col0, int* info, hipStream_t stream) {
  potrf_leaf128_kernel<<<1, 256, g_leaf_exclusive ? LEAF_LDS_WHOLE_CU : LEAF_LDS_BYTES, stream>>>(Ablk, lda, dinv, col0, info);
  return hipGetLastError();
}

hipError_t launch_potrf_leaf_strip128(double* Ablk, long lda, double* dinv, int col0, int* info, double* B, long ldb, int m,
                                      int* ready, hipStream_t stream) {
  if (m <= 0) return launch_potrf_leaf128(Ablk, lda, dinv, col0, info, stream);
  const size_t lds = g_leaf_exclusive ? LEAF_LDS_WHOLE_CU : (LEAF_LDS_BYTES > STRIP_LDS_BYTES ? LEAF_LDS_BYTES : STRIP_LDS_BYTES);
  potrf_leaf_strip128_kernel<<<1 + m / 64, 256, lds, stream>>>(Ablk, lda, dinv, col0, info, B, ldb, ready);
  return hipGetLastError();
}

hipError_t launch_trsm_strip128(const double* Lblk, long lda, const double* dinv, double* B, long ldb, int m,
                                hipStream_t stream) {
  if (m <= 0) return hipSuccess;
  trsm_strip128_kernel<<<m / 64, 256, STRIP_LDS_BYTES, stream>>>(Lblk, lda, dinv, B, ldb, 0, 0);
  return hipGetLastError();
}

hipError_t launch_trsm_strip128_batched(const double* Lblk, long lda, long strideL, const double* dinv, double* B,
                                        long ldb, long strideB, int m, int batch, hipStream_t stream) {
  if (m <= 0 || batch <= 0) return hipSuccess;
  trsm_strip128_kernel<<<dim3(m / 64, batch), 256, STRIP_LDS_BYTES, stream>>>(Lblk, lda, dinv, B, ldb, strideL, strideB);
  return hipGetLastError();
}

}  // namespace migp
