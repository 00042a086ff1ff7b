// fp64 tiled GEMM / SYRK for gfx950 on v_mfma_f64_16x16x4_f64 with VGPR accumulators.
//
// Measured on MI355X (profiles/r01_probe_*.txt): v_mfma_f64_16x16x4_f64 issues every 64.0 cycles
// (77.0 TFLOP/s chip-wide, 98 % of the 78.6 TFLOP/s fp64 peak) when its C/D operand lives in VGPRs,
// but every 130 cycles (38 TFLOP/s) when C/D is in AGPRs -- so this library is compiled with
// -mllvm -amdgpu-mfma-vgpr-form.  (v_mfma_f64_4x4x4_4b_f64 reaches 75.8 TFLOP/s from either file;
// its lane map is in profiles/r01_probe_mfma_f64_4x4x4_lanemap.txt.)
// Lane maps of the 16x16x4 form (one f64 of A and of B per lane, 4 f64 of C/D per lane):
//   A[i = l&15][k = l>>4]   B[k = l>>4][j = l&15]   D[row = (l>>4) + 4r][col = l&15], r = 0..3
//
// Work decomposition: 128x128 output tile per 512-thread workgroup (8 waves as 2x4, 64x32 per
// wave = 4x2 MFMA tiles = 32 accumulator doubles per lane), K consumed in chunks of 32 through a
// double-buffered LDS image stored k-major ([k][x], leading dimension 144 doubles so both halves
// of a ds_read_b64 wave access hit disjoint banks).  The next chunk travels global -> registers
// during the first half of the current chunk's MFMAs and registers -> LDS in its middle; MFMA
// operand fragments are double-buffered in registers one k4-step ahead; the C tile is fetched at the
// start of the last chunk so the read-modify-write epilogue does not expose HBM latency.
//
// This kernel serves: K3 Cholesky trailing / panel updates (NT, lower),
// K7 triangular inverse levels (NN with triangular k-ranges) and L^-T L^-1 (TN), K8 predict
// triangular-solve updates (NT).
#include <cstdlib>
#include <type_traits>
#include "migp_kernels.h"

namespace migp {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int TILE = 128;
constexpr int BK = 32;
constexpr int LDS_LD = 144;
constexpr int OPER_ELEMS = BK * LDS_LD;  // one operand chunk in LDS
constexpr int NTHREADS = 512;
constexpr int NQ = TILE * BK / 2 / NTHREADS;  // 16-byte pieces per thread per operand chunk


// Staging of one 128 x BK operand chunk (512 threads, NQ x 16 B each).  XMAJOR: memory is [x][k]
// (x = row of A or column of B), KMAJOR: memory is [k][x].  Per-thread byte offsets are 32-bit and
// loop-invariant; the chunk advance is a uniform (scalar) pointer increment.
// Thread t owns piece (q, t): XMAJOR  x = 32q + (t>>8)*16 + (t&15), k-pair = (t>>4)&15;
//                              KMAJOR  k = 8q + (t>>6),               x-pair = t&63.
// so the q-dependence is a uniform stride in global memory and an immediate offset in LDS.
template <bool KMAJOR>
__device__ __forceinline__ void chunk_offsets(long ld, int tid, unsigned& goff, unsigned& loff, long& gstride) {
  if (KMAJOR) {
    const int k = tid >> 6, xc = tid & 63;
    goff = (unsigned)((k * ld + 2 * xc) * 8);
    loff = (unsigned)((k * LDS_LD + 2 * xc) * 8);
    gstride = 8 * ld * 8;
  } else {
    const int xl = tid & 15, kc = (tid >> 4) & 15, xh = tid >> 8;
    goff = (unsigned)(((xh * 16 + xl) * ld + 2 * kc) * 8);
    loff = (unsigned)(((2 * kc) * LDS_LD + xh * 16 + xl) * 8);
    gstride = 32 * ld * 8;
  }
}

__device__ __forceinline__ void chunk_load(const char* __restrict__ base, unsigned goff, long gstride,
                                           double2_t (&r)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) r[q] = *reinterpret_cast<const double2_t*>(base + q * gstride + goff);
}

template <bool KMAJOR>
__device__ __forceinline__ void chunk_store(char* __restrict__ lds, unsigned loff, const double2_t (&r)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (KMAJOR) {
      *reinterpret_cast<double2_t*>(lds + loff + q * (8 * LDS_LD * 8)) = r[q];
    } else {
      *reinterpret_cast<double*>(lds + loff + q * (32 * 8)) = r[q].x;
      *reinterpret_cast<double*>(lds + loff + q * (32 * 8) + LDS_LD * 8) = r[q].y;
    }
  }
}

__device__ __forceinline__ void tile_from_index(const GemmParams& p, int idx, int& ti, int& tj) {
  if (p.tri) {
    const int ntri = p.nt * (p.nt + 1) / 2;
    if (idx < ntri) {
      int t = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
      while ((t + 1) * (t + 2) / 2 <= idx) ++t;
      while (t * (t + 1) / 2 > idx) --t;
      ti = t;
      tj = idx - t * (t + 1) / 2;
    } else {
      const int rem = idx - ntri;
      ti = p.nt + rem / p.nt;
      tj = rem % p.nt;
    }
  } else {
    ti = idx / p.nt;
    tj = idx % p.nt;
  }
}

// Band-column-major enumeration of the lower trapezoid (tri) or the full rectangle: bands of R tile rows; inside a band the columns left to right,
// inside a column the band's rows top to bottom.  The 64 tiles an XCD works on at a time then cover ~R row strips and
// ~64/R column strips (each fetched into that L2 once and hit by the others) instead of one row strip and 64 column strips.
__device__ __forceinline__ void tile_from_index_banded(const GemmParams& p, int idx, int R, int& ti, int& tj) {
  auto before = [&](int r) {  // tiles in tile rows [0, r)
    if (!p.tri) return r * p.nt;
    return r <= p.nt ? r * (r + 1) / 2 : p.nt * (p.nt + 1) / 2 + (r - p.nt) * p.nt;
  };
  int lo = 0, hi = (p.mt + R - 1) / R - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (before(mid * R) <= idx) lo = mid; else hi = mid - 1;
  }
  const int r0 = lo * R, r1 = min(r0 + R, p.mt), h = r1 - r0;
  int e = idx - before(r0);
  const int cfull = p.tri ? min(r0 + 1, p.nt) : p.nt;  // columns that hold all h rows of the band
  if (e < cfull * h) { tj = e / h; ti = r0 + e % h; return; }
  e -= cfull * h;
  for (tj = cfull; tj < p.nt; ++tj) {    // the band's own triangle: column tj holds rows tj .. r1-1
    const int cnt = r1 - tj;
    if (e < cnt) { ti = tj + e; return; }
    e -= cnt;
  }
  ti = r1 - 1; tj = min(ti, p.nt - 1);
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_f64_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;                   // [2][BK][LDS_LD]
  double* Bs = smem + 2 * OPER_ELEMS;  // [2][BK][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;  // 2 x 4 waves, 64 x 32 outputs each

  // XCD-aware remap: blocks b, b+8, b+16.. share an XCD (and its L2); give each XCD a contiguous
  // run of tile indices so neighbouring tiles (same A strip, adjacent B strips) hit in that L2.
  const int nblk = gridDim.x;
  int idx = blockIdx.x;
  if (p.kmode == 0) {
    const int b = blockIdx.x, x = b & 7, q = nblk >> 3, r = nblk & 7;
    idx = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }  // triangular k-ranges: per-tile work varies with the tile index, keep the round-robin deal balanced
  int ti, tj;
  tile_from_index(p, idx, ti, tj);
  if (p.kmode == 2) ti = p.mt - 1 - ti;  // longest-K tiles (largest ti) first
  if (p.kmode == 4 && !p.tri) { tj = p.nt - 1 - idx / p.mt; ti = idx % p.mt; }  // k < (tj+1)*128: longest-k COLUMNS first
  const int i0 = ti * TILE, j0 = tj * TILE;
  int kbeg = 0, kend = p.k;
  if (p.kmode == 1) kbeg = j0;
  else if (p.kmode == 2) kend = i0 + TILE;
  else if (p.kmode == 3) kbeg = i0;
  else if (p.kmode == 4) kend = j0 + TILE;

  const double* A = p.A + (long)blockIdx.z * p.strideA;
  const double* B = p.B + (long)blockIdx.z * p.strideB;
  double* C = p.C + (long)blockIdx.z * p.strideC;

  double4_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int nchunk = (kend - kbeg) / BK;
  unsigned gA, lA, gB, lB;
  long sA, sB;
  chunk_offsets<A_KMAJOR>(p.lda, tid, gA, lA, sA);
  chunk_offsets<B_KMAJOR>(p.ldb, tid, gB, lB, sB);
  // uniform chunk pointers (scalar registers)
  const char* Ag = reinterpret_cast<const char*>(A_KMAJOR ? A + (long)kbeg * p.lda + i0 : A + (long)i0 * p.lda + kbeg);
  const char* Bg = reinterpret_cast<const char*>(B_KMAJOR ? B + (long)kbeg * p.ldb + j0 : B + (long)j0 * p.ldb + kbeg);
  const long stepA = (A_KMAJOR ? (long)BK * p.lda : (long)BK) * 8;
  const long stepB = (B_KMAJOR ? (long)BK * p.ldb : (long)BK) * 8;
  char* Asb = reinterpret_cast<char*>(As);
  char* Bsb = reinterpret_cast<char*>(Bs);
  double2_t ra[NQ], rb[NQ];
  if (nchunk > 0) {
    chunk_load(Ag, gA, sA, ra);
    chunk_load(Bg, gB, sB, rb);
    chunk_store<A_KMAJOR>(Asb, lA, ra);
    chunk_store<B_KMAJOR>(Bsb, lB, rb);
  }
  __syncthreads();

  // lane l of acc[a][b] holds rows 16a + (l>>4) + 4r (r = 0..3), column 16b + (l&15) of the wave's
  // 64x32 sub-tile
  const int kq = lane >> 4;  // k within a k4 step == row offset of the result
  const int l15 = lane & 15;
  const double* a_ptr = As + kq * LDS_LD + wr * 64 + l15;  // + 16*a
  const double* b_ptr = Bs + kq * LDS_LD + wc * 32 + l15;  // + 16*b

  double af[2][4], bf[2][2];  // register double buffer of the MFMA operands
  auto load_frags = [&](int set, int boff, int kk) {
    const double* ap = a_ptr + boff + kk * 4 * LDS_LD;
    const double* bp = b_ptr + boff + kk * 4 * LDS_LD;
#pragma unroll
    for (int a = 0; a < 4; ++a) af[set][a] = ap[16 * a];
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[set][b] = bp[16 * b];
  };
  // one K chunk: MFMAs on buffer boff while the next chunk travels global -> registers -> other buffer
  auto chunk_iter = [&](int boff, bool more) {
#ifndef EXP_NOLOAD
    if (more) {
      Ag += stepA;
      Bg += stepB;
      chunk_load(Ag, gA, sA, ra);
      chunk_load(Bg, gB, sB, rb);
    }
#endif
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int cur = kk & 1;
      if (kk + 1 < BK / 4) load_frags(cur ^ 1, boff, kk + 1);
      // keep the next step's LDS reads ahead of this step's MFMAs (hipcc otherwise sinks them to
      // their first use and every k4 step starts with an exposed LDS round trip)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][a], bf[cur][b], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#ifndef EXP_STORE_KK
#define EXP_STORE_KK (BK / 8)
#endif
      if (kk == EXP_STORE_KK && more) {
        // the other buffer was last read in the previous iteration (behind its barrier): refill it now,
        // well ahead of the barrier that publishes it
        const int noff = (boff ^ OPER_ELEMS) * 8;
        chunk_store<A_KMAJOR>(Asb + noff, lA, ra);
        chunk_store<B_KMAJOR>(Bsb + noff, lB, rb);
      }
    }
#ifndef EXP_NOBARRIER
    __syncthreads();
#endif
    if (more) load_frags(0, boff ^ OPER_ELEMS, 0);
  };

  if (nchunk > 0) load_frags(0, 0, 0);
  for (int c = 0; c + 1 < nchunk; ++c) chunk_iter((c & 1) * OPER_ELEMS, true);

  // last chunk: fetch the C tile first so its latency hides under this chunk's MFMAs
  const double alpha = p.alpha, beta = p.beta;
  double* cbase = C + (long)(i0 + wr * 64 + kq) * p.ldc + j0 + wc * 32 + l15;
  double4_t cv[4][2];
  if (beta != 0.0) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[a][b][r] = cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b];
  }
  if (nchunk > 0) chunk_iter(((nchunk - 1) & 1) * OPER_ELEMS, false);

#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = alpha * acc[a][b][r];
        if (beta != 0.0) v += beta * cv[a][b][r];
        cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b] = v;
      }
}

// ---------------------------------------------------------------------------------------------
// Variant B: 256-thread workgroups (4 waves as 2x2, 64x64 per wave = 4x4 MFMA tiles, 64 accumulator
// doubles per lane), K chunks of 16, 72 KiB of LDS -> TWO workgroups per CU whose barriers, prologues
// and C read-modify-write epilogues overlap each other's MFMA loops.
namespace vb {
constexpr int BKB = 16;
constexpr int OPER_B = BKB * LDS_LD;
constexpr int NT_B = 256;
constexpr int NQB = TILE * BKB / 2 / NT_B;  // 4

template <bool KMAJOR>
__device__ __forceinline__ void chunk_offsets(long ld, int tid, unsigned& goff, unsigned& loff, long& gstride) {
  if (KMAJOR) {
    const int k = tid >> 6, xc = tid & 63;  // k = 4q + (t>>6)
    goff = (unsigned)((k * ld + 2 * xc) * 8);
    loff = (unsigned)((k * LDS_LD + 2 * xc) * 8);
    gstride = 4 * ld * 8;
  } else {
    const int xl = tid & 15, kc = (tid >> 4) & 7, xh = tid >> 7;  // x = 32q + 16*(t>>7) + (t&15)
    goff = (unsigned)(((xh * 16 + xl) * ld + 2 * kc) * 8);
    loff = (unsigned)(((2 * kc) * LDS_LD + xh * 16 + xl) * 8);
    gstride = 32 * ld * 8;
  }
}
__device__ __forceinline__ void chunk_load(const char* __restrict__ base, unsigned goff, long gstride,
                                           double2_t (&r)[NQB]) {
#pragma unroll
  for (int q = 0; q < NQB; ++q) r[q] = *reinterpret_cast<const double2_t*>(base + q * gstride + goff);
}
template <bool KMAJOR>
__device__ __forceinline__ void chunk_store(char* __restrict__ lds, unsigned loff, const double2_t (&r)[NQB]) {
#pragma unroll
  for (int q = 0; q < NQB; ++q) {
    if (KMAJOR) {
      *reinterpret_cast<double2_t*>(lds + loff + q * (4 * LDS_LD * 8)) = r[q];
    } else {
      *reinterpret_cast<double*>(lds + loff + q * (32 * 8)) = r[q].x;
      *reinterpret_cast<double*>(lds + loff + q * (32 * 8) + LDS_LD * 8) = r[q].y;
    }
  }
}
}  // namespace vb

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(vb::NT_B, 2) void gemm_f64_kernel_b(GemmParams p) {
  using vb::BKB; using vb::OPER_B; using vb::NQB;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;               // [2][BKB][LDS_LD]
  double* Bs = smem + 2 * OPER_B;  // [2][BKB][LDS_LD]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  const int nblk = gridDim.x;
  int idx = blockIdx.x;
  if (p.kmode == 0) {
    const int b = blockIdx.x, x = b & 7, q = nblk >> 3, r = nblk & 7;
    idx = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  int ti, tj;
  // (full rectangles measured neutral: 8192^3 75.2 vs 75.7 TFLOP/s -- the row-major order stays for them)
  if (p.band > 0 && p.tri && p.kmode == 0) tile_from_index_banded(p, idx, p.band, ti, tj);
  else tile_from_index(p, idx, ti, tj);
  if (p.kmode == 2) ti = p.mt - 1 - ti;
  if (p.kmode == 4 && !p.tri) { tj = p.nt - 1 - idx / p.mt; ti = idx % p.mt; }  // longest-k columns first (LPT order)
  const int i0 = ti * TILE, j0 = tj * TILE;
  int kbeg = 0, kend = p.k;
  if (p.kmode == 1) kbeg = j0;
  else if (p.kmode == 2) kend = i0 + TILE;
  else if (p.kmode == 3) kbeg = i0;
  else if (p.kmode == 4) kend = j0 + TILE;

  const double* A = p.A + (long)blockIdx.z * p.strideA;
  const double* B = p.B + (long)blockIdx.z * p.strideB;
  double* C = p.C + (long)blockIdx.z * p.strideC;

  double4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int nchunk = (kend - kbeg) / BKB;
  unsigned gA, lA, gB, lB;
  long sA, sB;
  vb::chunk_offsets<A_KMAJOR>(p.lda, tid, gA, lA, sA);
  vb::chunk_offsets<B_KMAJOR>(p.ldb, tid, gB, lB, sB);
  const char* Ag = reinterpret_cast<const char*>(A_KMAJOR ? A + (long)kbeg * p.lda + i0 : A + (long)i0 * p.lda + kbeg);
  const char* Bg = reinterpret_cast<const char*>(B_KMAJOR ? B + (long)kbeg * p.ldb + j0 : B + (long)j0 * p.ldb + kbeg);
  const long stepA = (A_KMAJOR ? (long)BKB * p.lda : (long)BKB) * 8;
  const long stepB = (B_KMAJOR ? (long)BKB * p.ldb : (long)BKB) * 8;
  char* Asb = reinterpret_cast<char*>(As);
  char* Bsb = reinterpret_cast<char*>(Bs);
  // Global -> register prefetch runs TWO chunks ahead (register set c & 1 holds chunk c until it is written to
  // LDS buffer c & 1 during chunk c - 1): ~7k cycles of latency tolerance instead of ~3k, enough for an
  // HBM / MALL miss under load while two workgroups share the CU.
  double2_t ra[2][NQB], rb[2][NQB];
  if (nchunk > 0) {
    vb::chunk_load(Ag, gA, sA, ra[0]);
    vb::chunk_load(Bg, gB, sB, rb[0]);
    if (nchunk > 1) {
      Ag += stepA;
      Bg += stepB;
      vb::chunk_load(Ag, gA, sA, ra[1]);
      vb::chunk_load(Bg, gB, sB, rb[1]);
    }
    vb::chunk_store<A_KMAJOR>(Asb, lA, ra[0]);
    vb::chunk_store<B_KMAJOR>(Bsb, lB, rb[0]);
  }
  __syncthreads();

  const int kq = lane >> 4, l15 = lane & 15;
  const double* a_ptr = As + kq * LDS_LD + wr * 64 + l15;
  const double* b_ptr = Bs + kq * LDS_LD + wc * 64 + l15;
  double af[2][4], bf[2][4];
  auto load_frags = [&](int set, int boff, int kk) {
    const double* ap = a_ptr + boff + kk * 4 * LDS_LD;
    const double* bp = b_ptr + boff + kk * 4 * LDS_LD;
#pragma unroll
    for (int a = 0; a < 4; ++a) af[set][a] = ap[16 * a];
#pragma unroll
    for (int b = 0; b < 4; ++b) bf[set][b] = bp[16 * b];
  };
  if (nchunk > 0) load_frags(0, 0, 0);

  // one chunk; S = c & 1 selects both the LDS buffer being consumed and the register set being refilled.
  // The body is branch-free (loads / stores of the last chunks are redundant instead of skipped) so that it is
  // one scheduling region, and sched_group_barrier spreads the memory instructions between the 64 MFMAs:
  //   k4-step 0: 8 global loads (chunk c+2) + the 4 ds_read2 of step 1     step 1, 2: the 4 ds_read2 of the next step
  //   k4-step 3: the 8 ds_write2 of chunk c+1
  // A burst of LDS / VMEM issue in both co-resident workgroups at once left the MFMA pipe 88 % busy; spread out
  // it is 93 % (8192^3: 70.3 -> 73.3 TFLOP/s, rocBLAS 72.9).
  auto chunk_body = [&](int c, auto S) {
    constexpr int s = decltype(S)::value;
    constexpr int boff = s * OPER_B;
    const bool adv = (c + 2 < nchunk);
    Ag += adv ? stepA : 0;
    Bg += adv ? stepB : 0;
#pragma unroll
    for (int kk = 0; kk < BKB / 4; ++kk) {
      const int cur = kk & 1;
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) {
        vb::chunk_load(Ag, gA, sA, ra[s]);
        vb::chunk_load(Bg, gB, sB, rb[s]);
      }
      if (kk + 1 < BKB / 4) load_frags(cur ^ 1, boff, kk + 1);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][a], bf[cur][b], acc[a][b], 0, 0, 0);
      if (kk == BKB / 4 - 1) {
        constexpr int noff = (boff ^ OPER_B) * 8;
        vb::chunk_store<A_KMAJOR>(Asb + noff, lA, ra[s ^ 1]);
        vb::chunk_store<B_KMAJOR>(Bsb + noff, lB, rb[s ^ 1]);
      }
      if (kk == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      } else if (kk + 1 < BKB / 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      } else {
        constexpr int nw = (A_KMAJOR ? NQB : 2 * NQB) + (B_KMAJOR ? NQB : 2 * NQB);  // ds_write instructions
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (g < nw) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    load_frags(0, boff ^ OPER_B, 0);
  };
  for (int c = 0; c < nchunk; c += 2) {  // k is a multiple of 128, so nchunk is even
    chunk_body(c, std::integral_constant<int, 0>());
    chunk_body(c + 1, std::integral_constant<int, 1>());
  }

  const double alpha = p.alpha, beta = p.beta;
  double* cbase = C + (long)(i0 + wr * 64 + kq) * p.ldc + j0 + wc * 64 + l15;
  if (beta != 0.0) {
    // read-modify-write in four row groups, the next group's loads in flight while this one is stored
    double4_t cv[2][4];
    auto load_group = [&](int set, int a) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[set][b][r] = cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b];
    };
    load_group(0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (a + 1 < 4) load_group((a + 1) & 1, a + 1);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b] = alpha * acc[a][b][r] + beta * cv[a & 1][b][r];
    }
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b] = alpha * acc[a][b][r];
  }
}

// ---------------------------------------------------------------------------------------------
// Variant C (option, off by default: mi_gp_set_option(h, 9, 1)): one 512-thread workgroup per CU (8 waves as 2x4,
// 64x32 outputs per wave = 4x2 MFMA tiles, 32 accumulator doubles per lane), K chunks of 16, the same 72 KiB LDS
// image and the same interleaved, branch-free schedule as variant B -- but 142 VGPRs and half a CU's LDS, so that
// every CU keeps room for one workgroup of the panel chain while a look-ahead trailing update runs.  Stand-alone
// it reaches 73.0 TFLOP/s on 8192^3 (B: 75.5) and 66.8 on the k=1024 trapezoid (B: 69.1).  As the look-ahead bulk
// kernel it changes nothing (N=16384: 30.8 vs 30.7 ms): the chain's kernels do find a slot at once, but next to
// MFMA-saturated waves they still run 1.7-2.2x slower than alone (leaf 75 us vs 34) -- it is execution
// contention on the CU, not waiting for a free slot, that stretches the panel stream.
// Persistent form (option 9 = n > 1): n workgroups, each asking for a whole CU's LDS, walk the tiles with stride n, which
// pins the trailing update to n CUs and leaves 256 - n to the panel chain.  With n = 224 (28 per XCD) the leaf runs at
// its stand-alone 33 us and the strip at 42 us next to it, but the in-panel GEMMs (24 % of all flops) are then confined to
// 32 CUs and the trailing update loses 1/8 of the chip: 29.9 vs 30.5 ms with plain launches, no change under graph replay;
// n = 232 or 192 are 2-3 ms worse, switching n by phase does not help (tools/dev_ab8.py).  Throughput work is conserved:
// a static CU split only moves the bottleneck.  n | 0x1000 keeps the half-CU LDS request (strips and in-panel GEMMs
// may share the bulk CUs) and only the leaf, asking for a whole CU (option 12), is confined to the 256 - n free ones:
// with n = 248 the leaf runs in 38 us and the strip in 39 us next to the trailing update and the panel stream is busy
// 24 instead of 30 of the 29.7 ms -- but the bulk kernel then carries the wait (1.62 vs 1.41 ms per launch) and the
// evaluation gains 0-2 %.  With every kernel's own efficiency as it is (bulk ~63, 64x64-tile ~48 TFLOP/s) perfect
// packing would be ~28 ms; what is left is kernel efficiency, not scheduling.  Kept as building blocks.
namespace vc {
constexpr int BKC = 16;
constexpr int OPER_C = BKC * LDS_LD;
constexpr int NT_C = 512;
constexpr int NQC = TILE * BKC / 2 / NT_C;  // 2

template <bool KMAJOR>
__device__ __forceinline__ void chunk_offsets(long ld, int tid, unsigned& goff, unsigned& loff, long& gstride) {
  if (KMAJOR) {
    const int k = tid >> 6, xc = tid & 63;  // k = 8q + (t>>6)
    goff = (unsigned)((k * ld + 2 * xc) * 8);
    loff = (unsigned)((k * LDS_LD + 2 * xc) * 8);
    gstride = 8 * ld * 8;
  } else {
    const int xl = tid & 15, kc = (tid >> 4) & 7, xh = tid >> 7;  // x = 64q + 16*(t>>7) + (t&15)
    goff = (unsigned)(((xh * 16 + xl) * ld + 2 * kc) * 8);
    loff = (unsigned)(((2 * kc) * LDS_LD + xh * 16 + xl) * 8);
    gstride = 64 * ld * 8;
  }
}
__device__ __forceinline__ void chunk_load(const char* __restrict__ base, unsigned goff, long gstride,
                                           double2_t (&r)[NQC]) {
#pragma unroll
  for (int q = 0; q < NQC; ++q) r[q] = *reinterpret_cast<const double2_t*>(base + q * gstride + goff);
}
template <bool KMAJOR>
__device__ __forceinline__ void chunk_store(char* __restrict__ lds, unsigned loff, const double2_t (&r)[NQC]) {
#pragma unroll
  for (int q = 0; q < NQC; ++q) {
    if (KMAJOR) {
      *reinterpret_cast<double2_t*>(lds + loff + q * (8 * LDS_LD * 8)) = r[q];
    } else {
      *reinterpret_cast<double*>(lds + loff + q * (64 * 8)) = r[q].x;
      *reinterpret_cast<double*>(lds + loff + q * (64 * 8) + LDS_LD * 8) = r[q].y;
    }
  }
}
}  // namespace vc

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(vc::NT_C, 1) void gemm_f64_kernel_c(GemmParams p) {
  using vc::BKC; using vc::OPER_C; using vc::NQC;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;               // [2][BKC][LDS_LD]
  double* Bs = smem + 2 * OPER_C;  // [2][BKC][LDS_LD]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;

  // p.ntiles > gridDim.x: persistent form -- workgroup b handles tiles b, b + gridDim.x, ... (a grid smaller than the
  // chip with a whole-CU LDS request pins the trailing update to that many CUs and leaves the others to the chain)
  const int ntiles = p.ntiles > 0 ? p.ntiles : (int)gridDim.x;
  // XCD-aware walk (workgroup b runs on XCD b % 8): each XCD owns one contiguous eighth of the tile list, so that
  // neighbouring tiles (same A strip / same B strip) meet in one L2; within it the XCD's workgroups stride
  const int xcd = blockIdx.x & 7, wx = blockIdx.x >> 3;
  const int nwx = ((int)gridDim.x + 7 - xcd) >> 3;                       // workgroups of this launch on this XCD
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tbeg = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tcnt = tq + (xcd < tr ? 1 : 0);
  const bool xmap = (p.kmode == 0);
  for (int t = xmap ? wx : (int)blockIdx.x; t < (xmap ? tcnt : ntiles); t += xmap ? nwx : (int)gridDim.x) {
  const int idx = xmap ? tbeg + t : t;
  int ti, tj;
  tile_from_index(p, idx, ti, tj);
  if (p.kmode == 2) ti = p.mt - 1 - ti;
  if (p.kmode == 4 && !p.tri) { tj = p.nt - 1 - idx / p.mt; ti = idx % p.mt; }  // longest-k columns first (LPT order)
  const int i0 = ti * TILE, j0 = tj * TILE;
  int kbeg = 0, kend = p.k;
  if (p.kmode == 1) kbeg = j0;
  else if (p.kmode == 2) kend = i0 + TILE;
  else if (p.kmode == 3) kbeg = i0;
  else if (p.kmode == 4) kend = j0 + TILE;

  const double* A = p.A + (long)blockIdx.z * p.strideA;
  const double* B = p.B + (long)blockIdx.z * p.strideB;
  double* C = p.C + (long)blockIdx.z * p.strideC;

  double4_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int nchunk = (kend - kbeg) / BKC;
  unsigned gA, lA, gB, lB;
  long sA, sB;
  vc::chunk_offsets<A_KMAJOR>(p.lda, tid, gA, lA, sA);
  vc::chunk_offsets<B_KMAJOR>(p.ldb, tid, gB, lB, sB);
  const char* Ag = reinterpret_cast<const char*>(A_KMAJOR ? A + (long)kbeg * p.lda + i0 : A + (long)i0 * p.lda + kbeg);
  const char* Bg = reinterpret_cast<const char*>(B_KMAJOR ? B + (long)kbeg * p.ldb + j0 : B + (long)j0 * p.ldb + kbeg);
  const long stepA = (A_KMAJOR ? (long)BKC * p.lda : (long)BKC) * 8;
  const long stepB = (B_KMAJOR ? (long)BKC * p.ldb : (long)BKC) * 8;
  char* Asb = reinterpret_cast<char*>(As);
  char* Bsb = reinterpret_cast<char*>(Bs);
  double2_t ra[2][NQC], rb[2][NQC];  // global -> register prefetch two chunks ahead (see variant B)
  if (nchunk > 0) {
    vc::chunk_load(Ag, gA, sA, ra[0]);
    vc::chunk_load(Bg, gB, sB, rb[0]);
    if (nchunk > 1) {
      Ag += stepA;
      Bg += stepB;
      vc::chunk_load(Ag, gA, sA, ra[1]);
      vc::chunk_load(Bg, gB, sB, rb[1]);
    }
    vc::chunk_store<A_KMAJOR>(Asb, lA, ra[0]);
    vc::chunk_store<B_KMAJOR>(Bsb, lB, rb[0]);
  }
  __syncthreads();

  const int kq = lane >> 4, l15 = lane & 15;
  const double* a_ptr = As + kq * LDS_LD + wr * 64 + l15;
  const double* b_ptr = Bs + kq * LDS_LD + wc * 32 + l15;
  double af[2][4], bf[2][2];
  auto load_frags = [&](int set, int boff, int kk) {
    const double* ap = a_ptr + boff + kk * 4 * LDS_LD;
    const double* bp = b_ptr + boff + kk * 4 * LDS_LD;
#pragma unroll
    for (int a = 0; a < 4; ++a) af[set][a] = ap[16 * a];
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[set][b] = bp[16 * b];
  };
  if (nchunk > 0) load_frags(0, 0, 0);

  auto chunk_body = [&](int c, auto S) {
    constexpr int s = decltype(S)::value;
    constexpr int boff = s * OPER_C;
    const bool adv = (c + 2 < nchunk);
    Ag += adv ? stepA : 0;
    Bg += adv ? stepB : 0;
#pragma unroll
    for (int kk = 0; kk < BKC / 4; ++kk) {
      const int cur = kk & 1;
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) {
        vc::chunk_load(Ag, gA, sA, ra[s]);
        vc::chunk_load(Bg, gB, sB, rb[s]);
      }
      if (kk + 1 < BKC / 4) load_frags(cur ^ 1, boff, kk + 1);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][a], bf[cur][b], acc[a][b], 0, 0, 0);
      if (kk == BKC / 4 - 1) {
        constexpr int noff = (boff ^ OPER_C) * 8;
        vc::chunk_store<A_KMAJOR>(Asb + noff, lA, ra[s ^ 1]);
        vc::chunk_store<B_KMAJOR>(Bsb + noff, lB, rb[s ^ 1]);
      }
      if (kk == 0) {  // 8 MFMAs, 4 global loads, 3 ds_read2
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      } else if (kk + 1 < BKC / 4) {  // 8 MFMAs, 3 ds_read2
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      } else {  // 8 MFMAs and the LDS writes of the next chunk
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    load_frags(0, boff ^ OPER_C, 0);
  };
  for (int c = 0; c < nchunk; c += 2) {  // k is a multiple of 128, so nchunk is even
    chunk_body(c, std::integral_constant<int, 0>());
    chunk_body(c + 1, std::integral_constant<int, 1>());
  }

  const double alpha = p.alpha, beta = p.beta;
  double* cbase = C + (long)(i0 + wr * 64 + kq) * p.ldc + j0 + wc * 32 + l15;
  if (beta != 0.0) {
    double4_t cv[2][2];
    auto load_group = [&](int set, int a) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[set][b][r] = cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b];
    };
    load_group(0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (a + 1 < 4) load_group((a + 1) & 1, a + 1);
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b] = alpha * acc[a][b][r] + beta * cv[a & 1][b][r];
    }
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b] = alpha * acc[a][b][r];
  }
  __syncthreads();  // the next tile's prologue overwrites the LDS buffers
  }
}

// ---------------------------------------------------------------------------------------------
// Variant S: 64x64 output tile per 256-thread workgroup (4 waves as 2x2, 32x32 per wave) for launches
// with too few 128x128 tiles to fill the chip (the in-panel updates of the Cholesky, a few dozen to a
// few hundred tiles, which sit on the factorisation's critical path): 4x the workgroups, 1/4 of the
// per-workgroup latency.  GemmParams.mt / nt are reinterpreted in 64-row tiles by the launcher.
namespace vs {
constexpr int TS = 64;
constexpr int BKS = 16;
constexpr int LDS_S = 80;  // 64 + 16: odd k rows land 16 bank-pairs away from even ones
constexpr int OPER_S = BKS * LDS_S;
constexpr int NQS = TS * BKS / 2 / 256;  // 2

template <bool KMAJOR>
__device__ __forceinline__ void chunk_offsets(long ld, int tid, unsigned& goff, unsigned& loff, long& gstride) {
  if (KMAJOR) {
    const int k = tid >> 5, xc = tid & 31;  // k = 8q + (t>>5)
    goff = (unsigned)((k * ld + 2 * xc) * 8);
    loff = (unsigned)((k * LDS_S + 2 * xc) * 8);
    gstride = 8 * ld * 8;
  } else {
    const int xl = tid & 15, kc = (tid >> 4) & 7, xh = tid >> 7;  // x = 32q + 16*(t>>7) + (t&15)
    goff = (unsigned)(((xh * 16 + xl) * ld + 2 * kc) * 8);
    loff = (unsigned)(((2 * kc) * LDS_S + xh * 16 + xl) * 8);
    gstride = 32 * ld * 8;
  }
}
__device__ __forceinline__ void chunk_load(const char* __restrict__ base, unsigned goff, long gstride,
                                           double2_t (&r)[NQS]) {
#pragma unroll
  for (int q = 0; q < NQS; ++q) r[q] = *reinterpret_cast<const double2_t*>(base + q * gstride + goff);
}
template <bool KMAJOR>
__device__ __forceinline__ void chunk_store(char* __restrict__ lds, unsigned loff, const double2_t (&r)[NQS]) {
#pragma unroll
  for (int q = 0; q < NQS; ++q) {
    if (KMAJOR) {
      *reinterpret_cast<double2_t*>(lds + loff + q * (8 * LDS_S * 8)) = r[q];
    } else {
      *reinterpret_cast<double*>(lds + loff + q * (32 * 8)) = r[q].x;
      *reinterpret_cast<double*>(lds + loff + q * (32 * 8) + LDS_S * 8) = r[q].y;
    }
  }
}
}  // namespace vs

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel_s(GemmParams p) {
  using vs::BKS; using vs::OPER_S; using vs::NQS; using vs::LDS_S; using vs::TS;
  __shared__ __attribute__((aligned(16))) double smem[4 * vs::OPER_S];
  double* As = smem;
  double* Bs = smem + 2 * OPER_S;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  int ti, tj;
  tile_from_index(p, blockIdx.x, ti, tj);
  if (p.kmode == 2) ti = p.mt - 1 - ti;
  if (p.kmode == 4 && !p.tri) { tj = p.nt - 1 - (int)blockIdx.x / p.mt; ti = (int)blockIdx.x % p.mt; }  // longest-k columns first (LPT order)
  const int i0 = ti * TS, j0 = tj * TS;
  int kbeg = 0, kend = p.k;
  if (p.kmode == 1) kbeg = j0;
  else if (p.kmode == 2) kend = i0 + TS;
  else if (p.kmode == 3) kbeg = i0;
  else if (p.kmode == 4) kend = j0 + TS;
  kbeg &= ~(BKS - 1);

  const double* A = p.A + (long)blockIdx.z * p.strideA;
  const double* B = p.B + (long)blockIdx.z * p.strideB;
  double* C = p.C + (long)blockIdx.z * p.strideC;

  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int nchunk = (kend - kbeg) / BKS;
  unsigned gA, lA, gB, lB;
  long sA, sB;
  vs::chunk_offsets<A_KMAJOR>(p.lda, tid, gA, lA, sA);
  vs::chunk_offsets<B_KMAJOR>(p.ldb, tid, gB, lB, sB);
  const char* Ag = reinterpret_cast<const char*>(A_KMAJOR ? A + (long)kbeg * p.lda + i0 : A + (long)i0 * p.lda + kbeg);
  const char* Bg = reinterpret_cast<const char*>(B_KMAJOR ? B + (long)kbeg * p.ldb + j0 : B + (long)j0 * p.ldb + kbeg);
  const long stepA = (A_KMAJOR ? (long)BKS * p.lda : (long)BKS) * 8;
  const long stepB = (B_KMAJOR ? (long)BKS * p.ldb : (long)BKS) * 8;
  char* Asb = reinterpret_cast<char*>(As);
  char* Bsb = reinterpret_cast<char*>(Bs);
  double2_t ra[NQS], rb[NQS];
  if (nchunk > 0) {
    vs::chunk_load(Ag, gA, sA, ra);
    vs::chunk_load(Bg, gB, sB, rb);
    vs::chunk_store<A_KMAJOR>(Asb, lA, ra);
    vs::chunk_store<B_KMAJOR>(Bsb, lB, rb);
  }
  // C tile early: these launches are latency-bound, the read hides under the whole k loop
  const int kq = lane >> 4, l15 = lane & 15;
  const double alpha = p.alpha, beta = p.beta;
  double* cbase = C + (long)(i0 + wr * 32 + kq) * p.ldc + j0 + wc * 32 + l15;
  double4_t cv[2][2];
  if (beta != 0.0) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[a][b][r] = cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b];
  }
  __syncthreads();

  const double* a_ptr = As + kq * LDS_S + wr * 32 + l15;
  const double* b_ptr = Bs + kq * LDS_S + wc * 32 + l15;
  // Same schedule as variant B at a quarter of the tile: branch-free chunk body, operand fragments double-buffered
  // one k4-step ahead, and the chunk's 4 global loads / 8 fragment reads / LDS writes spread between its 16 MFMAs.
  double fa[2][2], fb[2][2];
  auto load_frags = [&](int set, int boff, int kk) {
    const double* ap = a_ptr + boff + kk * 4 * LDS_S;
    const double* bp = b_ptr + boff + kk * 4 * LDS_S;
    fa[set][0] = ap[0]; fa[set][1] = ap[16]; fb[set][0] = bp[0]; fb[set][1] = bp[16];
  };
  if (nchunk > 0) load_frags(0, 0, 0);
  for (int c = 0; c < nchunk; ++c) {
    const int boff = (c & 1) * OPER_S;
    const bool adv = (c + 1 < nchunk);
    Ag += adv ? stepA : 0;
    Bg += adv ? stepB : 0;
#pragma unroll
    for (int kk = 0; kk < BKS / 4; ++kk) {
      const int cur = kk & 1;
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) {
        vs::chunk_load(Ag, gA, sA, ra);
        vs::chunk_load(Bg, gB, sB, rb);
      }
      if (kk + 1 < BKS / 4) load_frags(cur ^ 1, boff, kk + 1);
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
      if (kk == BKS / 4 - 1) {
        const int noff = (boff ^ OPER_S) * 8;
        vs::chunk_store<A_KMAJOR>(Asb + noff, lA, ra);
        vs::chunk_store<B_KMAJOR>(Bsb + noff, lB, rb);
      }
      if (kk == 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      } else if (kk + 1 < BKS / 4) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    load_frags(0, boff ^ OPER_S, 0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double v = alpha * acc[a][b][r];
        if (beta != 0.0) v += beta * cv[a][b][r];
        cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b] = v;
      }
}

static int tile_count(const GemmParams& p) {
  if (!p.tri) return p.mt * p.nt;
  return p.nt * (p.nt + 1) / 2 + (p.mt - p.nt) * p.nt;
}

static int g_variant = -1;
static int gemm_variant() {
  if (g_variant < 0) {
    const char* e = getenv("MIGP_GEMM_VARIANT");
    g_variant = (e && e[0] == 'A') ? 0 : 1;
  }
  return g_variant;
}

static int g_small_tiles = 1024;
static int g_band_rows = 8;  // > 0: band-column-major tile order for the uniform-k trapezoid launches of the 128-tile kernel
void set_gemm_band_rows(int v) { g_band_rows = v; }
constexpr size_t LDS_ONE_PER_CU = 82432;  // > half a CU (one workgroup per CU) and <= 160 KB - the 79 KB leaf image
constexpr size_t LDS_WHOLE_CU = 160 * 1024;  // 160 KB per CU: two of these do not fit, one + a 76 KB leaf does
void set_gemm_variant(int v) { g_variant = v; }
void set_gemm_small_tiles(int v) { g_small_tiles = v; }
static int tile_count(const GemmParams& p);
bool gemm_uses_small_tiles(const GemmParams& p, int batch) {
  return gemm_variant() == 1 && tile_count(p) * batch < g_small_tiles && p.kmode != 2;
}
int gemm_variant_get() { return gemm_variant(); }

hipError_t launch_gemm_f64(const GemmParams& p, int opA_kmajor, int opB_kmajor, int batch, hipStream_t stream) {
  const int nblk = tile_count(p);
  if (nblk <= 0 || batch <= 0) return hipSuccess;
  if (gemm_uses_small_tiles(p, batch)) {
    // few 128x128 tiles: cut them into 64x64 ones (same enumeration, tile units halve)
    GemmParams q = p;
    q.mt = 2 * p.mt;
    q.nt = 2 * p.nt;
    dim3 grid(tile_count(q), 1, batch), block(256);
    if (!opA_kmajor && !opB_kmajor) gemm_f64_kernel_s<false, false><<<grid, block, 0, stream>>>(q);
    else if (!opA_kmajor && opB_kmajor) gemm_f64_kernel_s<false, true><<<grid, block, 0, stream>>>(q);
    else if (opA_kmajor && opB_kmajor) gemm_f64_kernel_s<true, true><<<grid, block, 0, stream>>>(q);
    else gemm_f64_kernel_s<true, false><<<grid, block, 0, stream>>>(q);
    return hipGetLastError();
  }
  static const bool force_wide = getenv("MIGP_GEMM_WIDE") != nullptr;  // dev harnesses only
  if (gemm_variant() == 1 && (p.wide8 || force_wide)) {
    GemmParams pc = p;
    int g = nblk;
    size_t lds = sizeof(double) * 4 * vc::OPER_C;
    if (p.wide8 > 1 && batch == 1 && nblk > (p.wide8 & 0xfff)) {
      // persistent on (wide8 & 0xfff) CUs; bit 12 clear: each CU taken whole (nothing else fits next to the bulk
      // workgroup), bit 12 set: half a CU's LDS only (the bulk-free CUs are then merely the ones a whole-CU request
      // such as the exclusive leaf can still find)
      pc.ntiles = nblk;
      g = p.wide8 & 0xfff;
      if (!(p.wide8 & 0x1000)) lds = LDS_WHOLE_CU;
    }
    const GemmParams& p = pc;
    dim3 grid(g, 1, batch), block(vc::NT_C);
    if (!opA_kmajor && !opB_kmajor) gemm_f64_kernel_c<false, false><<<grid, block, lds, stream>>>(p);
    else if (!opA_kmajor && opB_kmajor) gemm_f64_kernel_c<false, true><<<grid, block, lds, stream>>>(p);
    else if (opA_kmajor && opB_kmajor) gemm_f64_kernel_c<true, true><<<grid, block, lds, stream>>>(p);
    else gemm_f64_kernel_c<true, false><<<grid, block, lds, stream>>>(p);
    return hipGetLastError();
  }
  if (gemm_variant() == 1) {
    dim3 grid(nblk, 1, batch), block(vb::NT_B);
    const size_t lds = p.one_per_cu ? LDS_ONE_PER_CU : sizeof(double) * 4 * vb::OPER_B;
    GemmParams pb = p;
    pb.band = g_band_rows;
    const GemmParams& p = pb;
    if (!opA_kmajor && !opB_kmajor) gemm_f64_kernel_b<false, false><<<grid, block, lds, stream>>>(p);
    else if (!opA_kmajor && opB_kmajor) gemm_f64_kernel_b<false, true><<<grid, block, lds, stream>>>(p);
    else if (opA_kmajor && opB_kmajor) gemm_f64_kernel_b<true, true><<<grid, block, lds, stream>>>(p);
    else gemm_f64_kernel_b<true, false><<<grid, block, lds, stream>>>(p);
    return hipGetLastError();
  }
  dim3 grid(nblk, 1, batch), block(NTHREADS);
  const size_t lds = sizeof(double) * 4 * OPER_ELEMS;
  if (!opA_kmajor && !opB_kmajor) gemm_f64_kernel<false, false><<<grid, block, lds, stream>>>(p);
  else if (!opA_kmajor && opB_kmajor) gemm_f64_kernel<false, true><<<grid, block, lds, stream>>>(p);
  else if (opA_kmajor && opB_kmajor) gemm_f64_kernel<true, true><<<grid, block, lds, stream>>>(p);
  else gemm_f64_kernel<true, false><<<grid, block, lds, stream>>>(p);
  return hipGetLastError();
}

hipError_t gemm_f64_enable_lds() {
  hipError_t e;
  {
    const int lds = (int)(sizeof(double) * 4 * OPER_ELEMS);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
  }
  {
    const int ldsc = (int)LDS_WHOLE_CU;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_c<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsc);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_c<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsc);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_c<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsc);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_c<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsc);
    if (e != hipSuccess) return e;
  }
  const int ldsb = (int)LDS_ONE_PER_CU;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_b<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_b<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_b<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_b<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
}

}  // namespace migp
