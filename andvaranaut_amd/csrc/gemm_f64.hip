// fp64 tiled GEMM / SYRK for gfx950 on v_mfma_f64_16x16x4_f64 with VGPR accumulators.
//
// Measured on MI355X (profiles/r01_probe_*.txt): v_mfma_f64_16x16x4_f64 issues every 64.0 cycles
// (77.0 TFLOP/s chip-wide, 98 % of the 78.6 TFLOP/s fp64 peak) when its C/D operand lives in VGPRs,
// but every 130 cycles (38 TFLOP/s) when C/D is in AGPRs -- so this library is compiled with
// -mllvm -amdgpu-mfma-vgpr-form.
// Lane maps of the 16x16x4 form (one f64 of A and of B per lane, 4 f64 of C/D per lane):
//   A[i = l&15][k = l>>4]   B[k = l>>4][j = l&15]   D[row = (l>>4) + 4r][col = l&15], r = 0..3
//
// Two kernels, one design:
//   gemm_f64_kernel_b : 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 4x4 MFMA tiles =
//                       64 accumulator doubles per lane), K consumed in chunks of 16 through a double-buffered LDS image
//                       stored k-major ([k][x], leading dimension 144 doubles so both halves of a ds_read_b64 wave access
//                       hit disjoint banks), 72 KiB of LDS -> two workgroups per CU.  Global -> register prefetch runs
//                       two chunks ahead, registers -> LDS one chunk ahead, MFMA operand fragments are double-buffered in
//                       registers one k4-step ahead, and sched_group_barrier spreads the memory instructions between
//                       the 64 MFMAs of a chunk.
//   gemm_f64_kernel_s : the same schedule on 64x64 tiles for launches with too few 128x128 tiles to fill the chip; its
//                       row-major operands are fetched as whole 128-byte row segments and stored XOR-swizzled (round 4).
//
// They serve: K3 Cholesky trailing / panel updates (NT, lower), K7 triangular
// inverse levels (NN with triangular k-ranges) and L^-T L^-1 (TN), K8 predict triangular-solve updates (NT).
// (Two rejected designs -- an 8-wave / one-workgroup-per-CU 128x128 kernel with BK = 32, 62 TFLOP/s, and an 8-wave half-CU
// kernel with persistent CU-subset modes, 73 TFLOP/s but no gain next to the panel chain -- lived here in round 1; DESIGN.md
// section 3 and 8 keep what they showed, git history keeps the code.)
#include <cstdlib>
#include <type_traits>
#include "migp_kernels.h"

namespace migp {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int TILE = 128;
constexpr int LDS_LD = 144;

__device__ __forceinline__ void tile_from_index(int tri, int nt, int idx, int& ti, int& tj) {
  if (tri) {
    const int ntri = nt * (nt + 1) / 2;
    if (idx < ntri) {
      int t = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
      while ((t + 1) * (t + 2) / 2 <= idx) ++t;
      while (t * (t + 1) / 2 > idx) --t;
      ti = t;
      tj = idx - t * (t + 1) / 2;
    } else {
      const int rem = idx - ntri;
      ti = nt + rem / nt;
      tj = rem % nt;
    }
  } else {
    ti = idx / nt;
    tj = idx % nt;
  }
}
__device__ __forceinline__ void tile_from_index(const GemmParams& p, int idx, int& ti, int& tj) {
  tile_from_index(p.tri, p.nt, idx, ti, tj);
}

// Band-column-major enumeration of the lower trapezoid (tri) or the full rectangle: bands of R tile rows; inside a band the columns left to right,
// inside a column the band's rows top to bottom.  The 64 tiles an XCD works on at a time then cover ~R row strips and
// ~64/R column strips (each fetched into that L2 once and hit by the others) instead of one row strip and 64 column strips.
__device__ __forceinline__ void tile_from_index_banded(int tri, int mt, int nt, int idx, int R, int& ti, int& tj) {
  auto before = [&](int r) {  // tiles in tile rows [0, r)
    if (!tri) return r * nt;
    return r <= nt ? r * (r + 1) / 2 : nt * (nt + 1) / 2 + (r - nt) * nt;
  };
  int lo = 0, hi = (mt + R - 1) / R - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (before(mid * R) <= idx) lo = mid; else hi = mid - 1;
  }
  const int r0 = lo * R, r1 = min(r0 + R, mt), h = r1 - r0;
  int e = idx - before(r0);
  const int cfull = tri ? min(r0 + 1, nt) : nt;  // columns that hold all h rows of the band
  if (e < cfull * h) { tj = e / h; ti = r0 + e % h; return; }
  e -= cfull * h;
  for (tj = cfull; tj < nt; ++tj) {    // the band's own triangle: column tj holds rows tj .. r1-1
    const int cnt = r1 - tj;
    if (e < cnt) { ti = tj + e; return; }
    e -= cnt;
  }
  ti = r1 - 1; tj = min(ti, nt - 1);
}

__device__ __forceinline__ void tile_from_index_banded(const GemmParams& p, int idx, int R, int& ti, int& tj) {
  tile_from_index_banded(p.tri, p.mt, p.nt, idx, R, ti, tj);
}

// First-columns-first enumeration of the lower trapezoid (GemmParams::fc): the fc leading tile columns row by row, then the
// sub-trapezoid of the remaining columns in the banded order.
__device__ __forceinline__ void tile_from_index_fc(int mt, int nt, int fc, int band, int idx, int& ti, int& tj) {
  const int ft = fc * (fc + 1) / 2 + (mt - fc) * fc;
  if (idx < ft) {
    tile_from_index(1, fc, idx, ti, tj);
  } else {
    if (band > 0) tile_from_index_banded(1, mt - fc, nt - fc, idx - ft, band, ti, tj);
    else tile_from_index(1, nt - fc, idx - ft, ti, tj);
    ti += fc;
    tj += fc;
  }
}

// Panel-list mode (GemmParams::pl): tile `idx` of the launch -> tile-unit coordinates of its A rows, B rows, C rows and C
// columns, and whether it is a diagonal tile of its panel's trapezoid.  The table walk is uniform per workgroup (scalar loads).
__device__ __forceinline__ void list_tile(const GemmParams& p, int idx, int& ra, int& rb, int& rc, int& cc, bool& diag) {
  const int want = idx + p.pl[p.pl_first].x;
  int lo = p.pl_first, hi = p.pl_first + p.pl_n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (p.pl[mid].x <= want) lo = mid; else hi = mid - 1;
  }
  const int4 d = p.pl[lo];
  int ti, tj;
  if (p.band > 0) tile_from_index_banded(1, p.pl_rows - d.y, d.w, want - d.x, p.band, ti, tj);
  else tile_from_index(1, d.w, want - d.x, ti, tj);
  ra = d.y - p.pl_abase + ti;
  rb = d.y - p.pl_abase + tj;
  rc = d.y + ti;
  cc = d.z + tj;
  diag = (ti == tj);
}

// ---------------------------------------------------------------------------------------------
// 128x128 tiles: 256-thread workgroups (4 waves as 2x2, 64x64 per wave = 4x4 MFMA tiles, 64 accumulator
// doubles per lane), K chunks of 16, 72 KiB of LDS -> TWO workgroups per CU whose barriers, prologues
// and C read-modify-write epilogues overlap each other's MFMA loops.
namespace vb {
constexpr int BKB = 16;
constexpr int OPER_B = BKB * LDS_LD;
constexpr int NT_B = 256;
constexpr int NQB = TILE * BKB / 2 / NT_B;  // 4
// Coalesced row-major operand loads + XOR-swizzled LDS image (chunk_offsets), as in the 64x64-tile kernel, where they are
// worth 1.3 % of an N = 16384 evaluation.  OFF here: this kernel needs half the bytes per flop and is not bound by what a CU
// can fetch -- with the swizzle an evaluation measures 25.72-25.89 ms against 25.81-25.84 without (same box, alternating),
// and the eight fragment bases instead of two cost 6 more spilled VGPRs (10 instead of 4) at the 256-register ceiling.
#ifndef MIGP_SWZ_B
#define MIGP_SWZ_B 0
#endif
constexpr bool SWZ = MIGP_SWZ_B != 0;

template <bool KMAJOR>
__device__ __forceinline__ void chunk_offsets(long ld, int tid, unsigned& goff, unsigned& loff, long& gstride) {
  if (KMAJOR) {
    const int k = tid >> 6, xc = tid & 63;  // k = 4q + (t>>6)
    goff = (unsigned)((k * ld + 2 * xc) * 8);
    loff = (unsigned)((k * LDS_LD + 2 * xc) * 8);
    gstride = 4 * ld * 8;
  } else if (SWZ) {
    // coalesced row segments + XOR-swizzled image, as in the 64x64-tile kernel (vs::chunk_offsets has the reasoning)
    const int kc = tid & 7, row = tid >> 3;  // x = 32q + (t>>3), k = 2 (t&7)
    goff = (unsigned)((row * ld + 2 * kc) * 8);
    loff = (unsigned)(((2 * kc) * LDS_LD + (row ^ (2 * kc))) * 8);
    gstride = 32 * ld * 8;
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
  idx += p.tile0;  // sub-range launch: the XCD-aware order above is taken inside the range
  int ti, tj;
  int rc, cc;  // C tile row / column (differ from the operand rows ti / tj only in panel-list mode)
  if (p.pl) {
    bool diag;
    list_tile(p, idx, ti, tj, rc, cc, diag);
  } else {
    // (full rectangles measured neutral: 8192^3 75.2 vs 75.7 TFLOP/s -- the row-major order stays for them)
    if (p.fc > 0) tile_from_index_fc(p.mt, p.nt, p.fc, p.band, idx, ti, tj);
    else if (p.band > 0 && p.tri && p.kmode == 0) tile_from_index_banded(p, idx, p.band, ti, tj);
    else tile_from_index(p, idx, ti, tj);
    if (p.kmode == 2) ti = p.mt - 1 - ti;
    if (p.kmode == 4 && !p.tri) { tj = p.nt - 1 - idx / p.mt; ti = idx % p.mt; }  // longest-k columns first (LPT order)
    rc = ti;
    cc = tj;
  }
  const int i0 = ti * TILE, j0 = tj * TILE;
  int kbeg = 0, kend = p.k;
  if (p.kmode == 1) kbeg = j0;
  else if (p.kmode == 2) kend = i0 + TILE;
  else if (p.kmode == 3) kbeg = i0;
  else if (p.kmode == 4) kend = j0 + TILE;

  // blockIdx.z = z1 + batch1 * z2 (two-level batch: nodes of a triangular-inverse level x problems of a batched evaluation)
  const long z1 = p.batch1 > 0 ? (long)(blockIdx.z % p.batch1) : (long)blockIdx.z, z2 = p.batch1 > 0 ? (long)(blockIdx.z / p.batch1) : 0;
  const double* A = p.A + z1 * p.strideA + z2 * p.strideA2;
  const double* B = p.B + z1 * p.strideB + z2 * p.strideB2;
  double* C = p.C + z1 * p.strideC + z2 * p.strideC2;

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
  // k-segmented operands: the step INTO chunk n is a jump to the next segment's base when n is a multiple of kseg / BKB
  const int segmask = p.kseg > 0 ? p.kseg / BKB - 1 : 0x7fffffff;
  const long segjump = (p.kseg_stride - (long)(p.kseg - BKB)) * 8;
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
  const double* a_ptr[4];  // per k4-step: the row-major operands' images are swizzled by the k pair, k & 14 = 4 kk + (kq & 2)
  const double* b_ptr[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int sa = (vb::SWZ && !A_KMAJOR) ? (4 * kk + (kq & 2)) : 0, sb = (vb::SWZ && !B_KMAJOR) ? (4 * kk + (kq & 2)) : 0;
    a_ptr[kk] = As + kq * LDS_LD + wr * 64 + (l15 ^ sa);
    b_ptr[kk] = Bs + kq * LDS_LD + wc * 64 + (l15 ^ sb);
  }
  double af[2][4], bf[2][4];
  auto load_frags = [&](int set, int boff, int kk) {
    const double* ap = a_ptr[kk] + boff + kk * 4 * LDS_LD;
    const double* bp = b_ptr[kk] + boff + kk * 4 * LDS_LD;
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
  // C read-modify-write: the first two of its four row groups are requested during the LAST two chunks of the k loop (in
  // place of those chunks' global prefetches, which would be redundant reloads), so that their round trip hides under
  // 128 MFMAs instead of following the loop
  const double alpha = p.alpha, beta = p.beta;
  double* cbase = C + (long)(rc * TILE + wr * 64 + kq) * p.ldc + cc * TILE + wc * 64 + l15;
  double4_t cv[2][4];
  auto load_group = [&](int set, int a) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[set][b][r] = cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b];
  };
  auto chunk_body = [&](int c, auto S, auto LASTC) {
    constexpr int s = decltype(S)::value;
    constexpr int lastc = decltype(LASTC)::value;  // 0: inside the loop; 1 / 2: second-to-last / last chunk
    constexpr int boff = s * OPER_B;
    const bool adv = (c + 2 < nchunk);
    const bool seg = ((c + 2) & segmask) == 0;
    Ag += adv ? (seg ? segjump : stepA) : 0;
    Bg += adv ? (seg ? segjump : stepB) : 0;
#pragma unroll
    for (int kk = 0; kk < BKB / 4; ++kk) {
      const int cur = kk & 1;
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) {
        if (lastc == 0) {
          vb::chunk_load(Ag, gA, sA, ra[s]);
          vb::chunk_load(Bg, gB, sB, rb[s]);
        } else {
          load_group(lastc - 1, lastc - 1);
        }
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
  int c = 0;
  for (; c + 2 < nchunk; c += 2) {  // k is a multiple of 128, so nchunk is even and >= 8
    chunk_body(c, std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
    chunk_body(c + 1, std::integral_constant<int, 1>(), std::integral_constant<int, 0>());
  }
  chunk_body(c, std::integral_constant<int, 0>(), std::integral_constant<int, 1>());
  chunk_body(c + 1, std::integral_constant<int, 1>(), std::integral_constant<int, 2>());

  if (beta != 0.0) {
    // row groups 0 and 1 are in registers; groups 2 and 3 are requested while 0 and 1 are stored
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          cbase[(long)(16 * a + 4 * r) * p.ldc + 16 * b] = alpha * acc[a][b][r] + beta * cv[a & 1][b][r];
      if (a + 2 < 4) load_group(a & 1, a + 2);
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
// 64x64 tiles: 64x64 output tile per 256-thread workgroup (4 waves as 2x2, 32x32 per wave) for launches
// with too few 128x128 tiles to fill the chip (the in-panel updates of the Cholesky, a few dozen to a
// few hundred tiles, which sit on the factorisation's critical path): 4x the workgroups, 1/4 of the
// per-workgroup latency.  GemmParams.mt / nt are reinterpreted in 64-row tiles by the launcher.
namespace vs {
constexpr int TS = 64;
constexpr int BKS = 16;
constexpr int LDS_S = 80;  // 64 + 16: odd k rows land 16 bank-pairs away from even ones
constexpr int OPER_S = BKS * LDS_S;
constexpr int NQS = TS * BKS / 2 / 256;  // 2
constexpr int NBUF = 3;                  // LDS buffers per operand (see the pipeline note in the kernel)
#ifndef MIGP_SWZ_S
#define MIGP_SWZ_S 1
#endif
constexpr bool SWZ = MIGP_SWZ_S != 0;    // coalesced row-major operand loads + XOR-swizzled LDS image (chunk_offsets)
#ifndef MIGP_XCD_MAP_S
#define MIGP_XCD_MAP_S 1
#endif

template <bool KMAJOR>
__device__ __forceinline__ void chunk_offsets(long ld, int tid, unsigned& goff, unsigned& loff, long& gstride) {
  if (KMAJOR) {
    const int k = tid >> 5, xc = tid & 31;  // k = 8q + (t>>5)
    goff = (unsigned)((k * ld + 2 * xc) * 8);
    loff = (unsigned)((k * LDS_S + 2 * xc) * 8);
    gstride = 8 * ld * 8;
  } else if (SWZ) {
    // Eight consecutive lanes fetch ONE row's 128-byte chunk segment (a whole cache line per row, 8 rows per wave
    // instruction) instead of sixteen lanes fetching 16 bytes of sixteen different rows.  The k-major LDS image is then
    // written with the column XOR-swizzled by the k pair, col = x ^ (k & 14): a wave's 16-lane write groups (8 k pairs x 2
    // rows) hit 16 distinct bank pairs, and a fragment read (16 consecutive x at one k, XORed with a constant below 16)
    // stays a permutation of its aligned 16-column block, so the read side keeps its conflict-free pattern.
    const int kc = tid & 7, row = tid >> 3;  // x = 32q + (t>>3), k = 2 (t&7)
    goff = (unsigned)((row * ld + 2 * kc) * 8);
    loff = (unsigned)(((2 * kc) * LDS_S + (row ^ (2 * kc))) * 8);
    gstride = 32 * ld * 8;
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

// FLUSH: the k-segmented update (GemmParams::kflush) -- an instantiation of its own, so that the plain kernel keeps the code
// and the registers it was tuned with (the flush as a run-time branch cost every launch 4 %: N = 16384 25.0 -> 25.9 ms)
template <bool A_KMAJOR, bool B_KMAJOR, bool FLUSH = false>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel_s(GemmParams p) {
  using vs::BKS; using vs::OPER_S; using vs::NQS; using vs::LDS_S; using vs::TS;
  if (p.hiprio) __builtin_amdgcn_s_setprio(3);  // panel-chain launches: win the SIMD's issue arbitration against bulk waves
  __shared__ __attribute__((aligned(16))) double smem[2 * vs::NBUF * vs::OPER_S];
  double* As = smem;
  double* Bs = smem + vs::NBUF * OPER_S;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  int ti, tj;
  int rc, cc;  // C tile row / column in 64-row units (differ from the operand rows ti / tj only in panel-list mode)
  // XCD-aware block -> tile map for uniform-k launches, as in the 128x128-tile kernel: workgroup b runs on XCD b % 8, and
  // each XCD takes a CONTIGUOUS range of the enumeration, so the tiles of one tile row (consecutive indices: they share
  // their A rows) meet in one L2 instead of eight.
  int bid = (int)blockIdx.x;
  if (MIGP_XCD_MAP_S && p.kmode == 0) {
    const int nblk = (int)gridDim.x, x = bid & 7, q = nblk >> 3, r = nblk & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  if (p.pl) {
    // panel-list launch (or the tail of one): this workgroup is one quadrant of 128x128 tile sub_base + bid / 4
    int ra, rb, prc, pcc;
    bool diag;
    const int quad = bid & 3;
    list_tile(p, p.sub_base + (bid >> 2), ra, rb, prc, pcc, diag);
    if (diag && quad == 1) return;  // the quadrant above the diagonal of a diagonal tile
    ti = 2 * ra + (quad >> 1);
    tj = 2 * rb + (quad & 1);
    rc = 2 * prc + (quad >> 1);
    cc = 2 * pcc + (quad & 1);
  } else {
    if (p.sub_base >= 0) {
      // tail of a 128x128-tile launch: this workgroup is one quadrant of parent tile sub_base + blockIdx.x / 4
      int pti, ptj;
      const int pidx = p.sub_base + (bid >> 2), quad = bid & 3;
      if (p.fc > 0) tile_from_index_fc(p.sub_mt, p.sub_nt, p.fc, p.band, pidx, pti, ptj);
      else if (p.band > 0 && p.tri) tile_from_index_banded(p.tri, p.sub_mt, p.sub_nt, pidx, p.band, pti, ptj);
      else tile_from_index(p.tri, p.sub_nt, pidx, pti, ptj);
      ti = 2 * pti + (quad >> 1);
      tj = 2 * ptj + (quad & 1);
      if (p.tri && tj > ti) return;  // the quadrant above the diagonal of a diagonal parent tile
      if (p.dead_last_half && ti == 2 * p.sub_mt - 1) return;  // the all-zero half of the y^T tile row
    } else {
      tile_from_index(p, bid, ti, tj);
      if (p.dead_last_half && ti == p.mt - 1) return;
    }
    if (p.kmode == 2) ti = p.mt - 1 - ti;
    if (p.kmode == 4 && !p.tri) { tj = p.nt - 1 - (int)blockIdx.x / p.mt; ti = (int)blockIdx.x % p.mt; }  // longest-k columns first (LPT order)
    rc = ti;
    cc = tj;
  }
  const int i0 = ti * TS, j0 = tj * TS;
  int kbeg = 0, kend = p.k;
  if (p.kmode == 1) kbeg = j0;
  else if (p.kmode == 2) kend = i0 + TS;
  else if (p.kmode == 3) kbeg = i0;
  else if (p.kmode == 4) kend = j0 + TS;
  kbeg &= ~(BKS - 1);

  // blockIdx.z = z1 + batch1 * z2 (two-level batch: nodes of a triangular-inverse level x problems of a batched evaluation)
  const long z1 = p.batch1 > 0 ? (long)(blockIdx.z % p.batch1) : (long)blockIdx.z, z2 = p.batch1 > 0 ? (long)(blockIdx.z / p.batch1) : 0;
  const double* A = p.A + z1 * p.strideA + z2 * p.strideA2;
  const double* B = p.B + z1 * p.strideB + z2 * p.strideB2;
  double* C = p.C + z1 * p.strideC + z2 * p.strideC2;

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
  const int segmask = p.kseg > 0 ? p.kseg / BKS - 1 : 0x7fffffff;  // k-segmented operands, as in the 128x128-tile kernel
  const long segjump = (p.kseg_stride - (long)(p.kseg - BKS)) * 8;
  char* Asb = reinterpret_cast<char*>(As);
  char* Bsb = reinterpret_cast<char*>(Bs);
  // Pipeline (chunk = 16 k = 16 MFMAs per wave, 0.43 us): chunk c is requested from global memory during chunk c - 4
  // (one-chunk prefetch, round 1: every chunk waited for its load, 1.4 us per chunk), parked in register set c % 3,
  // written to LDS buffer c % 3 at the end of chunk c - 2, and its first operand fragments are read during the last
  // k-step of chunk c - 1 -- BEFORE that chunk's barrier, which is possible because the buffer became visible one barrier
  // earlier.  With two LDS buffers the fragment read followed the barrier and every chunk began with an LDS round trip
  // (~150 cycles of 1024 + 150); the third buffer costs 20 KB (60 KB per workgroup, two workgroups per CU still fit).
  constexpr int NB = vs::NBUF;  // 3
  double2_t ra[NB][NQS], rb[NB][NQS];
  auto advance = [&](int next_chunk) {  // chunks beyond the last are redundant reloads of the last one, never consumed
    const bool adv = next_chunk < nchunk;
    const bool seg = (next_chunk & segmask) == 0;
    Ag += adv ? (seg ? segjump : stepA) : 0;
    Bg += adv ? (seg ? segjump : stepB) : 0;
  };
  if (nchunk > 0) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      vs::chunk_load(Ag, gA, sA, ra[u]);
      vs::chunk_load(Bg, gB, sB, rb[u]);
      advance(u + 1);
    }
    vs::chunk_store<A_KMAJOR>(Asb, lA, ra[0]);
    vs::chunk_store<B_KMAJOR>(Bsb, lB, rb[0]);
    vs::chunk_store<A_KMAJOR>(Asb + OPER_S * 8, lA, ra[1]);
    vs::chunk_store<B_KMAJOR>(Bsb + OPER_S * 8, lB, rb[1]);
    vs::chunk_load(Ag, gA, sA, ra[0]);  // chunk 3
    vs::chunk_load(Bg, gB, sB, rb[0]);
    advance(4);
  }
  // C tile early: these launches are latency-bound, the read hides under the whole k loop
  const int kq = lane >> 4, l15 = lane & 15;
  const double alpha = p.alpha, beta = p.beta;
  double* cbase = C + (long)(rc * TS + wr * 32 + kq) * p.ldc + cc * TS + wc * 32 + l15;
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

  // fragment bases per k4-step: the row-major operands' images are swizzled by the k pair (vs::chunk_offsets), k & 14 =
  // 4 kk + (kq & 2) for k = 4 kk + kq
  const double* a_ptr[4];
  const double* b_ptr[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int sa = (vs::SWZ && !A_KMAJOR) ? (4 * kk + (kq & 2)) : 0, sb = (vs::SWZ && !B_KMAJOR) ? (4 * kk + (kq & 2)) : 0;
    a_ptr[kk] = As + kq * LDS_S + wr * 32 + (l15 ^ sa);
    b_ptr[kk] = Bs + kq * LDS_S + wc * 32 + (l15 ^ sb);
  }
  // Same schedule as the 128x128 kernel at a quarter of the tile: branch-free chunk body, operand fragments double-buffered
  // one k4-step ahead, and the chunk's 4 global loads / 8 fragment reads / LDS writes spread between its 16 MFMAs.
  double fa[2][2], fb[2][2];
  auto load_frags = [&](int set, int boff, int kk) {
    const double* ap = a_ptr[kk] + boff + kk * 4 * LDS_S;
    const double* bp = b_ptr[kk] + boff + kk * 4 * LDS_S;
    fa[set][0] = ap[0]; fa[set][1] = ap[16]; fb[set][0] = bp[0]; fb[set][1] = bp[16];
  };
  if (nchunk > 0) load_frags(0, 0, 0);
  // one chunk c; S = c % 3: LDS buffer consumed; register set (S + 1) % 3 is refilled with chunk c + 4; register set and
  // LDS buffer (S + 2) % 3 take part in the hand-over of chunk c + 2
  const int kflush_chunks = FLUSH ? p.kflush / BKS : 1;
  auto chunk_body = [&](int c, auto Sc) {
    constexpr int sidx = decltype(Sc)::value;
    constexpr int boff = sidx * OPER_S;
    constexpr int s1 = (sidx + 1) % NB, s2 = (sidx + 2) % NB;
#pragma unroll
    for (int kk = 0; kk < BKS / 4; ++kk) {
      const int cur = kk & 1;
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 0) {
        vs::chunk_load(Ag, gA, sA, ra[s1]);
        vs::chunk_load(Bg, gB, sB, rb[s1]);
      }
      if (kk + 1 < BKS / 4) load_frags(cur ^ 1, boff, kk + 1);
      else load_frags(cur ^ 1, s1 * OPER_S, 0);  // first fragments of the next chunk, ahead of the barrier
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][0], fb[cur][0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][0], fb[cur][1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][1], fb[cur][0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[cur][1], fb[cur][1], acc[1][1], 0, 0, 0);
      if (kk == BKS / 4 - 1) {
        vs::chunk_store<A_KMAJOR>(Asb + s2 * OPER_S * 8, lA, ra[s2]);
        vs::chunk_store<B_KMAJOR>(Bsb + s2 * OPER_S * 8, lB, rb[s2]);
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
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
        }
      }
    }
    advance(c + 5);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FLUSH) if ((c + 1) % kflush_chunks == 0 && c + 1 < nchunk) {
      // k-segmented update (GemmParams::kflush): C takes this segment's sum NOW, rounded as a launch of its own would round
      // it, and the accumulators start the next segment at zero -- one launch then returns the bits of one launch per segment
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            double v = alpha * acc[a][b][r];
            v += beta * cv[a][b][r];
            cv[a][b][r] = v;
          }
          acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
        }
    }
    __syncthreads();
  };
  int c = 0;
  for (; c + 3 <= nchunk; c += 3) {
    chunk_body(c, std::integral_constant<int, 0>());
    chunk_body(c + 1, std::integral_constant<int, 1>());
    chunk_body(c + 2, std::integral_constant<int, 2>());
  }
  if (c < nchunk) chunk_body(c, std::integral_constant<int, 0>());  // c is a multiple of 3 here
  if (c + 1 < nchunk) chunk_body(c + 1, std::integral_constant<int, 1>());
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


constexpr size_t LDS_ONE_PER_CU = 82432;  // > half a CU (one workgroup per CU) and <= 160 KB - the 79 KB leaf image

static int tile_count(const GemmParams& p) {
  if (p.pl) return p.pl_tiles;
  if (!p.tri) return p.mt * p.nt;
  return p.nt * (p.nt + 1) / 2 + (p.mt - p.nt) * p.nt;
}


// launches with fewer 128x128 tiles than p.small_below run on 64x64 tiles (4x the workgroups, 1/4 of the latency)
bool gemm_uses_small_tiles(const GemmParams& p, int batch) {
  if (p.kflush > 0) return true;  // (the 64x64-tile kernel holds its C tile in registers from the start: only it can flush per k-segment)
  return tile_count(p) * batch < p.small_below && p.kmode != 2;
}

// The 128x128-tile kernel runs its tiles in rounds of 512 (two workgroups on each of 256 CUs, 218 us per round at
// k = 1024): a last round with few tiles costs a whole round (7 % of the nine bulk launches of an N = 16384
// factorisation).  Uniform-k launches therefore hand the tiles beyond the last full round to the 64x64-tile kernel (a
// quarter of the time per round): worth it up to 384 such tiles (three quarter-rounds).  The rule does not look at
// one_per_cu or the stream, so results stay bit-identical across the scheduling options.
constexpr int ROUND_TILES = 512, TAIL_MAX_TILES = 384;
constexpr int NARROW_TRAPEZOID_COLS = 8;
int gemm_tail_tiles(const GemmParams& p, int batch) {
  if (!p.tail_small || p.kmode != 0 || batch != 1 || gemm_uses_small_tiles(p, batch)) return 0;
  const int nblk = tile_count(p), rem = nblk % ROUND_TILES;
  return (nblk > ROUND_TILES && rem > 0 && rem <= TAIL_MAX_TILES) ? rem : 0;
}

hipError_t launch_gemm_f64(const GemmParams& p_in, int opA_kmajor, int opB_kmajor, int batch, hipStream_t stream, int part) {
  GemmParams p = p_in;
  // Narrow trapezoids (the next-panel and in-panel updates: at most 8 tile columns under a tall panel) keep the row-major
  // order: a tile row's <= 8 tiles share one 1 MB row strip of A and the column strips (<= 8 MB) are common to all rows,
  // while bands of 8 rows x 8 columns touch 16 MB per 64 tiles (15360 x 1024, k = 1024 on 128x128 tiles: 545 vs 558 us;
  // 8192 x 1024: 253 vs 264 us).  Wide trapezoids (the bulk updates) keep the band-column-major order of section 5.1.
  if (p.tri && !p.pl && p.nt <= NARROW_TRAPEZOID_COLS) p.band = 0;
  if (!(p.tri && !p.pl && p.kmode == 0 && p.fc > 0 && p.fc < p.nt)) p.fc = 0;
  if (p.kseg != 0 && (p.kseg < 0 || p.kseg % 128 || p.kmode != 0 || opA_kmajor || opB_kmajor || p.k % p.kseg))
    return hipErrorInvalidValue;  // k-segments: whole tile columns, the NT form, uniform k
  const int nblk = tile_count(p);
  if (nblk <= 0 || batch <= 0) return hipSuccess;
  const bool small = gemm_uses_small_tiles(p, batch);
  const int tail = gemm_tail_tiles(p, batch);
  // 64x64 tiles: the whole product (few 128x128 tiles: same enumeration, tile units halve) or the `ntail` last tiles of
  // a 128x128-tile launch
  auto launch_small = [&](int ntail) -> hipError_t {
    GemmParams q = p;
    q.mt = 2 * p.mt;
    q.nt = 2 * p.nt;
    int nwg = tile_count(q);
    if (ntail > 0) {
      q.sub_base = nblk - ntail;
      q.sub_mt = p.mt;
      q.sub_nt = p.nt;
      nwg = 4 * ntail;
    } else if (p.pl) {  // a whole panel-list launch on 64x64 tiles: four quadrant workgroups per 128x128 tile
      q.sub_base = 0;
      nwg = 4 * nblk;
    }
    dim3 grid(nwg, 1, batch), block(256);
    // one_per_cu: unused dynamic LDS on top of the 60 KB static image pushes the request over half a CU
    const size_t pad = p.one_per_cu ? LDS_ONE_PER_CU - sizeof(double) * 2 * vs::NBUF * vs::OPER_S : 0;
    if (q.kflush > 0) {
      if (opA_kmajor || opB_kmajor || q.kmode != 0 || q.beta == 0.0 || q.kflush % vs::BKS) return hipErrorInvalidValue;
      gemm_f64_kernel_s<false, false, true><<<grid, block, pad, stream>>>(q);
    } else if (!opA_kmajor && !opB_kmajor) gemm_f64_kernel_s<false, false><<<grid, block, pad, stream>>>(q);
    else if (!opA_kmajor && opB_kmajor) gemm_f64_kernel_s<false, true><<<grid, block, pad, stream>>>(q);
    else if (opA_kmajor && opB_kmajor) gemm_f64_kernel_s<true, true><<<grid, block, pad, stream>>>(q);
    else gemm_f64_kernel_s<true, false><<<grid, block, pad, stream>>>(q);
    return hipGetLastError();
  };
  if (small) return (part == 2 || p.tile0 > 0) ? hipSuccess : launch_small(0);  // small launches are never split
  // sub-range [t0, t1) of the enumeration: the 128x128-tile kernel takes what lies in front of the tail, the tail goes
  // with the range that reaches the end
  const int t0 = p.tile0, t1 = p.tile_cnt > 0 ? (t0 + p.tile_cnt < nblk ? t0 + p.tile_cnt : nblk) : nblk;
  const int big_end = t1 < nblk - tail ? t1 : nblk - tail;
  if (part != 2 && big_end > t0) {
    dim3 grid(big_end - t0, 1, batch), block(vb::NT_B);
    const size_t lds = p.one_per_cu ? LDS_ONE_PER_CU : sizeof(double) * 4 * vb::OPER_B;
    if (!opA_kmajor && !opB_kmajor) gemm_f64_kernel_b<false, false><<<grid, block, lds, stream>>>(p);
    else if (!opA_kmajor && opB_kmajor) gemm_f64_kernel_b<false, true><<<grid, block, lds, stream>>>(p);
    else if (opA_kmajor && opB_kmajor) gemm_f64_kernel_b<true, true><<<grid, block, lds, stream>>>(p);
    else gemm_f64_kernel_b<true, false><<<grid, block, lds, stream>>>(p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return (tail > 0 && part != 1 && t1 == nblk) ? launch_small(tail) : hipSuccess;
}

hipError_t gemm_f64_enable_lds() {
  const int ldsb = (int)LDS_ONE_PER_CU;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_s<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_s<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_s<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_s<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_s<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_b<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_b<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_b<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel_b<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
}

}  // namespace migp
