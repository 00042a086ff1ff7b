// fp64 tiled GEMM / SYRK for gfx950 on v_mfma_f64_4x4x4_4b_f64.
//
// Why the 4x4x4 form: measured on MI355X (profiles/r01_probe_*.txt) v_mfma_f64_16x16x4_f64
// issues every ~139 cycles (35.8 TFLOP/s chip-wide) while v_mfma_f64_4x4x4_4b_f64 issues every
// ~17 cycles (74.8 TFLOP/s, the 78.6 TFLOP/s fp64 peak).  The 4-block instruction only forms the
// four DIAGONAL 4x4 blocks of a 16x16 outer product, so each 16x16 output tile is built from four
// MFMAs whose B operand has its 4-column blocks rotated by s = 0..3 (the rotation is free: it is
// just a different LDS read address per lane).  Lane maps (decoded by one-hot probing,
// profiles/r01_probe_mfma_f64_4x4x4_lanemap.txt):
//   A operand lane p : A_blk[i][k]  with k = p>>4, blk = (p>>2)&3, i = p&3
//   B operand lane p : B_blk[k][j]  with k = p>>4, blk = (p>>2)&3, j = p&3
//   D result  lane l : D_blk[i][j]  with i = l>>4, blk = (l>>2)&3, j = l&3
//   blgp bit0 negates A, bit1 negates B; cbsz/abid have no effect on this opcode.
//
// Work decomposition: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per
// wave = 4x4 tiles of 16x16 = 64 accumulator doubles per lane), K consumed in chunks of 16 through
// a double-buffered LDS image stored k-major ([k][x], leading dimension 144 doubles so both
// halves of a ds_read_b64 wave access hit disjoint banks).  Two workgroups per CU so one
// workgroup's C read-modify-write epilogue overlaps the other's MFMA loop.
//
// This kernel serves: K3 Cholesky trailing / panel updates (NT, lower),
// K7 triangular inverse levels (NN with triangular k-ranges) and L^-T L^-1 (TN), K8 predict
// triangular-solve updates (NT).
#include "migp_kernels.h"

namespace migp {

typedef double double2_t __attribute__((ext_vector_type(2)));

constexpr int TILE = 128;
constexpr int BK = 16;
constexpr int LDS_LD = 144;
constexpr int OPER_ELEMS = BK * LDS_LD;  // one operand chunk in LDS

// Stage one 128 x 16 operand chunk from global memory into registers.
// XMAJOR: memory is [x][k] (x = row of A or column of B), KMAJOR: memory is [k][x].
template <bool KMAJOR>
__device__ __forceinline__ void chunk_load(const double* __restrict__ base, long ld, int x0, int k0, int tid,
                                           double2_t (&r)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = tid + 256 * q;
    if (KMAJOR) {
      const int k = p >> 6, xc = p & 63;
      r[q] = *reinterpret_cast<const double2_t*>(base + (long)(k0 + k) * ld + x0 + 2 * xc);
    } else {
      const int xl = p & 15, kc = (p >> 4) & 7, xh = p >> 7;
      r[q] = *reinterpret_cast<const double2_t*>(base + (long)(x0 + xh * 16 + xl) * ld + k0 + 2 * kc);
    }
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void chunk_store(double* __restrict__ lds, int tid, const double2_t (&r)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = tid + 256 * q;
    if (KMAJOR) {
      const int k = p >> 6, xc = p & 63;
      *reinterpret_cast<double2_t*>(lds + k * LDS_LD + 2 * xc) = r[q];
    } else {
      const int xl = p & 15, kc = (p >> 4) & 7, xh = p >> 7;
      lds[(2 * kc) * LDS_LD + xh * 16 + xl] = r[q].x;
      lds[(2 * kc + 1) * LDS_LD + xh * 16 + xl] = r[q].y;
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

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* As = smem;                   // [2][BK][LDS_LD]
  double* Bs = smem + 2 * OPER_ELEMS;  // [2][BK][LDS_LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware remap: blocks b, b+8, b+16.. share an XCD (and its L2); give each XCD a contiguous
  // run of tile indices so neighbouring tiles (same A strip, adjacent B strips) hit in that L2.
  const int nblk = gridDim.x;
  int idx;
  {
    const int b = blockIdx.x, x = b & 7, q = nblk >> 3, r = nblk & 7;
    idx = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
  }
  int ti, tj;
  tile_from_index(p, idx, ti, tj);
  if (p.kmode == 2) {  // longest-K tiles (largest ti) first
    ti = p.mt - 1 - ti;
  }
  const int i0 = ti * TILE, j0 = tj * TILE;
  int kbeg = 0, kend = p.k;
  if (p.kmode == 1) kbeg = j0;
  else if (p.kmode == 2) kend = i0 + TILE;
  else if (p.kmode == 3) kbeg = i0;

  const double* A = p.A + (long)blockIdx.z * p.strideA;
  const double* B = p.B + (long)blockIdx.z * p.strideB;
  double* C = p.C + (long)blockIdx.z * p.strideC;

  double acc[4][4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[a][b][s] = 0.0;

  const int nchunk = (kend - kbeg) / BK;
  double2_t ra[4], rb[4];
  if (nchunk > 0) {
    chunk_load<A_KMAJOR>(A, p.lda, i0, kbeg, tid, ra);
    chunk_load<B_KMAJOR>(B, p.ldb, j0, kbeg, tid, rb);
    chunk_store<A_KMAJOR>(As, tid, ra);
    chunk_store<B_KMAJOR>(Bs, tid, rb);
  }
  __syncthreads();

  // per-lane fragment offsets inside a chunk
  const int kq = lane >> 4;                 // k within a k4 step
  const int a_off = wr * 64 + (lane & 15);  // + 16*a
  const int blk = (lane >> 2) & 3, jj = lane & 3;
  int b_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) b_off[s] = wc * 64 + 4 * ((blk + s) & 3) + jj;  // + 16*b

  for (int c = 0; c < nchunk; ++c) {
    const int buf = c & 1;
    const bool more = (c + 1 < nchunk);
    if (more) {
      chunk_load<A_KMAJOR>(A, p.lda, i0, kbeg + (c + 1) * BK, tid, ra);
      chunk_load<B_KMAJOR>(B, p.ldb, j0, kbeg + (c + 1) * BK, tid, rb);
    }
    const double* Ac = As + buf * OPER_ELEMS;
    const double* Bc = Bs + buf * OPER_ELEMS;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const int krow = (kk * 4 + kq) * LDS_LD;
      double af[4], bf[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a) af[a] = Ac[krow + a_off + 16 * a];
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s) bf[b][s] = Bc[krow + b_off[s] + 16 * b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            acc[a][b][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[a], bf[b][s], acc[a][b][s], 0, 0, 0);
    }
    if (more) {
      chunk_store<A_KMAJOR>(As + (buf ^ 1) * OPER_ELEMS, tid, ra);
      chunk_store<B_KMAJOR>(Bs + (buf ^ 1) * OPER_ELEMS, tid, rb);
    }
    __syncthreads();
  }

  // epilogue: C = beta*C + alpha*acc.  lane (i = l>>4, blk, j) of acc[a][b][s] is
  // row 16a + 4blk + i, column 16b + 4((blk+s)&3) + j of the wave's 64x64 sub-tile.
  const int ii = lane >> 4;
  const double alpha = p.alpha, beta = p.beta;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const long row = i0 + wr * 64 + 16 * a + 4 * blk + ii;
    double* crow = C + row * p.ldc + j0 + wc * 64;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        double* cp = crow + 16 * b + 4 * ((blk + s) & 3) + jj;
        double v = alpha * acc[a][b][s];
        if (beta != 0.0) v += beta * (*cp);
        *cp = v;
      }
    }
  }
}

static int tile_count(const GemmParams& p) {
  if (!p.tri) return p.mt * p.nt;
  return p.nt * (p.nt + 1) / 2 + (p.mt - p.nt) * p.nt;
}

hipError_t launch_gemm_f64(const GemmParams& p, int opA_kmajor, int opB_kmajor, int batch, hipStream_t stream) {
  const int nblk = tile_count(p);
  if (nblk <= 0 || batch <= 0) return hipSuccess;
  dim3 grid(nblk, 1, batch), block(256);
  const size_t lds = sizeof(double) * 4 * OPER_ELEMS;
  if (!opA_kmajor && !opB_kmajor) gemm_f64_kernel<false, false><<<grid, block, lds, stream>>>(p);
  else if (!opA_kmajor && opB_kmajor) gemm_f64_kernel<false, true><<<grid, block, lds, stream>>>(p);
  else if (opA_kmajor && opB_kmajor) gemm_f64_kernel<true, true><<<grid, block, lds, stream>>>(p);
  else gemm_f64_kernel<true, false><<<grid, block, lds, stream>>>(p);
  return hipGetLastError();
}

hipError_t gemm_f64_enable_lds() {
  // 72 KiB of dynamic LDS per workgroup needs the opt-in attribute.
  const int lds = (int)(sizeof(double) * 4 * OPER_ELEMS);
  hipError_t e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f64_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace migp
