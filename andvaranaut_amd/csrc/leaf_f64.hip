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
constexpr int LEAF_LD = 130;  // LDS leading dimension (rows stay 16-byte aligned)
constexpr int SB = 16;        // sub-block width

__device__ __forceinline__ void wave_lds_fence() {
  // order this wave's LDS writes before its later LDS reads (DS ops execute in order per wave;
  // this only stops the compiler from reordering / caching across the point)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// info: 0 = ok, else 1-based global index of the first non-positive (or NaN) pivot (atomicMin'd).
__global__ __launch_bounds__(256, 1) void potrf_leaf128_kernel(double* __restrict__ Ablk, long lda,
                                                                double* __restrict__ dinv, int col0,
                                                                int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* S = smem;                          // [128][LEAF_LD]
  double* Ld = smem + LEAF * LEAF_LD;        // [16][17] factored diagonal sub-block
  double* invd = Ld + SB * 17;               // [128] 1 / L[c][c]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // load the lower triangle (whole rows up to the diagonal's 16-block, coalesced along rows)
  for (int e = tid; e < LEAF * (LEAF / 2); e += 256) {
    const int r = e >> 6, c2 = (e & 63) * 2;
    if (c2 <= r) {
      const double* src = Ablk + (long)r * lda + c2;
      S[r * LEAF_LD + c2] = src[0];
      S[r * LEAF_LD + c2 + 1] = src[1];
    }
  }
  __syncthreads();

  for (int jb = 0; jb < LEAF / SB; ++jb) {
    const int j0 = jb * SB;
    // ---- (A) factor the 16x16 diagonal sub-block: wave 0, right-looking, column at a time
    if (wave == 0) {
      const int r = lane & 15, g = lane >> 4;
      for (int j = 0; j < SB; ++j) {
        const double p = S[(j0 + j) * LEAF_LD + j0 + j];
        if (!(p > 0.0)) {
          if (lane == 0) atomicMin(info, col0 + j0 + j + 1);
        }
        const double inv = 1.0 / sqrt(p);
        const double arj = S[(j0 + r) * LEAF_LD + j0 + j];
        const double inv2 = inv * inv;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int c = g + 4 * t;
          if (c > j && c <= r) {
            const double acj = S[(j0 + c) * LEAF_LD + j0 + j];
            S[(j0 + r) * LEAF_LD + j0 + c] -= arj * acj * inv2;
          }
        }
        if (g == 0) {
          if (r > j) Ld[r * 17 + j] = arj * inv;
          else if (r == j) { Ld[r * 17 + j] = p * inv; invd[j0 + j] = inv; }
        }
        wave_lds_fence();
      }
    }
    __syncthreads();
    // ---- (B) rows below: X = B * L16^-T by forward substitution, one thread per row;
    //          rows inside the diagonal sub-block just copy the factor back into S
    {
      const int nrow = LEAF - j0;  // rows j0 .. 127
      if (tid < nrow) {
        const int r = j0 + tid;
        double* row = S + r * LEAF_LD + j0;
        if (tid < SB) {
          for (int c = 0; c <= tid; ++c) row[c] = Ld[tid * 17 + c];
        } else {
          double x[SB];
#pragma unroll
          for (int c = 0; c < SB; ++c) x[c] = row[c];
#pragma unroll
          for (int c = 0; c < SB; ++c) {
            double s = x[c];
#pragma unroll
            for (int k = 0; k < c; ++k) s -= x[k] * Ld[c * 17 + k];
            x[c] = s * invd[j0 + c];
          }
#pragma unroll
          for (int c = 0; c < SB; ++c) row[c] = x[c];
        }
      }
    }
    __syncthreads();
    // ---- (C) trailing update S[r][c] -= sum_k X[r][k] X[c][k], 4x4 micro-tiles, lower part only
    {
      const int t0 = j0 + SB;
      const int q = (LEAF - t0) / 4;  // micro-tiles per dimension
      const int nt = q * (q + 1) / 2;
      for (int e = tid; e < nt; e += 256) {
        int tr = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
        while ((tr + 1) * (tr + 2) / 2 <= e) ++tr;
        while (tr * (tr + 1) / 2 > e) --tr;
        const int tc = e - tr * (tr + 1) / 2;
        const int r0 = t0 + 4 * tr, c0 = t0 + 4 * tc;
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll
        for (int kc = 0; kc < SB; kc += 4) {
          double xr[4][4], xc[4][4];
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              xr[a][k] = S[(r0 + a) * LEAF_LD + j0 + kc + k];
              xc[a][k] = S[(c0 + a) * LEAF_LD + j0 + kc + k];
            }
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
              for (int b = 0; b < 4; ++b) acc[a][b] += xr[a][k] * xc[b][k];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) S[(r0 + a) * LEAF_LD + c0 + b] -= acc[a][b];
      }
    }
    __syncthreads();
  }

  // ---- write L back (lower triangle incl. diagonal)
  for (int e = tid; e < LEAF * (LEAF / 2); e += 256) {
    const int r = e >> 6, c2 = (e & 63) * 2;
    double* dst = Ablk + (long)r * lda + c2;
    if (c2 + 1 <= r) {
      dst[0] = S[r * LEAF_LD + c2];
      dst[1] = S[r * LEAF_LD + c2 + 1];
    } else if (c2 == r) {
      dst[0] = S[r * LEAF_LD + c2];
    }
  }
  // ---- inverses of the eight 16x16 diagonal sub-blocks: thread (b, c) solves column c of block b
  if (tid < LEAF) {
    const int b = tid >> 4, c = tid & 15, j0 = b * SB;
    double z[SB];
#pragma unroll
    for (int r = 0; r < SB; ++r) z[r] = 0.0;
#pragma unroll
    for (int r = 0; r < SB; ++r) {
      if (r == c) z[r] = invd[j0 + r];
      else if (r > c) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k)
          if (k >= c) s -= S[(j0 + r) * LEAF_LD + j0 + k] * z[k];
        z[r] = s * invd[j0 + r];
      }
    }
    double* out = dinv + (long)b * SB * SB;
#pragma unroll
    for (int r = 0; r < SB; ++r) out[r * SB + c] = z[r];
  }
}

// X * L^T = B, in place on B (m x 128, leading dimension ldb, m multiple of 64).
// One wave per 16 rows; tiles kept transposed: T_j[r] = B[row0 + (l&15)][16j + 4r + (l>>4)].
__global__ __launch_bounds__(256, 1) void trsm_strip128_kernel(const double* __restrict__ Lblk, long lda,
                                                                const double* __restrict__ dinv,
                                                                double* __restrict__ B, long ldb) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ls = smem;  // [128][LEAF_LD] lower triangle of L
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int e = tid; e < LEAF * (LEAF / 2); e += 256) {
    const int r = e >> 6, c2 = (e & 63) * 2;
    if (c2 <= r) {
      const double* src = Lblk + (long)r * lda + c2;
      Ls[r * LEAF_LD + c2] = src[0];
      Ls[r * LEAF_LD + c2 + 1] = src[1];
    }
  }
  const int n = lane & 15, q = lane >> 4;
  double* Brow = B + ((long)blockIdx.x * 64 + wave * 16 + n) * ldb;
  double4_t T[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[j][r] = Brow[16 * j + 4 * r + q];
  __syncthreads();

#pragma unroll
  for (int j = 0; j < 8; ++j) {
    // X_j = Dinv_j * T_j
    double4_t X = {0.0, 0.0, 0.0, 0.0};
    const double* dj = dinv + j * SB * SB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double a = dj[n * SB + 4 * s + q];  // Dinv_j[i = l&15][k = 4s + (l>>4)]
      X = __builtin_amdgcn_mfma_f64_16x16x4f64(a, T[j][s], X, 0, 0, 0);
    }
    T[j] = X;
    const double4_t Xn = -X;
#pragma unroll
    for (int i = j + 1; i < 8; ++i) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double a = Ls[(16 * i + n) * LEAF_LD + 16 * j + 4 * s + q];  // L[16i + (l&15)][16j + 4s + (l>>4)]
        T[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Xn[s], T[i], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) Brow[16 * j + 4 * r + q] = T[j][r];
}

constexpr size_t LEAF_LDS_BYTES = sizeof(double) * (LEAF * LEAF_LD + SB * 17 + LEAF);
constexpr size_t STRIP_LDS_BYTES = sizeof(double) * (LEAF * LEAF_LD);

hipError_t leaf_enable_lds() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_leaf128_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEAF_LDS_BYTES);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_strip128_kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)STRIP_LDS_BYTES);
}

hipError_t launch_potrf_leaf128(double* Ablk, long lda, double* dinv, int col0, int* info, hipStream_t stream) {
  potrf_leaf128_kernel<<<1, 256, LEAF_LDS_BYTES, stream>>>(Ablk, lda, dinv, col0, info);
  return hipGetLastError();
}

hipError_t launch_trsm_strip128(const double* Lblk, long lda, const double* dinv, double* B, long ldb, int m,
                                hipStream_t stream) {
  if (m <= 0) return hipSuccess;
  trsm_strip128_kernel<<<m / 64, 256, STRIP_LDS_BYTES, stream>>>(Lblk, lda, dinv, B, ldb);
  return hipGetLastError();
}

}  // namespace migp
