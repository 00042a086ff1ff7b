// Covariance assembly and the small reductions around the factorisation
// (K6).  Restates pymc.gp.cov.Stationary.square_dist / euclidean_dist and the ExpQuad / Matern52 /
// Matern32 / Exponential / RatQuad .full formulas that gpmcmc.py:282-307 composes:
//   Xs = X * (1/ls);  r2 = clip(-2 Xs Xs^T + (|Xs_i|^2 + |Xs_j|^2), 0, inf);  r = sqrt(r2 + 1e-12)
// One 64x64 output tile per 256-thread workgroup; the two 64-row x-blocks are staged (already scaled
// by 1/ls) in LDS, each thread owns a 4x4 strided micro-tile so that stores are 128-byte coalesced.
#include "migp_kernels.h"

namespace migp {

constexpr int AT = 64;      // assembly tile
constexpr int DCH = 32;     // input dimensions per LDS chunk
constexpr int DLD = DCH + 1;

__device__ __forceinline__ double base_kernel_eval(int kid, double r2, double alpha) {
  switch (kid) {
    case KID_RBF:
      return exp(-0.5 * r2);
    case KID_RATQUAD:
      return pow(1.0 + 0.5 * r2 * (1.0 / alpha), -1.0 * alpha);
    default: {
      const double r = sqrt(r2 + 1e-12);
      if (kid == KID_MATERN52) return (1.0 + 2.23606797749979 * r + 5.0 / 3.0 * (r * r)) * exp(-1.0 * 2.23606797749979 * r);
      if (kid == KID_MATERN32) return (1.0 + 1.7320508075688772 * r) * exp(-1.7320508075688772 * r);
      return exp(-0.5 * r);  // KID_EXPONENTIAL (PyMC's Exponential is exp(-r/2))
    }
  }
}

// theta layout: [ls(nkern*d), kv(nkern), alpha(nkern), gv, jitter]
__global__ __launch_bounds__(256) void assemble_kernel(KernSpec spec, const double* __restrict__ theta,
                                                       const double* __restrict__ X1, int n1,
                                                       const double* __restrict__ X2, int n2,
                                                       double* __restrict__ K, long ldk, int rows_pad,
                                                       int cols_pad, int sym, int noise_form,
                                                       int diag_shift, const double* __restrict__ extra_diag) {
  __shared__ double Xi[AT * DLD];
  __shared__ double Xj[AT * DLD];
  __shared__ double n2i[AT], n2j[AT];
  const int tid = threadIdx.x;
  int ti, tj;
  if (sym) {
    const int e = blockIdx.x;
    int t = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((t + 1) * (t + 2) / 2 <= e) ++t;
    while (t * (t + 1) / 2 > e) --t;
    ti = t;
    tj = e - t * (t + 1) / 2;
  } else {
    const int ntc = cols_pad / AT;
    ti = blockIdx.x / ntc;
    tj = blockIdx.x % ntc;
  }
  const int i0 = ti * AT, j0 = tj * AT;
  const int tx = tid & 15, ty = tid >> 4;
  const int d = spec.d, nk = spec.nkern;
  const double* ls = theta;
  const double* kv = theta + nk * d;
  const double* al = kv + nk;
  const double gv = theta[nk * d + 2 * nk];
  const double jitter = theta[nk * d + 2 * nk + 1];

  double Kacc[4][4];
  for (int c = 0; c < nk; ++c) {
    double dot[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) dot[a][b] = 0.0;
    double mynorm = 0.0;  // threads 0..63: |Xs_i|^2 of row tid, threads 64..127: |Xs_j|^2
    for (int m0 = 0; m0 < d; m0 += DCH) {
      const int dc = min(DCH, d - m0);
      __syncthreads();
      // 256 % DCH == 0: a thread always stages the same input dimension m, so its 1/l is one division per chunk
      const int m = tid % DCH;
      const double il = (m < dc) ? 1.0 / ls[c * d + m0 + m] : 0.0;
      for (int r = tid / DCH; r < AT; r += 256 / DCH) {
        double vi = 0.0, vj = 0.0;
        if (m < dc) {
          if (i0 + r < n1) vi = X1[(long)(i0 + r) * d + m0 + m] * il;
          if (j0 + r < n2) vj = X2[(long)(j0 + r) * d + m0 + m] * il;
        }
        Xi[r * DLD + m] = vi;
        Xj[r * DLD + m] = vj;
      }
      __syncthreads();
      if (tid < 2 * AT) {
        const double* row = (tid < AT) ? (Xi + tid * DLD) : (Xj + (tid - AT) * DLD);
        for (int m = 0; m < dc; ++m) mynorm += row[m] * row[m];
      }
      for (int m = 0; m < dc; ++m) {
        double xi[4], xj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xi[a] = Xi[(ty + 16 * a) * DLD + m];
#pragma unroll
        for (int b = 0; b < 4; ++b) xj[b] = Xj[(tx + 16 * b) * DLD + m];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) dot[a][b] += xi[a] * xj[b];
      }
    }
    if (tid < AT) n2i[tid] = mynorm;
    else if (tid < 2 * AT) n2j[tid - AT] = mynorm;
    __syncthreads();
    const int kid = spec.kid[c];
    const double kvc = kv[c], alc = al[c];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double r2 = -2.0 * dot[a][b] + (n2i[ty + 16 * a] + n2j[tx + 16 * b]);
        r2 = fmax(r2, 0.0);
        const double kval = kvc * base_kernel_eval(kid, r2, alc);
        if (c == 0) Kacc[a][b] = kval;
        else if (spec.op[c - 1] == 0) Kacc[a][b] = Kacc[a][b] + kval;
        else Kacc[a][b] = Kacc[a][b] * kval;
      }
  }
  const double sg = sqrt(gv);
  // diagonal of the global matrix: local (gi, gj) with gi + diag_shift == gj; sym mode has shift 0,
  // rectangular blocks of a distributed matrix pass row0 - col0, cross-covariances pass INT_MIN (none)
  const bool diag_on = sym || diag_shift != -2147483647 - 1;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int gi = i0 + ty + 16 * a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int gj = j0 + tx + 16 * b;
      double v = Kacc[a][b];
      if (gi >= n1 || gj >= n2) {
        v = (diag_on && gi + diag_shift == gj) ? 1.0 : 0.0;
      } else if (diag_on && gi + diag_shift == gj) {
        if (noise_form == 0) { v += sg * sg; v += jitter; }       // Marginal._build_marginal_likelihood
        else if (noise_form == 1) { v += jitter; v += sg * sg; }  // Marginal._build_conditional
        else { v += jitter + gv; }                                // gpmcmc.py:312 explicit form
        if (extra_diag) v += extra_diag[gi];                      // per-point noise vector (inverse_opt, gpmcmc.py:1134-1158)
      }
      K[(long)gi * ldk + gj] = v;
    }
  }
}

// rows [row0, row0+128) x cols [0, cols_pad): zero, except row row0 = y^T (first n entries)
__global__ void set_yrows_kernel(double* __restrict__ K, long ldk, int row0, int cols_pad,
                                 const double* __restrict__ y, int n) {
  const long total = 128L * cols_pad;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols_pad), c = (int)(e % cols_pad);
    double v = 0.0;
    if (r == 0 && c < n) v = y[c];
    K[(long)(row0 + r) * ldk + c] = v;
  }
}

// out[0] = LML, out[1] = sum log L_ii, out[2] = |beta|^2   (gpmcmc.py:316-318 / MvNormal.logp)
// One 256-thread workgroup: 16 independent diagonal gathers in flight per thread (the strided diagonal walk is pure
// latency), partial sums combined in a fixed order (bit-reproducible).
__global__ __launch_bounds__(256) void lml_reduce_kernel(const double* __restrict__ L, long ld,
                                                         const double* __restrict__ beta, int n,
                                                         double* __restrict__ out) {
  __shared__ double s1[256], s2[256];
  double a = 0.0, b = 0.0;
  for (int i0 = threadIdx.x; i0 < n; i0 += 256 * 16) {
    double dv[16], bv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + 256 * u;
      dv[u] = (i < n) ? L[(long)i * ld + i] : 1.0;
      bv[u] = (i < n) ? beta[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      a += log(dv[u]);
      b += bv[u] * bv[u];
    }
  }
  s1[threadIdx.x] = a;
  s2[threadIdx.x] = b;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      s1[threadIdx.x] += s1[threadIdx.x + w];
      s2[threadIdx.x] += s2[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[1] = s1[0];
    out[2] = s2[0];
    out[0] = -0.5 * (double)n * 1.8378770664093453 - 0.5 * s2[0] - s1[0];
  }
}

hipError_t launch_assemble(const KernSpec& spec, const double* theta, const double* X1, int n1, const double* X2,
                           int n2, double* K, long ldk, int rows_pad, int cols_pad, int sym, int noise_form,
                           hipStream_t stream, int diag_shift, const double* extra_diag) {
  int nblk;
  if (sym) {
    const int nt = rows_pad / AT;
    nblk = nt * (nt + 1) / 2;
  } else {
    nblk = (rows_pad / AT) * (cols_pad / AT);
  }
  assemble_kernel<<<nblk, 256, 0, stream>>>(spec, theta, X1, n1, X2, n2, K, ldk, rows_pad, cols_pad, sym, noise_form,
                                            sym ? 0 : diag_shift, sym ? extra_diag : nullptr);
  return hipGetLastError();
}

hipError_t launch_set_yrows(double* K, long ldk, int row0, int cols_pad, const double* y, int n, hipStream_t stream) {
  const long total = 128L * cols_pad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  set_yrows_kernel<<<blocks, 256, 0, stream>>>(K, ldk, row0, cols_pad, y, n);
  return hipGetLastError();
}

hipError_t launch_lml_reduce(const double* L, long ld, const double* beta, int n, double* out, hipStream_t stream) {
  lml_reduce_kernel<<<1, 256, 0, stream>>>(L, ld, beta, n, out);
  return hipGetLastError();
}

}  // namespace migp
