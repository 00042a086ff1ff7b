// Kernels around the factorisation for the hyper-parameter gradient and
// the posterior conditional (K8):
//   set_identity_blocks : identity into the 128x128 diagonal blocks of U (input of the leaf inverses)
//   trmv_upper          : alpha = U beta  with U = L^-T upper triangular  (== L^-T L^-1 y, gpmcmc.py:315)
//   grad_contract       : dLML/dtheta_k = 1/2 sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dtheta_k, all
//                         parameters in one pass over the lower triangle ("assembly shaped": the
//                         covariance and its derivatives are recomputed from X, only Kinv is read)
//   predict_reduce      : mu_i = A_i . beta,  var_i = kdiag - |A_i|^2 (+ gv)   (gpmcmc.py:766-778)
// The reference obtains the gradient by reverse-mode autodiff through the PyTensor graph inside
// pm.find_MAP / pm.sample (gpmcmc.py:345,351); the analytic form is restated in oracle/gp_oracle.py.
#include "migp_kernels.h"
#include "migp_math.h"

namespace migp {

constexpr int GT = 64;  // contraction tile
constexpr int GDCH = 32;
constexpr int GDLD = GDCH + 1;

__global__ void set_identity_blocks_kernel(double* __restrict__ U, long ld, long sZ) {
  double* blk = U + (long)blockIdx.z * sZ + (long)blockIdx.x * 128 * ld + (long)blockIdx.x * 128;
  for (int e = threadIdx.x; e < 128 * 128; e += blockDim.x) {
    const int r = e >> 7, c = e & 127;
    blk[(long)r * ld + c] = (r == c) ? 1.0 : 0.0;
  }
}

// alpha[i] = sum_{k >= i} U[i][k] * beta[k], one wave per row, rows [0, n); U is np x ld, columns
// beyond n hold the padding (identity diagonal / zeros) and beta is zero there by construction.
__global__ __launch_bounds__(256) void trmv_upper_kernel(const double* __restrict__ U, long ld,
                                                         const double* __restrict__ beta, int n,
                                                         double* __restrict__ alpha, long sZ, long sK, long salpha) {
  U += (long)blockIdx.z * sZ;  // batched evaluation: problem blockIdx.z
  beta += (long)blockIdx.z * sK;
  alpha += (long)blockIdx.z * salpha;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const double* u = U + (long)row * ld;
  double s = 0.0;
  const int k0 = row & ~63;
  for (int k = k0 + lane; k < n; k += 64) {
    if (k >= row) s += u[k] * beta[k];
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) alpha[row] = s;
}

// out[i] = sum_{k <= i} U[k][i] * x[k]  (U^T x = L^-1 x for U = L^-T), columns [0, n).  One workgroup per 64 columns;
// wave g walks rows k = g (mod 4), 8 independent loads in flight per lane: consecutive lanes read consecutive doubles of
// one row of U (512 B per wave load), x[k] is a broadcast; the four partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void trmv_upper_t_kernel(const double* __restrict__ U, long ld,
                                                           const double* __restrict__ x, int n,
                                                           double* __restrict__ out) {
  __shared__ double part[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + c;
  const int kmax = min(n - 1, blockIdx.x * 64 + 63);
  const bool live = i < n;
  double s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) s[u] = 0.0;
  int k = g;
  for (; k + 28 <= kmax; k += 32) {
    double uv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = k + 4 * u;
      uv[u] = (live && kk <= i) ? U[(long)kk * ld + i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s[u] += uv[u] * x[k + 4 * u];
  }
  for (; k <= kmax; k += 4)
    if (live && k <= i) s[0] += U[(long)k * ld + i] * x[k];
  part[g][c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (g == 0 && live) out[i] = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// d k / d r2 of the base kernels (matches oracle base_kernel_dr2) and the value itself
__device__ __forceinline__ void base_kernel_val_der(int kid, double r2, double alpha, double& k, double& dk,
                                                    double& dalpha) {
  dalpha = 0.0;
  if (kid == KID_RATQUAD) {
    const double u = 0.5 * r2 / alpha;
    k = pow(1.0 + u, -alpha);
    dk = -0.5 * k / (1.0 + u);
    dalpha = k * (-log1p(u) + u / (1.0 + u));
    return;
  }
  // the four exponential families share ONE exp evaluation: e = exp(-rate * s), s = r2 (RBF) or r = sqrt(r2 + 1e-12)
  const bool rbf = kid == KID_RBF;
  const double r = rbf ? 0.0 : sqrt_pos(r2 + 1e-12);
  const double rate = rbf ? 0.5 : kid == KID_MATERN52 ? 2.23606797749979 : kid == KID_MATERN32 ? 1.7320508075688772 : 0.5;
  const double e = exp_nonpos(-1.0 * rate * (rbf ? r2 : r));
  if (rbf) {
    k = e;
    dk = -0.5 * e;
  } else if (kid == KID_MATERN52) {
    k = (1.0 + 2.23606797749979 * r + 5.0 / 3.0 * (r * r)) * e;
    dk = -(5.0 / 6.0) * (1.0 + 2.23606797749979 * r) * e;
  } else if (kid == KID_MATERN32) {
    k = (1.0 + 1.7320508075688772 * r) * e;
    dk = -1.5 * e;
  } else {
    k = e;
    dk = -0.25 * e / r;
  }
}

// One 64x64 tile of the lower triangle per workgroup; thread (ty, tx) owns the 4x4 strided
// micro-tile rows ty+16a, cols tx+16b.  part[blockIdx.x][p] receives the block's partial sums.
// Parameter order p: ls(nk*d), kv(nk), alpha(nk), gv, jitter.
// Occupancy (round 6): the kernel is bound by LDS / exp latency, not by issue -- at 176 VGPRs (two waves per SIMD) the N = 16384 pass
// took 2.0 ms against ~0.6 ms of fp64 issue.  One and two components are built for three waves per SIMD (<= 168 VGPRs: the
// weights' products with dk/dr2 are formed once, G = wc * dkv, and the five per-element arrays of the fold die before the
// length-scale pass; RQ = false: no component is a rational quadratic, so d k / d alpha -- identically zero then -- is not carried);
// N = 16384 LML + gradient 70.6 -> 69.9 ms on one box.  Same arithmetic, same bits.
template <int NK, bool RQ>
__global__ __launch_bounds__(256, NK == 1 ? 3 : (NK == 2 && !RQ) ? 2 : 1) void grad_contract_kernel(KernSpec spec, const double* __restrict__ theta,
                                                            const double* __restrict__ X, int n,
                                                            const double* __restrict__ W, long ldw,
                                                            const double* __restrict__ alpha_v,
                                                            double* __restrict__ part, int rect_tw, int tj0,
                                                            long wrow0, long wcol0, int stheta, long sW, long salpha,
                                                            long spart) {
  theta += (long)blockIdx.z * stheta;  // batched evaluation: problem blockIdx.z
  W += (long)blockIdx.z * sW;
  alpha_v += (long)blockIdx.z * salpha;
  part += (long)blockIdx.z * spart;
  __shared__ double Xi[GT * GDLD];
  __shared__ double Xj[GT * GDLD];
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const int d = spec.d;
  const int nk = NK > 4 ? spec.nkern : NK;  // NK = 8: one instantiation for 5..8 components, run-time count
  const int P = nk * d + 2 * nk + 2;
  double* out = part + (long)blockIdx.x * P;
  int ti, tj;
  if (rect_tw > 0) {
    // column slab of a distributed K^-1: tile columns [tj0, tj0 + rect_tw), tile rows from tj0 down; W holds the slab
    // with its element (wrow0, wcol0) first.  Tiles above the diagonal contribute nothing.
    ti = tj0 + blockIdx.x / rect_tw;
    tj = tj0 + blockIdx.x % rect_tw;
    if (ti < tj) {
      for (int p = tid; p < P; p += 256) out[p] = 0.0;
      return;
    }
    W -= wrow0 * ldw + wcol0;
  } else {
    const int e = blockIdx.x;
    int t = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
    while ((t + 1) * (t + 2) / 2 <= e) ++t;
    while (t * (t + 1) / 2 > e) --t;
    ti = t;
    tj = e - t * (t + 1) / 2;
  }
  const int i0 = ti * GT, j0 = tj * GT;
  const int tx = tid & 15, ty = tid >> 4;
  const double* ls = theta;
  const double* kv = theta + nk * d;
  const double* al = kv + nk;

  // weights: w_ab = (alpha_i alpha_j - W_ij) * (1 below the diagonal, 1/2 on it, 0 above / padding)
  double wgt[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int gi = i0 + ty + 16 * a, gj = j0 + tx + 16 * b;
      double w = 0.0;
      if (gi < n && gj < n && gj <= gi) {
        w = alpha_v[gi] * alpha_v[gj] - W[(long)gi * ldw + gj];
        if (gi == gj) w *= 0.5;
      }
      wgt[a][b] = w;
    }

  // pass 1: per-component scaled squared distances r2[c] (direct form) -> value, derivative, fold
  double kval[NK][4][4], dkv[NK][4][4], dal[RQ ? NK : 1][4][4];
#pragma unroll
  for (int c = 0; c < nk; ++c) {
    double r2[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) r2[a][b] = 0.0;
    for (int m0 = 0; m0 < d; m0 += GDCH) {
      const int dc = min(GDCH, d - m0);
      __syncthreads();
      {  // 256 % GDCH == 0: a thread always stages the same input dimension, one division per chunk
        const int m = tid % GDCH;
        const double il = (m < dc) ? 1.0 / ls[c * d + m0 + m] : 0.0;
        for (int r = tid / GDCH; r < GT; r += 256 / GDCH) {
          double vi = 0.0, vj = 0.0;
          if (m < dc) {
            if (i0 + r < n) vi = X[(long)(i0 + r) * d + m0 + m] * il;
            if (j0 + r < n) vj = X[(long)(j0 + r) * d + m0 + m] * il;
          }
          Xi[r * GDLD + m] = vi;
          Xj[r * GDLD + m] = vj;
        }
      }
      __syncthreads();
      for (int m = 0; m < dc; ++m) {
        double xi[4], xj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xi[a] = Xi[(ty + 16 * a) * GDLD + m];
#pragma unroll
        for (int b = 0; b < 4; ++b) xj[b] = Xj[(tx + 16 * b) * GDLD + m];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const double df = xi[a] - xj[b];
            r2[a][b] += df * df;
          }
      }
    }
    const int kid = spec.kid[c];
    const double kvc = kv[c], alc = al[c];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        double k, dk, da;
        base_kernel_val_der(kid, r2[a][b], alc, k, dk, da);
        kval[c][a][b] = kvc * k;
        dkv[c][a][b] = kvc * dk;
        if constexpr (RQ) dal[c][a][b] = kvc * da;
      }
  }
  // coefficient dK/dK_c of the left-to-right fold, times the weight
  double wc[NK][4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      double pref[NK];
      double T = kval[0][a][b];
      pref[0] = 1.0;
#pragma unroll
      for (int c = 1; c < nk; ++c) {
        pref[c] = (spec.op[c - 1] == 0) ? 1.0 : T;
        T = (spec.op[c - 1] == 0) ? T + kval[c][a][b] : T * kval[c][a][b];
      }
#pragma unroll
      for (int c = 0; c < nk; ++c) {
        double coef = pref[c];
#pragma unroll
        for (int c2 = c + 1; c2 < nk; ++c2)
          if (spec.op[c2 - 1] == 1) coef *= kval[c2][a][b];
        wc[c][a][b] = wgt[a][b] * coef;
      }
    }

  // block reduction helper
  auto block_sum = [&](double v) -> double {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };

  // kv, alpha, gv, jitter
#pragma unroll
  for (int c = 0; c < nk; ++c) {
    double skv = 0.0, sal = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        skv += wc[c][a][b] * kval[c][a][b];
        if constexpr (RQ) sal += wc[c][a][b] * dal[c][a][b];
      }
    skv = block_sum(skv);
    if constexpr (RQ) sal = block_sum(sal);
    if (tid == 0) {
      out[nk * d + c] = skv / kv[c];
      out[nk * d + nk + c] = sal;
    }
  }
  {
    double sd = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        if (i0 + ty + 16 * a == j0 + tx + 16 * b) sd += wgt[a][b];  // already carries the 1/2
    sd = block_sum(sd);
    if (tid == 0) {
      out[nk * d + 2 * nk] = sd;
      out[nk * d + 2 * nk + 1] = sd;
    }
  }
  // length scales: dK/dl_{c,m} = wc * kv dk/dr2 * (-2/l_m) * ((x_im - x_jm)/l_m)^2; G = wc * kv dk/dr2 once per element
#pragma unroll
  for (int c = 0; c < nk; ++c)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) dkv[c][a][b] *= wc[c][a][b];
  // (one component with d <= GDCH: the scaled coordinates of pass 1 are still in LDS -- nothing to stage again)
  const bool staged = NK == 1 && d <= GDCH;
#pragma unroll
  for (int c = 0; c < nk; ++c) {
    for (int m0 = 0; m0 < d; m0 += GDCH) {
      const int dc = min(GDCH, d - m0);
      __syncthreads();
      if (!staged) {  // 256 % GDCH == 0: a thread always stages the same input dimension, one division per chunk
        const int m = tid % GDCH;
        const double il = (m < dc) ? 1.0 / ls[c * d + m0 + m] : 0.0;
        for (int r = tid / GDCH; r < GT; r += 256 / GDCH) {
          double vi = 0.0, vj = 0.0;
          if (m < dc) {
            if (i0 + r < n) vi = X[(long)(i0 + r) * d + m0 + m] * il;
            if (j0 + r < n) vj = X[(long)(j0 + r) * d + m0 + m] * il;
          }
          Xi[r * GDLD + m] = vi;
          Xj[r * GDLD + m] = vj;
        }
        __syncthreads();
      }
      // per-dimension sums: wave partials of all dimensions of the chunk go to LDS, ONE barrier, then thread m adds
      // its four (same order as a per-dimension block sum, two barriers per dimension less)
      for (int m = 0; m < dc; ++m) {
        double xi[4], xj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xi[a] = Xi[(ty + 16 * a) * GDLD + m];
#pragma unroll
        for (int b = 0; b < 4; ++b) xj[b] = Xj[(tx + 16 * b) * GDLD + m];
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const double df = xi[a] - xj[b];
            s += dkv[c][a][b] * (df * df);
          }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((tid & 63) == 0) red[4 * m + (tid >> 6)] = s;
      }
      __syncthreads();
      if (tid < dc) {
        const double s = red[4 * tid] + red[4 * tid + 1] + red[4 * tid + 2] + red[4 * tid + 3];
        out[c * d + m0 + tid] = s * (-2.0 / ls[c * d + m0 + tid]);
      }
    }
  }
}

// dLML/dX (n x d, row-major) for the inputs-as-parameters consumers (input warping, gpmcmc.py:211-233; the
// free observation rows of inverse_opt, gpmcmc.py:1096-1101):
//   dLML/dx_im = sum_j (alpha_i alpha_j - Kinv_ij) dK_ij/dx_im
//              = sum_c sum_j Wsym_ij coef_c kv_c k_c'(r2_c) * 2 (x_im - x_jm) / l_cm^2
// (row i and column i of the symmetric sum 1/2 tr(W dK) contribute equally).  One workgroup per block of
// 64 rows walks all 64-column blocks: pass 1 builds the coefficient tile C_c (same 4x4 micro-tiles as
// grad_contract) and parks it in LDS, pass 2 is the small dense product C_c (64x64) . (x_i - X_j) with
// thread (row, m mod 4).  Kinv is stored as its lower triangle only; the upper part is read transposed.
constexpr int GXCH = 16;  // input dimensions per LDS chunk
constexpr int GXLD = GXCH + 1;
constexpr int GX_MAXD = 128;  // output dimensions per grad_x pass (window)
constexpr size_t PREDICT_GRAD_MAX_LDS = 61440;  // dynamic LDS of predict_grad_kernel: (nkern + 1) * d doubles

// Input dimensions beyond GX_MAXD: the host launches one pass per WINDOW of up to 128 output dimensions [w0, w0 + dw); every
// pass recomputes the coefficient tiles from all d dimensions (pass 1) and contracts them with its window (pass 2).
template <int NK, int NCH>  // NCH: chunks of 16 input dimensions covered by one window (dw <= 16 * NCH)
__global__ __launch_bounds__(256) void grad_x_kernel(KernSpec spec, const double* __restrict__ theta,
                                                     const double* __restrict__ X, int n,
                                                     const double* __restrict__ W, long ldw,
                                                     const double* __restrict__ alpha_v, double* __restrict__ gx, int w0) {
  // gridDim.y > 1: this workgroup walks column blocks blockIdx.y, blockIdx.y + gridDim.y, ... and writes its partial
  // result to slab blockIdx.y of gx ([gridDim.y][n][d]); gx_reduce_kernel adds the slabs in order
  __shared__ double Xi[GT * GXLD];
  __shared__ double Xj[GT * GXLD];
  __shared__ double Ct[GT * (GT + 1)];
  __shared__ double ils[NK * GX_MAXD];  // 1 / l_cm of the window's dimensions
  __shared__ double ilc[NK * GXCH];     // 1 / l_cm of the pass-1 chunk being staged
  const int tid = threadIdx.x;
  const int d = spec.d;
  const int nk = NK > 4 ? spec.nkern : NK;  // NK = 8: one instantiation for 5..8 components, run-time count
  const int dw = min(GX_MAXD, d - w0);  // this pass's output dimensions
  const int nt = (n + GT - 1) / GT;
  const int ib = blockIdx.x, i0 = ib * GT;
  const int tx = tid & 15, ty = tid >> 4;  // pass-1 micro-tile: rows ty+16a, cols tx+16b
  const int pr = tid & 63, pq = tid >> 6;  // pass-2: row pr, dimensions m = pq (mod 4)
  const double* kv = theta + nk * d;
  const double* al = kv + nk;
  for (int e = tid; e < nk * dw; e += 256) ils[(e / dw) * GX_MAXD + e % dw] = 1.0 / theta[(e / dw) * d + w0 + e % dw];
  double acc[NCH][GXCH / 4];
#pragma unroll
  for (int mc = 0; mc < NCH; ++mc)
#pragma unroll
    for (int u = 0; u < GXCH / 4; ++u) acc[mc][u] = 0.0;

  gx += (long)blockIdx.y * n * d;
  for (int jb = blockIdx.y; jb < nt; jb += gridDim.y) {
    const int j0 = jb * GT;
    __syncthreads();  // previous block's pass 2 is done with Ct / Xj
    // symmetric weight tile, coalesced along whichever index is contiguous in the stored lower triangle
    for (int e = tid; e < GT * GT; e += 256) {
      int r, c;
      if (jb <= ib) { r = e >> 6; c = e & 63; } else { c = e >> 6; r = e & 63; }
      const int gi = i0 + r, gj = j0 + c;
      double w = 0.0;
      if (gi < n && gj < n) {
        const double kin = (gj <= gi) ? W[(long)gi * ldw + gj] : W[(long)gj * ldw + gi];
        w = alpha_v[gi] * alpha_v[gj] - kin;
      }
      Ct[r * (GT + 1) + c] = w;
    }
    // pass 1: scaled squared distances per component
    double r2[NK][4][4];
#pragma unroll
    for (int c = 0; c < nk; ++c)
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r2[c][a][b] = 0.0;
    for (int m0 = 0; m0 < d; m0 += GXCH) {
      const int dc = min(GXCH, d - m0);
      if (m0 > 0) __syncthreads();
      for (int e = tid; e < GT * GXCH; e += 256) {
        const int r = e / GXCH, m = e % GXCH;
        double vi = 0.0, vj = 0.0;
        if (m < dc) {
          if (i0 + r < n) vi = X[(long)(i0 + r) * d + m0 + m];
          if (j0 + r < n) vj = X[(long)(j0 + r) * d + m0 + m];
        }
        Xi[r * GXLD + m] = vi;
        Xj[r * GXLD + m] = vj;
      }
      if (tid < nk * GXCH) {
        const int c = tid / GXCH, m = tid % GXCH;
        ilc[tid] = (m < dc) ? 1.0 / theta[c * d + m0 + m] : 0.0;
      }
      __syncthreads();
      for (int m = 0; m < dc; ++m) {
        double xi[4], xj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) xi[a] = Xi[(ty + 16 * a) * GXLD + m];
#pragma unroll
        for (int b = 0; b < 4; ++b) xj[b] = Xj[(tx + 16 * b) * GXLD + m];
#pragma unroll
        for (int c = 0; c < nk; ++c) {
          const double il = ilc[c * GXCH + m];
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              const double df = (xi[a] - xj[b]) * il;
              r2[c][a][b] += df * df;
            }
        }
      }
    }
    // coefficient tiles cf[c] = Wsym * (dK/dK_c of the fold) * kv_c dk_c/dr2
    double cf[NK][4][4];
    auto coef_elem = [&](int a, int b) {
      double kval[NK], dkv[NK];
#pragma unroll
      for (int c = 0; c < nk; ++c) {
        double k, dk, da;
        base_kernel_val_der(spec.kid[c], r2[c][a][b], al[c], k, dk, da);
        kval[c] = kv[c] * k;
        dkv[c] = kv[c] * dk;
      }
      double pref[NK];
      double T = kval[0];
      pref[0] = 1.0;
#pragma unroll
      for (int c = 1; c < nk; ++c) {
        pref[c] = (spec.op[c - 1] == 0) ? 1.0 : T;
        T = (spec.op[c - 1] == 0) ? T + kval[c] : T * kval[c];
      }
      const double w = Ct[(ty + 16 * a) * (GT + 1) + tx + 16 * b];
#pragma unroll
      for (int c = 0; c < nk; ++c) {
        double coef = pref[c];
#pragma unroll
        for (int c2 = c + 1; c2 < nk; ++c2)
          if (spec.op[c2 - 1] == 1) coef *= kval[c2];
        cf[c][a][b] = w * coef * dkv[c];
      }
    };
    if constexpr (NK >= 2) {
      // NOT unrolled for composite kernels.  Round 3: with the inlined exp / sqrt sequences of migp_math.h the fully unrolled
      // form (sixteen elements x NK families, 80-150 KB of code, 256 VGPRs) returned NONDETERMINISTIC dLML/dX for
      // four-component kernels -- entries of 1e13 that changed from call to call on identical inputs, while the same
      // source at -O1, with this loop rolled, or with exp behind a call is bit-reproducible and matches the oracle to
      // 2e-12; neither extra barriers nor SGPR spilling to memory changed it, so it is a code-generation problem of
      // this toolchain on the giant basic block, not a race in the source.  The rolled loop keeps the block small;
      // tests/test_gpu_data_grad.py repeats the evaluation and demands identical bits.
#pragma unroll 1
      for (int a = 0; a < 4; ++a)
#pragma unroll 1
        for (int b = 0; b < 4; ++b) coef_elem(a, b);
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) coef_elem(a, b);
    }
    // pass 2, one component at a time through the LDS tile
#pragma unroll
    for (int c = 0; c < nk; ++c) {
      __syncthreads();  // everyone has read the weights (c == 0) / finished the previous component
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) Ct[(ty + 16 * a) * (GT + 1) + tx + 16 * b] = cf[c][a][b];
#pragma unroll
      for (int mc = 0; mc < NCH; ++mc) {
        const int m0 = mc * GXCH;
        if (m0 >= dw) break;
        const int dc = min(GXCH, dw - m0);
        if (NCH > 1) {  // several chunks: bring this one back (a single chunk -- d <= 16, w0 = 0 -- is still resident)
          __syncthreads();
          for (int e = tid; e < GT * GXCH; e += 256) {
            const int r = e / GXCH, m = e % GXCH;
            double vi = 0.0, vj = 0.0;
            if (m < dc) {
              if (i0 + r < n) vi = X[(long)(i0 + r) * d + w0 + m0 + m];
              if (j0 + r < n) vj = X[(long)(j0 + r) * d + w0 + m0 + m];
            }
            Xi[r * GXLD + m] = vi;
            Xj[r * GXLD + m] = vj;
          }
        }
        __syncthreads();
        double t[GXCH / 4], xr[GXCH / 4];
#pragma unroll
        for (int u = 0; u < GXCH / 4; ++u) {
          t[u] = 0.0;
          xr[u] = Xi[pr * GXLD + pq + 4 * u];
        }
        for (int j = 0; j < GT; ++j) {
          const double cj = Ct[pr * (GT + 1) + j];
#pragma unroll
          for (int u = 0; u < GXCH / 4; ++u) t[u] += cj * (xr[u] - Xj[j * GXLD + pq + 4 * u]);
        }
#pragma unroll
        for (int u = 0; u < GXCH / 4; ++u) {
          const int m = m0 + pq + 4 * u;
          if (pq + 4 * u < dc) {
            const double il = ils[c * GX_MAXD + m];
            acc[mc][u] += 2.0 * il * il * t[u];
          }
        }
      }
    }
  }
  if (i0 + pr < n) {
#pragma unroll
    for (int mc = 0; mc < NCH; ++mc)
#pragma unroll
      for (int u = 0; u < GXCH / 4; ++u) {
        const int m = mc * GXCH + pq + 4 * u;
        if (m < dw) gx[(long)(i0 + pr) * d + w0 + m] = acc[mc][u];
      }
  }
}

// grad[p] = sum_b part[b][p] in a fixed order (the weights already are 1 below the diagonal and 1/2
// on it, which is 1/2 sum over the full symmetric matrix).
__global__ void grad_final_kernel(const double* __restrict__ part, int nblk, int P, double* __restrict__ grad, long spart,
                                  int sgrad, unsigned* __restrict__ done, double* __restrict__ flag, int sflag, double seq) {
  part += (long)blockIdx.z * spart;  // batched evaluation: problem blockIdx.z
  grad += (long)blockIdx.z * sgrad;
  const int p = blockIdx.x;
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += part[(long)b * P + p];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    grad[p] = red[0];
    if (done && seq != 0.0) {
      // LAST kernel of an LML + gradient evaluation: the workgroup that completes the gradient publishes the evaluation's
      // sequence number, released at system scope (the host may spin on it instead of synchronising the stream)
      __threadfence_system();
      done += blockIdx.z;
      if (atomicAdd(done, 1u) == (unsigned)P - 1u) {
        *done = 0u;  // ready for the next evaluation
        __hip_atomic_store(flag + (long)blockIdx.z * sflag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// mean[i] = A_i . beta ; var[i] = kdiag - |A_i|^2 (+ noise), one wave per prediction point
__global__ __launch_bounds__(256) void predict_reduce_kernel(const double* __restrict__ A, long lda,
                                                             const double* __restrict__ beta, int n, int m,
                                                             double kdiag, double noise, double* __restrict__ mean,
                                                             double* __restrict__ var) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= m) return;
  const double* a = A + (long)row * lda;
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < n; k += 64) {
    const double v = a[k];
    s1 += v * beta[k];
    s2 += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s1 += __shfl_down(s1, off, 64);
    s2 += __shfl_down(s2, off, 64);
  }
  if (lane == 0) {
    mean[row] = s1;
    var[row] = kdiag - s2 + noise;
  }
}

hipError_t launch_set_identity_blocks(double* U, long ld, int nblocks, hipStream_t stream, const Batch* bt) {
  set_identity_blocks_kernel<<<dim3(nblocks, 1, bt ? bt->nb : 1), 256, 0, stream>>>(U, ld, bt ? bt->sZ : 0);
  return hipGetLastError();
}

hipError_t launch_trmv_upper(const double* U, long ld, const double* beta, int n, double* alpha, hipStream_t stream,
                             const Batch* bt) {
  trmv_upper_kernel<<<dim3((n + 3) / 4, 1, bt ? bt->nb : 1), 256, 0, stream>>>(U, ld, beta, n, alpha, bt ? bt->sZ : 0, bt ? bt->sK : 0,
                                                                               bt ? bt->salpha : 0);
  return hipGetLastError();
}

int grad_contract_blocks(int n) {
  const int nt = (n + GT - 1) / GT;
  return nt * (nt + 1) / 2;
}

static bool has_ratquad(const KernSpec& spec) {
  for (int c = 0; c < spec.nkern; ++c)
    if (spec.kid[c] == KID_RATQUAD) return true;
  return false;
}

hipError_t launch_grad_contract(const KernSpec& spec, const double* theta, const double* X, int n, const double* W,
                                long ldw, const double* alpha, double* part, double* grad, hipStream_t stream, const Batch* bt,
                                unsigned* done, double* flag, double seq) {
  const int nblk = grad_contract_blocks(n);
  const int P = spec.nkern * spec.d + 2 * spec.nkern + 2;
  const dim3 grid(nblk, 1, bt ? bt->nb : 1);
  const int sth = bt ? bt->stheta : 0;
  const long sW = bt ? bt->sW : 0, sal = bt ? bt->salpha : 0, sp = bt ? bt->spart : 0;
  const bool rq = has_ratquad(spec);
  switch (spec.nkern) {
    case 1:
      if (rq) grad_contract_kernel<1, true><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      else grad_contract_kernel<1, false><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      break;
    case 2:
      if (rq) grad_contract_kernel<2, true><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      else grad_contract_kernel<2, false><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      break;
    case 3:
      if (rq) grad_contract_kernel<3, true><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      else grad_contract_kernel<3, false><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      break;
    case 4:
      if (rq) grad_contract_kernel<4, true><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      else grad_contract_kernel<4, false><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      break;
    default:
      if (rq) grad_contract_kernel<8, true><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      else grad_contract_kernel<8, false><<<grid, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, 0, 0, 0, 0, sth, sW, sal, sp);
      break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  grad_final_kernel<<<dim3(P, 1, bt ? bt->nb : 1), 256, 0, stream>>>(part, nblk, P, grad, sp, sth, done, flag, bt ? bt->sout : 0, seq);
  return hipGetLastError();
}

// Column slab [col0, col0 + cols) of the lower triangle (rows >= col0): W points at element (row0, col0) of K^-1,
// row0 <= col0, both multiples of 64.  part: grad_contract_slab_blocks() x ntheta doubles.
int grad_contract_slab_blocks(int n, int col0, int cols) {
  const int nt = (n + GT - 1) / GT;
  const int tj0 = col0 / GT, tw = min((cols + GT - 1) / GT, nt - tj0);
  return tw <= 0 ? 0 : (nt - tj0) * tw;
}

hipError_t launch_grad_contract_slab(const KernSpec& spec, const double* theta, const double* X, int n, const double* W,
                                     long ldw, int row0, int col0, int cols, const double* alpha, double* part,
                                     double* grad, hipStream_t stream) {
  const int nt = (n + GT - 1) / GT;
  const int tj0 = col0 / GT, tw = min((cols + GT - 1) / GT, nt - tj0);
  const int P = spec.nkern * spec.d + 2 * spec.nkern + 2;
  if (tw <= 0) return hipMemsetAsync(grad, 0, sizeof(double) * P, stream);
  const int nblk = (nt - tj0) * tw;
  const bool rq = has_ratquad(spec);
  switch (spec.nkern) {
    case 1:
      if (rq) grad_contract_kernel<1, true><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      else grad_contract_kernel<1, false><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      break;
    case 2:
      if (rq) grad_contract_kernel<2, true><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      else grad_contract_kernel<2, false><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      break;
    case 3:
      if (rq) grad_contract_kernel<3, true><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      else grad_contract_kernel<3, false><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      break;
    case 4:
      if (rq) grad_contract_kernel<4, true><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      else grad_contract_kernel<4, false><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      break;
    default:
      if (rq) grad_contract_kernel<8, true><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      else grad_contract_kernel<8, false><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, part, tw, tj0, row0, col0, 0, 0, 0, 0);
      break;
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  grad_final_kernel<<<P, 256, 0, stream>>>(part, nblk, P, grad, 0, 0, nullptr, nullptr, 0, 0.0);
  return hipGetLastError();
}

// Gradient of the posterior conditional w.r.t. ONE prediction point x* (converted inputs):
//   d mu / d x*_m  =  sum_i alpha_i dk(x_i, x*)/dx*_m,     d var / d x*_m = -2 sum_i w_i dk(x_i, x*)/dx*_m,
// with alpha = K^-1 y, w = K^-1 k(X, x*) and dk/dx*_m = sum_c coef_c kv_c k_c'(r2_c) 2 (x*_m - x_im) / l_cm^2
// (Stationary.diag is constant, so the prior variance does not move).  This is the differentiable single-point
// predictive that BO's refinement step builds in PyTensor at gpmcmc.py:766-778 and maximises with pm.find_MAP.
// One workgroup per point; dimensions in chunks of 16 held in registers, block-reduced through LDS.
template <int NK>
__global__ __launch_bounds__(256) void predict_grad_kernel(KernSpec spec, const double* __restrict__ theta,
                                                           const double* __restrict__ X, int n,
                                                           const double* __restrict__ xstar,
                                                           const double* __restrict__ alpha_v,
                                                           const double* __restrict__ wv, long ldwv,
                                                           double* __restrict__ dmean, double* __restrict__ dvar) {
  extern __shared__ double pg_dyn[];  // xs[d], then ils[NK][d]: sized by the launcher (any d that fits 64 KB of LDS)
  __shared__ double red[2][GXCH][4];
  const int tid = threadIdx.x;
  const int d = spec.d;
  const int nk = NK > 4 ? spec.nkern : NK;  // NK = 8: one instantiation for 5..8 components, run-time count
  double* xs = pg_dyn;
  double* ils = pg_dyn + d;
  const int p = blockIdx.x;
  const double* kv = theta + nk * d;
  const double* al = kv + nk;
  const double* w = wv + (long)p * ldwv;
  for (int e = tid; e < d; e += 256) xs[e] = xstar[(long)p * d + e];
  for (int e = tid; e < nk * d; e += 256) ils[e] = 1.0 / theta[e];
  __syncthreads();
  for (int m0 = 0; m0 < d; m0 += GXCH) {
    const int dc = min(GXCH, d - m0);
    double am[GXCH], av[GXCH];
#pragma unroll
    for (int u = 0; u < GXCH; ++u) am[u] = av[u] = 0.0;
    for (int i = tid; i < n; i += 256) {
      const double* xi = X + (long)i * d;
      double kval[NK], dkv[NK];
#pragma unroll
      for (int c = 0; c < nk; ++c) {
        double r2 = 0.0;
        for (int m = 0; m < d; ++m) {
          const double df = (xs[m] - xi[m]) * ils[c * d + m];
          r2 += df * df;
        }
        double k, dk, da;
        base_kernel_val_der(spec.kid[c], r2, al[c], k, dk, da);
        kval[c] = kv[c] * k;
        dkv[c] = kv[c] * dk;
      }
      double pref[NK];
      double T = kval[0];
      pref[0] = 1.0;
#pragma unroll
      for (int c = 1; c < nk; ++c) {
        pref[c] = (spec.op[c - 1] == 0) ? 1.0 : T;
        T = (spec.op[c - 1] == 0) ? T + kval[c] : T * kval[c];
      }
      const double ai = alpha_v[i], wi = -2.0 * w[i];
#pragma unroll
      for (int c = 0; c < nk; ++c) {
        double coef = pref[c];
#pragma unroll
        for (int c2 = c + 1; c2 < nk; ++c2)
          if (spec.op[c2 - 1] == 1) coef *= kval[c2];
        const double g = coef * dkv[c];
#pragma unroll
        for (int u = 0; u < GXCH; ++u) {
          if (u < dc) {
            const double il = ils[c * d + m0 + u];
            const double dkx = g * 2.0 * (xs[m0 + u] - xi[m0 + u]) * il * il;
            am[u] += ai * dkx;
            av[u] += wi * dkx;
          }
        }
      }
    }
    // fixed-order block reduction: wave shuffle, then the four wave sums
#pragma unroll
    for (int u = 0; u < GXCH; ++u) {
      double a = am[u], b = av[u];
      for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
      }
      if ((tid & 63) == 0) { red[0][u][tid >> 6] = a; red[1][u][tid >> 6] = b; }
    }
    __syncthreads();
    if (tid < dc) {
      dmean[(long)p * d + m0 + tid] = red[0][tid][0] + red[0][tid][1] + red[0][tid][2] + red[0][tid][3];
      dvar[(long)p * d + m0 + tid] = red[1][tid][0] + red[1][tid][1] + red[1][tid][2] + red[1][tid][3];
    }
    __syncthreads();
  }
}

hipError_t launch_predict_grad(const KernSpec& spec, const double* theta, const double* X, int n, const double* xstar,
                               int m, const double* alpha, const double* w, long ldw, double* dmean, double* dvar,
                               hipStream_t stream) {
  const size_t lds = sizeof(double) * (size_t)(spec.nkern + 1) * spec.d;
  if (lds > PREDICT_GRAD_MAX_LDS) return hipErrorInvalidValue;  // d <= 1536 with four components
  switch (spec.nkern) {
    case 1: predict_grad_kernel<1><<<m, 256, lds, stream>>>(spec, theta, X, n, xstar, alpha, w, ldw, dmean, dvar); break;
    case 2: predict_grad_kernel<2><<<m, 256, lds, stream>>>(spec, theta, X, n, xstar, alpha, w, ldw, dmean, dvar); break;
    case 3: predict_grad_kernel<3><<<m, 256, lds, stream>>>(spec, theta, X, n, xstar, alpha, w, ldw, dmean, dvar); break;
    case 4: predict_grad_kernel<4><<<m, 256, lds, stream>>>(spec, theta, X, n, xstar, alpha, w, ldw, dmean, dvar); break;
    default: predict_grad_kernel<8><<<m, 256, lds, stream>>>(spec, theta, X, n, xstar, alpha, w, ldw, dmean, dvar); break;
  }
  return hipGetLastError();
}

hipError_t launch_trmv_upper_t(const double* U, long ld, const double* x, int n, double* out, hipStream_t stream) {
  trmv_upper_t_kernel<<<(n + 63) / 64, 256, 0, stream>>>(U, ld, x, n, out);
  return hipGetLastError();
}

__global__ void gx_reduce_kernel(const double* __restrict__ part, int nsplit, long len, double* __restrict__ gx) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < len; e += (long)gridDim.x * blockDim.x) {
    double s = part[e];
    for (int k = 1; k < nsplit; ++k) s += part[(long)k * len + e];
    gx[e] = s;
  }
}

int grad_x_splits(int n, int d) {
  const int nrb = (n + GT - 1) / GT;
  int s = (512 + nrb - 1) / nrb;
  if (s > 8) s = 8;
  if (s > nrb) s = nrb;
  while (s > 1 && (size_t)s * n * d * sizeof(double) > ((size_t)256 << 20)) --s;
  return s < 1 ? 1 : s;
}

// scratch: [grad_x_splits(n, d)][n][d] doubles when more than one split is used (may be null otherwise)
hipError_t launch_grad_x(const KernSpec& spec, const double* theta, const double* X, int n, const double* W, long ldw,
                         const double* alpha, double* gx_out, double* scratch, hipStream_t stream) {
  const int nsplit = scratch ? grad_x_splits(n, spec.d) : 1;
  double* gx = nsplit > 1 ? scratch : gx_out;
  const dim3 nblk((n + GT - 1) / GT, nsplit);
  // one pass per window of GX_MAXD output dimensions (a single pass up to d = 128)
#define GX_LAUNCH(NK_)                                                                                                     \
  do {                                                                                                                    \
    if (spec.d <= GXCH) grad_x_kernel<NK_, 1><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, gx, 0);          \
    else if (spec.d <= 2 * GXCH) grad_x_kernel<NK_, 2><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, gx, 0); \
    else                                                                                                                  \
      for (int w0 = 0; w0 < spec.d; w0 += GX_MAXD)                                                                        \
        grad_x_kernel<NK_, GX_MAXD / GXCH><<<nblk, 256, 0, stream>>>(spec, theta, X, n, W, ldw, alpha, gx, w0);            \
  } while (0)
  switch (spec.nkern) {
    case 1: GX_LAUNCH(1); break;
    case 2: GX_LAUNCH(2); break;
    case 3: GX_LAUNCH(3); break;
    case 4: GX_LAUNCH(4); break;
    default: GX_LAUNCH(8); break;
  }
#undef GX_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || nsplit == 1) return e;
  const long len = (long)n * spec.d;
  gx_reduce_kernel<<<(int)((len + 255) / 256 < 1024 ? (len + 255) / 256 : 1024), 256, 0, stream>>>(scratch, nsplit, len, gx_out);
  return hipGetLastError();
}

hipError_t launch_predict_reduce(const double* A, long lda, const double* beta, int n, int m, double kdiag,
                                 double noise, double* mean, double* var, hipStream_t stream) {
  predict_reduce_kernel<<<(m + 3) / 4, 256, 0, stream>>>(A, lda, beta, n, m, kdiag, noise, mean, var);
  return hipGetLastError();
}

}  // namespace migp
