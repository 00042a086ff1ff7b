// Covariance assembly and the small reductions around the factorisation
// (K6).  Restates pymc.gp.cov.Stationary.square_dist / euclidean_dist and the ExpQuad / Matern52 /
// Matern32 / Exponential / RatQuad .full formulas that gpmcmc.py:282-307 composes:
//   Xs = X * (1/ls);  r2 = clip(-2 Xs Xs^T + (|Xs_i|^2 + |Xs_j|^2), 0, inf);  r = sqrt(r2 + 1e-12)
// One 64x64 output tile per 256-thread workgroup; the two 64-row x-blocks are staged (already scaled by 1/ls) in LDS.
// Round 2: the Xs Xs^T tile is formed on v_mfma_f64_16x16x4_f64 (the fp64 matrix rate equals the vector rate on this
// chip, so MFMA buys no flops -- it buys LDS traffic: 2 operand reads per 2048 flops instead of 8, and the dot phase was
// LDS-bandwidth-bound); the column-side operand rows are permuted so that a lane's four results are four ADJACENT
// columns of one row (two 16-byte stores instead of four 8-byte ones); exp and sqrt are branch-free fp64 sequences of
// ~19 and ~10 instructions (<= 1 ulp, checked against libm over the argument range) instead of the libm calls.
#include "migp_kernels.h"
#include "migp_math.h"

namespace migp {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int AT = 64;      // assembly tile
constexpr int DCH = 32;     // input dimensions per LDS chunk (multiple of 4)
constexpr int DLD = 36;     // LDS row stride in doubles: 4 mod 32, so 16 rows x 4 columns of a fragment read cover all bank pairs twice

__device__ __forceinline__ double base_kernel_eval(int kid, double r2, double alpha) {
  switch (kid) {
    case KID_RBF:
      return exp_nonpos(-0.5 * r2);
    case KID_RATQUAD:
      return pow(1.0 + 0.5 * r2 * (1.0 / alpha), -1.0 * alpha);
    default: {
      const double r = sqrt_pos(r2 + 1e-12);
      if (kid == KID_MATERN52) return (1.0 + 2.23606797749979 * r + 5.0 / 3.0 * (r * r)) * exp_nonpos(-1.0 * 2.23606797749979 * r);
      if (kid == KID_MATERN32) return (1.0 + 1.7320508075688772 * r) * exp_nonpos(-1.7320508075688772 * r);
      return exp_nonpos(-0.5 * r);  // KID_EXPONENTIAL (PyMC's Exponential is exp(-r/2))
    }
  }
}

// theta layout: [ls(nkern*d), kv(nkern), alpha(nkern), gv, jitter]
// Work split: a workgroup owns a run of up to TPW consecutive 64x64 tiles of one tile row; wave w owns rows 16 w .. 16 w + 15
// of a tile and all 64 columns as four 16-column groups cg.  MFMA operands (lane = (n = lane & 15, q = lane >> 4), k-step s
// covers input dimensions 4 s + q):
//   B operand = the lane's row point  Xi[16 w + n][4 s + q];   A operand = column point  Xj[16 cg + g(n)][4 s + q],
//   g(n) = 4 (n & 3) + (n >> 2), so that the D layout (row q + 4 r of A's index, column n of B's) is
//   dot(row point 16 w + n, column point 16 cg + 4 q + r): element r = 0..3 of a lane = four adjacent columns.
// Single-component covariances with d <= 32 (the common case) keep the scaled row block and its norms resident for the
// whole run and fetch the NEXT tile's column block into registers while the current tile is evaluated: a workgroup of the
// one-tile-per-workgroup form lived ~14 us of which ~1 us was arithmetic (three dependent global round trips per tile).
// KID_STATIC >= 0: single-component covariance of that family (no runtime switch, no pow() in the register budget);
// -1: any composition (runtime kernel ids, '+' / '*' folds).
constexpr int TPW = 8;                  // tiles per workgroup
constexpr int SROWS = AT / (256 / DCH);  // rows a thread stages per 64-row block: 8

// RESIDENT: the host guarantees nkern == 1 and d <= DCH (only the prefetching form is compiled), else only the general form.

template <int KID_STATIC, bool RESIDENT>
__global__ __launch_bounds__(256, (KID_STATIC >= 0 && KID_STATIC != KID_RATQUAD) ? 3 : 2) void assemble_kernel(
    KernSpec spec, const double* __restrict__ theta_in, const double* __restrict__ X1, int n1, const double* __restrict__ X2, int n2,
    double* __restrict__ K_in, long ldk, int rows_pad, int cols_pad, int sym, int noise_form, int diag_shift,
    const double* __restrict__ extra_diag, long sK, int stheta) {
  const double* __restrict__ theta = theta_in + (long)blockIdx.z * stheta;  // batched evaluation: problem blockIdx.z
  double* __restrict__ K = K_in + (long)blockIdx.z * sK;
  __shared__ __attribute__((aligned(16))) double Xi[AT * DLD];
  __shared__ __attribute__((aligned(16))) double Xj[AT * DLD];
  __shared__ double n2i[AT], n2j[AT];
  const int tid = threadIdx.x;
  // run (ti, chunk): tiles tj = TPW chunk .. ; in sym mode tile row ti holds ti + 1 tiles, i.e. ti / TPW + 1 runs, and the
  // TPW rows of a row group g = ti / TPW all hold g + 1 runs: 4 g (g + 1) runs (TPW = 8) lie before group g
  int ti, chunk, tj_end;
  if (sym) {
    const int e = blockIdx.x;
    int g = (int)((sqrt(1.0 + (double)e) - 1.0) * 0.5);
    while ((TPW / 2) * (g + 1) * (g + 2) <= e) ++g;
    while ((TPW / 2) * g * (g + 1) > e) --g;
    const int rem = e - (TPW / 2) * g * (g + 1);
    ti = TPW * g + rem / (g + 1);
    chunk = rem % (g + 1);
    tj_end = ti + 1;
  } else {
    const int nrun = (cols_pad / AT + TPW - 1) / TPW;
    ti = blockIdx.x / nrun;
    chunk = blockIdx.x % nrun;
    tj_end = cols_pad / AT;
  }
  // gridDim.y > 1 (small problems, round 6): a run is split over that many workgroups -- N = 4096 is 288 runs, one per CU and 40 us
  // each; which workgroup evaluates a tile does not enter its arithmetic
  const int per = TPW / (int)gridDim.y;
  const int tj0 = chunk * TPW + (int)blockIdx.y * per;
  const int tj1 = min(tj0 + per, min(chunk * TPW + TPW, tj_end));
  if (tj0 >= tj1) return;
  const int i0 = ti * AT;
  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int gperm = 4 * (n & 3) + (n >> 2);
  const int d = spec.d, nk = spec.nkern;
  const double* ls = theta;
  const double* kv = theta + nk * d;
  const double* al = kv + nk;
  const double gv = theta[nk * d + 2 * nk];
  const double jitter = theta[nk * d + 2 * nk + 1];
  const double sg = sqrt(gv);
  // diagonal of the global matrix: local (gi, gj) with gi + diag_shift == gj; sym mode has shift 0,
  // rectangular blocks of a distributed matrix pass row0 - col0, cross-covariances pass INT_MIN (none)
  const bool diag_on = sym || diag_shift != -2147483647 - 1;
  const int gi = i0 + 16 * wave + n;
  const int sm = tid % DCH, sr = tid / DCH;  // staging role: input dimension sm of rows sr + 8 u

  // |Xs_row|^2 of a staged 64-row block, four threads per row (quad partial sums combined by DPP)
  auto block_norms = [&](const double* Xs, double* out, int dc, bool first) {
    const int row = tid >> 2, part = tid & 3;
    double s = 0.0;
    for (int mm = part; mm < dc; mm += 4) s = __builtin_fma(Xs[row * DLD + mm], Xs[row * DLD + mm], s);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (part == 0) out[row] = first ? s : out[row] + s;
  };
  auto mfma_tile = [&](double4_t (&dot)[4], int dc) {
    const int nstep = (dc + 3) >> 2;
    const double* bp = Xi + (16 * wave + n) * DLD + q;
    const double* ap = Xj + gperm * DLD + q;
    for (int s4 = 0; s4 < nstep; ++s4) {
      const double b = bp[4 * s4];
#pragma unroll
      for (int cg = 0; cg < 4; ++cg)
        dot[cg] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[16 * cg * DLD + 4 * s4], b, dot[cg], 0, 0, 0);
    }
  };
  auto fold = [&](double4_t (&Kacc)[4], const double4_t (&dot)[4], int c) {
    const int kid = KID_STATIC >= 0 ? KID_STATIC : spec.kid[c];
    const double kvc = kv[c], alc = al[c];
    const double ni = n2i[16 * wave + n];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {  // four outputs at a time (scheduling fence below): enough independent exp / sqrt
                                      // chains, a quarter of the registers of sixteen interleaved ones
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double r2 = -2.0 * dot[cg][r] + (ni + n2j[16 * cg + 4 * q + r]);
        r2 = fmax(r2, 0.0);
        const double kval = kvc * base_kernel_eval(kid, r2, alc);
        if (c == 0) Kacc[cg][r] = kval;
        else if (spec.op[c - 1] == 0) Kacc[cg][r] = Kacc[cg][r] + kval;
        else Kacc[cg][r] = Kacc[cg][r] * kval;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto store_tile = [&](const double4_t (&Kacc)[4], int j0) {
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      double v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gj = j0 + 16 * cg + 4 * q + r;
        double x = Kacc[cg][r];
        if (gi >= n1 || gj >= n2) {
          x = (diag_on && gi + diag_shift == gj) ? 1.0 : 0.0;
        } else if (diag_on && gi + diag_shift == gj) {
          if (noise_form == 0) { x += sg * sg; x += jitter; }       // Marginal._build_marginal_likelihood
          else if (noise_form == 1) { x += jitter; x += sg * sg; }  // Marginal._build_conditional
          else { x += jitter + gv; }                                // gpmcmc.py:312 explicit form
          if (extra_diag) x += extra_diag[gi];                      // per-point noise vector (inverse_opt, gpmcmc.py:1134-1158)
        }
        v[r] = x;
      }
      double* dst = K + (long)gi * ldk + j0 + 16 * cg + 4 * q;  // ldk even, column a multiple of 4: 16-byte aligned
      *reinterpret_cast<double2_t*>(dst) = (double2_t){v[0], v[1]};
      *reinterpret_cast<double2_t*>(dst + 2) = (double2_t){v[2], v[3]};
    }
  };

  if (RESIDENT) {
    // ---- resident row block, column blocks prefetched one tile ahead
    const double il = (sm < d) ? 1.0 / ls[sm] : 0.0;
    double xr[SROWS];
    auto fetch = [&](const double* X, int nx, int r0) {
#pragma unroll
      for (int u = 0; u < SROWS; ++u) {
        const int row = r0 + sr + (256 / DCH) * u;
        xr[u] = (sm < d && row < nx) ? X[(long)row * d + sm] : 0.0;
      }
    };
    auto park = [&](double* Xs) {
#pragma unroll
      for (int u = 0; u < SROWS; ++u) Xs[(sr + (256 / DCH) * u) * DLD + sm] = xr[u] * il;
    };
    fetch(X1, n1, i0);
    park(Xi);
    fetch(X2, n2, tj0 * AT);
    __syncthreads();
    block_norms(Xi, n2i, d, true);
    for (int tj = tj0; tj < tj1; ++tj) {
      if (tj > tj0) __syncthreads();  // the previous tile's operand reads of Xj / n2j are done
      park(Xj);
      __syncthreads();
      block_norms(Xj, n2j, d, true);
      if (tj + 1 < tj1) fetch(X2, n2, (tj + 1) * AT);  // travels while this tile is evaluated
      __syncthreads();
      double4_t dot[4], Kacc[4];
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) dot[cg] = (double4_t){0.0, 0.0, 0.0, 0.0};
      mfma_tile(dot, d);
      fold(Kacc, dot, 0);
      store_tile(Kacc, tj * AT);
    }
    return;
  }
  // ---- general form: any composition, any d; everything is restaged per tile, component and chunk of 32 dimensions
  for (int tj = tj0; tj < tj1; ++tj) {
    const int j0 = tj * AT;
    double4_t Kacc[4];
    for (int c = 0; c < nk; ++c) {
      double4_t dot[4];
#pragma unroll
      for (int cg = 0; cg < 4; ++cg) dot[cg] = (double4_t){0.0, 0.0, 0.0, 0.0};
      for (int m0 = 0; m0 < d; m0 += DCH) {
        const int dc = min(DCH, d - m0);
        __syncthreads();
        // a thread always stages the same input dimension, so its 1/l is one division per chunk; dimensions dc .. DCH-1
        // are staged as zeros (the MFMA steps run over dc rounded up to a multiple of 4)
        const double il = (sm < dc) ? 1.0 / ls[c * d + m0 + sm] : 0.0;
#pragma unroll
        for (int u = 0; u < SROWS; ++u) {
          const int r = sr + (256 / DCH) * u;
          double vi = 0.0, vj = 0.0;
          if (sm < dc) {
            if (i0 + r < n1) vi = X1[(long)(i0 + r) * d + m0 + sm] * il;
            if (j0 + r < n2) vj = X2[(long)(j0 + r) * d + m0 + sm] * il;
          }
          Xi[r * DLD + sm] = vi;
          Xj[r * DLD + sm] = vj;
        }
        __syncthreads();
        block_norms(Xi, n2i, dc, m0 == 0);
        block_norms(Xj, n2j, dc, m0 == 0);
        mfma_tile(dot, dc);
      }
      __syncthreads();
      fold(Kacc, dot, c);
    }
    store_tile(Kacc, j0);
  }
}

// rows [row0, row0+128) x cols [0, cols_pad): zero, except row row0 = y^T (first n entries)
__global__ void set_yrows_kernel(double* __restrict__ K, long ldk, int row0, int cols_pad,
                                 const double* __restrict__ y, int n, int* __restrict__ info,
                                 const double* __restrict__ theta_src, double* __restrict__ theta_dst, int ntheta, long sK,
                                 int sinfo, int stheta) {
  K += (long)blockIdx.z * sK;  // batched evaluation: problem blockIdx.z
  if (info) info += (long)blockIdx.z * sinfo;
  if (theta_src) {
    theta_src += (long)blockIdx.z * stheta;
    theta_dst += (long)blockIdx.z * stheta;
  }
  // info (optional): the evaluation's bad-pivot word starts as "none" here (a memset less per evaluation)
  if (info && blockIdx.x == 0 && threadIdx.x == 0) info[0] = 0x7f7f7f7f;
  // theta (optional): this evaluation's hyper-parameters travel from the handle's pinned host buffer to the device copy
  // every later kernel reads -- a copy launch less per evaluation (this kernel runs first)
  if (theta_src && blockIdx.x == 0)
    for (int i = threadIdx.x; i < ntheta; i += blockDim.x) theta_dst[i] = theta_src[i];
  const long total = 128L * cols_pad;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / cols_pad), c = (int)(e % cols_pad);
    double v = 0.0;
    if (r == 0 && c < n) v = y[c];
    K[(long)(row0 + r) * ldk + c] = v;
  }
}

// out[0] = LML, out[1] = sum log L_ii, out[2] = |beta|^2   (gpmcmc.py:316-318 / MvNormal.logp)
// LR_BLOCKS workgroups per problem, each sums a contiguous slice of the diagonal (the strided diagonal walk and fp64 log are
// pure latency: one workgroup took 52 us at N = 16384); the last one to finish (a ticket in `sync`) adds the slices' partial
// sums in slice order -- the summation order is fixed, the result bit-reproducible.
constexpr int LR_BLOCKS = LML_REDUCE_BLOCKS;
__global__ __launch_bounds__(256) void lml_reduce_kernel(const double* __restrict__ L, long ld,
                                                         const double* __restrict__ beta, int n,
                                                         double* __restrict__ out, const int* __restrict__ info, long sK,
                                                         int sout, int sinfo, double* __restrict__ part,
                                                         unsigned* __restrict__ sync, double seq) {
  L += (long)blockIdx.z * sK;  // batched evaluation: problem blockIdx.z
  beta += (long)blockIdx.z * sK;
  out += (long)blockIdx.z * sout;
  if (info) info += (long)blockIdx.z * sinfo;
  if (part) part += (long)blockIdx.z * 2 * LR_BLOCKS;
  if (sync) sync += blockIdx.z;
  __shared__ double s1[256], s2[256];
  __shared__ unsigned ticket;
  const int per = gridDim.x == 1 ? n : ((n + LR_BLOCKS - 1) / LR_BLOCKS + 255) / 256 * 256;
  const int lo = blockIdx.x * per, hi = min(n, lo + per);
  double a = 0.0, b = 0.0;
  for (int i0 = lo + threadIdx.x; i0 < hi; i0 += 256 * 4) {
    double dv[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
      dv[u] = (i < hi) ? L[(long)i * ld + i] : 1.0;
      bv[u] = (i < hi) ? beta[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
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
  if (gridDim.x == 1) {  // no scratch given (block-level entry mi_gp_lml_partial): one workgroup does it all
    if (threadIdx.x == 0) {
      out[1] = s1[0];
      out[2] = s2[0];
      out[0] = -0.5 * (double)n * 1.8378770664093453 - 0.5 * s2[0] - s1[0];
      if (info) out[3] = (double)info[0];
    }
    return;
  }
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = s1[0];
    part[2 * blockIdx.x + 1] = s2[0];
    __threadfence();
    ticket = atomicAdd(sync, 1u);
  }
  __syncthreads();
  if (ticket != LR_BLOCKS - 1) return;
  if (threadIdx.x == 0) {
    __threadfence();
    double t1 = 0.0, t2 = 0.0;
    for (int k = 0; k < LR_BLOCKS; ++k) {
      t1 += __hip_atomic_load(part + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      t2 += __hip_atomic_load(part + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    out[1] = t1;
    out[2] = t2;
    out[0] = -0.5 * (double)n * 1.8378770664093453 - 0.5 * t2 - t1;
    if (info) out[3] = (double)info[0];  // the bad-pivot word rides in the same download as the scalars
    *sync = 0u;                          // ready for the next evaluation
    // the evaluation's sequence number, LAST and released at system scope: a host that spins on it (pinned, coherent memory)
    // sees the scalars above once it sees the number -- without the round trip of a stream synchronisation
    if (seq != 0.0) __hip_atomic_store(out + 4, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

hipError_t launch_assemble(const KernSpec& spec, const double* theta, const double* X1, int n1, const double* X2,
                           int n2, double* K, long ldk, int rows_pad, int cols_pad, int sym, int noise_form,
                           hipStream_t stream, int diag_shift, const double* extra_diag, const Batch* bt) {
  int nblk;  // runs of up to TPW tiles of one tile row
  if (sym) {
    const int nt = rows_pad / AT, G = nt / TPW, rem = nt % TPW;
    nblk = (TPW / 2) * G * (G + 1) + rem * (G + 1);
  } else {
    nblk = (rows_pad / AT) * ((cols_pad / AT + TPW - 1) / TPW);
  }
  const int ds = sym ? 0 : diag_shift;
  const double* ed = sym ? extra_diag : nullptr;
  const bool resident = spec.nkern == 1 && spec.d <= DCH;
  // few runs (N <= 6144 for one problem): split each over 2 / 4 / 8 workgroups so that every CU has several to overlap
  const long nwg = (long)nblk * (bt ? bt->nb : 1);
  const int split = nwg >= 2048 ? 1 : nwg >= 1024 ? 2 : nwg >= 512 ? 4 : 8;
  const dim3 grid(nblk, split, bt ? bt->nb : 1);
  const long sK = bt ? bt->sK : 0;
  const int sth = bt ? bt->stheta : 0;
#define MIGP_ASM(KID)                                                                                                              \
  do {                                                                                                                             \
    if (resident)                                                                                                                  \
      assemble_kernel<KID, true><<<grid, 256, 0, stream>>>(spec, theta, X1, n1, X2, n2, K, ldk, rows_pad, cols_pad, sym, noise_form, ds, ed, sK, sth);  \
    else                                                                                                                           \
      assemble_kernel<KID, false><<<grid, 256, 0, stream>>>(spec, theta, X1, n1, X2, n2, K, ldk, rows_pad, cols_pad, sym, noise_form, ds, ed, sK, sth); \
  } while (0)
  if (spec.nkern != 1) MIGP_ASM(-1);
  else if (spec.kid[0] == KID_RBF) MIGP_ASM(KID_RBF);
  else if (spec.kid[0] == KID_MATERN52) MIGP_ASM(KID_MATERN52);
  else if (spec.kid[0] == KID_MATERN32) MIGP_ASM(KID_MATERN32);
  else if (spec.kid[0] == KID_EXPONENTIAL) MIGP_ASM(KID_EXPONENTIAL);
  else MIGP_ASM(KID_RATQUAD);
#undef MIGP_ASM
  return hipGetLastError();
}

hipError_t launch_set_yrows(double* K, long ldk, int row0, int cols_pad, const double* y, int n, hipStream_t stream,
                            int* info, const double* theta_src, double* theta_dst, int ntheta, const Batch* bt) {
  const long total = 128L * cols_pad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  set_yrows_kernel<<<dim3(blocks, 1, bt ? bt->nb : 1), 256, 0, stream>>>(K, ldk, row0, cols_pad, y, n, info, theta_src, theta_dst, ntheta,
                                                                         bt ? bt->sK : 0, bt ? bt->sinfo : 0, bt ? bt->stheta : 0);
  return hipGetLastError();
}

hipError_t launch_lml_reduce(const double* L, long ld, const double* beta, int n, double* out, hipStream_t stream,
                             const int* info, const Batch* bt, double* part, unsigned* sync, double seq) {
  lml_reduce_kernel<<<dim3((part && sync) ? LR_BLOCKS : 1, 1, bt ? bt->nb : 1), 256, 0, stream>>>(L, ld, beta, n, out, info, bt ? bt->sK : 0,
                                                                             bt ? bt->sout : 0, bt ? bt->sinfo : 0, part, sync, seq);
  return hipGetLastError();
}

}  // namespace migp
