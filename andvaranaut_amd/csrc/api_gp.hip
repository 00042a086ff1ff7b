// Handle-level C-ABI (include/mi_gp.h): covariance assembly -> blocked right-looking Cholesky ->
// log marginal likelihood.  Replaces what pm.find_MAP / pm.sample evaluate per step through
// pm.gp.Marginal.marginal_likelihood (gpmcmc.py:321-323, 345, 351).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "migp_kernels.h"
#include "../../include/mi_gp.h"

using namespace migp;

struct mi_gp_handle {
  mi_gp_config cfg;
  KernSpec spec;
  int n, np, ntc;       // points, padded points, 128-column tiles
  int ntheta;
  int device;
  hipStream_t stream;
  mi_gp_buffers buf;
  bool have_data;
  // handle-owned small scratch
  double* theta_dev;    // [ntheta]
  double* out_dev;      // [16] scalars
  double* dinv_dev;     // [ntc][8][16][16]
  int* info_dev;
  double* out_host;     // pinned [16]
  int* info_host;       // pinned
  double* theta_host;   // pinned
  // profiling
  int prof_level;
  hipEvent_t ev[8];
  std::vector<hipEvent_t> gemm_ev;  // pairs
  size_t gemm_ev_used;
  double gemm_flops_acc;
  double t_assemble_ms, t_chol_ms, t_reduce_ms, t_gemm_ms, t_total_ms, gemm_flops, n_gemm;
  char err[256];
};

static int hfail(mi_gp_handle* h, hipError_t e, const char* where) {
  snprintf(h->err, sizeof(h->err), "%s: %s", where, hipGetErrorString(e));
  return -2;
}
#define HCK(call, where)                          \
  do {                                            \
    hipError_t e__ = (call);                      \
    if (e__ != hipSuccess) return hfail(h, e__, where); \
  } while (0)

extern "C" const char* mi_gp_last_error(mi_gp_handle* h) { return h ? h->err : "null handle"; }

extern "C" int mi_gp_create(const mi_gp_config* cfg, mi_gp_handle** out) {
  if (!cfg || !out) return -1;
  if (cfg->n <= 0 || cfg->d <= 0 || cfg->nkern <= 0 || cfg->nkern > MAX_KERN) return -1;
  for (int i = 0; i < cfg->nkern; ++i)
    if (cfg->kernel_ids[i] < 0 || cfg->kernel_ids[i] > KID_RATQUAD) return -1;
  mi_gp_handle* h = new mi_gp_handle();
  memset(h->err, 0, sizeof(h->err));
  h->cfg = *cfg;
  h->spec.nkern = cfg->nkern;
  h->spec.d = cfg->d;
  for (int i = 0; i < MAX_KERN; ++i) {
    h->spec.kid[i] = i < cfg->nkern ? cfg->kernel_ids[i] : 0;
    h->spec.op[i] = i < cfg->nkern ? cfg->ops[i] : 0;
  }
  h->n = cfg->n;
  h->np = (cfg->n + 127) / 128 * 128;
  h->ntc = h->np / 128;
  h->ntheta = cfg->nkern * cfg->d + 2 * cfg->nkern + 2;
  h->device = cfg->device;
  h->have_data = false;
  h->prof_level = 0;
  h->gemm_ev_used = 0;
  hipError_t e = hipSetDevice(h->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMalloc(&h->theta_dev, sizeof(double) * h->ntheta);
  if (e == hipSuccess) e = hipMalloc(&h->out_dev, sizeof(double) * 16);
  if (e == hipSuccess) e = hipMalloc(&h->dinv_dev, sizeof(double) * 2048 * (size_t)h->ntc);
  if (e == hipSuccess) e = hipMalloc(&h->info_dev, sizeof(int) * 4);
  if (e == hipSuccess) e = hipHostMalloc(&h->out_host, sizeof(double) * 16);
  if (e == hipSuccess) e = hipHostMalloc(&h->info_host, sizeof(int) * 4);
  if (e == hipSuccess) e = hipHostMalloc(&h->theta_host, sizeof(double) * h->ntheta);
  for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreate(&h->ev[i]);
  if (e == hipSuccess) e = gemm_f64_enable_lds();
  if (e == hipSuccess) e = leaf_enable_lds();
  if (e != hipSuccess) {
    snprintf(h->err, sizeof(h->err), "mi_gp_create: %s", hipGetErrorString(e));
    fprintf(stderr, "%s\n", h->err);
    delete h;
    return -2;
  }
  *out = h;
  return 0;
}

extern "C" int mi_gp_destroy(mi_gp_handle* h) {
  if (!h) return 0;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  hipFree(h->theta_dev); hipFree(h->out_dev); hipFree(h->dinv_dev); hipFree(h->info_dev);
  hipHostFree(h->out_host); hipHostFree(h->info_host); hipHostFree(h->theta_host);
  for (int i = 0; i < 8; ++i) hipEventDestroy(h->ev[i]);
  for (auto& ev : h->gemm_ev) hipEventDestroy(ev);
  hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

extern "C" long mi_gp_padded_n(const mi_gp_handle* h) { return h ? h->np : -1; }
extern "C" int mi_gp_num_theta(const mi_gp_handle* h) { return h ? h->ntheta : -1; }
extern "C" void* mi_gp_stream(const mi_gp_handle* h) { return h ? (void*)h->stream : nullptr; }

extern "C" int mi_gp_set_data(mi_gp_handle* h, const mi_gp_buffers* b) {
  if (!h || !b || !b->X_dev || !b->y_dev || !b->K_dev) return -1;
  if (b->lda < h->np || (b->lda & 1)) {
    snprintf(h->err, sizeof(h->err), "mi_gp_set_data: lda must be even and >= padded n (%d)", h->np);
    return -1;
  }
  h->buf = *b;
  h->have_data = true;
  return 0;
}

extern "C" int mi_gp_set_profiling(mi_gp_handle* h, int level) {
  if (!h) return -1;
  h->prof_level = level;
  return 0;
}

// ---------------------------------------------------------------- driver pieces
static hipError_t prof_gemm(mi_gp_handle* h, const GemmParams& p, int ak, int bk, int batch, double flops) {
  if (h->prof_level >= 2) {
    if (h->gemm_ev_used + 2 > h->gemm_ev.size()) {
      for (int i = 0; i < 64; ++i) {
        hipEvent_t e;
        hipError_t r = hipEventCreate(&e);
        if (r != hipSuccess) return r;
        h->gemm_ev.push_back(e);
      }
    }
    hipEventRecord(h->gemm_ev[h->gemm_ev_used], h->stream);
    hipError_t r = launch_gemm_f64(p, ak, bk, batch, h->stream);
    hipEventRecord(h->gemm_ev[h->gemm_ev_used + 1], h->stream);
    h->gemm_ev_used += 2;
    h->gemm_flops_acc += flops;
    return r;
  }
  return launch_gemm_f64(p, ak, bk, batch, h->stream);
}

// trapezoid update  A[r0:, c0:c0+nc] -= P P_c^T  with P = A[r0:, k0:k0+kw] (tile units)
static hipError_t syrk_trapezoid(mi_gp_handle* h, double* A, long lda, int ntr, int r0, int nc, int k0, int kw) {
  GemmParams p;
  p.A = A + (long)r0 * 128 * lda + (long)k0 * 128;
  p.B = p.A;
  p.C = A + (long)r0 * 128 * lda + (long)r0 * 128;
  p.lda = p.ldb = p.ldc = lda;
  p.strideA = p.strideB = p.strideC = 0;
  p.mt = ntr - r0;
  p.nt = nc;
  p.k = kw * 128;
  p.tri = 1;
  p.kmode = 0;
  p.alpha = -1.0;
  p.beta = 1.0;
  // algorithmic flops (nb*m^2 for the lower-triangle SYRK, 2*nb*rows*cols for the block
  // below it, one y^T row for the folded-in forward solve); the MFMA work issued is slightly larger
  // (full diagonal tiles, a 128-row tile for the y row).
  const double c = nc * 128.0, rows_real = (p.mt - 1) * 128.0;
  const double flops = (double)p.k * (c * (c + 1.0) + 2.0 * (rows_real - c) * c + 2.0 * c);
  return prof_gemm(h, p, 0, 0, 1, flops);
}

// factor tile columns [c0, c0+w) of the (ntr x ntc)-tile trapezoid, recursively halving w
static hipError_t chol_panel(mi_gp_handle* h, double* A, long lda, int ntr, int c0, int w) {
  hipError_t e;
  if (w == 1) {
    double* blk = A + (long)c0 * 128 * lda + (long)c0 * 128;
    double* dinv = h->dinv_dev + (size_t)c0 * 2048;
    e = launch_potrf_leaf128(blk, lda, dinv, c0 * 128, h->info_dev, h->stream);
    if (e != hipSuccess) return e;
    const int m = (ntr - c0 - 1) * 128;
    return launch_trsm_strip128(blk, lda, dinv, blk + 128 * lda, lda, m, h->stream);
  }
  const int w1 = w / 2, w2 = w - w1;
  e = chol_panel(h, A, lda, ntr, c0, w1);
  if (e != hipSuccess) return e;
  e = syrk_trapezoid(h, A, lda, ntr, c0 + w1, w2, c0, w1);
  if (e != hipSuccess) return e;
  return chol_panel(h, A, lda, ntr, c0 + w1, w2);
}

static hipError_t cholesky(mi_gp_handle* h, double* A, long lda, int ntr, int ntc) {
  int W = h->cfg.panel_tiles > 0 ? h->cfg.panel_tiles : 2;
  for (int J = 0; J < ntc; J += W) {
    const int w = (ntc - J < W) ? (ntc - J) : W;
    hipError_t e = chol_panel(h, A, lda, ntr, J, w);
    if (e != hipSuccess) return e;
    if (J + w < ntc) {
      e = syrk_trapezoid(h, A, lda, ntr, J + w, ntc - J - w, J, w);
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

static int upload_theta(mi_gp_handle* h, const double* theta) {
  for (int i = 0; i < h->ntheta; ++i) {
    if (!std::isfinite(theta[i])) { snprintf(h->err, sizeof(h->err), "theta[%d] is not finite", i); return -1; }
    h->theta_host[i] = theta[i];
  }
  HCK(hipMemcpyAsync(h->theta_dev, h->theta_host, sizeof(double) * h->ntheta, hipMemcpyHostToDevice, h->stream), "theta upload");
  return 0;
}

// assemble + factor the augmented trapezoid [[K],[y^T]]; leaves L in K_dev, beta = L^-1 y in row np
static int factor_internal(mi_gp_handle* h, const double* theta, int noise_form) {
  if (!h->have_data) { snprintf(h->err, sizeof(h->err), "mi_gp_set_data has not been called"); return -1; }
  HCK(hipSetDevice(h->device), "hipSetDevice");
  if (int r = upload_theta(h, theta)) return r;
  const bool prof = h->prof_level >= 1;
  h->gemm_ev_used = 0;
  h->gemm_flops_acc = 0.0;
  HCK(hipMemsetAsync(h->info_dev, 0x7f, sizeof(int) * 4, h->stream), "info reset");
  if (prof) hipEventRecord(h->ev[0], h->stream);
  HCK(launch_assemble(h->spec, h->theta_dev, h->buf.X_dev, h->n, h->buf.X_dev, h->n, h->buf.K_dev, h->buf.lda, h->np,
                      h->np, 1, noise_form, h->stream), "assemble");
  HCK(launch_set_yrows(h->buf.K_dev, h->buf.lda, h->np, h->np, h->buf.y_dev, h->n, h->stream), "set_yrows");
  if (prof) hipEventRecord(h->ev[1], h->stream);
  HCK(cholesky(h, h->buf.K_dev, h->buf.lda, h->ntc + 1, h->ntc), "cholesky");
  if (prof) hipEventRecord(h->ev[2], h->stream);
  HCK(launch_lml_reduce(h->buf.K_dev, h->buf.lda, h->buf.K_dev + (long)h->np * h->buf.lda, h->n, h->out_dev, h->stream), "lml_reduce");
  if (prof) hipEventRecord(h->ev[3], h->stream);
  HCK(hipMemcpyAsync(h->out_host, h->out_dev, sizeof(double) * 16, hipMemcpyDeviceToHost, h->stream), "out download");
  HCK(hipMemcpyAsync(h->info_host, h->info_dev, sizeof(int) * 4, hipMemcpyDeviceToHost, h->stream), "info download");
  HCK(hipStreamSynchronize(h->stream), "stream sync");
  if (prof) {
    float ms;
    hipEventElapsedTime(&ms, h->ev[0], h->ev[1]); h->t_assemble_ms = ms;
    hipEventElapsedTime(&ms, h->ev[1], h->ev[2]); h->t_chol_ms = ms;
    hipEventElapsedTime(&ms, h->ev[2], h->ev[3]); h->t_reduce_ms = ms;
    hipEventElapsedTime(&ms, h->ev[0], h->ev[3]); h->t_total_ms = ms;
    double g = 0.0;
    for (size_t i = 0; i + 1 < h->gemm_ev_used; i += 2) {
      hipEventElapsedTime(&ms, h->gemm_ev[i], h->gemm_ev[i + 1]);
      g += ms;
    }
    h->t_gemm_ms = g;
    h->gemm_flops = h->gemm_flops_acc;
    h->n_gemm = (double)(h->gemm_ev_used / 2);
  }
  const int info = h->info_host[0];
  if (info != 0x7f7f7f7f) return info;  // 1-based index of the first bad pivot
  return 0;
}

extern "C" int mi_gp_lml(mi_gp_handle* h, const double* theta, double* lml_out) {
  if (!h || !theta || !lml_out) return -1;
  const int r = factor_internal(h, theta, 0);
  if (r < 0) return r;
  if (r > 0) { *lml_out = -INFINITY; return r; }
  *lml_out = h->out_host[0];
  return 0;
}

extern "C" int mi_gp_lml_parts(mi_gp_handle* h, double* logdet, double* quad) {
  if (!h) return -1;
  if (logdet) *logdet = h->out_host[1];
  if (quad) *quad = h->out_host[2];
  return 0;
}

// out: [assemble_ms, chol_ms, reduce_ms, total_ms, gemm_ms, gemm_flops, n_gemm_launches]
extern "C" int mi_gp_timers(mi_gp_handle* h, double* out, int n) {
  if (!h || !out) return -1;
  const double v[7] = {h->t_assemble_ms, h->t_chol_ms, h->t_reduce_ms, h->t_total_ms, h->t_gemm_ms, h->gemm_flops, h->n_gemm};
  for (int i = 0; i < n && i < 7; ++i) out[i] = v[i];
  return 0;
}
