// Thin trapezoid updates of the panel chain (round 5): C[ti, tj] -= P[ti] P[tj]^T for the tile columns tj < nt of a
// trapezoid whose k is short (128, 256) and whose columns are few (1, 2 tile columns) -- the in-panel updates between two
// leaves, the next super-panel's first column, column mode's one update per column.  On the 64x64-tile GEMM kernel such a
// launch is 50-250 workgroups that each walk the whole k through LDS: 8 us for k = 128, 12-16 for k = 256 over two columns
// (kernel trace of N = 4096, profiles/NOTES_r05.md), nearly all of it latency -- the chain waits for every one of them.
// Here a workgroup owns 16 rows x 64 columns (wave w: one 16x16 tile).  Its rows come in as 1 KB row loads and reach MFMA operand
// order through LDS; the B operand is read from the operand-order copy that the strip in front of the update wrote (lane quarter q
// covers k = 16 kb + 4 q + s of a 16-wide k-block: 1 KB of consecutive addresses per wave load); each tile keeps four partial
// accumulators (one per MFMA step: a dependent fp64 MFMA issues after ~250 cycles, an independent one after 64).  8 x mt x 2 nt
// workgroups: a whole round of the chip from 16 tile rows on.  (Forms that loaded row-major operands straight into MFMA
// registers, for k up to 1024, were bound by the texture addresser -- every quarter-wave touching 16 rows -- and are gone.)
//
// Optional edge of the panel stream folded into the launch (option 26 = 2; it was a one-lane launch of its own): workgroup
// (0, 0, 0) raises *wr to val -- "everything queued on this stream before me is done".  (Round 5 also had every workgroup poll
// a second slot before its first load: a batched launch then filled every SIMD's registers with pollers while the main
// stream's bulk update, whose end the poll waited for, could not place a workgroup any more.  That edge is a poll at the end
// of the panel's last LEAF now -- one workgroup.)
#include <hip/hip_runtime.h>

#include "migp_kernels.h"

namespace migp {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

struct ThinArgs {
  const double* P;  // panel: element (0, 0) of the trapezoid's row tile 0, first k column
  double* C;        // element (0, 0) of the trapezoid (row tile 0, column tile 0)
  long ld, sZ;
  int mt, nt;
  unsigned* wr;
  unsigned val;
  const double* lsw;  // B operand in operand order
  const double* lsw2; // K = 256: the operand-order block of the second 128 k
  long sL;
};

template <int K>
__global__ __launch_bounds__(256) void syrk_thin_kernel(ThinArgs g) {
  __builtin_amdgcn_s_setprio(3);
  // row slices -> XCDs in contiguous ranges (workgroup b runs on XCD b % 8), as the strip and the GEMM kernels map their rows
  int rs = (int)blockIdx.x;
  {
    const int nblk = (int)gridDim.x, x = rs & 7, qq = nblk >> 3, r = nblk & 7;
    rs = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + (rs >> 3);
  }
  const int cg = (int)blockIdx.y;
  const int tid = threadIdx.x;
  if (g.wr != nullptr && blockIdx.x == 0 && cg == 0 && blockIdx.z == 0 && tid == 0)
    __hip_atomic_store(g.wr, g.val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((cg >> 1) > (rs >> 3)) return;  // above the diagonal tile
  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const long zoff = (long)blockIdx.z * g.sZ;
  // B operand: 16x16 tiles of 256 doubles, tile (cb & 7, kb) of the 128-row block cb >> 3, two runs of 64 lanes x 2 doubles each
  const int cb = cg * 4 + wave;
  const double* bsw = g.lsw + (long)blockIdx.z * g.sL + (long)(cb >> 3) * 16384 + (long)((cb & 7) * 8) * 256 + 2 * lane;
  const double* bsw2 = K == 256 ? g.lsw2 + (long)blockIdx.z * g.sL + (long)((cb & 7) * 8) * 256 + 2 * lane : nullptr;
  double* out = g.C + zoff + (long)(rs * 16 + q) * g.ld + cg * 64 + wave * 16 + n;
  constexpr int NC = K / 64;
  double2_t a[NC][4][2], b[NC][4][2];
  // the workgroup's 16 x K rows as 1 KB row loads (four rows per wave), through LDS into operand order (rows K + 2 doubles
  // apart) -- as the strip does it
  constexpr int ALD = K + 2;  // (K = 128 or 256: a row's dword stride is 4 mod 64 either way)
  __shared__ __attribute__((aligned(16))) double As[16 * ALD];
  double2_t stage[K / 32];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int hk = 0; hk < K / 128; ++hk)
      stage[i * (K / 128) + hk] =
          *reinterpret_cast<const double2_t*>(g.P + zoff + (long)(rs * 16 + 4 * wave + i) * g.ld + 128 * hk + 2 * lane);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const double* src = (K == 256 && c >= 2) ? bsw2 + 256 * (4 * (c - 2) + kb) : bsw + 256 * (4 * c + kb);
      b[c][kb][0] = *reinterpret_cast<const double2_t*>(src);
      b[c][kb][1] = *reinterpret_cast<const double2_t*>(src + 128);
    }
  }
  double cold[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) cold[r] = out[(long)(4 * r) * g.ld];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int hk = 0; hk < K / 128; ++hk)
      *reinterpret_cast<double2_t*>(As + (4 * wave + i) * ALD + 128 * hk + 2 * lane) = stage[i * (K / 128) + hk];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      a[c][kb][0] = *reinterpret_cast<const double2_t*>(As + n * ALD + 64 * c + 16 * kb + 4 * q);
      a[c][kb][1] = *reinterpret_cast<const double2_t*>(As + n * ALD + 64 * c + 16 * kb + 4 * q + 2);
    }
  }
  const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  double4_t p[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      p[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][kb][0].x, b[c][kb][0].x, p[0], 0, 0, 0);
      p[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][kb][0].y, b[c][kb][0].y, p[1], 0, 0, 0);
      p[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][kb][1].x, b[c][kb][1].x, p[2], 0, 0, 0);
      p[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c][kb][1].y, b[c][kb][1].y, p[3], 0, 0, 0);
    }
  }
  const double4_t s = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(long)(4 * r) * g.ld] = cold[r] - s[r];
}

hipError_t launch_syrk_thin(const double* P, double* C, long ld, int mt, int nt, int k, hipStream_t stream, const Batch* bt,
                            unsigned* wr, unsigned val, const double* lsw, const double* lsw2) {
  ThinArgs g;
  g.P = P;
  g.C = C;
  g.ld = ld;
  g.sZ = bt ? bt->sK : 0;
  g.mt = mt;
  g.nt = nt;
  g.wr = wr;
  g.val = val;
  g.lsw = lsw;
  g.lsw2 = lsw2;
  g.sL = bt ? bt->sdinv : 0;
  if (lsw == nullptr || !((k == 128 && nt <= 2) || (k == 256 && lsw2 != nullptr && nt == 1))) return hipErrorInvalidValue;
  const dim3 grid(mt * 8, nt * 2, bt ? bt->nb : 1);
  if (k == 128) syrk_thin_kernel<128><<<grid, 256, 0, stream>>>(g);
  else syrk_thin_kernel<256><<<grid, 256, 0, stream>>>(g);
  return hipGetLastError();
}

}  // namespace migp
