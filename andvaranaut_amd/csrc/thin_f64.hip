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


// ---------------------------------------------------------------------------------------------------------------------------------
// Strip + thin update in ONE launch (round 6).  A tile column of the panel chain is leaf -> strip -> update of the next column ->
// next leaf, and every launch on that chain costs ~3 us of kernel boundary on top of ~2-3 us of work.  Here the strip of column j
// (X = B M^T, trsm_strip128_body<1> of leaf_f64.hip) and the k = 128 / 256 update of column j + 1 that follows it (syrk_thin_kernel
// above) are one kernel: workgroup g owns 16 rows, computes their strip, keeps X_g in LDS as the update's A operand, and updates
// its 16 x 128 slice of the next column.  The update's B operand is the strip of the FIRST 128 rows (tile row j + 1): the eight
// workgroups that own them (blocks 0..7: dispatched first) publish their rows in operand order with write-through (sc1) stores, drain,
// and raise one flag word each; every workgroup polls the eight words (one lane each, sc1 loads, bounded), meets at a barrier and
// reads the operand with sc1 loads -- the in-launch hand-off MI355X_MICROARCH.md lists as valid across XCDs (per-XCD L2s are not
// coherent).  A workgroup waits only for workgroups of its own launch with smaller block ids, so nothing it waits for can be kept
// out by it.  Per element the arithmetic is that of the two kernels (same operands, same order of the partial sums): the fused and
// the split schedule return the same bits -- scheduling only.
// 512 threads at <= 128 registers (two waves per SIMD: a workgroup fits beside a GEMM workgroup of the main stream): each wave one
// column block of the strip, then the 16 x 16 tile of tile column w of the update (what workgroup (., w >> 2) wave w & 3 of
// syrk_thin_kernel computes); for k = 256 the half of the update that needs nothing fresh runs while the flags travel.
struct FusedArgs {
  const double* minv;      // leaf inverse of column j (operand order)
  double* B;               // element (tile row j + 1, column j): m rows x 128, solved in place
  const double* Pprev;     // K == 256: element (tile row j + 1, column j - 1), the A operand's first 128 k
  double* C;               // element (tile row j + 1, column j + 1)
  long ld, sK;             // leading dimension, stride between the problems of a batch
  double* lsw;             // this strip's operand-order copy (first lsw_blocks 16-row groups); its first block is the B operand
  const double* lsw_prev;  // K == 256: the block of column j - 1's copy that holds tile row j + 1 (first 128 k of the B operand)
  long sL;                 // batch stride of minv / lsw / lsw_prev
  int lsw_blocks;
  unsigned* flags;         // [problems][FUSE_FLAG_WORDS]
  unsigned tag;            // this launch's value of the flag words
  int* info;
  int sinfo, poll_log2;
};

typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
// 16-byte loads / 8-byte stores that another XCD's workgroup of the same launch can be on the other side of (sc1)
__device__ __forceinline__ double2_t load_sc1_b128(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
  return __builtin_bit_cast(double2_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 1 << 4));
}
__device__ __forceinline__ void store_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int K>
__global__ __launch_bounds__(512, 2) void strip_thin_kernel(FusedArgs g) {
  __builtin_amdgcn_s_setprio(3);
  // blocks 0..7 are the producers (row groups 0..7 = tile row j + 1); the others -> XCDs in contiguous ranges, as the strip maps them
  int rs = (int)blockIdx.x;
  if (rs >= FUSE_PRODUCERS) {
    const int b = rs - FUSE_PRODUCERS, nblk = (int)gridDim.x - FUSE_PRODUCERS, x = b & 7, qq = nblk >> 3, r = nblk & 7;
    rs = FUSE_PRODUCERS + (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + (b >> 3);
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, q = lane >> 4;
  const long z = blockIdx.z;
  const double* minv = g.minv + z * g.sL;
  double* B = g.B + z * g.sK;
  double* C = g.C + z * g.sK;
  double* lsw = g.lsw + z * g.sL;
  constexpr int ALD = K + 2, KOFF = K - 128, NC = K / 64;
  __shared__ __attribute__((aligned(16))) double As[16 * 130];  // the strip's rows of B
  __shared__ __attribute__((aligned(16))) double Xs[16 * ALD];  // the update's A operand: [P_(j-1) rows |] X_g
  __shared__ int poll_fail;
  // The strip's eight column blocks over eight waves.  Waves s and s + 4 share SIMD s: blocks 7 - s and s, (8 - s) + (s + 1) = 9
  // k-blocks of MFMAs per SIMD, the split the four-wave strip kernel has per wave.  (Which wave computes a tile does not enter its
  // arithmetic.)
  const int jb = wave < 4 ? 7 - wave : wave - 4;
  // ---- loads: the strip's operands, the update's old C values and, for K = 256, its rows of column j - 1
  double2_t stage[2], sprev[2];
  double2_t mt[8][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) stage[i] = *reinterpret_cast<const double2_t*>(B + ((long)rs * 16 + 2 * wave + i) * g.ld + 2 * lane);
  {
    const double* mp = minv + (long)(jb * 8) * 256 + 2 * lane;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      if (kb <= jb) {
        mt[kb][0] = *reinterpret_cast<const double2_t*>(mp + 256 * kb);
        mt[kb][1] = *reinterpret_cast<const double2_t*>(mp + 256 * kb + 128);
      }
    }
  }
  if constexpr (K == 256) {
    const double* pp = g.Pprev + z * g.sK;
#pragma unroll
    for (int i = 0; i < 2; ++i) sprev[i] = *reinterpret_cast<const double2_t*>(pp + ((long)rs * 16 + 2 * wave + i) * g.ld + 2 * lane);
  }
  const int cb = wave;  // this wave's tile column of the 128 x 128 tile row it updates
  double* out = C + ((long)rs * 16 + q) * g.ld + cb * 16 + n;
  double cold[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) cold[r] = out[(long)(4 * r) * g.ld];
  if (tid == 0) poll_fail = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) *reinterpret_cast<double2_t*>(As + (2 * wave + i) * 130 + 2 * lane) = stage[i];
  if constexpr (K == 256) {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<double2_t*>(Xs + (2 * wave + i) * ALD + 2 * lane) = sprev[i];
  }
  __syncthreads();  // (every wave's rows have left memory: the strip may store to them in place)
  // ---- the strip: X_g = B_g M^T, one column block per wave, four partial accumulators (one per MFMA step of a k-block)
  const bool producer = rs < FUSE_PRODUCERS;
  const double4_t zero4 = {0.0, 0.0, 0.0, 0.0};
  {
    double4_t ps[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      if (kb <= jb) {
        const double2_t a0 = *reinterpret_cast<const double2_t*>(As + n * 130 + 16 * kb + 4 * q);
        const double2_t a1 = *reinterpret_cast<const double2_t*>(As + n * 130 + 16 * kb + 4 * q + 2);
        ps[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, mt[kb][0].x, ps[0], 0, 0, 0);
        ps[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, mt[kb][0].y, ps[1], 0, 0, 0);
        ps[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, mt[kb][1].x, ps[2], 0, 0, 0);
        ps[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, mt[kb][1].y, ps[3], 0, 0, 0);
      }
    }
    const double4_t x = (ps[0] + ps[1]) + (ps[2] + ps[3]);
    // D layout: lane holds X[q + 4 r][16 jb + n]
    double* xo = B + ((long)rs * 16 + q) * g.ld + 16 * jb + n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xo[(long)(4 * r) * g.ld] = x[r];
      Xs[(q + 4 * r) * ALD + KOFF + 16 * jb + n] = x[r];
    }
    if (rs < g.lsw_blocks) {
      double* sw = lsw + (long)(rs >> 3) * 16384;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i0 = minv_index(16 * (rs & 7) + q + 4 * r, 16 * jb + n);
        if (producer) store_sc1(sw + i0, x[r]);  // read by the other workgroups of THIS launch: write-through
        else sw[i0] = x[r];                      // (the second block: read by the next column's launch)
      }
    }
  }
  // ---- K = 256: the update's first 128 k -- column j - 1's rows, both operands written by earlier launches -- run in the shadow
  // of the hand-off (same accumulators, same order as in syrk_thin_kernel: chunks 0 and 1 first)
  double4_t p[4] = {zero4, zero4, zero4, zero4};
  auto chunk = [&](int c, const double2_t (&bt)[4][2]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const double2_t a0 = *reinterpret_cast<const double2_t*>(Xs + n * ALD + 64 * c + 16 * kb + 4 * q);
      const double2_t a1 = *reinterpret_cast<const double2_t*>(Xs + n * ALD + 64 * c + 16 * kb + 4 * q + 2);
      p[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.x, bt[kb][0].x, p[0], 0, 0, 0);
      p[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0.y, bt[kb][0].y, p[1], 0, 0, 0);
      p[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.x, bt[kb][1].x, p[2], 0, 0, 0);
      p[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1.y, bt[kb][1].y, p[3], 0, 0, 0);
    }
  };
  double2_t bo[2][4][2];
  if constexpr (K == 256) {
    const double* bp = g.lsw_prev + z * g.sL + (long)(cb * 8) * 256 + 2 * lane;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        bo[c][kb][0] = *reinterpret_cast<const double2_t*>(bp + 256 * (4 * c + kb));
        bo[c][kb][1] = *reinterpret_cast<const double2_t*>(bp + 256 * (4 * c + kb) + 128);
      }
  }
  // ---- the producers publish: every storing wave drains its stores, the workgroup meets, ONE lane raises the workgroup's flag
  unsigned* flags = g.flags + z * FUSE_FLAG_WORDS;
  if (producer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // (also: the X part of Xs is complete)
  if (producer && tid == 0) __hip_atomic_store(flags + rs, g.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if constexpr (K == 256) {
    chunk(0, bo[0]);
    chunk(1, bo[1]);
  }
  // ---- everybody waits for the eight flags: lanes 0..7 of wave 0 poll one word each
  if (wave == 0) {
    bool ok = true;
    if (lane < FUSE_PRODUCERS) {
      long spins = 0;
      const long limit = 1L << g.poll_log2;
      while (__hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.tag) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > limit) {
          ok = false;
          break;
        }
      }
    }
    if (__any(!ok) && lane == 0) {
      poll_fail = 1;
      atomicMin(g.info + z * g.sinfo, SIGNAL_TIMEOUT_INFO);
    }
  }
  __syncthreads();
  if (poll_fail) return;  // (the evaluation is lost and reported; nothing stale is read)
  // ---- the last 128 k: X of tile row j + 1 as the B operand, fresh from the producers (sc1 loads)
  {
    double2_t bn[2][4][2];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(lsw, 0, 16384 * 8, 0x00020000);
    const int base = ((cb * 8) * 256 + 2 * lane) * 8;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        bn[c][kb][0] = load_sc1_b128(rsrc, base + 256 * 8 * (4 * c + kb));
        bn[c][kb][1] = load_sc1_b128(rsrc, base + 256 * 8 * (4 * c + kb) + 128 * 8);
      }
    chunk(NC - 2, bn[0]);
    chunk(NC - 1, bn[1]);
  }
  const double4_t sacc = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(long)(4 * r) * g.ld] = cold[r] - sacc[r];
}

hipError_t launch_strip_thin(const double* minv, double* B, const double* Pprev, double* C, long ld, int m, int k, double* lsw,
                             const double* lsw_prev, int lsw_blocks, unsigned* flags, unsigned tag, int* info, int poll_log2,
                             hipStream_t stream, const Batch* bt) {
  if (m < 128 || m % 16 || lsw == nullptr || lsw_blocks < FUSE_PRODUCERS || !(k == 128 || (k == 256 && Pprev && lsw_prev)))
    return hipErrorInvalidValue;
  FusedArgs g;
  g.minv = minv; g.B = B; g.Pprev = Pprev; g.C = C; g.ld = ld; g.sK = bt ? bt->sK : 0;
  g.lsw = lsw; g.lsw_prev = lsw_prev; g.sL = bt ? bt->sdinv : 0; g.lsw_blocks = lsw_blocks;
  g.flags = flags; g.tag = tag; g.info = info; g.sinfo = bt ? bt->sinfo : 0; g.poll_log2 = poll_log2;
  const dim3 grid(m / 16, 1, bt ? bt->nb : 1);
  if (k == 128) strip_thin_kernel<128><<<grid, 512, 0, stream>>>(g);
  else strip_thin_kernel<256><<<grid, 512, 0, stream>>>(g);
  return hipGetLastError();
}

}  // namespace migp
