// Dense "tail" GEMMs: the 1x1 convolutions of the Discriminator's dz / dxz stacks (reference mnist.py:115-117,124-127:
// Conv2d(512,512,1), Conv2d(1024,1024,1) on 1x1 maps) and their data gradients.
//
//   out[m][n] = epilogue( sum_k a[m][k] * b[n][k] )        a: activations [M][lda], b: packed weights [N][ldb]
//
// M is the batch (512 or 1024 rows), so a 64x64 tiling has at most 256 tiles: ONE block per CU.  The implicit-GEMM
// kernel (gconv.hip) hides its barrier / LDS / MFMA-dependency latencies behind the other blocks resident on the CU; a
// lone block runs its 16-MFMA dependency chain per k-tile at ~50 % of the MFMA rate (measured 0.85 us per k-tile against
// 0.43), and splitting K over the grid to get co-resident blocks pays the gain back in slab traffic and hand-off
// latency (measured: S = 1, 2, 4 all 27.6-28.2 us on 1024^3).  Here the co-resident waves come from the block itself:
// DKG = 4 groups of 4 waves (1024 threads), every group a complete 2x2-wave 64x64 tile pipeline (its own double-buffered
// LDS tiles and staging registers) over a quarter of K, so that every SIMD holds four independent MFMA chains; the four
// partial tiles meet in LDS (fixed order: deterministic) and group 0 runs the epilogue.  Same fragment layout, same
// fp32 MFMA (v_mfma_f32_32x32x2_f32, exact k-ordered fma chain inside a group) as gconv.hip.
#include "ali_common.h"
#include <string.h>

namespace ali {

constexpr int DK = 32;        // k-tile
constexpr int DLD = DK + 4;   // LDS row pitch (floats): 16-B aligned, ds_read_b128 conflict-free
constexpr int DKG = 4;        // K-groups per block
constexpr int DT = 64;        // tile edge

struct DenseDesc {
  const float* a;
  const float* b;
  float* out;
  AliEpilogue ep;
  int M, N, K, lda, ldb, ldo;
  int tm, tn;         // tiles along M and N
  int rows_per_img;   // rows that share a Dropout2d mask row (H*W of the 1x1 convolution's map)
  unsigned a_bytes, b_bytes;
};

__global__ __launch_bounds__(256 * DKG, 1) void dense_gemm_kernel(const DenseDesc d) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];     // [DKG][2][2*DT*DLD]: per group, per buffer: A rows then B rows
  const int tid = threadIdx.x, grp = tid >> 8, t = tid & 255, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // Tile of this block.  Slot u of the (1-D) grid runs on XCD u % 8 (scratch/ub/place.hip); each XCD has its own 4 MB
  // L2 and both operands are re-read by every tile of their row / column.  The tiles are numbered in panels of `pw`
  // n-tiles (n fastest inside a panel) and every XCD takes a contiguous range of that numbering: a compact rectangle
  // (4 x 8 tiles of the 16 x 16 for 1024^3: 3 MB of operand rows) instead of a stripe through the whole matrix --
  // otherwise the launch is bound by L2 misses (134 MB of operand reads for 8 MB of operands), not by latency.
  const int T = d.tm * d.tn;
  int L = blockIdx.x;
  if ((T & 7) == 0) L = (L & 7) * (T >> 3) + (L >> 3);
  const int pw = d.tn < 8 ? d.tn : 8;
  const int pan = L / (pw * d.tm), rem = L - pan * (pw * d.tm);
  const int pwl = min(pw, d.tn - pan * pw);                  // (the last panel may be narrower)
  const int mt = rem / pwl, nt = pan * pw + (rem - mt * pwl);
  const int m0 = mt * DT, n0 = nt * DT;
  float* gs = dsm + grp * (2 * 2 * DT * DLD);
  const int nkt = d.K / DK;
  const int per = (nkt + DKG - 1) / DKG;          // iterations of EVERY group (block-wide barriers): a group whose range
  const int q0 = grp * per;                        // ends early multiplies zero tiles (out-of-range loads return 0)
  const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc((void*)d.a, 0, d.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)d.b, 0, d.b_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  const int c4 = t & 7, r0 = t >> 3;               // this thread's 16-byte chunk and first row (second: + 32)
  unsigned aoff[2], boff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + r0 + 32 * i, n = n0 + r0 + 32 * i;
    aoff[i] = m < d.M ? (unsigned)(((long long)m * d.lda + c4 * 4) * 4) : OOB;
    boff[i] = n < d.N ? (unsigned)(((long long)n * d.ldb + c4 * 4) * 4) : OOB;
  }
  f32x4 sa[2][2], sb[2][2];                         // two staging sets: the tiles i+1 and i+2 are in flight
  auto fetch = [&](int i, int set) {
    const int q = q0 + i;
    const bool live = i < per && q < nkt;
    const unsigned ko = (unsigned)q * (DK * 4);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const unsigned oa = (live && aoff[x] != OOB) ? aoff[x] + ko : OOB;
      const unsigned ob = (live && boff[x] != OOB) ? boff[x] + ko : OOB;
      sa[set][x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra_, (int)oa, 0, 0));
      sb[set][x] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb_, (int)ob, 0, 0));
    }
  };
  auto stash = [&](int buf, int set) {
    float* A = gs + buf * (2 * DT * DLD);
    float* B = A + DT * DLD;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      *reinterpret_cast<f32x4*>(A + (r0 + 32 * x) * DLD + c4 * 4) = sa[set][x];
      *reinterpret_cast<f32x4*>(B + (r0 + 32 * x) * DLD + c4 * 4) = sb[set][x];
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lrow = lane & 31, lh = lane >> 5;

  fetch(0, 0);
  fetch(1, 1);
  stash(0, 0);
  __syncthreads();
  for (int i = 0; i < per; ++i) {
    const int buf = i & 1;
    // tile i is multiplied out of LDS[buf]; tile i+2 is fetched into the set tile i came from; tile i+1, fetched an
    // iteration ago, goes to LDS[buf^1] behind the MFMAs
    if (buf == 0) fetch(i + 2, 0); else fetch(i + 2, 1);
    const float* A = gs + buf * (2 * DT * DLD) + (wm * 32 + lrow) * DLD + lh * 4;
    const float* B = gs + buf * (2 * DT * DLD) + DT * DLD + (wn * 32 + lrow) * DLD + lh * 4;
#pragma unroll
    for (int kg = 0; kg < DK / 8; ++kg) {
      const f32x4 fa = *reinterpret_cast<const f32x4*>(A + kg * 8);
      const f32x4 fb = *reinterpret_cast<const f32x4*>(B + kg * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], acc, 0, 0, 0);
    }
    if (buf == 0) stash(1, 1); else stash(0, 0);
    __syncthreads();
  }
  // the four groups' partial tiles -> group 0, through LDS (all operand tiles are dead behind the last barrier)
  float* red = dsm;                                 // [DKG-1][4 waves][16][64]
  if (grp > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(((grp - 1) * 4 + wave) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (grp > 0) return;
#pragma unroll
  for (int g = 0; g < DKG - 1; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[((g * 4 + wave) * 16 + r) * 64 + lane];
  const AliEpilogue& ep = d.ep;
  const int n = n0 + wn * 32 + lrow;
  if (n >= d.N) return;
  const float bias = ep.bias ? ep.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m >= d.M) continue;
    float v = apply_act(acc[r] + bias, ep.act, ep.slope);
    if (ep.mask) v *= ep.mask[(long long)(m / d.rows_per_img) * ep.mask_ld + n];
    if (ep.dact_y) v *= act_grad_from_output(ep.dact_y[(long long)m * d.ldo + n], ep.dact, ep.dslope);
    d.out[(long long)m * d.ldo + n] = v;
  }
}

// (gconv.hip) a 1x1 stride-1 convolution / data gradient whose 64x64 tiling gives at most one block per CU
int dense_gemm_launch(const float* a, const float* b, float* out, const AliEpilogue& ep, int M, int N, int K, int lda,
                      int ldb, int ldo, int rows_per_img, hipStream_t stream) {
  static bool attr_set = false;
  const size_t lds = (size_t)DKG * 2 * 2 * DT * DLD * sizeof(float);
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dense_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      set_error("dense_gemm: cannot reserve %zu bytes of LDS", lds);
      return ALI_ERR_LAUNCH;
    }
    attr_set = true;
  }
  DenseDesc d;
  memset(&d, 0, sizeof(d));
  d.a = a; d.b = b; d.out = out; d.ep = ep;
  d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldo = ldo; d.rows_per_img = rows_per_img;
  d.a_bytes = (unsigned)((((long long)M - 1) * lda + K) * 4);
  d.b_bytes = (unsigned)((((long long)N - 1) * ldb + K) * 4);
  d.tm = (M + DT - 1) / DT;
  d.tn = (N + DT - 1) / DT;
  hipLaunchKernelGGL(dense_gemm_kernel, dim3(d.tm * d.tn), dim3(256 * DKG), lds, stream, d);
  return check_launch("dense_gemm_kernel");
}

}  // namespace ali
