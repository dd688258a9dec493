// Weight-gradient implicit GEMM (reference: autograd of nn.Conv2d /
// nn.ConvTranspose2d / nn.Linear inside loss.backward(), image_scms/mnist.py:229,
// 235,240 and copies).
//
//   dWp[(tap, gc)][dc] = sum_{pix=(b,p,q)}  X[b, p*st-pad+r, q*st-pad+s, gc] * DY[b,p,q,dc]
//
// X is the "gathered" operand (conv input / ConvT output-grad), DY the "dense"
// one (conv output-grad / ConvT input).  GEMM rows m' = tap*Cg + gc, columns dc,
// reduction over pixels: the reduction index is the slow (row) index of both
// NHWC operands, so tiles are staged k-major in LDS ([pixel][channel]) and MFMA
// fragments are 32 consecutive floats (conflict-free ds_read_b32).
// Split over the pixel range (grid.z) with a deterministic slab reduction that
// also scatters to the reference parameter layout.
#include "ali_common.h"
#include <string.h>
#include <algorithm>
#include <vector>

namespace ali {

constexpr int WBK = 16;  // pixels per k-tile

struct WDesc {
  const float* x;   // gathered  [B,H,W,Cg]
  const float* dy;  // dense     [B,P,Q,Cd]
  float* dst;
  float* ws;
  int B, H, W, Cg, P, Q, Cd;
  int R, S, stride, pad;
  int Cg_log, Cd_log;
  long long s_dc, s_gc, s_tap;
  int Mtot;            // R*S*Cg
  int npix, pix_per_split, splitk;
  long long slab;      // floats between consecutive slabs: Mtot*Cd plus a pad that breaks the power-of-two stride
  float* db;          // optional: db[dc] = sum_pix dy[pix][dc] (bias gradient), fused into the m-tile-0 blocks
  float* dbws;        // [splitk][Cd] slabs when splitk > 1
  const int* pixtab;  // optional [npix][2] (ali_wgrad_pixtab)
  const _Float16* x16;   // fp16 twins of x / dy (same shapes), or null: the F16 == 2 kernels read these
  const _Float16* dy16;
  int ldd;            // floats between consecutive pixels of dy (Cd, or more when dy is a column range of wider rows)
};

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VECA, bool VECB>
__global__ __launch_bounds__(256) void wgrad_kernel(const WDesc d) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int A4 = BM / 4, B4 = BN / 4;          // float4 per pixel row
  constexpr int AP = (WBK * A4 + 255) / 256, BP = (WBK * B4 + 255) / 256;
  constexpr int AROWS = 256 / A4, BROWS = 256 / B4;  // pixel rows covered per pass

  __shared__ __attribute__((aligned(16))) float As[2][WBK * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][WBK * LDB];
  __shared__ long long s_rowdst[BM];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int pix_begin = blockIdx.z * d.pix_per_split;
  const int pix_end = min(d.npix, pix_begin + d.pix_per_split);
  const int PQ = d.P * d.Q;

  for (int r = t; r < BM; r += 256) {
    const int m = m0 + r;
    long long off = -1;
    if (m < d.Mtot) {
      const int tap = m / d.Cg, gc = m - tap * d.Cg;
      if (gc < d.Cg_log) off = gc * d.s_gc + tap * d.s_tap;
    }
    s_rowdst[r] = off;
  }

  // A loader: element column(s) fixed per thread
  const int a4 = t % A4, arow0 = t / A4;
  int a_dh[4], a_dw[4], a_gc[4];
  bool a_ok[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = m0 + a4 * 4 + e;
    a_ok[e] = m < d.Mtot;
    const int mm = a_ok[e] ? m : 0;
    const int tap = mm / d.Cg;
    a_gc[e] = mm - tap * d.Cg;
    a_dh[e] = tap / d.S - d.pad;
    a_dw[e] = tap % d.S - d.pad;
  }
  const int b4 = t % B4, brow0 = t / B4;
  __syncthreads();

  f32x4 ra[AP], rb[BP];
  auto load_tile = [&](int pix0) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      const int pix = pix0 + arow0 + i * AROWS;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (arow0 + i * AROWS < WBK && pix < pix_end) {
        const int b = pix / PQ, rem = pix - b * PQ;
        const int p = rem / d.Q, q = rem - p * d.Q;
        const int h0 = p * d.stride, w0 = q * d.stride;
        if (VECA) {
          const int ih = h0 + a_dh[0], iw = w0 + a_dw[0];
          if (a_ok[0] && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
            v = *reinterpret_cast<const f32x4*>(d.x + ((long long)(b * d.H + ih) * d.W + iw) * d.Cg + a_gc[0]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ih = h0 + a_dh[e], iw = w0 + a_dw[e];
            if (a_ok[e] && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
              v[e] = d.x[((long long)(b * d.H + ih) * d.W + iw) * d.Cg + a_gc[e]];
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const int pix = pix0 + brow0 + j * BROWS;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (brow0 + j * BROWS < WBK && pix < pix_end) {
        const int n = n0 + b4 * 4;
        const float* src = d.dy + (long long)pix * d.Cd + n;
        if (VECB) {
          if (n < d.Cd) v = *reinterpret_cast<const f32x4*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < d.Cd) v[e] = src[e];
        }
      }
      rb[j] = v;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AP; ++i)
      if (arow0 + i * AROWS < WBK) *reinterpret_cast<f32x4*>(&As[buf][(arow0 + i * AROWS) * LDA + a4 * 4]) = ra[i];
#pragma unroll
    for (int j = 0; j < BP; ++j)
      if (brow0 + j * BROWS < WBK) *reinterpret_cast<f32x4*>(&Bs[buf][(brow0 + j * BROWS) * LDB + b4 * 4]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (pix_begin < pix_end) {
    load_tile(pix_begin);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    const int lcol = lane & 31, lh = lane >> 5;
    for (int pix0 = pix_begin; pix0 < pix_end; pix0 += WBK) {
      const bool has_next = pix0 + WBK < pix_end;
      if (has_next) load_tile(pix0 + WBK);
      const float* Ab = &As[buf][lh * LDA + wm * WM + lcol];
      const float* Bb = &Bs[buf][lh * LDB + wn * WN + lcol];
#pragma unroll
      for (int ks = 0; ks < WBK / 2; ++ks) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = Ab[ks * 2 * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bb[ks * 2 * LDB + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (has_next) store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  const bool partial = d.splitk > 1;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 32 + (lane & 31);
    if (n >= (partial ? d.Cd : d.Cd_log)) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = acc[i][j][r];
        if (partial) {
          const int m = m0 + row;
          if (m < d.Mtot) d.ws[(long long)blockIdx.z * d.slab + (long long)m * d.Cd + n] = v;
        } else {
          const long long off = s_rowdst[row];
          if (off >= 0) d.dst[off + n * d.s_dc] = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fast path (both channel strides % 4 == 0): same loop discipline as gconv.hip -- fp32 MFMA shares the vector issue
// path, so VALU work per k-tile is kept minimal (buffer loads with hardware range check, float-reciprocal pixel
// decode) and every LDS / global instruction sits behind an individual MFMA.  k-tile = 32 pixels.
constexpr int WBK2 = 32;

__device__ __forceinline__ void w_divmod(int m, int dv, float rcp, int& q, int& r) {
  q = (int)((float)m * rcp);
  r = m - q * dv;
  if (r < 0) { --q; r += dv; }
  else if (r >= dv) { ++q; r -= dv; }
}

// F16 (needs TAB): the fp16-MFMA variant (BASELINE config 5).  Both operands are rounded to fp16 on their way into
// LDS, where they keep the [pixel][channel] order they have in memory (8-byte stores); v_mfma_f32_32x32x16_f16 wants 8
// consecutive PIXELS per lane for one channel, i.e. the transposed image, which ds_read_b64_tr_b16 delivers for free:
// per 16-lane group it reads a 4-pixel x 16-channel block and hands lane i channel i of the 4 pixels (verified on the
// hardware by scratch/ub/tr16.hip).  Row pitch = channels + 32 halves: the 4 pixel rows of a block start 16 banks apart
// (pitch/4 = 16 or 48 mod 64), so a half-wave's 32 x 8 bytes cover all 64 banks once.  fp32 accumulation, fp32 slabs.
using wf16x4 = __attribute__((ext_vector_type(4))) _Float16;
using wf16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef short ws4v __attribute__((__vector_size__(4 * sizeof(short))));

// F16 == 2: both operands come from their fp16 twins in memory (WDesc.x16 / dy16, left by the fp16 launches that produced
// the tensors): a 16-byte gather carries 8 channels, no conversion work, half the bytes.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool TAB, int F16 = 0>
__device__ __forceinline__ void wgrad_fast_body(const WDesc& d, const unsigned x_bytes, const unsigned dy_bytes,
                                                const int bx_, const int by_, const int bz_) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int A4 = BM / 4, B4 = BN / 4;            // float4 per pixel row
  constexpr int AROWS = 256 / A4, BROWS = 256 / B4;  // pixel rows covered per pass
  constexpr int AP = WBK2 / AROWS, BP = WBK2 / BROWS;
  constexpr int NL = AP + BP;
  constexpr int NMF = (WBK2 / 2) * TM * TN;
  static_assert(AP >= 1 && BP >= 1, "tile");

  __shared__ __attribute__((aligned(16))) float As[2][WBK2 * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][WBK2 * LDB];
  __shared__ long long s_rowdst[BM];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int m0 = bx_ * BM, n0 = by_ * BN;
  const int pix_begin = bz_ * d.pix_per_split;
  const int pix_end = min(d.npix, pix_begin + d.pix_per_split);
  const int PQ = d.P * d.Q;
  const float rPQ = 1.0f / (float)PQ, rQ = 1.0f / (float)d.Q;

  for (int r = t; r < BM; r += 256) {
    const int m = m0 + r;
    long long off = -1;
    if (m < d.Mtot) {
      const int tap = m / d.Cg, gc = m - tap * d.Cg;
      if (gc < d.Cg_log) off = gc * d.s_gc + tap * d.s_tap;
    }
    s_rowdst[r] = off;
  }
  // this thread's A column: tap (dh, dw) and gathered channel are fixed
  const int a4 = t % A4, arow0 = t / A4;
  const int mA = m0 + a4 * 4;
  const bool a_ok = mA < d.Mtot;
  const int tapA = a_ok ? mA / d.Cg : 0;
  const int gcA = mA - tapA * d.Cg;
  const int dhA = tapA / d.S - d.pad, dwA = tapA % d.S - d.pad;
  const int b4 = t % B4, brow0 = t / B4;
  const int nB = n0 + b4 * 4;
  const bool b_ok = nB < d.Cd;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)d.x, 0, x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)d.dy, 0, dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  __syncthreads();

  f32x4 ra[AP], rb[BP];
  const bool do_db = d.db != nullptr && bx_ == 0;
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
  // TAB: the pixel's (byte offset, packed position) comes from the per-geometry table, fetched one k-tile ahead of the
  // gather that uses it (entries past pix_end read as 0 = a position outside the map); this thread's tap adds constants
  using i32x2 = __attribute__((ext_vector_type(2))) int;
  i32x2 te[AP];
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)d.pixtab, 0, TAB ? (unsigned)pix_end * 8u : 0u, 0x00020000);
  const int tapoffA = ((dhA * d.W + dwA) * d.Cg + gcA) * 4;      // (dhA, dwA) carry the -pad; the table does not
  auto load_te = [&](int pix0, int i) {
    const int pix = pix0 + arow0 + i * AROWS;
    te[i] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(rt, pix * 8, 0, 0));
  };
  auto load_a = [&](int pix0, int i) {
    unsigned off = OOB;
    if (TAB) {
      // entry: x = byte offset of x[b, p*stride, q*stride, 0]; y = (p*stride + 0x4000) | (q*stride + 0x4000) << 16
      const int ih = (te[i].y & 0xffff) - 0x4000 + dhA, iw = (int)((unsigned)te[i].y >> 16) - 0x4000 + dwA;
      if (a_ok && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W) off = (unsigned)(te[i].x + tapoffA);
    } else {
      const int pix = pix0 + arow0 + i * AROWS;
      if (a_ok && pix < pix_end) {
        int b, rem, p, q;
        w_divmod(pix, PQ, rPQ, b, rem);
        w_divmod(rem, d.Q, rQ, p, q);
        const int ih = p * d.stride + dhA, iw = q * d.stride + dwA;
        if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
          off = (unsigned)(((b * d.H + ih) * d.W + iw) * d.Cg + gcA) * 4u;
      }
    }
    ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0));
  };
  auto load_b = [&](int pix0, int j) {
    const int pix = pix0 + brow0 + j * BROWS;
    const unsigned off = (b_ok && pix < pix_end) ? (unsigned)(pix * d.ldd + nB) * 4u : OOB;
    rb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, (int)off, 0, 0));
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&As[buf][(arow0 + i * AROWS) * LDA + a4 * 4]) = ra[i];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      *reinterpret_cast<f32x4*>(&Bs[buf][(brow0 + j * BROWS) * LDB + b4 * 4]) = rb[j];
      if (do_db) dbacc += rb[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (F16 == 2 && TAB) {
    if (pix_begin < pix_end) {
      constexpr int PA = BM + 32, PB = BN + 32;                 // halves per pixel row (see the F16 == 1 loop below)
      constexpr int A8 = BM / 8, B8 = BN / 8;                   // 16-byte gathers per pixel row
      constexpr int AROWS8 = 256 / A8, BROWS8 = 256 / B8;
      constexpr int AP8 = WBK2 / AROWS8, BP8 = WBK2 / BROWS8;
      static_assert(AP8 >= 1 && BP8 >= 1, "fp16-in-memory wgrad needs tiles of at least 64 x 64");
      _Float16* A16 = reinterpret_cast<_Float16*>(&As[0][0]);   // [2][32][PA]
      _Float16* B16 = reinterpret_cast<_Float16*>(&Bs[0][0]);   // [2][32][PB]
      const __amdgpu_buffer_rsrc_t rx16 = __builtin_amdgcn_make_buffer_rsrc((void*)d.x16, 0, x_bytes / 2, 0x00020000);
      const __amdgpu_buffer_rsrc_t ry16 = __builtin_amdgcn_make_buffer_rsrc((void*)d.dy16, 0, dy_bytes / 2, 0x00020000);
      const int a8 = t % A8, arow8 = t / A8;
      const int mA8 = m0 + a8 * 8;                              // 8 consecutive gathered channels of one tap (8 | Cg)
      const bool a_ok8 = mA8 < d.Mtot;
      const int tapA8 = a_ok8 ? mA8 / d.Cg : 0;
      const int gcA8 = mA8 - tapA8 * d.Cg;
      const int dhA8 = tapA8 / d.S - d.pad, dwA8 = tapA8 % d.S - d.pad;
      const int tapoffA8 = ((dhA8 * d.W + dwA8) * d.Cg + gcA8) * 2;
      const int b8 = t % B8, brow8 = t / B8;
      const int nB8 = n0 + b8 * 8;
      const bool b_ok8 = nB8 < d.Cd;
      i32x2 te8[AP8];
      auto load_te8 = [&](int pix0, int i) {
        const int pix = pix0 + arow8 + i * AROWS8;
        te8[i] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(rt, pix * 8, 0, 0));
      };
      auto fetch8 = [&](int pix0, f32x4* sa, f32x4* sb) {
#pragma unroll
        for (int i = 0; i < AP8; ++i) {
          unsigned off = OOB;
          const int ih = (te8[i].y & 0xffff) - 0x4000 + dhA8, iw = (int)((unsigned)te8[i].y >> 16) - 0x4000 + dwA8;
          if (a_ok8 && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W) off = (unsigned)((te8[i].x >> 1) + tapoffA8);
          sa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx16, (int)off, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < BP8; ++j) {
          const int pix = pix0 + brow8 + j * BROWS8;
          const unsigned off = (b_ok8 && pix < pix_end) ? (unsigned)(pix * d.Cd + nB8) * 2u : OOB;
          sb[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ry16, (int)off, 0, 0));
        }
      };
      f32x4 dbacc8 = {0.f, 0.f, 0.f, 0.f}, dbacc8b = {0.f, 0.f, 0.f, 0.f};   // bias gradient: channels nB8..+3, +4..+7
      auto store8 = [&](int buf, const f32x4* sa, const f32x4* sb) {
#pragma unroll
        for (int i = 0; i < AP8; ++i)
          *reinterpret_cast<f32x4*>(&A16[(buf * WBK2 + arow8 + i * AROWS8) * PA + a8 * 8]) = sa[i];
#pragma unroll
        for (int j = 0; j < BP8; ++j) {
          *reinterpret_cast<f32x4*>(&B16[(buf * WBK2 + brow8 + j * BROWS8) * PB + b8 * 8]) = sb[j];
          if (do_db) {
            const wf16x8 h = __builtin_bit_cast(wf16x8, sb[j]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { dbacc8[e] += (float)h[e]; dbacc8b[e] += (float)h[4 + e]; }
          }
        }
      };
      f32x4 sa0[AP8], sb0[BP8], sa1[AP8], sb1[BP8];
#pragma unroll
      for (int i = 0; i < AP8; ++i) load_te8(pix_begin, i);
      fetch8(pix_begin, sa0, sb0);
#pragma unroll
      for (int i = 0; i < AP8; ++i) load_te8(pix_begin + WBK2, i);
      fetch8(pix_begin + WBK2, sa1, sb1);
#pragma unroll
      for (int i = 0; i < AP8; ++i) load_te8(pix_begin + 2 * WBK2, i);
      store8(0, sa0, sb0);
      __syncthreads();
      int buf = 0;
      const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1, th = lane >> 5;
      const int a_col = wm * WM + 16 * tg + 4 * tp, b_col = wn * WN + 16 * tg + 4 * tp;
      auto frag = [&](const _Float16* img, int pitch, int col, int k0) -> wf16x8 {
        const _Float16* p0 = img + (k0 + 8 * th + tq) * pitch + col;
        const ws4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws4v*)p0);
        const ws4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws4v*)(p0 + 4 * pitch));
        const wf16x4 l4 = __builtin_bit_cast(wf16x4, lo), h4 = __builtin_bit_cast(wf16x4, hi);
        return wf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
      };
      auto iter8 = [&](int pix0, f32x4* fa_, f32x4* fb_, const f32x4* oa, const f32x4* ob) {
        fetch8(pix0 + 2 * WBK2, fa_, fb_);
#pragma unroll
        for (int i = 0; i < AP8; ++i) load_te8(pix0 + 3 * WBK2, i);
        const _Float16* Ai = A16 + buf * WBK2 * PA;
        const _Float16* Bi = B16 + buf * WBK2 * PB;
#pragma unroll
        for (int ks = 0; ks < WBK2 / 16; ++ks) {
          wf16x8 ha[TM], hb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) ha[i] = frag(Ai, PA, a_col + i * 32, ks * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j) hb[j] = frag(Bi, PB, b_col + j * 32, ks * 16);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[i], hb[j], acc[i][j], 0, 0, 0);
        }
        store8(buf ^ 1, oa, ob);
        __syncthreads();
        buf ^= 1;
      };
      for (int pix0 = pix_begin; pix0 < pix_end;) {
        iter8(pix0, sa0, sb0, sa1, sb1);
        pix0 += WBK2;
        if (pix0 >= pix_end) break;
        iter8(pix0, sa1, sb1, sa0, sb0);
        pix0 += WBK2;
      }
      if (do_db) {   // column sums of this thread's 8 channels over its pixel rows -> the fp32 path's fold layout
        __syncthreads();
        float* red = &As[0][0];                     // [BROWS8][BN] floats <= tile size
        *reinterpret_cast<f32x4*>(&red[brow8 * BN + b8 * 8]) = dbacc8;
        *reinterpret_cast<f32x4*>(&red[brow8 * BN + b8 * 8 + 4]) = dbacc8b;
        __syncthreads();
        if (t < BN) {
          float sum = 0.f;
          for (int r = 0; r < BROWS8; ++r) sum += red[r * BN + t];
          const int n = n0 + t;
          if (n < d.Cd_log) {
            if (d.splitk > 1) d.dbws[(long long)bz_ * d.Cd + n] = sum;
            else d.db[n] = sum;
          }
        }
        __syncthreads();
      }
    }
  } else if constexpr (F16 == 1 && TAB) {
    if (pix_begin < pix_end) {
      constexpr int PA = BM + 32, PB = BN + 32;                 // halves per pixel row
      static_assert(PA * 2 <= LDA * 4 && PB * 2 <= LDB * 4, "fp16 rows fit the fp32 tile buffers");
      _Float16* A16 = reinterpret_cast<_Float16*>(&As[0][0]);   // [2][32][PA]
      _Float16* B16 = reinterpret_cast<_Float16*>(&Bs[0][0]);   // [2][32][PB]
      f32x4 ra1[AP], rb1[BP];                                    // second staging set: two k-tiles of gathers in flight
      auto cvt_store = [&](int buf, const f32x4* sa, const f32x4* sb) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          const wf16x4 v = {(_Float16)sa[i][0], (_Float16)sa[i][1], (_Float16)sa[i][2], (_Float16)sa[i][3]};
          *reinterpret_cast<wf16x4*>(&A16[(buf * WBK2 + arow0 + i * AROWS) * PA + a4 * 4]) = v;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
          const wf16x4 v = {(_Float16)sb[j][0], (_Float16)sb[j][1], (_Float16)sb[j][2], (_Float16)sb[j][3]};
          *reinterpret_cast<wf16x4*>(&B16[(buf * WBK2 + brow0 + j * BROWS) * PB + b4 * 4]) = v;
          if (do_db) dbacc += sb[j];
        }
      };
      auto fetch = [&](int pix0, f32x4* sa, f32x4* sb) {      // gathers of the tile at pix0 (te holds its table entries)
#pragma unroll
        for (int i = 0; i < AP; ++i) { load_a(pix0, i); sa[i] = ra[i]; }
#pragma unroll
        for (int j = 0; j < BP; ++j) { load_b(pix0, j); sb[j] = rb[j]; }
      };
      // (load_a / load_b write ra / rb; set 1 copies them -- register renaming, no instruction survives)
      f32x4 sa0[AP], sb0[BP];
#pragma unroll
      for (int i = 0; i < AP; ++i) load_te(pix_begin, i);
      fetch(pix_begin, sa0, sb0);
#pragma unroll
      for (int i = 0; i < AP; ++i) load_te(pix_begin + WBK2, i);
      fetch(pix_begin + WBK2, ra1, rb1);
#pragma unroll
      for (int i = 0; i < AP; ++i) load_te(pix_begin + 2 * WBK2, i);
      cvt_store(0, sa0, sb0);
      __syncthreads();
      int buf = 0;
      // transposed fragment reads: lane l = 16*grp + 4*q + p addresses pixel row q, channels 4p..4p+3 of its group's
      // 16-channel block; it receives channel (l & 15) of the block's 4 pixels.  grp & 1 selects the 16-channel half
      // of the 32-wide MFMA operand, grp >> 1 (= lane >> 5) the 8-pixel half of the 16-pixel k-step.
      const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1, th = lane >> 5;
      const int a_col = wm * WM + 16 * tg + 4 * tp, b_col = wn * WN + 16 * tg + 4 * tp;
      auto frag = [&](const _Float16* img, int pitch, int col, int k0) -> wf16x8 {
        const _Float16* p0 = img + (k0 + 8 * th + tq) * pitch + col;
        const ws4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws4v*)p0);
        const ws4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws4v*)(p0 + 4 * pitch));
        const wf16x4 l4 = __builtin_bit_cast(wf16x4, lo), h4 = __builtin_bit_cast(wf16x4, hi);
        return wf16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
      };
      auto iteration16 = [&](int pix0, f32x4* fa_, f32x4* fb_, const f32x4* oa, const f32x4* ob) {
        // tile pix0 is multiplied out of LDS[buf]; tile pix0 + 2*WBK2 is fetched into (fa_, fb_); tile pix0 + WBK2,
        // fetched an iteration ago into (oa, ob), goes to LDS[buf ^ 1]
        fetch(pix0 + 2 * WBK2, fa_, fb_);
#pragma unroll
        for (int i = 0; i < AP; ++i) load_te(pix0 + 3 * WBK2, i);
        const _Float16* Ai = A16 + buf * WBK2 * PA;
        const _Float16* Bi = B16 + buf * WBK2 * PB;
#pragma unroll
        for (int ks = 0; ks < WBK2 / 16; ++ks) {
          wf16x8 ha[TM], hb[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) ha[i] = frag(Ai, PA, a_col + i * 32, ks * 16);
#pragma unroll
          for (int j = 0; j < TN; ++j) hb[j] = frag(Bi, PB, b_col + j * 32, ks * 16);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[i], hb[j], acc[i][j], 0, 0, 0);
        }
        cvt_store(buf ^ 1, oa, ob);
        __syncthreads();
        buf ^= 1;
      };
      for (int pix0 = pix_begin; pix0 < pix_end;) {
        iteration16(pix0, sa0, sb0, ra1, rb1);
        pix0 += WBK2;
        if (pix0 >= pix_end) break;
        iteration16(pix0, ra1, rb1, sa0, sb0);
        pix0 += WBK2;
      }
    }
  } else if (pix_begin < pix_end) {
    if (TAB) {
#pragma unroll
      for (int i = 0; i < AP; ++i) load_te(pix_begin, i);
    }
#pragma unroll
    for (int i = 0; i < AP; ++i) load_a(pix_begin, i);
#pragma unroll
    for (int j = 0; j < BP; ++j) load_b(pix_begin, j);
    if (TAB) {
#pragma unroll
      for (int i = 0; i < AP; ++i) load_te(pix_begin + WBK2, i);
    }
    store_tile(0);
    __syncthreads();
    int buf = 0;
    const int lcol = lane & 31, lh = lane >> 5;
    float fa[2][TM], fb[2][TN];
    for (int pix0 = pix_begin; pix0 < pix_end; pix0 += WBK2) {
      const int nxt = pix0 + WBK2;   // rows >= pix_end load zeros (range-checked offsets)
      const float* Ac = &As[buf][lh * LDA + wm * WM + lcol];
      const float* Bc = &Bs[buf][lh * LDB + wn * WN + lcol];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[0][i] = Ac[i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = Bc[j * 32];
#pragma unroll
      for (int s = 0; s < NMF; ++s) {
        const int ks = s / (TM * TN), i = (s / TN) % TM, j = s % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        {
          constexpr int Q = NMF / 4;
          if (s < Q) {
#pragma unroll
            for (int x = 0; x < NL; ++x) {
              if ((x * Q) / NL != s) continue;
              if (x < AP) load_a(nxt, x); else load_b(nxt, x - AP);
            }
          }
          if (TAB && s >= Q && s < 2 * Q) {   // second quarter: table entries of the tile after next
#pragma unroll
            for (int x = 0; x < AP; ++x)
              if ((x * Q) / AP == s - Q) load_te(nxt + WBK2, x);
          }
        }
        {  // fragments of k-step ks+1: TM+TN single-float reads, one behind each MFMA of step ks
          const int sg = s - ks * TM * TN;
          if (ks + 1 < WBK2 / 2 && sg < TM + TN) {
            if (sg < TM) fa[(ks + 1) & 1][sg] = Ac[(ks + 1) * 2 * LDA + sg * 32];
            else fb[(ks + 1) & 1][sg - TM] = Bc[(ks + 1) * 2 * LDB + (sg - TM) * 32];
          }
          if (TM * TN < TM + TN && ks + 1 < WBK2 / 2 && sg == TM * TN - 1) {   // 1x1 wave tile: 2 reads behind 1 MFMA
#pragma unroll
            for (int x = TM * TN; x < TM + TN; ++x) {
              if (x < TM) fa[(ks + 1) & 1][x] = Ac[(ks + 1) * 2 * LDA + x * 32];
              else fb[(ks + 1) & 1][x - TM] = Bc[(ks + 1) * 2 * LDB + (x - TM) * 32];
            }
          }
        }
        {
          constexpr int Q = NMF / 4;
          const int sw = s - 3 * Q;
          if (sw >= 0) {
#pragma unroll
            for (int x = 0; x < NL; ++x) {
              if ((x * Q) / NL != sw) continue;
              if (x < AP) *reinterpret_cast<f32x4*>(&As[buf ^ 1][(arow0 + x * AROWS) * LDA + a4 * 4]) = ra[x];
              else {
                *reinterpret_cast<f32x4*>(&Bs[buf ^ 1][(brow0 + (x - AP) * BROWS) * LDB + b4 * 4]) = rb[x - AP];
                if (do_db) dbacc += rb[x - AP];
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      buf ^= 1;
    }
  }

  const bool partial = d.splitk > 1;
  if (do_db && F16 != 2) {   // fold the per-thread column sums over the BROWS row lanes (fixed order), then one store per column
    __syncthreads();
    float* red = &As[0][0];                       // BROWS x BN floats <= tile size
    *reinterpret_cast<f32x4*>(&red[brow0 * BN + b4 * 4]) = dbacc;
    __syncthreads();
    if (t < BN) {
      float sum = 0.f;
      for (int r = 0; r < BROWS; ++r) sum += red[r * BN + t];
      const int n = n0 + t;
      if (n < d.Cd_log) {
        if (partial) d.dbws[(long long)bz_ * d.Cd + n] = sum;
        else d.db[n] = sum;
      }
    }
  }
  if (partial) {
    // raw partial tile -> this split's slab; the slabs are summed by wgrad_reduce_tile_kernel / wgrad_fold_multi_kernel.
    // (Measured and dropped: the last-arriving block of a tile summing the slabs itself, as gconv.hip's split-K does --
    // store-ack, counter RMW, device-scope loads and the scattered stores are four dependent memory round trips at
    // the tail of the launch, which cost exactly what the second launch costs.)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 32 + (lane & 31);
      if (n >= d.Cd) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (m < d.Mtot) d.ws[(long long)bz_ * d.slab + (long long)m * d.Cd + n] = acc[i][j][r];
        }
      }
    }
    return;
  }
  // Final tile -> parameter layout dst[dc*s_dc + gc*s_gc + tap*s_tap], transposed through LDS one 32x32 MFMA tile per
  // wave at a time (the operand tiles are dead): a lane then walks the gathered channels of ONE dense channel, whose
  // destinations are s_gc floats apart (contiguous for 1x1 kernels) instead of a whole filter apart.
  __syncthreads();
  float* tr = &As[0][0] + wave * (32 * 33);
  static_assert(4 * 32 * 33 <= 2 * WBK2 * LDA, "transposition scratch fits the A tiles");
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        tr[(lane & 31) * 33 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = acc[i][j][r];
      __syncthreads();
      const long long off = s_rowdst[wm * WM + i * 32 + (lane & 31)];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int nl = 2 * k + (lane >> 5);
        const int n = n0 + wn * WN + j * 32 + nl;
        const float v = tr[nl * 33 + (lane & 31)];
        if (off >= 0 && n < d.Cd_log) d.dst[off + n * d.s_dc] = v;
      }
      __syncthreads();
    }
  }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool TAB, int F16 = 0>
__global__ __launch_bounds__(256, 2) void wgrad_fast_kernel(const WDesc d, unsigned x_bytes, unsigned dy_bytes) {
  wgrad_fast_body<BM, BN, WAVES_M, WAVES_N, TAB, F16>(d, x_bytes, dy_bytes, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several weight-gradient GEMMs in ONE launch (ali_wgrad_launch_multi): the weight gradients of a backward pass depend
// on nothing but saved activations and the layers' output gradients, and nothing depends on them before the optimiser
// step -- so their launches are collected and issued together, longest blocks first: one launch instead of one per
// layer, and the small layers' grids fill the gaps the large ones leave.
constexpr int kWJobs = 12;
struct WJob { WDesc d; unsigned xb, yb; int gx, gy, blk0, pad_; };
struct WJobs { int n, pad_; WJob j[kWJobs]; };
static_assert(sizeof(WJobs) <= 4000, "kernel argument segment");
template <int BM, int BN, int WAVES_M, int WAVES_N, bool TAB, int F16 = 0>
__global__ __launch_bounds__(256, 2) void wgrad_fast_multi_kernel(const WJobs jobs) {
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < jobs.n; ++i)
    if ((int)blockIdx.x >= jobs.j[i].blk0) k = i;
  const WJob& J = jobs.j[k];
  const int lin = blockIdx.x - J.blk0;
  const int bx = lin % J.gx, t2 = lin / J.gx;
  wgrad_fast_body<BM, BN, WAVES_M, WAVES_N, TAB, F16>(J.d, J.xb, J.yb, bx, t2 % J.gy, t2 / J.gy);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the Discriminator's first conv (mnist.py:118: 5 planes padded to 8 -> 32 channels, 5x5, stride 1) as
// a per-image LDS-resident kernel, the counterpart of conv_first_kernel (gconv.hip).  As a GEMM it is 200 x 32 over
// 295 k pixels with 75 of the 200 rows channel padding, two 128-row tiles, one block per CU: 69 us.  Here a block keeps
// the image's live planes in LDS (pitch CL, odd), streams the 32-channel output gradient through LDS FW_GR rows at a
// time (double buffered, fetched one chunk ahead), and wave w accumulates dW[k][n], n = 32 w + lane (n = tap * CL + c,
// 125 of 128 live), over all pixels of all its images in ONE MFMA accumulator: A = g^T (k x pixel), B = shifted image
// reads (pixel x (tap, c)), both one ds_read_b32 per MFMA.  Every block leaves a slab [tap * CL + c][k] (+ its
// bias-gradient row) for the usual slab fold.
constexpr int FW_GR = 4;
template <int CL, int RS>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const WDesc d) {
  extern __shared__ __attribute__((aligned(16))) float fw_smem[];
  float* img = fw_smem;                                  // [H*W][CL]
  float* gch = fw_smem + d.H * d.W * CL;                 // [2][FW_GR*Q][33]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = wave * 32 + (lane & 31), kk = lane >> 5;
  constexpr int KL = RS * RS * CL;
  const bool live = n < KL;
  const int tap = live ? n / CL : 0, c = live ? n - tap * CL : 0;
  const int boff = ((tap / RS) * d.W + (tap % RS)) * CL + c;
  f32x16 acc, acc1;                                      // two accumulators: two independent MFMA chains
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = 0.f;
  f32x4 dbacc = {0.f, 0.f, 0.f, 0.f};
  const int HW = d.H * d.W;
  const int nchunk = (d.P + FW_GR - 1) / FW_GR;
  const int gch_sz = FW_GR * d.Q * 33;
  constexpr int PF = (FW_GR * 32 * 8 + 255) / 256;      // float4 per thread per chunk (Q <= 32)
  f32x4 pf[PF];
  // chunk (b, ci): FW_GR rows of the output gradient of image b; fetched into registers one chunk ahead of its use
  auto fetch = [&](int b, int ci) {
    const int p0 = ci * FW_GR;
    const int npx = min(FW_GR, d.P - p0) * d.Q;
    const float* gsrc = d.dy + ((long long)(b * d.P + p0) * d.Q) * 32;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int i = t + u * 256;
      pf[u] = i < npx * 8 ? *reinterpret_cast<const f32x4*>(gsrc + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int buf = 0;
  if ((int)blockIdx.x < d.B) fetch(blockIdx.x, 0);
  for (int b = blockIdx.x; b < d.B; b += gridDim.x) {
    __syncthreads();                                     // the previous image's last chunk has been consumed
    const float* src = d.x + (long long)b * HW * 8;
    for (int i = t; i < HW * 2; i += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
      float* dst = img + (i >> 1) * CL + (i & 1) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((i & 1) * 4 + e < CL) dst[e] = v[e];
    }
    for (int ci = 0; ci < nchunk; ++ci) {
      const int p0 = ci * FW_GR;
      const int rows = min(FW_GR, d.P - p0), npx = rows * d.Q;
      float* gc = gch + buf * gch_sz;
#pragma unroll
      for (int u = 0; u < PF; ++u) {                     // (i & 7 == t & 7: a thread always stages the same 4 channels)
        const int i = t + u * 256;
        if (i < npx * 8) {
          float* dst = gc + (i >> 3) * 33 + (i & 7) * 4;
          dst[0] = pf[u][0]; dst[1] = pf[u][1]; dst[2] = pf[u][2]; dst[3] = pf[u][3];
          dbacc += pf[u];
        }
      }
      __syncthreads();                                   // chunk (and, for ci == 0, the image) visible; the other buffer is free
      if (ci + 1 < nchunk) fetch(b, ci + 1);             // the next chunk travels while this one is multiplied
      else if (b + (int)gridDim.x < d.B) fetch(b + gridDim.x, 0);
      // pixels (row pr, column q = kk, kk + 2, ...) of the chunk, four MFMA operand pairs requested at a time (the loop is
      // one dependent MFMA chain: without the batching every step would also wait for its two LDS reads)
      for (int pr = 0; pr < rows; ++pr) {
        const float* arow = gc + (pr * d.Q) * 33 + (lane & 31);
        const float* irow = img + ((p0 + pr) * d.W) * CL + boff;
        for (int q0 = kk; q0 < d.Q; q0 += 8) {
          float av[4], bw[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int q = q0 + 2 * u;
            const bool in = q < d.Q;
            av[u] = in ? arow[q * 33] : 0.f;
            bw[u] = (in && live) ? irow[q * CL] : 0.f;
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bw[0], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bw[1], acc1, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bw[2], acc, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bw[3], acc1, 0, 0, 0);
        }
      }
      buf ^= 1;
    }
  }
  // slab of this block: [n][k] (+ pad rows never read), then its bias-gradient row
  float* slab = d.ws + (long long)blockIdx.x * d.slab;
  if (live) {
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[n * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk] = acc[r] + acc1[r];
  }
  if (d.db) {
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(gch);          // [32 thread groups][8 channel chunks]
    red[(t >> 3) * 8 + (t & 7)] = dbacc;
    __syncthreads();
    if (t < 8) {
      f32x4 s4 = red[t];
      for (int q = 1; q < 32; ++q) s4 += red[q * 8 + t];
      *reinterpret_cast<f32x4*>(d.dbws + (long long)blockIdx.x * 32 + t * 4) = s4;
    }
  }
}

__global__ void wgrad_pixtab_kernel(int npix, int P, int Q, int H, int W, int Cg, int stride, int* __restrict__ out) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int b = pix / (P * Q), rem = pix - b * (P * Q);
  const int p = rem / Q, q = rem - p * Q;
  out[2 * pix] = ((b * H + p * stride) * W + q * stride) * Cg * 4;
  out[2 * pix + 1] = (p * stride + 0x4000) | ((q * stride + 0x4000) << 16);
}

// dst[dc*s_dc + gc*s_gc + tap*s_tap] = sum_s ws[s][tap*Cg+gc][dc]
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, int S, int Mtot, int Cg, int Cd, int Cg_log,
                                    int Cd_log, long long s_dc, long long s_gc, long long s_tap,
                                    float* __restrict__ dst, const float* __restrict__ dbws, float* __restrict__ db,
                                    long long slab) {
  const long long total = (long long)Mtot * Cd;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  if (db && i < Cd_log) {
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += dbws[(long long)s * Cd + i];
    db[i] = v;
  }
  for (; i < total; i += step) {
    const int dc = (int)(i % Cd);
    const int m = (int)(i / Cd);
    const int tap = m / Cg, gc = m - tap * Cg;
    if (dc >= Cd_log || gc >= Cg_log) continue;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += ws[(long long)s * slab + i];
    dst[dc * s_dc + gc * s_gc + tap * s_tap] = v;
  }
}

// Same reduction, laid out for the memory system: a block owns the [G gathered channels x all T taps] x [TD dense
// channels] tile.  Slab rows are read as TD-float runs (float4 per lane), the S slabs are split over SGN thread groups
// (fixed order inside a group, groups combined in order: deterministic), and the tile is written back through LDS so
// that each dense channel stores one contiguous run of G*T floats of the reference weight layout.
// slabs of a power-of-two size would all start in the same HBM channel / L2 set: 96 floats (384 B) between them
constexpr int kSlabPad = 96;
constexpr int kRedRows = 64, kRedGroups = 16, kRedLds = 256;   // rows per tile, slab groups, groups*rows bound
template <int TD>
__device__ __forceinline__ void reduce_tile_body(const float* __restrict__ ws, int S, int Mtot, int Cg, int Cd, int Cg_log,
                                                 int Cd_log, long long s_dc, long long s_gc, long long s_tap,
                                                 float* __restrict__ dst, const float* __restrict__ dbws,
                                                 float* __restrict__ db, int G, int T, int SGN, long long slab,
                                                 int bx, int by, int lin, float* part) {
  const int tid = threadIdx.x;
  if (db && lin * 32 < Cd_log) {
    // bias gradient: block `lin` folds columns [32*lin, 32*lin + 32) -- 8 thread groups take every 8th slab (four loads
    // in flight each), combined in group order: a fixed summation order, and no chain of S dependent loads
    const int c = lin * 32 + (tid & 31), sg = tid >> 5;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < Cd_log) {
      int sl = sg;
      for (; sl + 24 < S; sl += 32) {
        a0 += dbws[(long long)sl * Cd + c];
        a1 += dbws[(long long)(sl + 8) * Cd + c];
        a2 += dbws[(long long)(sl + 16) * Cd + c];
        a3 += dbws[(long long)(sl + 24) * Cd + c];
      }
      for (; sl < S; sl += 8) a0 += dbws[(long long)sl * Cd + c];
    }
    part[tid] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (tid < 32 && c < Cd_log) {
      float v = part[tid];
#pragma unroll
      for (int g8 = 1; g8 < 8; ++g8) v += part[g8 * 32 + tid];
      db[c] = v;
    }
    __syncthreads();
  }
  const int gc0 = bx * G, dc0 = by * TD;
  const int RT = G * T;
  constexpr int C4 = TD / 4;
  const int slots = RT * C4;
  for (int p = tid; p < slots * SGN; p += 256) {
    const int sg = p / slots, sl = p - sg * slots;
    const int r = sl / C4, c4 = sl - r * C4;
    const int gl = r / T, tap = r - gl * T;
    const int gc = gc0 + gl, dc = dc0 + c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gc < Cg && dc < Cd) {
      const float* src = ws + (long long)(tap * Cg + gc) * Cd + dc;
      int sidx = sg;
      for (; sidx + 7 * SGN < S; sidx += 8 * SGN) {      // eight independent loads in flight, summed in slab order
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (long long)(sidx + u * SGN) * slab);
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
      }
      for (; sidx + 3 * SGN < S; sidx += 4 * SGN) {
        const float4 v0 = *reinterpret_cast<const float4*>(src + (long long)sidx * slab);
        const float4 v1 = *reinterpret_cast<const float4*>(src + (long long)(sidx + SGN) * slab);
        const float4 v2 = *reinterpret_cast<const float4*>(src + (long long)(sidx + 2 * SGN) * slab);
        const float4 v3 = *reinterpret_cast<const float4*>(src + (long long)(sidx + 3 * SGN) * slab);
        acc.x = (((acc.x + v0.x) + v1.x) + v2.x) + v3.x;
        acc.y = (((acc.y + v0.y) + v1.y) + v2.y) + v3.y;
        acc.z = (((acc.z + v0.z) + v1.z) + v2.z) + v3.z;
        acc.w = (((acc.w + v0.w) + v1.w) + v2.w) + v3.w;
      }
      for (; sidx < S; sidx += SGN) {
        const float4 v = *reinterpret_cast<const float4*>(src + (long long)sidx * slab);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    float* q = &part[(sg * RT + r) * (TD + 1) + c4 * 4];
    q[0] = acc.x; q[1] = acc.y; q[2] = acc.z; q[3] = acc.w;
  }
  __syncthreads();
  for (int o = tid; o < TD * RT; o += 256) {
    const int dcl = o / RT, r = o - dcl * RT;
    float v = part[r * (TD + 1) + dcl];
    for (int sg = 1; sg < SGN; ++sg) v += part[(sg * RT + r) * (TD + 1) + dcl];
    const int gl = r / T, tap = r - gl * T;
    const int gc = gc0 + gl, dc = dc0 + dcl;
    if (gc < Cg_log && dc < Cd_log) dst[dc * s_dc + gc * s_gc + tap * s_tap] = v;
  }
}

template <int TD>
__global__ void __launch_bounds__(256)
wgrad_reduce_tile_kernel(const float* __restrict__ ws, int S, int Mtot, int Cg, int Cd, int Cg_log, int Cd_log,
                         long long s_dc, long long s_gc, long long s_tap, float* __restrict__ dst,
                         const float* __restrict__ dbws, float* __restrict__ db, int G, int T, int SGN, long long slab) {
  __shared__ float part[kRedLds * (TD + 1)];   // [SGN][RT][TD+1], SGN*RT <= kRedLds
  reduce_tile_body<TD>(ws, S, Mtot, Cg, Cd, Cg_log, Cd_log, s_dc, s_gc, s_tap, dst, dbws, db, G, T, SGN, slab,
                       blockIdx.x, blockIdx.y, blockIdx.y * gridDim.x + blockIdx.x, part);
}

// The same reduction for up to kFoldJobs weight-gradient launches in ONE launch (ali_wgrad_fold_multi): a backward pass
// leaves the slabs of all its layers in place and folds them together in front of the optimiser step -- a dozen
// 10-us launches (each mostly launch latency: a few MB of slabs) become one that streams them at HBM rate.
constexpr int kFoldJobs = 12;
struct FoldJob {
  const float* ws; float* dst; const float* dbws; float* db;
  long long slab, s_dc, s_gc, s_tap;
  int S, Mtot, Cg, Cd, Cg_log, Cd_log, G, T, SGN, nbx, tile0, TD;
};
struct FoldJobs { int n, total; FoldJob j[kFoldJobs]; };
__global__ void __launch_bounds__(256) wgrad_fold_multi_kernel(const FoldJobs jobs) {
  __shared__ float part[kRedLds * 33];
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < jobs.n; ++i)
    if ((int)blockIdx.x >= jobs.j[i].tile0) k = i;
  const FoldJob& J = jobs.j[k];
  const int lin = blockIdx.x - J.tile0;
  const int by = lin / J.nbx, bx = lin - by * J.nbx;
#define FOLD_TD(TD_)                                                                                                \
  reduce_tile_body<TD_>(J.ws, J.S, J.Mtot, J.Cg, J.Cd, J.Cg_log, J.Cd_log, J.s_dc, J.s_gc, J.s_tap, J.dst, J.dbws, J.db, \
                        J.G, J.T, J.SGN, J.slab, bx, by, lin, part)
  if (J.TD == 32) FOLD_TD(32);          // block-uniform
  else if (J.TD == 16) FOLD_TD(16);
  else if (J.TD == 8) FOLD_TD(8);
  else FOLD_TD(4);
#undef FOLD_TD
}

}  // namespace ali

using namespace ali;

// tile of a weight-gradient launch: [gathered channel x tap] rows by dense channels
static void wgrad_tile(const AliConvGeom* g, bool fast, bool f16, int& bm, int& bn) {
  bn = g->K > 64 ? 128 : (g->K > 32 ? 64 : 32);
  bm = 128;
  if (!fast) return;
  const int Mtot = g->R * g->S * g->C;
  const long long npix = (long long)g->B * g->P * g->Q * (tuning().tile_m_scale > 0 ? tuning().tile_m_scale : 1);
  if (tuning().wbm > 0 && tuning().wbn > 0) {   // forced (tests): any of the instantiated variants
    const int fm = tuning().wbm, fn = tuning().wbn;
    if ((fm == 64 && fn == 64) || (fm == 128 && (fn == 128 || fn == 64 || fn == 32))) { bm = fm; bn = fn; return; }
  }
  // more, smaller tiles while the grid is shallow (see gconv.hip: waits are only hidden by co-resident waves)
  const long long b64 = (long long)((Mtot + 63) / 64) * ((g->K + 63) / 64);
  // ... but a long pixel reduction (spectrogram layers) re-reads both operands once per tile pair: larger tiles,
  // the grid depth comes from the split over pixels
  long long small_lim = npix >= 200000 ? 16 : 2 * kNumCU;
  if (tuning().wgrad_small >= 0) small_lim = tuning().wgrad_small;
  const bool f16_tiles = f16 && g->K > 64 && Mtot >= 128;   // fp16 loop: bytes-bound, big tiles
  if (g->K > 32 && b64 <= small_lim && !f16_tiles) { bm = 64; bn = 64; }
  else if (g->K > 64) { bm = 128; bn = 128; }
  else if (g->K > 32) { bm = 128; bn = 64; }
  else { bm = 128; bn = 32; }
}

static bool wgrad_fast_ok(const AliConvGeom* g, int ldd) {
  const long long x_elems = (long long)g->B * g->H * g->W * g->C;
  const long long dy_elems = ((long long)g->B * g->P * g->Q - 1) * ldd + g->K;
  return (g->C % 4) == 0 && (g->K % 4) == 0 && x_elems < (1LL << 30) && dy_elems < (1LL << 30) &&
         (long long)g->B * g->P * g->Q < (1 << 24);
}

extern "C" int32_t ali_wgrad_deferrable(const AliConvGeom* g, int32_t mfma_f16) {
  if (!g || g->B <= 0 || g->R * g->S > kMaxTaps || mfma_f16) return 0;
  if (!wgrad_fast_ok(g, g->K) || g->H >= 0x2000 || g->W >= 0x2000 || g->pad >= 0x2000) return 0;
  int bm, bn;
  wgrad_tile(g, true, false, bm, bn);
  return bm == 64 ? 1 : 0;
}

extern "C" int ali_conv_bwd_weight(const AliConvGeom* g, const float* x, const float* dy, float* dst, int32_t Cg_log,
                                   int32_t Cd_log, int64_t s_dc, int64_t s_gc, int64_t s_tap, float* db,
                                   const int32_t* pixtab, int32_t mfma_f16, const void* x16, const void* dy16,
                                   int32_t dy_ld, AliWgradFold* fold, AliWgradJob* job, int32_t split_target,
                                   void* ws, size_t ws_bytes, ali_stream_t stream_) {
  if (!g || !x || !dy || !dst || g->R * g->S > kMaxTaps || g->B <= 0) {
    set_error("ali_conv_bwd_weight: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  if (fold) fold->S = 0;
  if (job) job->opaque[0] = 0;
  hipStream_t stream = (hipStream_t)stream_;
  void* const ws_all = ws;
  const size_t ws_all_bytes = ws_bytes;
  ws = ws_payload(ws);                      // the workspace head holds the GEMM kernels' split-K counters
  ws_bytes = ws_payload_bytes(ws_bytes);
  WDesc d;
  memset(&d, 0, sizeof(d));
  d.x = x; d.dy = dy; d.dst = dst; d.ws = reinterpret_cast<float*>(ws);
  d.B = g->B; d.H = g->H; d.W = g->W; d.Cg = g->C; d.P = g->P; d.Q = g->Q; d.Cd = g->K;
  d.R = g->R; d.S = g->S; d.stride = g->stride; d.pad = g->pad;
  d.Cg_log = Cg_log; d.Cd_log = Cd_log; d.s_dc = s_dc; d.s_gc = s_gc; d.s_tap = s_tap;
  d.Mtot = g->R * g->S * g->C;
  d.npix = g->B * g->P * g->Q;
  const bool veca = (g->C % 4) == 0, vecb = (g->K % 4) == 0;
  d.ldd = dy_ld > 0 ? dy_ld : g->K;
  const long long x_elems = (long long)g->B * g->H * g->W * g->C;
  const long long dy_elems = ((long long)g->B * g->P * g->Q - 1) * d.ldd + g->K;
  if (d.ldd != g->K && (d.ldd < g->K || (d.ldd % 4) || !veca || !vecb)) {
    set_error("ali_conv_bwd_weight: dy_ld needs the vector kernels (channel counts % 4 == 0) and dy_ld >= K, % 4 == 0");
    return ALI_ERR_BAD_ARG;
  }
  const bool fast = wgrad_fast_ok(g, d.ldd);
  if (!fast && d.ldd != g->K) { set_error("ali_conv_bwd_weight: dy_ld on a tensor too large for the vector kernels"); return ALI_ERR_BAD_ARG; }
  int bm, bn;
  wgrad_tile(g, fast, mfma_f16 && pixtab, bm, bn);
  // the Discriminator's first conv (5 live planes in 8, 32 output channels, 5x5 stride 1 on maps up to 32 x 32): per-image
  // LDS-resident kernel; its per-block slabs take the usual fold (deferred with `fold`, or right here)
  const bool first_lds = fast && !mfma_f16 && tuning().no_first_wgrad == 0 && g->C == 8 && Cg_log == 5 && g->K == 32 &&
                         g->stride == 1 && g->pad == 0 && g->R == 5 && g->S == 5 && g->H <= 32 && g->W <= 32 &&
                         g->B >= 64 && d.ldd == g->K && g->P == g->H - 4 && g->Q == g->W - 4;
  if (first_lds) {
    int S = g->B < kNumCU ? g->B : kNumCU;
    d.Mtot = 25 * 5; d.Cg = 5;
    d.slab = (long long)d.Mtot * g->K + kSlabPad;
    if (((size_t)S * d.slab + (size_t)S * g->K) * sizeof(float) > ws_bytes) { set_error("ali_conv_bwd_weight: workspace too small"); return ALI_ERR_WORKSPACE; }
    d.splitk = S;
    d.db = db;
    d.dbws = d.ws + (size_t)S * d.slab;
    const size_t lds = ((size_t)g->H * g->W * 5 + (size_t)2 * FW_GR * g->Q * 33) * sizeof(float);
    hipLaunchKernelGGL((conv_first_wgrad_kernel<5, 5>), dim3(S), dim3(256), lds, stream, d);
    int rc1 = check_launch("conv_first_wgrad_kernel");
    if (rc1) return rc1;
    if (fold) {
      fold->ws = d.ws; fold->dst = dst; fold->dbws = d.dbws; fold->db = d.db;
      fold->slab = d.slab; fold->s_dc = s_dc; fold->s_gc = s_gc; fold->s_tap = s_tap;
      fold->S = S; fold->Mtot = d.Mtot; fold->Cg = 5; fold->Cd = g->K; fold->Cg_log = 5; fold->Cd_log = Cd_log;
      fold->T = 25; fold->reserved = 0;
      fold->ws_used = (uint64_t)kWsReserved + ((uint64_t)S * d.slab + (uint64_t)S * g->K) * sizeof(float);
      return ALI_OK;
    }
    const int T = 25, G = 1, TD = 32;
    int SGN = kRedLds / (G * T);
    if (SGN > kRedGroups) SGN = kRedGroups;
    if (SGN > S) SGN = S;
    hipLaunchKernelGGL(wgrad_reduce_tile_kernel<32>, dim3(5, (g->K + TD - 1) / TD), dim3(256), 0, stream, d.ws, S, d.Mtot, 5,
                       d.Cd, 5, Cd_log, (long long)s_dc, (long long)s_gc, (long long)s_tap, dst, d.dbws, d.db, G, T, SGN,
                       d.slab);
    return check_launch("wgrad_reduce_tile_kernel");
  }
  const int wbk = fast ? WBK2 : WBK;
  const int tiles_m = (d.Mtot + bm - 1) / bm, tiles_n = (g->K + bn - 1) / bn;
  const long long blocks = (long long)tiles_m * tiles_n;
  int S = 1;
  const int nkt = (d.npix + wbk - 1) / wbk;
  // split over pixels until the launch has `target` blocks: 4 per CU on its own; a job of a combined launch needs fewer
  // (the other layers' blocks fill the machine) -- the caller says how many (split_target, from the number of jobs its
  // passes have: measured best on the MNIST pass of 8 jobs: 256 of 128 / 256 / 512 / 1024 / 2048)
  int target = 4 * kNumCU;
  if (job && fold && split_target > 0) target = split_target < kNumCU ? kNumCU : (split_target > 4 * kNumCU ? 4 * kNumCU : split_target);
  if (tuning().wgrad_blocks > 0) target = tuning().wgrad_blocks;
  if (blocks < target && nkt >= 4) {
    S = (int)((target + blocks - 1) / blocks);
    if (S > nkt / 2) S = nkt / 2;
    {
      // the slab fold reads S * Mtot * K floats: keep S modest unless the pixel range is so long that two weight
      // tiles could not fill the chip otherwise (first layers of the spectrogram models)
      int cap = d.npix >= (1 << 20) ? 512 : 128;
      if (tuning().wgrad_scap > 0) cap = tuning().wgrad_scap;
      if (S > cap) S = cap;
    }
    while (S > 1 && (size_t)S * (((size_t)d.Mtot + 1) * g->K + kSlabPad) * sizeof(float) > ws_bytes) --S;
    if (S < 1) S = 1;
  }
  d.splitk = S;
  int per = (nkt + S - 1) / S;
  d.pix_per_split = per * wbk;
  d.db = fast ? db : nullptr;
  d.slab = (long long)d.Mtot * g->K + kSlabPad;
  d.dbws = d.ws + (size_t)S * d.slab;
  dim3 grid(tiles_m, tiles_n, S), block(256);
  if (fast) {
    const unsigned xb = (unsigned)(x_elems * 4), yb = (unsigned)(dy_elems * 4);
    // the table packs positions as 16-bit fields: maps up to 8191 x 8191
    d.pixtab = (pixtab && g->H < 0x2000 && g->W < 0x2000 && g->pad < 0x2000) ? pixtab : nullptr;
    const bool f16 = mfma_f16 && d.pixtab;
    const bool mem16 = f16 && x16 && dy16 && (g->C % 8) == 0 && (g->K % 8) == 0 && bn >= 64 && d.ldd == g->K;
    d.x16 = mem16 ? reinterpret_cast<const _Float16*>(x16) : nullptr;
    d.dy16 = mem16 ? reinterpret_cast<const _Float16*>(dy16) : nullptr;
#define FLAUNCH1(BM_, BN_, WMM, WNN)                                                                              \
  do {                                                                                                            \
    if (f16) hipLaunchKernelGGL((wgrad_fast_kernel<BM_, BN_, WMM, WNN, true, 1>), grid, block, 0, stream, d, xb, yb);     \
    else if (d.pixtab) hipLaunchKernelGGL((wgrad_fast_kernel<BM_, BN_, WMM, WNN, true>), grid, block, 0, stream, d, xb, yb);   \
    else hipLaunchKernelGGL((wgrad_fast_kernel<BM_, BN_, WMM, WNN, false>), grid, block, 0, stream, d, xb, yb);   \
  } while (0)
#define FLAUNCH(BM_, BN_, WMM, WNN)                                                                               \
  do {                                                                                                            \
    if (mem16) hipLaunchKernelGGL((wgrad_fast_kernel<BM_, BN_, WMM, WNN, true, 2>), grid, block, 0, stream, d, xb, yb);   \
    else FLAUNCH1(BM_, BN_, WMM, WNN);                                                                            \
  } while (0)
    static_assert(sizeof(WJob) + 8 <= sizeof(AliWgradJob), "AliWgradJob too small");
    if (job && fold && bm == 64 && d.pixtab && !f16 && (S == 1 || g->R * g->S <= kRedRows)) {
      // deferred: the caller launches it with the other weight gradients of the pass (ali_wgrad_launch_multi)
      WJob wj;
      memset(&wj, 0, sizeof(wj));
      wj.d = d; wj.xb = xb; wj.yb = yb; wj.gx = tiles_m; wj.gy = tiles_n; wj.blk0 = S;   // (blk0 carries S until launch)
      job->opaque[0] = 1;
      memcpy(&job->opaque[1], &wj, sizeof(wj));
    }
    else if (bm == 64) FLAUNCH(64, 64, 2, 2);
    else if (bn == 128) FLAUNCH(128, 128, 2, 2);
    else if (bn == 64) FLAUNCH(128, 64, 2, 2);
    else FLAUNCH1(128, 32, 4, 1);
#undef FLAUNCH
#undef FLAUNCH1
  } else {
#define WLAUNCH(BN_, WMM, WNN)                                                                        \
  do {                                                                                                 \
    if (veca && vecb) hipLaunchKernelGGL((wgrad_kernel<128, BN_, WMM, WNN, true, true>), grid, block, 0, stream, d);   \
    else if (veca) hipLaunchKernelGGL((wgrad_kernel<128, BN_, WMM, WNN, true, false>), grid, block, 0, stream, d);     \
    else if (vecb) hipLaunchKernelGGL((wgrad_kernel<128, BN_, WMM, WNN, false, true>), grid, block, 0, stream, d);     \
    else hipLaunchKernelGGL((wgrad_kernel<128, BN_, WMM, WNN, false, false>), grid, block, 0, stream, d);               \
  } while (0)
    if (bn == 128) WLAUNCH(128, 2, 2);
    else if (bn == 64) WLAUNCH(64, 2, 2);
    else WLAUNCH(32, 4, 1);
#undef WLAUNCH
  }
  int rc = check_launch("wgrad_kernel");
  if (rc) return rc;
  if (S > 1 && fold && fast && g->R * g->S <= kRedRows) {
    // deferred: the caller keeps the slabs (ws) untouched and folds them with ali_wgrad_fold_multi
    fold->ws = d.ws; fold->dst = dst; fold->dbws = d.dbws; fold->db = d.db;
    fold->slab = d.slab; fold->s_dc = s_dc; fold->s_gc = s_gc; fold->s_tap = s_tap;
    fold->S = S; fold->Mtot = d.Mtot; fold->Cg = d.Cg; fold->Cd = d.Cd; fold->Cg_log = Cg_log; fold->Cd_log = Cd_log;
    fold->T = g->R * g->S; fold->reserved = 0;
    fold->ws_used = (uint64_t)kWsReserved + ((uint64_t)S * d.slab + (uint64_t)S * g->K) * sizeof(float);
    return ALI_OK;
  }
  if (S > 1) {
    const long long total = (long long)d.Mtot * g->K;
    const int T = g->R * g->S;
    if (fast && T <= kRedRows) {
      // tile: G gathered channels x T taps (>= 16 rows, <= 64) by TD dense channels.  Reads dominate (S slabs in, one
      // tile out): take the widest TD that still gives the grid two blocks per CU, and grow G while it stays that deep.
      int G = (16 + T - 1) / T;
      auto nblk = [&](int G_, int TD_) { return (long long)((g->C + G_ - 1) / G_) * ((g->K + TD_ - 1) / TD_); };
      while (2 * G * T <= kRedRows && nblk(2 * G, 32) >= 2 * kNumCU) G *= 2;
      int TD = 32;
      while (TD > 4 && nblk(G, TD) < 2 * kNumCU) TD >>= 1;
      int SGN = kRedLds / (G * T);
      if (SGN > kRedGroups) SGN = kRedGroups;
      if (SGN > S) SGN = S;
      dim3 rgrid((g->C + G - 1) / G, (g->K + TD - 1) / TD);
#define RLAUNCH(TD_)                                                                                              \
  hipLaunchKernelGGL(wgrad_reduce_tile_kernel<TD_>, rgrid, dim3(256), 0, stream, d.ws, S, d.Mtot, d.Cg, d.Cd, Cg_log, \
                     Cd_log, (long long)s_dc, (long long)s_gc, (long long)s_tap, dst, d.dbws, d.db, G, T, SGN, d.slab)
      if (TD == 32) RLAUNCH(32);
      else if (TD == 16) RLAUNCH(16);
      else if (TD == 8) RLAUNCH(8);
      else RLAUNCH(4);
#undef RLAUNCH
    } else {
      int nb = (int)((total + 255) / 256);
      if (nb > 4096) nb = 4096;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, stream, d.ws, S, d.Mtot, d.Cg, d.Cd, Cg_log,
                         Cd_log, (long long)s_dc, (long long)s_gc, (long long)s_tap, dst, d.dbws, d.db, d.slab);
    }
    rc = check_launch("wgrad_reduce_kernel");
  }
  if (!rc && db && !fast)   // generic path: bias gradient by the stand-alone column-sum kernels
    rc = ali_colsum(dy, (int64_t)d.npix, Cd_log, g->K, db, ws_all, ws_all_bytes, stream_);
  return rc;
}

extern "C" int ali_wgrad_launch_multi(int32_t n, const AliWgradJob* jobs, ali_stream_t stream_) {
  if (n < 0 || (n > 0 && !jobs)) { set_error("ali_wgrad_launch_multi: bad argument"); return ALI_ERR_BAD_ARG; }
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<WJob> all;
  for (int i = 0; i < n; ++i) {
    if (jobs[i].opaque[0] != 1) { set_error("ali_wgrad_launch_multi: job %d was not deferred", i); return ALI_ERR_BAD_ARG; }
    WJob wj;
    memcpy(&wj, &jobs[i].opaque[1], sizeof(wj));
    all.push_back(wj);
  }
  // longest blocks first (the dispatcher hands blocks out in order: the short ones fill the tail)
  std::stable_sort(all.begin(), all.end(), [](const WJob& a, const WJob& b) { return a.d.pix_per_split > b.d.pix_per_split; });
  for (size_t j0 = 0; j0 < all.size(); j0 += kWJobs) {
    WJobs wj;
    memset(&wj, 0, sizeof(wj));
    long long blocks = 0;
    for (size_t i = j0; i < all.size() && i < j0 + kWJobs; ++i) {
      WJob& J = wj.j[wj.n++];
      J = all[i];
      const int S = J.blk0;
      J.blk0 = (int)blocks;
      blocks += (long long)J.gx * J.gy * S;
      if (blocks > (1LL << 30)) { set_error("ali_wgrad_launch_multi: grid too large"); return ALI_ERR_BAD_ARG; }
    }
    hipLaunchKernelGGL((wgrad_fast_multi_kernel<64, 64, 2, 2, true>), dim3((unsigned)blocks), dim3(256), 0, stream, wj);
    int rc = check_launch("wgrad_fast_multi_kernel");
    if (rc) return rc;
  }
  return ALI_OK;
}

extern "C" int ali_wgrad_fold_multi(int32_t n, const AliWgradFold* jobs, ali_stream_t stream_) {
  if (n < 0 || (n > 0 && !jobs)) { set_error("ali_wgrad_fold_multi: bad argument"); return ALI_ERR_BAD_ARG; }
  hipStream_t stream = (hipStream_t)stream_;
  for (int j0 = 0; j0 < n; j0 += kFoldJobs) {
    FoldJobs fj;
    memset(&fj, 0, sizeof(fj));
    int tiles = 0;
    for (int i = j0; i < n && i < j0 + kFoldJobs; ++i) {
      const AliWgradFold& a = jobs[i];
      if (a.S < 1 || a.T < 1 || a.T > kRedRows || !a.ws || !a.dst || a.Cg < 1 || a.Cd < 1 || (a.Cd % 4) != 0) {
        set_error("ali_wgrad_fold_multi: bad job %d", i);
        return ALI_ERR_BAD_ARG;
      }
      FoldJob& J = fj.j[fj.n++];
      J.ws = a.ws; J.dst = a.dst; J.dbws = a.dbws; J.db = a.db;
      J.slab = a.slab; J.s_dc = a.s_dc; J.s_gc = a.s_gc; J.s_tap = a.s_tap;
      J.S = a.S; J.Mtot = a.Mtot; J.Cg = a.Cg; J.Cd = a.Cd; J.Cg_log = a.Cg_log; J.Cd_log = a.Cd_log; J.T = a.T;
      // tile: G gathered channels x T taps (16..64 rows) by 32 dense channels; G grows while the job stays deep
      // ... and narrower than 32 dense channels when a block would otherwise walk more than ~128 KB of slabs (the
      // first layers' hundreds of slabs for a handful of tiles): the launch lasts as long as its longest block
      int G = (16 + a.T - 1) / a.T;
      int TD = 32;
      while (TD > 4 && (long long)a.S * G * a.T * TD * 4 > (128 << 10)) TD >>= 1;
      auto nblk = [&](int G_) { return (long long)((a.Cg + G_ - 1) / G_) * ((a.Cd + TD - 1) / TD); };
      while (TD == 32 && 2 * G * a.T <= kRedRows && nblk(2 * G) >= 2 * kNumCU) G *= 2;
      J.TD = TD;
      int SGN = kRedLds / (G * a.T);
      if (SGN > kRedGroups) SGN = kRedGroups;
      if (SGN > a.S) SGN = a.S;
      J.G = G; J.SGN = SGN;
      J.nbx = (a.Cg + G - 1) / G;
      J.tile0 = tiles;
      const long long nb = nblk(G);
      if (nb * 32 < a.Cd_log || tiles + nb > (1 << 30)) { set_error("ali_wgrad_fold_multi: job %d too large", i); return ALI_ERR_BAD_ARG; }
      tiles += (int)nb;
    }
    fj.total = tiles;
    if (tiles > 0) hipLaunchKernelGGL(wgrad_fold_multi_kernel, dim3(tiles), dim3(256), 0, stream, fj);
    int rc = check_launch("wgrad_fold_multi_kernel");
    if (rc) return rc;
  }
  return ALI_OK;
}

extern "C" int ali_wgrad_pixtab(const AliConvGeom* g, int32_t* out, ali_stream_t stream) {
  if (!g || !out || g->B <= 0 || g->P <= 0 || g->Q <= 0 || g->stride <= 0 ||
      (long long)g->B * g->H * g->W * g->C >= (1LL << 29) || (long long)g->B * g->P * g->Q >= (1LL << 28)) {
    set_error("ali_wgrad_pixtab: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  const int npix = g->B * g->P * g->Q;
  hipLaunchKernelGGL(wgrad_pixtab_kernel, dim3((npix + 255) / 256), dim3(256), 0, (hipStream_t)stream, npix, g->P, g->Q, g->H,
                     g->W, g->C, g->stride, out);
  return check_launch("wgrad_pixtab_kernel");
}
