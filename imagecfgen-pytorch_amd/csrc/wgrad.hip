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
          if (m < d.Mtot) d.ws[((long long)blockIdx.z * d.Mtot + m) * d.Cd + n] = v;
        } else {
          const long long off = s_rowdst[row];
          if (off >= 0) d.dst[off + n * d.s_dc] = v;
        }
      }
    }
  }
}

// dst[dc*s_dc + gc*s_gc + tap*s_tap] = sum_s ws[s][tap*Cg+gc][dc]
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, int S, int Mtot, int Cg, int Cd, int Cg_log,
                                    int Cd_log, long long s_dc, long long s_gc, long long s_tap,
                                    float* __restrict__ dst) {
  const long long total = (long long)Mtot * Cd;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += step) {
    const int dc = (int)(i % Cd);
    const int m = (int)(i / Cd);
    const int tap = m / Cg, gc = m - tap * Cg;
    if (dc >= Cd_log || gc >= Cg_log) continue;
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += ws[(long long)s * total + i];
    dst[dc * s_dc + gc * s_gc + tap * s_tap] = v;
  }
}

}  // namespace ali

using namespace ali;

extern "C" int ali_conv_bwd_weight(const AliConvGeom* g, const float* x, const float* dy, float* dst, int32_t Cg_log,
                                   int32_t Cd_log, int64_t s_dc, int64_t s_gc, int64_t s_tap, void* ws,
                                   size_t ws_bytes, ali_stream_t stream_) {
  if (!g || !x || !dy || !dst || g->R * g->S > kMaxTaps || g->B <= 0) {
    set_error("ali_conv_bwd_weight: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  hipStream_t stream = (hipStream_t)stream_;
  WDesc d;
  memset(&d, 0, sizeof(d));
  d.x = x; d.dy = dy; d.dst = dst; d.ws = reinterpret_cast<float*>(ws);
  d.B = g->B; d.H = g->H; d.W = g->W; d.Cg = g->C; d.P = g->P; d.Q = g->Q; d.Cd = g->K;
  d.R = g->R; d.S = g->S; d.stride = g->stride; d.pad = g->pad;
  d.Cg_log = Cg_log; d.Cd_log = Cd_log; d.s_dc = s_dc; d.s_gc = s_gc; d.s_tap = s_tap;
  d.Mtot = g->R * g->S * g->C;
  d.npix = g->B * g->P * g->Q;
  const bool veca = (g->C % 4) == 0, vecb = (g->K % 4) == 0;
  const int bn = g->K > 64 ? 128 : (g->K > 32 ? 64 : 32);
  const int bm = 128;
  const int tiles_m = (d.Mtot + bm - 1) / bm, tiles_n = (g->K + bn - 1) / bn;
  const long long blocks = (long long)tiles_m * tiles_n;
  int S = 1;
  const int nkt = (d.npix + WBK - 1) / WBK;
  if (blocks < 2 * kNumCU && nkt >= 8) {
    S = (int)((2 * kNumCU + blocks - 1) / blocks);
    if (S > nkt / 4) S = nkt / 4;
    if (S > 64) S = 64;
    while (S > 1 && (size_t)S * d.Mtot * g->K * sizeof(float) > ws_bytes) --S;
    if (S < 1) S = 1;
  }
  d.splitk = S;
  int per = (nkt + S - 1) / S;
  d.pix_per_split = per * WBK;
  dim3 grid(tiles_m, tiles_n, S), block(256);
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
  int rc = check_launch("wgrad_kernel");
  if (rc) return rc;
  if (S > 1) {
    const long long total = (long long)d.Mtot * g->K;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, stream, d.ws, S, d.Mtot, d.Cg, d.Cd, Cg_log,
                       Cd_log, (long long)s_dc, (long long)s_gc, (long long)s_tap, dst);
    rc = check_launch("wgrad_reduce_kernel");
  }
  return rc;
}
