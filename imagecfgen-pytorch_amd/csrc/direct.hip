// Direct (non-GEMM) kernels for the one-channel ends of the stacks, where an implicit GEMM would waste
// 31/32 of every MFMA tile:
//   * ConvTranspose2d(64 -> 1, k4) + Tanh, the Generator's last layer (reference mnist.py:72-73), forward,
//     data gradient and weight gradient;
//   * the single input plane of a first Conv2d's data gradient that is actually consumed (the image plane
//     on the D(G(z)) path, the embedding plane for Discriminator.digit_embedding; mnist.py:108,142-151) and the
//     per-input-channel weight gradient of that first layer.
// All are stride-1 "full" correlations between a C-channel NHWC map and a 1-channel map:
//   fwd   : out[b,h,w]      = act(bias + sum_{r,s,k} big[b, h+pad-r, w+pad-s, k] * w[r*S+s][k])
//   dgrad : gbig[b,p,q,k]   = act'(y[b,p,q,k]) * sum_{r,s} small[b, p+r-pad, q+s-pad] * w[r*S+s][k]
//   wgrad : dw[k][r*S+s]    = sum_{b,p,q} big[b,p,q,k] * small[b, p+r-pad, q+s-pad]
// They run on the VALU with the operands staged in LDS; algorithmically they are HBM bound
// (fwd/dgrad move the C-channel map once: 4*C bytes per pixel for 2*C*R*S flops).
#include "ali_common.h"
#include <algorithm>

namespace ali {


struct T1Desc {
  const float* big;
  const float* small;
  const float* w;      // [R*S][K]
  const float* bias;   // 1 float or null
  const float* dact_y; // [B,P,Q,K] or null
  float* out;
  float* part;
  int B, P, Q, K;      // big map
  int H, W;            // small map
  int R, S, pad;
  int sstride;         // element stride of the small map (1, or the channel count when it is one plane of an NHWC tensor)
  int ostride;         // element stride of the fwd output
  int act; float slope;
  int dact; float dslope;
  const float* rowscale;   // fwd: optional per-sample factor rowscale[b * rowscale_ld] on the output (a Dropout2d mask column)
  int rowscale_ld;
};

// ---------------------------------------------------------------- forward: big (K ch) -> small (1 ch)
// Block = T1F_RB x T1F_CB output pixels of one image; 32 channels of the (halo'd) big patch are staged in LDS per pass.
// A thread owns 4 channels (k4 = thread % 8) of a run of 4 output pixels: the filter taps of its channels sit in
// registers (TAPS x 4 floats), a filter row costs 4+S-1 reads of the patch for 4*S*4 fma (10 fma per LDS read instead
// of 2 with one pixel per thread: that kernel was LDS-bandwidth bound at 0.9 TB/s of input), and the 8 channel lanes
// of a pixel run meet in three shuffles.
constexpr int T1F_RB = 4, T1F_CB = 32, T1F_KC = 32, T1F_LDP = T1F_KC + 4;
template <int TAPS, int S_>
__global__ __launch_bounds__(256) void tconv1_fwd_kernel(const T1Desc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R_ = TAPS / S_;
  constexpr int PR = T1F_RB + R_ - 1, PC = T1F_CB + S_ - 1;   // staged patch of big pixels
  float* patch = smem;                                      // [PR*PC][T1F_LDP]
  const int t = threadIdx.x;
  // Block -> (image, row band, column block).  Neighbouring row bands share R-1 rows of the big map: dispatched in the
  // natural order (band fastest) they land on eight different XCDs and every L2 fetches the shared rows again (1.65x
  // the map's bytes from HBM).  The dispatcher deals blocks round-robin over the XCDs (MI355X_MICROARCH.md: blocks b and
  // b + 8 share one), so the j-th block an XCD receives takes band j % bands of image (j / bands) * 8 + xcd: the bands
  // of an image run back to back on ONE L2.  (Speed only: any placement computes the same thing.)
  int b = blockIdx.z, by = blockIdx.y, bx = blockIdx.x;
  if ((gridDim.z & 7) == 0) {
    const int per = gridDim.x * gridDim.y;
    const int lin = bx + gridDim.x * (by + gridDim.y * b);
    const int xcd = lin & 7, j = lin >> 3;
    const int il = j / per, rem = j - il * per;
    b = il * 8 + xcd;
    by = rem / gridDim.x;
    bx = rem - by * gridDim.x;
  }
  const int h0 = by * T1F_RB, w0 = bx * T1F_CB;
  // patch origin in the big map: row = h0 + pad - (R-1), col = w0 + pad - (S-1)
  const int pr0 = h0 + d.pad - (R_ - 1), pc0 = w0 + d.pad - (S_ - 1);
  const int k4 = t & 7, grp = t >> 3;                       // 32 groups = T1F_RB rows x 8 runs of 4 pixels
  const int gy = grp >> 3, gx = (grp & 7) * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < d.K; kc += T1F_KC) {
    __syncthreads();
    for (int i = t; i < PR * PC * (T1F_KC / 4); i += 256) {
      const int c4 = i % (T1F_KC / 4);
      const int pix = i / (T1F_KC / 4);
      const int pr = pix / PC, pc = pix - pr * PC;
      const int ih = pr0 + pr, iw = pc0 + pc;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)ih < (unsigned)d.P && (unsigned)iw < (unsigned)d.Q)
        v = *reinterpret_cast<const f32x4*>(d.big + ((long long)(b * d.P + ih) * d.Q + iw) * d.K + kc + c4 * 4);
      *reinterpret_cast<f32x4*>(patch + pix * T1F_LDP + c4 * 4) = v;
    }
    f32x4 wr[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) wr[tp] = *reinterpret_cast<const f32x4*>(d.w + tp * d.K + kc + k4 * 4);
    __syncthreads();
    f32x4 a4[4];
#pragma unroll
    for (int px = 0; px < 4; ++px) a4[px] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R_; ++r) {
      // output (gy, gx+px), tap (r, s) reads patch row gy + R-1-r, column gx + px + S-1-s
      const float* prow = patch + ((gy + R_ - 1 - r) * PC + gx) * T1F_LDP + k4 * 4;
      f32x4 xv[4 + S_ - 1];
#pragma unroll
      for (int c = 0; c < 4 + S_ - 1; ++c) xv[c] = *reinterpret_cast<const f32x4*>(prow + c * T1F_LDP);
#pragma unroll
      for (int s = 0; s < S_; ++s)
#pragma unroll
        for (int px = 0; px < 4; ++px) a4[px] += xv[px + S_ - 1 - s] * wr[r * S_ + s];
    }
#pragma unroll
    for (int px = 0; px < 4; ++px) acc[px] += (a4[px][0] + a4[px][1]) + (a4[px][2] + a4[px][3]);
  }
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    float v = acc[px];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    acc[px] = v;
  }
  if (k4 == 0) {
    const int h = h0 + gy;
    const float bias = d.bias ? d.bias[0] : 0.f;
    const float rs = d.rowscale ? d.rowscale[(long long)b * d.rowscale_ld] : 1.f;
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const int wq = w0 + gx + px;
      if (h < d.H && wq < d.W)
        d.out[((long long)(b * d.H + h) * d.W + wq) * d.ostride] = apply_act(acc[px] + bias, d.act, d.slope) * rs;
    }
  }
}

// ---------------------------------------------------------------- forward, scatter form on the matrix cores
// The gather form above re-reads the big map (R + RB - 1) / RB times and waits for memory twice per block: 1.1-1.8 TB/s.
// Scatter form: the contribution of input pixel p to the output through tap t is contrib[p][t] = sum_k big[p][k] * w[t][k]
// -- a [P*Q x K] x [K x taps] product with every row of the big map read exactly ONCE, straight from memory into MFMA
// operand registers (v_mfma_f32_16x16x4_f32: 16 pixels x 16 taps per tile, exact fp32).  One block owns one image: its
// contributions stay in LDS ([P*Q][LDC], 42-58 KB), and after a barrier every output pixel sums its <= R*S taps in a
// fixed order, adds the bias, applies the activation.  HBM traffic = the map once + the 1-channel output.
//   A (16 x 4 per MFMA): lane l holds pixel m = l % 16, k-slot l / 16; one 16-byte load gives a lane channels
//     16 j + 4 (l / 16) + e, e = 0..3 -- the k-slots of four MFMA steps (the weights use the same permutation);
//   B: lane l holds tap n = l % 16 of the same k-slot; all K/4 values per tap tile sit in registers for the whole block;
//   D: lane l holds tap n = l % 16 of pixels 4 (l / 16) + i, i = 0..3.
// NT = 16-tap tiles (1: up to 16 taps, 2: up to 32), KC = K / 16 (2 or 4).
using f32x4v = __attribute__((ext_vector_type(4))) float;
template <int NT, int KC, int NW>
__global__ __launch_bounds__(64 * NW) void tconv1_fwd_mfma_kernel(const T1Desc d, int LDC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* contrib = smem;                                  // [P*Q][LDC]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.x;
  const int PQ = d.P * d.Q, T = d.R * d.S, K = d.K;
  const int m = lane & 15, kq = lane >> 4;
  // weights: wreg[nt][j][e] = w[tap = 16 nt + m][channel 16 j + 4 kq + e]   (taps >= T: 0)
  float wreg[NT][KC][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int tap = nt * 16 + m;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      f32x4v v = {0.f, 0.f, 0.f, 0.f};
      if (tap < T) v = *reinterpret_cast<const f32x4v*>(d.w + (long long)tap * K + 16 * j + 4 * kq);
#pragma unroll
      for (int e = 0; e < 4; ++e) wreg[nt][j][e] = v[e];
    }
  }
  const float* src = d.big + (long long)b * PQ * K + 4 * kq;
  const int ntile = (PQ + 15) >> 4;
  auto load_tile = [&](int tile, f32x4v (&a)[KC]) {
    const int pix = tile * 16 + m;
    const bool ok = tile < ntile && pix < PQ;
    const float* p = src + (long long)(ok ? pix : 0) * K;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      a[j] = *reinterpret_cast<const f32x4v*>(p + 16 * j);
      if (!ok) a[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto compute_tile = [&](int tile, const f32x4v (&a)[KC]) {
    f32x4v acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][e], wreg[nt][j][e], acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int tap = nt * 16 + m;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pix = tile * 16 + 4 * kq + i;
        if (tap < T && pix < PQ) contrib[pix * LDC + tap] = acc[nt][i];
      }
    }
  };
  // tiles wave, wave + NW, ...: three tiles of loads in flight per wave (3 x K x 64 B = 12 KB at K = 64)
  f32x4v a0[KC], a1[KC], a2[KC];
  load_tile(wave, a0);
  load_tile(wave + NW, a1);
  load_tile(wave + 2 * NW, a2);
  for (int tile = wave; tile < ntile; tile += 3 * NW) {
    compute_tile(tile, a0);
    load_tile(tile + 3 * NW, a0);
    if (tile + NW < ntile) compute_tile(tile + NW, a1);
    load_tile(tile + 4 * NW, a1);
    if (tile + 2 * NW < ntile) compute_tile(tile + 2 * NW, a2);
    load_tile(tile + 5 * NW, a2);
  }
  __syncthreads();
  const float bias = d.bias ? d.bias[0] : 0.f;
  const float rs = d.rowscale ? d.rowscale[(long long)b * d.rowscale_ld] : 1.f;
  float* outb = d.out + (long long)b * d.H * d.W * d.ostride;
  for (int o = t; o < d.H * d.W; o += 64 * NW) {
    const int oh = o / d.W, ow = o - oh * d.W;
    float v = 0.f;
    for (int r = 0; r < d.R; ++r) {
      const int ih = oh + d.pad - r;
      if ((unsigned)ih >= (unsigned)d.P) continue;
      for (int sx = 0; sx < d.S; ++sx) {
        const int iw = ow + d.pad - sx;
        if ((unsigned)iw < (unsigned)d.Q) v += contrib[(ih * d.Q + iw) * LDC + r * d.S + sx];
      }
    }
    outb[(long long)o * d.ostride] = apply_act(v + bias, d.act, d.slope) * rs;
  }
}

// ---------------------------------------------------------------- transposed convolution to 1-2 channels, any stride
// The Generator tails of the spectrogram stacks (ConvTranspose2d(64 -> 1, 5, stride 2, pad 2, out_pad 1) on 64^2 ... 256^2
// maps, audio_mnist.py:243 and its whalecalls / esrf_acoustic copies) and the consumed input plane of their first
// Conv2d's data gradient (audio_mnist.py:186).  They used to be a 1x1 GEMM with N = taps columns writing a
// [pixels][taps] contribution tensor plus ali_col2im reading it back: 100 B per input pixel each way, 0.77 ms per ESRF
// launch for 0.25 ms of input.  Here the contributions never leave the CU -- the scatter form of tconv1_fwd_mfma_kernel,
// tiled: a block owns OBH x OBW output pixels of one image, multiplies the (OBH / stride + halo) x (OBW / stride + halo)
// input pixels that reach them by the [64 x N] tap matrix on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32;
// operand layout as above), keeps the products in LDS ([pixels][LDC]), and every output pixel then sums its taps in
// ali_col2im's order.  HBM traffic = the input once (+ ~20 % halo, mostly L2 hits) + the 1-2 channel output.
struct TSDesc {
  const float* x; const float* w; const float* bias; float* out;
  int B, H, W, Hout, Wout, NC, ostride, R, S, stride, pad, act; float slope;
  int OBH, OBW, ntx, N, LDC;
};

template <int NT>
__global__ __launch_bounds__(256) void tconv_scatter_kernel(const TSDesc d) {
  constexpr int KC = 4, NW = 4, K = 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* contrib = smem;                                  // [nh * nw][LDC]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.y;
  const int ty = blockIdx.x / d.ntx, tx = blockIdx.x - ty * d.ntx;
  const int o0 = ty * d.OBH, c0 = tx * d.OBW;
  const int st = d.stride, pad = d.pad, LDC = d.LDC, N = d.N;
  // input pixels that reach this tile: ih * st - pad + r in [o0, o0 + OBH) for some tap r
  auto lo = [&](int o, int taps) { const int v = o + pad - (taps - 1); return v <= 0 ? 0 : (v + st - 1) / st; };
  const int ih_lo = lo(o0, d.R), iw_lo = lo(c0, d.S);
  const int ih_hi = min(d.H - 1, (min(o0 + d.OBH, d.Hout) - 1 + pad) / st);
  const int iw_hi = min(d.W - 1, (min(c0 + d.OBW, d.Wout) - 1 + pad) / st);
  const int nh = max(ih_hi - ih_lo + 1, 0), nw = max(iw_hi - iw_lo + 1, 0);
  const int npx = nh * nw;
  const int m = lane & 15, kq = lane >> 4;
  // weights: wreg[nt][j][e] = w[column n = 16 nt + m][channel 16 j + 4 kq + e]   (columns >= N: 0)
  float wreg[NT][KC][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = nt * 16 + m;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      f32x4v v = {0.f, 0.f, 0.f, 0.f};
      if (n < N) v = *reinterpret_cast<const f32x4v*>(d.w + (long long)n * K + 16 * j + 4 * kq);
#pragma unroll
      for (int e = 0; e < 4; ++e) wreg[nt][j][e] = v[e];
    }
  }
  const float* src = d.x + ((long long)(b * d.H + ih_lo) * d.W + iw_lo) * K + 4 * kq;
  const int ntile = (npx + 15) >> 4;
  const float rnw = 1.0f / (float)max(nw, 1);
  auto load_tile = [&](int tile, f32x4v (&a)[KC]) {
    const int pix = tile * 16 + m;
    const bool ok = tile < ntile && pix < npx;
    int r = (int)((float)pix * rnw);                      // pix / nw, corrected: pix < 2^16
    r -= (r * nw > pix);
    r += ((r + 1) * nw <= pix);
    const int c = pix - r * nw;
    const float* p = src + ((long long)r * d.W + c) * K;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      a[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      if (ok) a[j] = *reinterpret_cast<const f32x4v*>(p + 16 * j);       // (a tile may have no input pixel at all)
    }
  };
  auto compute_tile = [&](int tile, const f32x4v (&a)[KC]) {
    f32x4v acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][e], wreg[nt][j][e], acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + m;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pix = tile * 16 + 4 * kq + i;
        if (n < N && pix < npx) contrib[pix * LDC + n] = acc[nt][i];
      }
    }
  };
  f32x4v a0[KC], a1[KC], a2[KC];
  load_tile(wave, a0);
  load_tile(wave + NW, a1);
  load_tile(wave + 2 * NW, a2);
  for (int tile = wave; tile < ntile; tile += 3 * NW) {
    compute_tile(tile, a0);
    load_tile(tile + 3 * NW, a0);
    if (tile + NW < ntile) compute_tile(tile + NW, a1);
    load_tile(tile + 4 * NW, a1);
    if (tile + 2 * NW < ntile) compute_tile(tile + 2 * NW, a2);
    load_tile(tile + 5 * NW, a2);
  }
  __syncthreads();
  const int NC = d.NC;
  const int obw_shift = 31 - __builtin_clz(d.OBW);        // OBW is a power of two
  for (int o = t; o < d.OBH * d.OBW; o += 256) {
    const int oy = o0 + (o >> obw_shift), ox = c0 + (o & (d.OBW - 1));
    if (oy >= d.Hout || ox >= d.Wout) continue;
    float acc0 = d.bias ? d.bias[0] : 0.f, acc1 = (d.bias && NC > 1) ? d.bias[1] : 0.f;
    const int r0 = (oy + pad) % st, s0 = (ox + pad) % st;
    for (int r = r0; r < d.R; r += st) {
      const int ih = (oy + pad - r) / st;
      if (oy + pad - r < 0 || ih >= d.H) continue;
      for (int sx = s0; sx < d.S; sx += st) {
        const int iw = (ox + pad - sx) / st;
        if (ox + pad - sx < 0 || iw >= d.W) continue;
        const float* cp = contrib + ((ih - ih_lo) * nw + (iw - iw_lo)) * LDC + (r * d.S + sx) * NC;
        acc0 += cp[0];
        if (NC > 1) acc1 += cp[1];
      }
    }
    float* op = d.out + ((long long)(b * d.Hout + oy) * d.Wout + ox) * d.ostride;
    op[0] = apply_act(acc0, d.act, d.slope);
    if (NC > 1) op[1] = apply_act(acc1, d.act, d.slope);
  }
}

// ---------------------------------------------------------------- data gradient: small (1 ch) -> big (K ch)
// One thread owns 4 channels (k4 = thread % (K/4): the grid stride is a multiple of K/4) and keeps their filter taps in
// registers (TAPS x 4 floats), so an output costs TAPS scalar reads of the 1-channel map and 4*TAPS fma -- no LDS traffic
// in the loop; the kernel then moves at the rate of its 4*K-byte output rows (+ the act' operand).
template <int TAPS>
__global__ __launch_bounds__(256) void tconv1_dgrad_kernel(const T1Desc d) {
  const int K4 = d.K / 4;
  const int k4 = threadIdx.x % K4;
  f32x4 wr[TAPS];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) wr[tp] = *reinterpret_cast<const f32x4*>(d.w + tp * d.K + k4 * 4);
  const long long total = (long long)d.B * d.P * d.Q * K4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long pix = i / K4;
    const int q = (int)(pix % d.Q);
    const long long t2 = pix / d.Q;
    const int p = (int)(t2 % d.P);
    const int b = (int)(t2 / d.P);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    const float* sb = d.small + (long long)b * d.H * d.W * d.sstride;
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
      const int r = tp / d.S, s = tp - r * d.S;
      const int ih = p + r - d.pad, iw = q + s - d.pad;
      const bool ok = (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
      const float g = ok ? sb[(long long)(ih * d.W + iw) * d.sstride] : 0.f;
      a += g * wr[tp];
    }
    const long long o = pix * d.K + k4 * 4;
    if (d.dact_y) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(d.dact_y + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= act_grad_from_output(y[e], d.dact, d.dslope);
    }
    *reinterpret_cast<f32x4*>(d.out + o) = a;
  }
}

// Band form (what the launcher uses when the zero-padded rows of the 1-channel map that a band of output rows reaches fit
// 32 KB of LDS -- always on the small maps these kernels serve): a block owns RB output rows of ONE image, the taps come
// from LDS (one ds_read_b32 per tap, the K/4 threads of a pixel read the same word) instead of 16-25 scalar memory loads
// per 16-byte store -- the kernel issued 18 memory instructions per KB it wrote and three 64-bit divisions per item
// (3.0 TB/s); now its memory instructions are the act' operand and the store, index arithmetic is 32-bit.
template <int TAPS>
__global__ __launch_bounds__(256) void tconv1_dgrad_band_kernel(const T1Desc d, int RB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int SW = d.Q + d.S - 1;                            // band image: entry (i, j) = small[p0 + i - pad][j - pad]
  const int t = threadIdx.x;
  const int K4 = d.K / 4;
  const int k4 = t % K4;
  f32x4 wr[TAPS];
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp) wr[tp] = *reinterpret_cast<const f32x4*>(d.w + tp * d.K + k4 * 4);
  const int b = blockIdx.y, p0 = blockIdx.x * RB;
  const int rows = min(RB, d.P - p0);
  const int SH = rows + d.R - 1;
  const float* sb = d.small + (long long)b * d.H * d.W * d.sstride;
  for (int i = t; i < SH * SW; i += 256) {
    const int sr = i / SW, sc = i - sr * SW;
    const int ih = p0 + sr - d.pad, iw = sc - d.pad;
    float v = 0.f;
    if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W) v = sb[(long long)(ih * d.W + iw) * d.sstride];
    smem[i] = v;
  }
  __syncthreads();
  const int ppt = 256 / K4;                               // pixels per pass of the block
  const int npix = rows * d.Q;
  const long long obase = ((long long)(b * d.P + p0) * d.Q) * d.K + k4 * 4;
  for (int pix = t / K4; pix < npix; pix += ppt) {
    const int p = pix / d.Q, q = pix - p * d.Q;
    const float* sp = smem + p * SW + q;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) {
      const int r = tp / d.S, sx = tp - r * d.S;
      a += sp[r * SW + sx] * wr[tp];
    }
    const long long o = obase + (long long)pix * d.K;
    if (d.dact_y) {
      const f32x4 y = *reinterpret_cast<const f32x4*>(d.dact_y + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) a[e] *= act_grad_from_output(y[e], d.dact, d.dslope);
    }
    *reinterpret_cast<f32x4*>(d.out + o) = a;
  }
}

// ---------------------------------------------------------------- weight gradient: partial [nblk][NC][K*T]
// A block walks over (image, band of T1W_RB big rows) work items and keeps its sums in registers, so only
// gridDim.x slabs have to be folded.  thread = (k, tap group g) owning taps g, g+G, ... for every small channel.
constexpr int T1W_RB = 4;
constexpr int T1W_MAXACC = 7;
constexpr int T1W_MAXNC = 8;
template <int NC>
__global__ __launch_bounds__(256) void tconv1_wgrad_kernel(const T1Desc d, int nitems, int bands) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = d.R * d.S;
  const int K = d.K;
  const int G = 256 / K;                 // host guarantees K in {32, 64, 128, 256}
  const int SR = T1W_RB + d.R - 1;       // small rows staged
  const int SW = d.Q + d.S - 1;          // small columns touched
  float* bigl = smem;                    // [T1W_RB*Q][K]
  float* sml = smem + T1W_RB * d.Q * K;  // [NC][SR][SW] zero padded
  const int t = threadIdx.x;
  const int k = t % K, g = t / K;
  float acc[NC][T1W_MAXACC];
  int toff[T1W_MAXACC];
#pragma unroll
  for (int a = 0; a < T1W_MAXACC; ++a) {
    const int tap = g + a * G;
    const int tp = tap < T ? tap : 0;
    toff[a] = (tp / d.S) * SW + (tp % d.S);
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c][a] = 0.f;
  }
  const int nacc = (T - g + G - 1) / G;  // taps owned by this thread
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / bands, p0 = (item % bands) * T1W_RB;
    const int rows = min(T1W_RB, d.P - p0);
    __syncthreads();
    for (int i = t; i < rows * d.Q * (K / 4); i += 256) {
      const int c4 = i % (K / 4);
      const int pix = i / (K / 4);
      const f32x4 v = *reinterpret_cast<const f32x4*>(d.big + ((long long)(b * d.P + p0) * d.Q + pix) * K + c4 * 4);
      *reinterpret_cast<f32x4*>(bigl + pix * K + c4 * 4) = v;
    }
    for (int i = t; i < NC * SR * SW; i += 256) {
      const int c = i / (SR * SW), rem = i % (SR * SW);
      const int sr = rem / SW, sc = rem % SW;
      const int ih = p0 + sr - d.pad, iw = sc - d.pad;
      float v = 0.f;
      if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
        v = d.small[((long long)(b * d.H + ih) * d.W + iw) * d.sstride + c];
      sml[i] = v;
    }
    __syncthreads();
    for (int pr = 0; pr < rows; ++pr) {
      for (int q = 0; q < d.Q; ++q) {
        const float bv = bigl[(pr * d.Q + q) * K + k];
        const float* sp = sml + pr * SW + q;
#pragma unroll
        for (int a = 0; a < T1W_MAXACC; ++a)
          if (a < nacc) {
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c][a] += bv * sp[c * SR * SW + toff[a]];
          }
      }
    }
  }
  float* part = d.part + (long long)blockIdx.x * (NC * K * T);
#pragma unroll
  for (int a = 0; a < T1W_MAXACC; ++a) {
    const int tap = g + a * G;
    if (tap < T) {
#pragma unroll
      for (int c = 0; c < NC; ++c) part[(c * K + k) * T + tap] = acc[c][a];
    }
  }
}

// Register-blocked variant (needs 256/K >= R): thread = (k, tap row r) keeps acc[NC][S]; per 4 consecutive pixels it
// reads 4 big values and, per small channel, one 8-float row segment, for 4*S*NC FMAs (7 FMAs per LDS read instead of 1).
template <int NC, int S_>
__global__ __launch_bounds__(256) void tconv1_wgrad_rb_kernel(const T1Desc d, int nitems, int bands) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K = d.K;
  const int SR = T1W_RB + d.R - 1;
  const int Q4 = (d.Q + 3) & ~3;
  const int SW = (Q4 + S_ - 1 + 3) & ~3;   // row stride of the staged small band (16-byte aligned rows)
  float* bigl = smem;                      // [T1W_RB][Q4][K], zero padded
  float* sml = smem + T1W_RB * Q4 * K;     // [NC][SR][SW], zero padded
  const int t = threadIdx.x;
  const int k = t % K, r = t / K;
  const bool active = r < d.R;
  float acc[NC][S_];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int s = 0; s < S_; ++s) acc[c][s] = 0.f;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int b = item / bands, p0 = (item % bands) * T1W_RB;
    const int rows = min(T1W_RB, d.P - p0);
    __syncthreads();
    for (int i = t; i < T1W_RB * Q4 * (K / 4); i += 256) {
      const int c4 = i % (K / 4);
      const int pix = i / (K / 4);
      const int pr = pix / Q4, q = pix % Q4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pr < rows && q < d.Q)
        v = *reinterpret_cast<const f32x4*>(d.big + ((long long)(b * d.P + p0 + pr) * d.Q + q) * K + c4 * 4);
      *reinterpret_cast<f32x4*>(bigl + pix * K + c4 * 4) = v;
    }
    for (int i = t; i < NC * SR * SW; i += 256) {
      const int c = i / (SR * SW), rem = i % (SR * SW);
      const int sr = rem / SW, sc = rem % SW;
      const int ih = p0 + sr - d.pad, iw = sc - d.pad;
      float v = 0.f;
      if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
        v = d.small[((long long)(b * d.H + ih) * d.W + iw) * d.sstride + c];
      sml[i] = v;
    }
    __syncthreads();
    if (active) {
      for (int pr = 0; pr < rows; ++pr) {
        for (int q = 0; q < Q4; q += 4) {
          float bv[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) bv[j] = bigl[((pr * Q4) + q + j) * K + k];
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const float* sp = sml + (c * SR + pr + r) * SW + q;
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp);
            const f32x4 s1 = *reinterpret_cast<const f32x4*>(sp + 4);
            const float seg[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
            for (int s = 0; s < S_; ++s)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[c][s] += bv[j] * seg[j + s];
          }
        }
      }
    }
  }
  if (active) {
    const int T = d.R * d.S;
    float* part = d.part + (long long)blockIdx.x * (NC * K * T);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int s = 0; s < S_; ++s) part[(c * K + k) * T + r * S_ + s] = acc[c][s];
  }
}

// ---------------------------------------------------------------- weight gradient on the matrix cores (K = 64, one small channel)
// dw[k][tap] = sum over pixels of big[pix][k] * small[pix + tap] is a [K x pixels] x [pixels x taps] product: per step of
// 4 pixels one v_mfma_f32_16x16x4_f32 per group of 16 channels.  The VALU kernels above are LDS-latency bound (1.3 TB/s
// of the big map); here a lane's ONE 16-byte load per step -- big[pix0 + l/16][4 (l%16) .. +3] -- is its A operand for
// the four channel groups {4 m + e} (e = 0..3), the 1-channel map sits zero-padded in LDS and gives the B operand with
// one ds_read_b32 (tap n = l % 16 of pixel pix0 + l/16), eight steps of loads are in flight per wave.  A block walks
// over images, keeps its sums in registers, folds its waves through LDS in a fixed order and leaves one slab.
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void tconv1_wgrad_mfma_kernel(const T1Desc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int K = 64;
  const int SH = d.P + d.R - 1, SW = d.Q + d.S - 1;      // zero-padded small image: entry (i, j) = small[i - pad][j - pad]
  float* simg = smem;                                     // [SH][SW]
  float* red = smem + ((SH * SW + 3) & ~3);               // [NW][4][NT][64][4] fold scratch
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int PQ = d.P * d.Q, T = d.R * d.S;
  int toff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int tap = nt * 16 + m;
    const int tp = tap < T ? tap : 0;
    toff[nt] = (tp / d.S) * SW + (tp % d.S);
  }
  f32x4v acc[4][NT];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[e][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const float rQ = 1.0f / (float)d.Q;
  const int nstep = (PQ + 3) >> 2;
  constexpr int DEPTH = 8;
  for (int b = blockIdx.x; b < d.B; b += gridDim.x) {
    __syncthreads();
    for (int i = t; i < SH * SW; i += 64 * NW) {
      const int sr = i / SW, sc = i - sr * SW;
      const int ih = sr - d.pad, iw = sc - d.pad;
      float v = 0.f;
      if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
        v = d.small[((long long)(b * d.H + ih) * d.W + iw) * d.sstride];
      simg[i] = v;
    }
    __syncthreads();
    const float* src = d.big + (long long)b * PQ * K + 4 * m;
    auto load_step = [&](int st) -> f32x4v {
      const int pix = st * 4 + kq;
      if (st < nstep && pix < PQ) return *reinterpret_cast<const f32x4v*>(src + (long long)pix * K);
      return f32x4v{0.f, 0.f, 0.f, 0.f};
    };
    f32x4v a[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) a[u] = load_step(wave + u * NW);
    for (int st0 = wave; st0 < nstep; st0 += DEPTH * NW) {
#pragma unroll
      for (int u = 0; u < DEPTH; ++u) {
        const int st = st0 + u * NW;
        const f32x4v av = a[u];
        a[u] = load_step(st + DEPTH * NW);
        if (st < nstep) {
          int pix = st * 4 + kq;
          if (pix >= PQ) pix = PQ - 1;                       // (its A values are zero)
          int p = (int)((float)pix * rQ);
          int q = pix - p * d.Q;
          if (q < 0) { --p; q += d.Q; } else if (q >= d.Q) { ++p; q -= d.Q; }
          const float* sp = simg + p * SW + q;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const float bv = sp[toff[nt]];
#pragma unroll
            for (int e = 0; e < 4; ++e)
              acc[e][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv, acc[e][nt], 0, 0, 0);
          }
        }
      }
    }
  }
  // fold the waves in a fixed order: acc[e][nt][i] = dw[channel 4 * (4 kq + i) + e][tap 16 nt + m]
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      *reinterpret_cast<f32x4v*>(red + (((wave * 4 + e) * NT + nt) * 64 + lane) * 4) = acc[e][nt];
  __syncthreads();
  float* part = d.part + (long long)blockIdx.x * (K * T);
  for (int o = t; o < 4 * NT * 64 * 4; o += 64 * NW) {
    const int i = o & 3, l = (o >> 2) & 63, nt = (o >> 8) % NT, e = (o >> 8) / NT;
    float v = 0.f;
    for (int w = 0; w < NW; ++w) v += red[(((w * 4 + e) * NT + nt) * 64 + l) * 4 + i];
    const int ch = 4 * (4 * (l >> 4) + i) + e, tap = nt * 16 + (l & 15);
    if (tap < T) part[ch * T + tap] = v;
  }
}

// Weight gradient of a STRIDED transposed convolution to one channel (the Generator tails of the spectrogram stacks,
// ConvTranspose2d(64 -> 1, 5, stride 2); audio_mnist.py:243):  dw[k][r*S+s] = sum_{b,p,q} big[b,p,q,k] * small[b, p*st - pad + r,
// q*st - pad + s].  As a GEMM it has ONE gathered channel (1/32 of a tile: 1.25 ms per ESRF launch for 0.25 ms of
// reading big once).  Same contraction over pixels as tconv1_wgrad_mfma_kernel -- one 16-byte load per lane per 4 pixels
// is the A operand of four channel groups, v_mfma_f32_16x16x4_f32 -- but a block owns a BAND of big rows of one image
// (the maps are 64^2 ... 256^2 x 64 channels: an image does not fit a block) and keeps the zero-padded rows of the small
// map that the band reaches in LDS.  One slab per block, folded by t1_reduce_kernel in block order.
struct TSWDesc {
  const float* big; const float* small; float* part;
  int B, P, Q, H, W, R, S, pad, stride, sstride, RB, bands;
};
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void tconvs_wgrad_mfma_kernel(const TSWDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int K = 64;
  const int st = d.stride;
  const int SW = (d.Q - 1) * st + d.S;                    // band image: entry (i, j) = small[p0 * st - pad + i][j - pad]
  const int SH = (d.RB - 1) * st + d.R;
  float* simg = smem;                                     // [SH][SW]
  float* red = smem;                                      // [NW][4][NT][64][4] fold scratch (the band image is dead by then)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m = lane & 15, kq = lane >> 4;
  const int T = d.R * d.S;
  const int b = blockIdx.x / d.bands, band = blockIdx.x - b * d.bands;
  const int p0 = band * d.RB;
  const int rows = min(d.RB, d.P - p0);
  const int npix = rows * d.Q;
  int toff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int tap = nt * 16 + m;
    const int tp = tap < T ? tap : 0;
    toff[nt] = (tp / d.S) * SW + (tp % d.S);
  }
  f32x4v acc[4][NT];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[e][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  for (int i = t; i < SH * SW; i += 64 * NW) {
    const int sr = i / SW, sc = i - sr * SW;
    const int ih = p0 * st - d.pad + sr, iw = sc - d.pad;
    float v = 0.f;
    if ((unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W)
      v = d.small[((long long)(b * d.H + ih) * d.W + iw) * d.sstride];
    simg[i] = v;
  }
  __syncthreads();
  const float rQ = 1.0f / (float)d.Q;
  const int nstep = (npix + 3) >> 2;
  constexpr int DEPTH = 8;
  const float* src = d.big + ((long long)(b * d.P + p0) * d.Q) * K + 4 * m;
  auto load_step = [&](int s4) -> f32x4v {
    const int pix = s4 * 4 + kq;
    if (s4 < nstep && pix < npix) return *reinterpret_cast<const f32x4v*>(src + (long long)pix * K);
    return f32x4v{0.f, 0.f, 0.f, 0.f};
  };
  f32x4v a[DEPTH];
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) a[u] = load_step(wave + u * NW);
  for (int s0 = wave; s0 < nstep; s0 += DEPTH * NW) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int s4 = s0 + u * NW;
      const f32x4v av = a[u];
      a[u] = load_step(s4 + DEPTH * NW);
      if (s4 < nstep) {
        int pix = s4 * 4 + kq;
        if (pix >= npix) pix = npix - 1;                       // (its A values are zero)
        int p = (int)((float)pix * rQ);
        int q = pix - p * d.Q;
        if (q < 0) { --p; q += d.Q; } else if (q >= d.Q) { ++p; q -= d.Q; }
        const float* sp = simg + (p * SW + q) * st;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float bv = sp[toff[nt]];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[e][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv, acc[e][nt], 0, 0, 0);
        }
      }
    }
  }
  // fold the waves in a fixed order: acc[e][nt][i] = dw[channel 4 * (4 kq + i) + e][tap 16 nt + m]
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      *reinterpret_cast<f32x4v*>(red + (((wave * 4 + e) * NT + nt) * 64 + lane) * 4) = acc[e][nt];
  __syncthreads();
  float* part = d.part + (long long)blockIdx.x * (K * T);
  for (int o = t; o < 4 * NT * 64 * 4; o += 64 * NW) {
    const int i = o & 3, l = (o >> 2) & 63, nt = (o >> 8) % NT, e = (o >> 8) / NT;
    float v = 0.f;
    for (int w = 0; w < NW; ++w) v += red[(((w * 4 + e) * NT + nt) * 64 + l) * 4 + i];
    const int ch = 4 * (4 * (l >> 4) + i) + e, tap = nt * 16 + (l & 15);
    if (tap < T) part[ch * T + tap] = v;
  }
}

// out[c*s_c + k*s_k + tap*s_tap] = sum_b part[b][(c*K + k)*T + tap]   (one wave per output, fixed order)
__global__ void t1_reduce_kernel(const float* __restrict__ part, int nblk, int K, int T, int NC, float* __restrict__ out,
                                 long long s_k, long long s_tap, long long s_c) {
  const int o = blockIdx.x;
  const int C = NC * K * T;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[(long long)b * C + o];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0) {
    const int tap = o % T, k = (o / T) % K, c = o / (T * K);
    out[(long long)c * s_c + (long long)k * s_k + (long long)tap * s_tap] = (float)s;
  }
}

static int t1_check(const char* what, int B, int P, int Q, int K, int H, int W, int R, int S, int pad) {
  if (B <= 0 || P <= 0 || Q <= 0 || K <= 0 || (K % 4) || H <= 0 || W <= 0 || R <= 0 || S <= 0 || R > 5 || S > 5 || pad < 0 ||
      H != P + R - 1 - 2 * pad || W != Q + S - 1 - 2 * pad) {
    set_error("%s: bad geometry", what);
    return ALI_ERR_BAD_ARG;
  }
  return ALI_OK;
}

}  // namespace ali

using namespace ali;

extern "C" int ali_tconv1_fwd(const float* big, const float* w_tk, const float* bias, float* out, int32_t B, int32_t P,
                              int32_t Q, int32_t K, int32_t R, int32_t S, int32_t pad, int32_t ostride, int32_t act,
                              float slope, const float* rowscale, int32_t rowscale_ld, ali_stream_t stream) {
  const int H = P + R - 1 - 2 * pad, W = Q + S - 1 - 2 * pad;
  if (!big || !w_tk || !out || ostride < 1) { set_error("ali_tconv1_fwd: bad argument"); return ALI_ERR_BAD_ARG; }
  int rc = t1_check("ali_tconv1_fwd", B, P, Q, K, H, W, R, S, pad);
  if (rc) return rc;
  T1Desc d = {};
  d.big = big; d.w = w_tk; d.bias = bias; d.out = out;
  d.B = B; d.P = P; d.Q = Q; d.K = K; d.H = H; d.W = W; d.R = R; d.S = S; d.pad = pad;
  d.ostride = ostride; d.act = act; d.slope = slope;
  d.rowscale = rowscale; d.rowscale_ld = rowscale_ld;
  if ((K % T1F_KC) != 0 || R != S) { set_error("ali_tconv1_fwd: needs K %% 32 == 0 and a square kernel"); return ALI_ERR_BAD_ARG; }
  {
    // scatter form on the matrix cores: one block per image, its tap contributions in LDS (maps up to ~32 x 32)
    const int T = R * S;
    const int LDC = (T <= 16 ? 17 : T) | 1;               // odd pitch: the gather's pixel-strided reads spread over the banks
    const size_t lds_sc = (size_t)P * Q * LDC * sizeof(float);
    if ((K == 32 || K == 64) && T <= 32 && lds_sc <= 64 * 1024 && tuning().no_t1_mfma == 0) {
#define T1M(NT_, KC_) hipLaunchKernelGGL((tconv1_fwd_mfma_kernel<NT_, KC_, 8>), dim3(B), dim3(512), lds_sc, (hipStream_t)stream, d, LDC)
      if (T <= 16 && K == 64) T1M(1, 4);
      else if (T <= 16) T1M(1, 2);
      else if (K == 64) T1M(2, 4);
      else T1M(2, 2);
#undef T1M
      return check_launch("tconv1_fwd_mfma_kernel");
    }
  }
  const size_t lds = (size_t)(T1F_RB + R - 1) * (T1F_CB + S - 1) * T1F_LDP * sizeof(float);
  dim3 grid((W + T1F_CB - 1) / T1F_CB, (H + T1F_RB - 1) / T1F_RB, B);
#define T1F(T_, S__) hipLaunchKernelGGL((tconv1_fwd_kernel<T_, S__>), grid, dim3(256), lds, (hipStream_t)stream, d)
  if (R == 5) T1F(25, 5);
  else if (R == 4) T1F(16, 4);
  else if (R == 3) T1F(9, 3);
  else if (R == 2) T1F(4, 2);
  else T1F(1, 1);
#undef T1F
  return check_launch("tconv1_fwd_kernel");
}

extern "C" int32_t ali_tconv_scatter_ok(int32_t C, int32_t NC, int32_t R, int32_t S, int32_t stride) {
  return (C == 64 && NC >= 1 && NC <= 2 && R >= 1 && S >= 1 && R <= 8 && S <= 8 && NC * R * S <= 64 && stride >= 1 &&
          stride <= 4 && tuning().no_t1_mfma == 0) ? 1 : 0;      // (ALI_NO_T1_MFMA=1: the two-launch form, A/B)
}

extern "C" int ali_tconv_scatter(const float* x, const float* w_nc, const float* bias, float* out, int32_t B, int32_t H,
                                 int32_t W, int32_t C, int32_t Hout, int32_t Wout, int32_t NC, int32_t ostride, int32_t R,
                                 int32_t S, int32_t stride, int32_t pad, int32_t act, float slope, ali_stream_t stream) {
  if (!x || !w_nc || !out || B <= 0 || B > 65535 || H <= 0 || W <= 0 || Hout <= 0 || Wout <= 0 || ostride < NC || pad < 0 ||
      !ali_tconv_scatter_ok(C, NC, R, S, stride)) {
    set_error("ali_tconv_scatter: bad argument (needs 64 input channels, 1-2 output channels, <= 64 tap columns)");
    return ALI_ERR_BAD_ARG;
  }
  TSDesc d = {};
  d.x = x; d.w = w_nc; d.bias = bias; d.out = out;
  d.B = B; d.H = H; d.W = W; d.Hout = Hout; d.Wout = Wout; d.NC = NC; d.ostride = ostride; d.R = R; d.S = S;
  d.stride = stride; d.pad = pad; d.act = act; d.slope = slope;
  d.N = NC * R * S;
  d.LDC = d.N | 1;
  // output tile: as large as 64 KB of contributions allow (two blocks per CU), at most 32 x 64
  int obh = 32, obw = 64;
  auto npx = [&](int bh, int bw) {
    return (long long)((bh - 1 + R - 1) / stride + 1) * ((bw - 1 + S - 1) / stride + 1);
  };
  while (npx(obh, obw) * d.LDC * 4 > 64 * 1024 && (obh > 4 || obw > 8)) {
    if (obw > 2 * obh && obw > 8) obw >>= 1; else if (obh > 4) obh >>= 1; else obw >>= 1;
  }
  if (npx(obh, obw) * d.LDC * 4 > 64 * 1024 || npx(obh, obw) >= 65536) { set_error("ali_tconv_scatter: filter too large"); return ALI_ERR_BAD_ARG; }
  d.OBH = obh; d.OBW = obw;
  d.ntx = (Wout + obw - 1) / obw;
  const long long nblk = (long long)d.ntx * ((Hout + obh - 1) / obh);
  if (nblk >= (1LL << 31)) { set_error("ali_tconv_scatter: output too large"); return ALI_ERR_BAD_ARG; }
  const size_t lds = (size_t)npx(obh, obw) * d.LDC * sizeof(float);
  dim3 grid((unsigned)nblk, B);
#define TSC(NT_) hipLaunchKernelGGL((tconv_scatter_kernel<NT_>), grid, dim3(256), lds, (hipStream_t)stream, d)
  if (d.N <= 16) TSC(1);
  else if (d.N <= 32) TSC(2);
  else if (d.N <= 48) TSC(3);
  else TSC(4);
#undef TSC
  return check_launch("tconv_scatter_kernel");
}

// (see tconvs_wgrad_mfma_kernel) returns the workspace bytes the launch needs, 0 when the shape is not served
extern "C" int64_t ali_tconv_scatter_wgrad_ws(int32_t B, int32_t P, int32_t Q, int32_t K, int32_t R, int32_t S,
                                              int32_t stride) {
  if (K != 64 || R < 1 || S < 1 || R * S > 32 || stride < 1 || stride > 4 || B <= 0 || P <= 0 || Q <= 0 ||
      tuning().no_t1_mfma != 0) return 0;
  int rb = 8;
  auto lds = [&](int r) {
    const long long img = (((long long)((r - 1) * stride + R) * ((Q - 1) * stride + S)) + 3) & ~3LL;
    return (size_t)std::max(img, 4LL * 4 * ((R * S + 15) / 16) * 64 * 4) * sizeof(float);
  };
  while (rb > 1 && lds(rb) > 64 * 1024) rb >>= 1;
  if (lds(rb) > 64 * 1024) return 0;
  const long long nblk = (long long)B * ((P + rb - 1) / rb);
  if (nblk > (1 << 20)) return 0;
  return (int64_t)nblk * K * R * S * (int64_t)sizeof(float) + (int64_t)kWsReserved;   // (incl. the reserved head)
}

extern "C" int ali_tconv_scatter_wgrad(const float* big, const float* small, int32_t sstride, float* dw, int64_t s_k,
                                       int64_t s_tap, int32_t B, int32_t P, int32_t Q, int32_t K, int32_t H, int32_t W,
                                       int32_t R, int32_t S, int32_t stride, int32_t pad, void* ws, size_t ws_bytes,
                                       ali_stream_t stream) {
  int64_t need = ali_tconv_scatter_wgrad_ws(B, P, Q, K, R, S, stride);
  if (need > 0) need -= (int64_t)kWsReserved;
  if (!big || !small || !dw || sstride < 1 || H <= 0 || W <= 0 || pad < 0 || need <= 0) {
    set_error("ali_tconv_scatter_wgrad: bad argument / unsupported shape (ali_tconv_scatter_wgrad_ws)");
    return ALI_ERR_BAD_ARG;
  }
  ws = ws_payload(ws);                       // (the head of every workspace holds the GEMM kernels' arrival counters)
  ws_bytes = ws_payload_bytes(ws_bytes);
  if (!ws || ws_bytes < (size_t)need) { set_error("ali_tconv_scatter_wgrad: workspace too small"); return ALI_ERR_WORKSPACE; }
  const int T = R * S;
  int rb = 8;
  const int mnt = (T + 15) / 16;
  auto lds = [&](int r) {
    const long long img = (((long long)((r - 1) * stride + R) * ((Q - 1) * stride + S)) + 3) & ~3LL;
    return (size_t)std::max(img, 4LL * 4 * mnt * 64 * 4) * sizeof(float);
  };
  while (rb > 1 && lds(rb) > 64 * 1024) rb >>= 1;
  TSWDesc d = {};
  d.big = big; d.small = small; d.part = reinterpret_cast<float*>(ws);
  d.B = B; d.P = P; d.Q = Q; d.H = H; d.W = W; d.R = R; d.S = S; d.pad = pad; d.stride = stride; d.sstride = sstride;
  d.RB = rb; d.bands = (P + rb - 1) / rb;
  const int nblk = B * d.bands;
  hipStream_t st = (hipStream_t)stream;
  if (mnt == 1) hipLaunchKernelGGL((tconvs_wgrad_mfma_kernel<1, 4>), dim3(nblk), dim3(256), lds(rb), st, d);
  else hipLaunchKernelGGL((tconvs_wgrad_mfma_kernel<2, 4>), dim3(nblk), dim3(256), lds(rb), st, d);
  hipLaunchKernelGGL(t1_reduce_kernel, dim3(K * T), dim3(64), 0, st, d.part, nblk, K, T, 1, dw, (long long)s_k,
                     (long long)s_tap, 0LL);
  return check_launch("tconvs_wgrad_mfma");
}

extern "C" int ali_tconv1_dgrad(const float* small, int32_t sstride, const float* w_tk, const float* dact_y,
                                int32_t dact, float dslope, float* gbig, int32_t B, int32_t P, int32_t Q, int32_t K,
                                int32_t R, int32_t S, int32_t pad, ali_stream_t stream) {
  const int H = P + R - 1 - 2 * pad, W = Q + S - 1 - 2 * pad;
  if (!small || !w_tk || !gbig || sstride < 1) { set_error("ali_tconv1_dgrad: bad argument"); return ALI_ERR_BAD_ARG; }
  int rc = t1_check("ali_tconv1_dgrad", B, P, Q, K, H, W, R, S, pad);
  if (rc) return rc;
  T1Desc d = {};
  d.small = small; d.sstride = sstride; d.w = w_tk; d.dact_y = dact_y; d.dact = dact; d.dslope = dslope; d.out = gbig;
  d.B = B; d.P = P; d.Q = Q; d.K = K; d.H = H; d.W = W; d.R = R; d.S = S; d.pad = pad;
  const long long total = (long long)B * P * Q * (K / 4);
  long long nb = (total + 255) / 256;
  if (nb > 4096) nb = 4096;              // (one output per thread was measured slower: 75 vs 54 us, the tap preload dominates)
  if (256 % (K / 4) != 0) { set_error("ali_tconv1_dgrad: K/4 must divide 256"); return ALI_ERR_BAD_ARG; }
  const int T = R * S;
  {
    // band form: RB output rows per block so that >= ~2 blocks per CU exist, LDS <= 32 KB
    int rbv = P;
    while (rbv > 1 && (long long)B * ((P + rbv - 1) / rbv) < 2 * kNumCU) rbv = (rbv + 1) / 2;
    const size_t lds = (size_t)(rbv + R - 1) * (Q + S - 1) * sizeof(float);
    if (lds <= 32 * 1024 && B <= 65535 && tuning().no_t1_mfma == 0) {
      dim3 grid((P + rbv - 1) / rbv, B);
#define T1B(T_) hipLaunchKernelGGL(tconv1_dgrad_band_kernel<T_>, grid, dim3(256), lds, (hipStream_t)stream, d, rbv)
      if (T == 16) T1B(16);
      else if (T == 25) T1B(25);
      else if (T == 9) T1B(9);
      else if (T == 4) T1B(4);
      else if (T == 1) T1B(1);
      else { set_error("ali_tconv1_dgrad: unsupported kernel size %dx%d", R, S); return ALI_ERR_BAD_ARG; }
#undef T1B
      return check_launch("tconv1_dgrad_band_kernel");
    }
  }
#define T1D(T_) hipLaunchKernelGGL(tconv1_dgrad_kernel<T_>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, d)
  if (T == 16) T1D(16);
  else if (T == 25) T1D(25);
  else if (T == 9) T1D(9);
  else if (T == 4) T1D(4);
  else if (T == 1) T1D(1);
  else { set_error("ali_tconv1_dgrad: unsupported kernel size %dx%d", R, S); return ALI_ERR_BAD_ARG; }
#undef T1D
  return check_launch("tconv1_dgrad_kernel");
}

extern "C" int ali_tconv1_wgrad(const float* big, const float* small, int32_t sstride, int32_t nc, float* dw, int64_t s_k,
                                int64_t s_tap, int64_t s_c, int32_t B, int32_t P, int32_t Q, int32_t K, int32_t R,
                                int32_t S, int32_t pad, void* ws, size_t ws_bytes, ali_stream_t stream) {
  ws = ws_payload(ws);
  ws_bytes = ws_payload_bytes(ws_bytes);
  const int H = P + R - 1 - 2 * pad, W = Q + S - 1 - 2 * pad;
  if (!big || !small || !dw || sstride < 1 || nc < 1 || nc > T1W_MAXNC || nc > sstride) {
    set_error("ali_tconv1_wgrad: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  int rc = t1_check("ali_tconv1_wgrad", B, P, Q, K, H, W, R, S, pad);
  if (rc) return rc;
  const int T = R * S;
  if (!(K == 32 || K == 64 || K == 128 || K == 256) || (T + 256 / K - 1) / (256 / K) > T1W_MAXACC) {
    set_error("ali_tconv1_wgrad: unsupported K=%d / taps=%d", K, T);
    return ALI_ERR_BAD_ARG;
  }
  const int bands = (P + T1W_RB - 1) / T1W_RB;
  const int nitems = bands * B;
  const int nblk = nitems < 8 * kNumCU ? nitems : 8 * kNumCU;   // (latency bound: as many resident blocks as LDS lets in)
  if (!ws || ws_bytes < (size_t)nblk * nc * K * T * sizeof(float)) { set_error("ali_tconv1_wgrad: workspace too small"); return ALI_ERR_WORKSPACE; }
  T1Desc d = {};
  d.big = big; d.small = small; d.sstride = sstride; d.part = reinterpret_cast<float*>(ws);
  d.B = B; d.P = P; d.Q = Q; d.K = K; d.H = H; d.W = W; d.R = R; d.S = S; d.pad = pad;
  hipStream_t st = (hipStream_t)stream;
  const int Q4 = (Q + 3) & ~3;
  const int SWp = (Q4 + S - 1 + 3) & ~3;
  const size_t lds_rb = ((size_t)T1W_RB * Q4 * K + (size_t)nc * (T1W_RB + R - 1) * SWp) * sizeof(float);
  const bool rb = (256 / K) >= R && S <= 5 && S >= 3 && lds_rb <= 64 * 1024 && (nc == 1 || nc == 5 || nc == 7);
  constexpr int MW = 8;                                   // waves per block of the matrix-core form
  const int mnt = (T + 15) / 16;
  const size_t lds_m = ((size_t)(((P + R - 1) * (Q + S - 1) + 3) & ~3) + (size_t)MW * 4 * mnt * 64 * 4) * sizeof(float);
  if (nc == 1 && K == 64 && T <= 32 && lds_m <= 64 * 1024 && tuning().no_t1_mfma == 0 &&
      ws_bytes >= (size_t)std::min(B, 2 * kNumCU) * K * T * sizeof(float)) {
    const int nb = std::min(B, 2 * kNumCU);
    if (mnt == 1) hipLaunchKernelGGL((tconv1_wgrad_mfma_kernel<1, MW>), dim3(nb), dim3(64 * MW), lds_m, st, d);
    else hipLaunchKernelGGL((tconv1_wgrad_mfma_kernel<2, MW>), dim3(nb), dim3(64 * MW), lds_m, st, d);
    hipLaunchKernelGGL(t1_reduce_kernel, dim3(K * T), dim3(64), 0, st, d.part, nb, K, T, 1, dw, (long long)s_k,
                       (long long)s_tap, (long long)s_c);
    return check_launch("tconv1_wgrad_mfma");
  }
  if (rb) {
#define WRB(NC_, S_) hipLaunchKernelGGL((tconv1_wgrad_rb_kernel<NC_, S_>), dim3(nblk), dim3(256), lds_rb, st, d, nitems, bands)
    if (nc == 1 && S == 3) WRB(1, 3); else if (nc == 1 && S == 4) WRB(1, 4); else if (nc == 1) WRB(1, 5);
    else if (nc == 5 && S == 3) WRB(5, 3); else if (nc == 5 && S == 4) WRB(5, 4); else if (nc == 5) WRB(5, 5);
    else if (nc == 7 && S == 3) WRB(7, 3); else if (nc == 7 && S == 4) WRB(7, 4); else WRB(7, 5);
#undef WRB
  } else {
  const size_t lds = ((size_t)T1W_RB * Q * K + (size_t)nc * (T1W_RB + R - 1) * (Q + S - 1)) * sizeof(float);
#define WG(NC_) hipLaunchKernelGGL((tconv1_wgrad_kernel<NC_>), dim3(nblk), dim3(256), lds, st, d, nitems, bands)
  switch (nc) {
    case 1: WG(1); break; case 2: WG(2); break; case 3: WG(3); break; case 4: WG(4); break;
    case 5: WG(5); break; case 6: WG(6); break; case 7: WG(7); break; default: WG(8); break;
  }
#undef WG
  }
  hipLaunchKernelGGL(t1_reduce_kernel, dim3(nc * K * T), dim3(64), 0, st, d.part, nblk, K, T, nc, dw, (long long)s_k,
                     (long long)s_tap, (long long)s_c);
  return check_launch("tconv1_wgrad");
}
