// Gather-GEMM: the implicit-GEMM kernel behind Conv2d forward, Conv2d data
// gradient and ConvTranspose2d forward (reference: image_scms/mnist.py:31-39,
// 64-72, 100-135 and the c2d/ct2d stacks of audio_mnist.py / whalecalls.py /
// esrf_acoustic.py).  fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32 -- the
// result is a k-ordered fp32 fma chain, so parity with the CPU reference is at
// rounding level.
//
//   C[m][n] = sum_{t in taps(phase(m))} sum_c  In[pix(m) + d_t][c] * Wp[n][wt_t][c]
//
// m enumerates output pixels phase by phase (stride-2 transposed convolutions are
// split into their 4 sub-pixel phases so no multiplication by inserted zeros is
// ever executed); k = (tap, channel) is flattened, channel fastest, which is the
// contiguous direction of both the NHWC activation and the packed weight.
//
// Tile: 256 threads = 4 waves; BM x BN x 32, double-buffered LDS ([row][k], rows
// padded to 36 floats => conflict-free ds_read_b128), one barrier per k-tile,
// global->register prefetch of tile k+1 issued before the MFMAs of tile k.
#include "ali_common.h"
#include <stdarg.h>
#include <string.h>

namespace ali {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // 36 floats = 144 B rows: 16-B aligned, b128 conflict-free

struct Phase {
  int Hq, Wq;          // extent of this phase's output sub-grid
  int oh0, ow0, ostep; // output pixel = (oh0 + qh*ostep, ow0 + qw*ostep)
  int mult;            // input pixel  = (qh*mult + dh, qw*mult + dw)
  int ntaps, tile0, M; // taps, first M-tile, rows in this phase
  signed char dh[kMaxTaps], dw[kMaxTaps];
  unsigned char wt[kMaxTaps];
};

struct GDesc {
  const float* in;
  const float* w;
  float* out;
  float* ws;
  AliEpilogue ep;
  int B, Hin, Win, Cin;
  int Hout, Wout, Cout, ldo;
  int ldw;
  int nphase, splitk, kt_per_split;
  long long out_elems;
  Phase ph[4];
};

template <int BM, int BN, int WAVES_M, int WAVES_N, bool VEC>
__global__ __launch_bounds__(256) void gconv_kernel(const GDesc d) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AP = BM / 32, BP = BN / 32;  // float4 loads per thread per tile
  static_assert(WAVES_M * WAVES_N == 4, "4 waves");

  __shared__ __attribute__((aligned(16))) float As[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDK];
  __shared__ long long s_rowoff[BM];
  __shared__ int s_rowimg[BM];
  __shared__ int s_tap[kMaxTaps];  // dh | dw<<8 | wt<<16

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  int p = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < d.nphase && (int)blockIdx.x >= d.ph[i].tile0) p = i;
  const Phase& P = d.ph[p];
  const int m0 = ((int)blockIdx.x - P.tile0) * BM;
  const int n0 = blockIdx.y * BN;
  const int Cin = d.Cin, Hin = d.Hin, Win = d.Win;
  const int Ktot = P.ntaps * Cin;
  const int nkt = (Ktot + BK - 1) / BK;
  const int kt_begin = blockIdx.z * d.kt_per_split;
  const int kt_end = min(nkt, kt_begin + d.kt_per_split);

  if (t < kMaxTaps) {
    int v = 0;
    if (t < P.ntaps) v = (P.dh[t] & 0xff) | ((P.dw[t] & 0xff) << 8) | ((int)P.wt[t] << 16);
    s_tap[t] = v;
  }
  for (int r = t; r < BM; r += 256) {
    int m = m0 + r;
    long long off = -1;
    int img = 0;
    if (m < P.M) {
      int qw = m % P.Wq;
      int t2 = m / P.Wq;
      int qh = t2 % P.Hq;
      img = t2 / P.Hq;
      off = ((long long)(img * d.Hout + P.oh0 + qh * P.ostep) * d.Wout + (P.ow0 + qw * P.ostep)) * d.ldo;
    }
    s_rowoff[r] = off;
    s_rowimg[r] = img;
  }

  // per-thread gather rows
  const int c4 = t & 7;
  const int r0 = t >> 3;
  long long abase[AP];
  int aih[AP], aiw[AP];
  bool avalid[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    int m = m0 + r0 + 32 * i;
    avalid[i] = m < P.M;
    int mm = avalid[i] ? m : 0;
    int qw = mm % P.Wq;
    int t2 = mm / P.Wq;
    int qh = t2 % P.Hq;
    int img = t2 / P.Hq;
    aih[i] = qh * P.mult;
    aiw[i] = qw * P.mult;
    abase[i] = ((long long)(img * Hin + aih[i]) * Win + aiw[i]) * Cin;
  }
  __syncthreads();

  f32x4 ra[AP], rb[BP];
  auto load_tile = [&](int kt) {
    const int kflat = kt * BK + c4 * 4;
    if (VEC) {
      const bool kvalid = kflat < Ktot;
      const int tap = kvalid ? kflat / Cin : 0;
      const int c = kflat - tap * Cin;
      const int tv = s_tap[tap];
      const int dh = (signed char)(tv & 0xff), dw = (signed char)((tv >> 8) & 0xff), wt = (tv >> 16) & 0xff;
      const long long doff = (long long)(dh * Win + dw) * Cin + c;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int ih = aih[i] + dh, iw = aiw[i] + dw;
        const bool ok = kvalid && avalid[i] && (unsigned)ih < (unsigned)Hin && (unsigned)iw < (unsigned)Win;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(d.in + abase[i] + doff);
        ra[i] = v;
      }
      const long long woff = (long long)wt * Cin + c;
#pragma unroll
      for (int j = 0; j < BP; ++j) {
        const int n = n0 + r0 + 32 * j;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kvalid && n < d.Cout) v = *reinterpret_cast<const f32x4*>(d.w + (long long)n * d.ldw + woff);
        rb[j] = v;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kf = kflat + e;
        const bool kvalid = kf < Ktot;
        const int tap = kvalid ? kf / Cin : 0;
        const int c = kf - tap * Cin;
        const int tv = s_tap[tap];
        const int dh = (signed char)(tv & 0xff), dw = (signed char)((tv >> 8) & 0xff), wt = (tv >> 16) & 0xff;
        const long long doff = (long long)(dh * Win + dw) * Cin + c;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          const int ih = aih[i] + dh, iw = aiw[i] + dw;
          const bool ok = kvalid && avalid[i] && (unsigned)ih < (unsigned)Hin && (unsigned)iw < (unsigned)Win;
          ra[i][e] = ok ? d.in[abase[i] + doff] : 0.f;
        }
        const long long woff = (long long)wt * Cin + c;
#pragma unroll
        for (int j = 0; j < BP; ++j) {
          const int n = n0 + r0 + 32 * j;
          rb[j][e] = (kvalid && n < d.Cout) ? d.w[(long long)n * d.ldw + woff] : 0.f;
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AP; ++i)
      *reinterpret_cast<f32x4*>(&As[buf][(r0 + 32 * i) * LDK + c4 * 4]) = ra[i];
#pragma unroll
    for (int j = 0; j < BP; ++j)
      *reinterpret_cast<f32x4*>(&Bs[buf][(r0 + 32 * j) * LDK + c4 * 4]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    const int lrow = lane & 31, lh = lane >> 5;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const bool has_next = kt + 1 < kt_end;
      if (has_next) load_tile(kt + 1);
      const float* Ab = &As[buf][(wm * WM + lrow) * LDK + lh * 4];
      const float* Bb = &Bs[buf][(wn * WN + lrow) * LDK + lh * 4];
#pragma unroll
      for (int kg = 0; kg < BK / 8; ++kg) {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK + kg * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK + kg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
      }
      if (has_next) store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // epilogue: acc(row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31)
  const AliEpilogue& ep = d.ep;
  const bool partial = d.splitk > 1;
  float* outp = partial ? d.ws + (long long)blockIdx.z * d.out_elems : d.out;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 32 + (lane & 31);
    if (n >= d.Cout) continue;
    const float bias = (!partial && ep.bias) ? ep.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const long long off = s_rowoff[row];
        if (off < 0) continue;
        float v = acc[i][j][r];
        if (!partial) {
          v = apply_act(v + bias, ep.act, ep.slope);
          if (ep.mask) v *= ep.mask[(long long)s_rowimg[row] * ep.mask_ld + n];
          if (ep.dact_y) v *= act_grad_from_output(ep.dact_y[off + n], ep.dact, ep.dslope);
        }
        outp[off + n] = v;
      }
    }
  }
}

// out = epilogue(sum_s ws[s]) over the flat NHWC output
__global__ void splitk_reduce_kernel(const float* __restrict__ ws, int S, long long out_elems, float* __restrict__ out,
                                     AliEpilogue ep, int ldo, int rows_per_img) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < out_elems; i += step) {
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += ws[(long long)s * out_elems + i];
    const int n = (int)(i % ldo);
    const long long pix = i / ldo;
    if (ep.bias) v += ep.bias[n];
    v = apply_act(v, ep.act, ep.slope);
    if (ep.mask) v *= ep.mask[(pix / rows_per_img) * ep.mask_ld + n];
    if (ep.dact_y) v *= act_grad_from_output(ep.dact_y[i], ep.dact, ep.dslope);
    out[i] = v;
  }
}

struct TileCfg { int bm, bn; };

static TileCfg pick_tile(long long M, int N) {
  TileCfg c;
  c.bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
  c.bm = 128;
  if (c.bn == 128 && M * (long long)((N + 127) / 128) < 128LL * 256) c.bm = 64;  // too few tiles: halve M tile
  if (c.bn == 64 && M < 128LL * 256) c.bm = 64;
  return c;
}

static int finalize_and_launch(GDesc& d, void* ws, size_t ws_bytes, hipStream_t stream, bool vec) {
  long long Mtot = 0;
  for (int i = 0; i < d.nphase; ++i) Mtot += d.ph[i].M;
  TileCfg tc = pick_tile(Mtot, d.Cout);
  if (!vec) { tc.bm = 128; if (tc.bn == 128) tc.bn = 64; }
  int tiles = 0, max_nkt = 0;
  for (int i = 0; i < d.nphase; ++i) {
    d.ph[i].tile0 = tiles;
    tiles += (d.ph[i].M + tc.bm - 1) / tc.bm;
    int nkt = (d.ph[i].ntaps * d.Cin + BK - 1) / BK;
    if (nkt > max_nkt) max_nkt = nkt;
  }
  const int ntile_n = (d.Cout + tc.bn - 1) / tc.bn;
  d.out_elems = (long long)d.B * d.Hout * d.Wout * d.ldo;
  // split-K when the grid cannot fill 256 CUs x 2 resident blocks
  int S = 1;
  const long long blocks = (long long)tiles * ntile_n;
  if (blocks < 2 * kNumCU && max_nkt >= 8) {
    S = (int)((2 * kNumCU + blocks - 1) / blocks);
    if (S > max_nkt / 4) S = max_nkt / 4;
    if (S > 32) S = 32;
    while (S > 1 && (size_t)S * d.out_elems * sizeof(float) > ws_bytes) --S;
    if (S < 1) S = 1;
  }
  d.splitk = S;
  d.kt_per_split = (max_nkt + S - 1) / S;
  if (d.kt_per_split < 1) d.kt_per_split = 1;
  d.ws = reinterpret_cast<float*>(ws);
  dim3 grid(tiles, ntile_n, S), block(256);
  if (tiles == 0 || d.out_elems == 0) return ALI_OK;
#define LAUNCH(BM_, BN_, WMM, WNN)                                                            \
  do {                                                                                          \
    if (vec) hipLaunchKernelGGL((gconv_kernel<BM_, BN_, WMM, WNN, true>), grid, block, 0, stream, d); \
    else hipLaunchKernelGGL((gconv_kernel<BM_, BN_, WMM, WNN, false>), grid, block, 0, stream, d);    \
  } while (0)
  if (tc.bm == 128 && tc.bn == 128) LAUNCH(128, 128, 2, 2);
  else if (tc.bm == 128 && tc.bn == 64) LAUNCH(128, 64, 2, 2);
  else if (tc.bm == 128 && tc.bn == 32) LAUNCH(128, 32, 4, 1);
  else if (tc.bm == 64 && tc.bn == 128) LAUNCH(64, 128, 2, 2);
  else LAUNCH(64, 64, 2, 2);
#undef LAUNCH
  int rc = check_launch("gconv_kernel");
  if (rc) return rc;
  if (S > 1) {
    long long n = d.out_elems;
    int nb = (int)((n + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, stream, d.ws, S, d.out_elems, d.out, d.ep, d.ldo,
                       d.Hout * d.Wout);
    rc = check_launch("splitk_reduce_kernel");
  }
  return rc;
}

static bool geom_ok(const AliConvGeom* g) {
  if (!g) return false;
  if (g->B <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->P <= 0 || g->Q <= 0 || g->K <= 0) return false;
  if (g->R <= 0 || g->S <= 0 || g->R * g->S > kMaxTaps || g->stride <= 0 || g->pad < 0) return false;
  if (g->pad > 100 || g->R > 100 || g->S > 100) return false;  // taps are stored as signed bytes
  return true;
}

static void fill_epilogue(GDesc& d, const AliEpilogue* ep) {
  if (ep) d.ep = *ep;
  else memset(&d.ep, 0, sizeof(d.ep));
}

}  // namespace ali

using namespace ali;

extern "C" const char* ali_last_error(void) { return ali::get_error(); }
extern "C" int ali_version(void) { return 1; }

extern "C" size_t ali_conv_workspace_bytes(const AliConvGeom* g, int32_t which) {
  if (!geom_ok(g)) return 0;
  if (which == 0) return (size_t)32 * g->B * g->P * g->Q * g->K * sizeof(float);
  if (which == 1) return (size_t)32 * g->B * g->H * g->W * g->C * sizeof(float);
  return (size_t)64 * g->R * g->S * g->C * g->K * sizeof(float);
}

extern "C" int ali_conv_fwd(const AliConvGeom* g, const float* x, const float* w, float* y, const AliEpilogue* ep,
                            void* ws, size_t ws_bytes, ali_stream_t stream) {
  if (!geom_ok(g) || !x || !w || !y) { set_error("ali_conv_fwd: bad argument"); return ALI_ERR_BAD_ARG; }
  GDesc d;
  memset(&d, 0, sizeof(d));
  d.in = x; d.w = w; d.out = y;
  fill_epilogue(d, ep);
  d.B = g->B; d.Hin = g->H; d.Win = g->W; d.Cin = g->C;
  d.Hout = g->P; d.Wout = g->Q; d.Cout = g->K; d.ldo = g->K;
  d.ldw = g->R * g->S * g->C;
  d.nphase = 1;
  Phase& P = d.ph[0];
  P.Hq = g->P; P.Wq = g->Q; P.oh0 = 0; P.ow0 = 0; P.ostep = 1; P.mult = g->stride;
  P.M = g->B * g->P * g->Q;
  P.ntaps = 0;
  for (int r = 0; r < g->R; ++r)
    for (int s = 0; s < g->S; ++s) {
      P.dh[P.ntaps] = (signed char)(r - g->pad);
      P.dw[P.ntaps] = (signed char)(s - g->pad);
      P.wt[P.ntaps] = (unsigned char)(r * g->S + s);
      ++P.ntaps;
    }
  return finalize_and_launch(d, ws, ws_bytes, (hipStream_t)stream, (g->C % 4) == 0);
}

extern "C" int ali_conv_bwd_data(const AliConvGeom* g, const float* dy, const float* w, float* dx,
                                 const AliEpilogue* ep, void* ws, size_t ws_bytes, ali_stream_t stream) {
  if (!geom_ok(g) || !dy || !w || !dx) { set_error("ali_conv_bwd_data: bad argument"); return ALI_ERR_BAD_ARG; }
  GDesc d;
  memset(&d, 0, sizeof(d));
  d.in = dy; d.w = w; d.out = dx;
  fill_epilogue(d, ep);
  d.B = g->B; d.Hin = g->P; d.Win = g->Q; d.Cin = g->K;
  d.Hout = g->H; d.Wout = g->W; d.Cout = g->C; d.ldo = g->C;
  d.ldw = g->R * g->S * g->K;
  const int st = g->stride;
  if (st > 2) { set_error("ali_conv_bwd_data: stride %d unsupported", st); return ALI_ERR_BAD_ARG; }
  d.nphase = 0;
  for (int ph = 0; ph < st; ++ph)
    for (int pw = 0; pw < st; ++pw) {
      const int Hq = (g->H - ph + st - 1) / st, Wq = (g->W - pw + st - 1) / st;
      if (Hq <= 0 || Wq <= 0) continue;
      Phase& P = d.ph[d.nphase++];
      P.Hq = Hq; P.Wq = Wq; P.oh0 = ph; P.ow0 = pw; P.ostep = st; P.mult = 1;
      P.M = g->B * Hq * Wq;
      P.ntaps = 0;
      for (int r = 0; r < g->R; ++r) {
        const int nh = ph + g->pad - r;
        if (((nh % st) + st) % st) continue;
        for (int s = 0; s < g->S; ++s) {
          const int nw = pw + g->pad - s;
          if (((nw % st) + st) % st) continue;
          // exact division (nh, nw are multiples of st)
          P.dh[P.ntaps] = (signed char)(nh / st);
          P.dw[P.ntaps] = (signed char)(nw / st);
          P.wt[P.ntaps] = (unsigned char)(r * g->S + s);
          ++P.ntaps;
        }
      }
    }
  return finalize_and_launch(d, ws, ws_bytes, (hipStream_t)stream, (g->K % 4) == 0);
}
