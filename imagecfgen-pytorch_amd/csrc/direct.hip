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
  const int b = blockIdx.z;
  const int h0 = blockIdx.y * T1F_RB, w0 = blockIdx.x * T1F_CB;
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
  if (nb > 4096) nb = 4096;
  if (256 % (K / 4) != 0) { set_error("ali_tconv1_dgrad: K/4 must divide 256"); return ALI_ERR_BAD_ARG; }
  const int T = R * S;
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
  const int nblk = nitems < 2 * kNumCU ? nitems : 2 * kNumCU;
  if (!ws || ws_bytes < (size_t)nblk * nc * K * T * sizeof(float)) { set_error("ali_tconv1_wgrad: workspace too small"); return ALI_ERR_WORKSPACE; }
  T1Desc d = {};
  d.big = big; d.small = small; d.sstride = sstride; d.part = reinterpret_cast<float*>(ws);
  d.B = B; d.P = P; d.Q = Q; d.K = K; d.H = H; d.W = W; d.R = R; d.S = S; d.pad = pad;
  hipStream_t st = (hipStream_t)stream;
  const int Q4 = (Q + 3) & ~3;
  const int SWp = (Q4 + S - 1 + 3) & ~3;
  const size_t lds_rb = ((size_t)T1W_RB * Q4 * K + (size_t)nc * (T1W_RB + R - 1) * SWp) * sizeof(float);
  const bool rb = (256 / K) >= R && S <= 5 && S >= 3 && lds_rb <= 64 * 1024 && (nc == 1 || nc == 5 || nc == 7);
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
