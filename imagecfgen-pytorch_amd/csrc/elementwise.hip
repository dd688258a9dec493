// HBM-bound pointwise / reduction kernels of the ALI step: weight re-layout,
// activation backward, bias-gradient column sums, Dropout2d masks, BatchNorm2d
// (training + eval, forward + backward), BCE-with-logits, Adam, and the
// conditioning-plane assembly.  Reference call sites are listed per entry point
// in include/ali_hip.h.  All reductions are deterministic (fixed partial slabs,
// fixed summation order; no float atomics).
#include "ali_common.h"
#include <algorithm>
#include <string.h>

namespace ali {

constexpr int kEwBlock = 256;
static inline int ew_grid(long long n, int per_thread = 1) {
  long long b = (n + (long long)kEwBlock * per_thread - 1) / ((long long)kEwBlock * per_thread);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int T, int C,
                                    int Cpad, long long s_n, long long s_tap, long long s_c) {
  const long long total = (long long)N * T * Cpad;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += step) {
    const int c = (int)(i % Cpad);
    const long long nt = i / Cpad;
    const int tp = (int)(nt % T);
    const long long n = nt / T;
    dst[i] = c < C ? src[n * s_n + tp * s_tap + c * s_c] : 0.f;
  }
}

// several re-layout jobs in one launch (the weight packs refreshed after an Adam step)
// Every re-layout of the path is, per batch entry b, a 2-D transpose in[b][i][j] -> out[b][j][i] with the out rows
// zero-padded from I to Ipad columns (Conv2d forward pack: b = n, i = c, j = tap; its data-gradient pack and the
// ConvTranspose2d forward pack: one batch, i = the channel that ends up contiguous, j = (other channel, tap)).  Such
// jobs go through 32 x 32 LDS tiles: 128-byte runs on both sides (the element-wise form reads with a stride of `taps`
// floats: 0.6 TB/s on the 1.3 GB of ESRF weights).  `dst16`: optional fp16 twin of dst, written alongside.
struct PackJob {
  const float* src; float* dst; _Float16* dst16;
  int N, T, C, Cpad; long long s_n, s_tap, s_c;
  int blk0;
  int tiled;                       // 1: batched transpose below (32 x 32 tiles); 2: 64 x 64 tiles, 16-byte accesses
  int batch, I, J, Ipad, ti, tj;   // tiles per batch entry: ti x tj
  long long in_b, in_i, out_b;     // in[b*in_b + i*in_i + j], out[b*out_b + j*out_j + i]   (out_j = Ipad for tiled == 1)
  long long out_j;
};
struct PackJobs { PackJob j[40]; int n; };
__global__ void __launch_bounds__(256) pack_weights_multi_kernel(PackJobs jobs) {
  __shared__ float tile[32][33];
  __shared__ float tile64[64][65];
  int k = 0;
  while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].blk0) ++k;
  const PackJob& J = jobs.j[k];
  if (J.tiled == 2) {
    // 64 x 64 tile, 16-byte loads along j (contiguous in the source) and 16-byte stores along i (contiguous in the
    // destination), 16 KB in flight per block: the data-gradient pack [C][T][K] of a master weight kept in the forward
    // pack's order [K][T][C] (FlatGroup.layouts) is, per tap, such a transpose (the element-wise form read it with a
    // stride of T*C floats from 256 blocks: 1 TB/s on the 7 GB of ESRF weights).
    int rel = (int)blockIdx.x - J.blk0;
    const int per = J.ti * J.tj;
    const int b = rel / per;
    rel -= b * per;
    const int it = rel / J.tj, jt = rel - it * J.tj;
    const int i0 = it * 64, j0 = jt * 64;
    const float* in = J.src + (long long)b * J.in_b;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = threadIdx.x + 256 * q, r = idx >> 4, c4 = idx & 15;
      const int i = i0 + r, j = j0 + 4 * c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < J.I && j < J.J) v = *reinterpret_cast<const f32x4*>(in + (long long)i * J.in_i + j);
      tile64[r][4 * c4 + 0] = v[0]; tile64[r][4 * c4 + 1] = v[1]; tile64[r][4 * c4 + 2] = v[2]; tile64[r][4 * c4 + 3] = v[3];
    }
    __syncthreads();
    float* out = J.dst + (long long)b * J.out_b;
    _Float16* out16 = J.dst16 ? J.dst16 + (long long)b * J.out_b : nullptr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = threadIdx.x + 256 * q, r = idx >> 4, c4 = idx & 15;
      const int j = j0 + r, i = i0 + 4 * c4;
      if (j < J.J && i < J.Ipad) {
        const f32x4 v = {tile64[4 * c4 + 0][r], tile64[4 * c4 + 1][r], tile64[4 * c4 + 2][r], tile64[4 * c4 + 3][r]};
        *reinterpret_cast<f32x4*>(out + (long long)j * J.out_j + i) = v;
        if (out16) {
          using h4 = __attribute__((ext_vector_type(4))) _Float16;
          *reinterpret_cast<h4*>(out16 + (long long)j * J.out_j + i) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
      }
    }
    return;
  }
  if (J.tiled) {
    int rel = (int)blockIdx.x - J.blk0;
    const int per = J.ti * J.tj;
    const int b = rel / per;
    rel -= b * per;
    const int it = rel / J.tj, jt = rel - it * J.tj;
    const int i0 = it * 32, j0 = jt * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
    const float* in = J.src + (long long)b * J.in_b;
    if (J.in_i == J.J && J.tj == 1) {
      // few columns (taps) and rows back to back in memory: the tile's rows are ONE contiguous run of rows*J floats
      const int rows = min(32, J.I - i0), n = rows > 0 ? rows * J.J : 0;
      for (int r = ty; r < 32; r += 8) tile[r][tx] = 0.f;
      __syncthreads();
      const float* run = in + (long long)i0 * J.J;
      for (int f = threadIdx.x; f < n; f += 256) {
        const int r = f / J.J;
        tile[r][f - r * J.J] = run[f];
      }
    } else {
      for (int r = ty; r < 32; r += 8) {                                 // rows i, columns j (contiguous in src)
        const int i = i0 + r, j = j0 + tx;
        tile[r][tx] = (i < J.I && j < J.J) ? in[(long long)i * J.in_i + j] : 0.f;
      }
    }
    __syncthreads();
    float* out = J.dst + (long long)b * J.out_b;
    _Float16* out16 = J.dst16 ? J.dst16 + (long long)b * J.out_b : nullptr;
    for (int r = ty; r < 32; r += 8) {                                   // rows j, columns i (contiguous in dst)
      const int j = j0 + r, i = i0 + tx;
      if (j < J.J && i < J.Ipad) {
        const float v = tile[tx][r];                                     // zero where i >= I (padding channels)
        out[(long long)j * J.Ipad + i] = v;
        if (out16) out16[(long long)j * J.Ipad + i] = (_Float16)v;
      }
    }
    return;
  }
  const int nblk = (k + 1 < jobs.n ? jobs.j[k + 1].blk0 : (int)gridDim.x) - J.blk0;
  const long long total = (long long)J.N * J.T * J.Cpad;
  long long i = (long long)((int)blockIdx.x - J.blk0) * blockDim.x + threadIdx.x;
  const long long step = (long long)nblk * blockDim.x;
  for (; i < total; i += step) {
    const int c = (int)(i % J.Cpad);
    const long long nt = i / J.Cpad;
    const int tp = (int)(nt % J.T);
    const long long n = nt / J.T;
    const float v = c < J.C ? J.src[n * J.s_n + tp * J.s_tap + c * J.s_c] : 0.f;
    J.dst[i] = v;
    if (J.dst16) J.dst16[i] = (_Float16)v;
  }
}

__global__ void act_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gpre,
                               long long n, int act, float slope) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += step) gpre[i] = gy[i] * act_grad_from_output(y[i], act, slope);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

// ---- column sums: stage 1 writes [nblk][C] partials, stage 2 folds them ----
// block = 256 threads as (256/CT) row lanes x CT channel lanes, CT = min(C,256) rounded to pow2 lanes
__global__ void colsum_partial_kernel(const float* __restrict__ x, long long rows, int C, int ld,
                                      float* __restrict__ part) {
  __shared__ float red[kEwBlock];
  const int t = threadIdx.x;
  for (int c0 = 0; c0 < C; c0 += kEwBlock) {
    const int cw = min(C - c0, kEwBlock);         // channels handled this sweep
    int lanes_c = 1;
    while (lanes_c < cw) lanes_c <<= 1;           // pow2 >= cw, <= 256
    const int lanes_r = kEwBlock / lanes_c;
    const int tc = t % lanes_c, tr = t / lanes_c;
    float s = 0.f;
    if (tc < cw)
      for (long long r = (long long)blockIdx.x * lanes_r + tr; r < rows; r += (long long)gridDim.x * lanes_r)
        s += x[r * ld + c0 + tc];
    red[t] = s;
    __syncthreads();
    for (int off = lanes_r / 2; off > 0; off >>= 1) {
      if (tr < off) red[t] += red[t + off * lanes_c];
      __syncthreads();
    }
    if (tr == 0 && tc < cw) part[(long long)blockIdx.x * C + c0 + tc] = red[t];
    __syncthreads();
  }
}
// one 64-lane wave per channel: lanes stride over the partial slabs, fixed-order wave reduction in double
__global__ void colsum_final_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ out) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += (double)part[(long long)b * C + c];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[c] = (float)s;
}

__global__ void rowmask_mul_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                   float* __restrict__ out, long long n, int rows_per_img, int C) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  const long long per_img = (long long)rows_per_img * C;
  for (; i < n; i += step) {
    const long long img = i / per_img;
    out[i] = x[i] * mask[img * C + (int)(i % C)];
  }
}

// splitmix64-based counter RNG: one draw per (seed, offset + i)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void dropout_mask_kernel(uint64_t seed, uint64_t offset, const long long* __restrict__ dev_counter, float p,
                                    float* __restrict__ out, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  const float keep = 1.f - p, inv = 1.f / (1.f - p);
  const uint64_t key = mix64(mix64(seed) ^ (dev_counter ? (uint64_t)dev_counter[0] * 0xD1B54A32D192ED03ull : 0ull));
  for (; i < n; i += step) {
    const uint64_t r = mix64(key ^ (offset + (uint64_t)i));
    const float u = (float)(r >> 40) * (1.f / 16777216.f);
    out[i] = u < keep ? inv : 0.f;
  }
}

// all Dropout2d masks of one iteration in one launch: blockIdx.y = segment (one [rows][cpad] mask), blockIdx.x strides
// over its elements.  Column c < clog of row b takes the draw dropout_mask_kernel(offset = draw_start) gives index
// b*clog + c (so the masks are bit-identical to one launch per mask); padding columns are 1.
struct MaskSegs { long long out_end[64]; long long draw_start[64]; float p[64]; int clog[64]; int cpad[64]; int n; };
__global__ void dropout_mask_multi_kernel(uint64_t seed, const long long* __restrict__ dev_counter, MaskSegs segs,
                                          float* __restrict__ out) {
  const int s = blockIdx.y;
  const long long lo = s ? segs.out_end[s - 1] : 0, n = segs.out_end[s] - lo;
  const long long d0 = segs.draw_start[s];
  const int clog = segs.clog[s], cpad = segs.cpad[s];
  const float p = segs.p[s], keep = 1.f - p, inv = 1.f / (1.f - p);
  const uint64_t key = mix64(mix64(seed) ^ (dev_counter ? (uint64_t)dev_counter[0] * 0xD1B54A32D192ED03ull : 0ull));
  const long long step = (long long)gridDim.x * blockDim.x;
  float* o = out + lo;
  for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += step) {
    long long di = j;
    if (clog != cpad) {
      const long long b = j / cpad;
      const int c = (int)(j - b * cpad);
      if (c >= clog) { o[j] = 1.f; continue; }
      di = b * clog + c;
    }
    const uint64_t r = mix64(key ^ (uint64_t)(d0 + di));
    const float u = (float)(r >> 40) * (1.f / 16777216.f);
    o[j] = u < keep ? inv : 0.f;
  }
}

// ---- BatchNorm2d ------------------------------------------------------------
// stage 1: per-block partial (sum, sumsq) of x~ = x*mask per channel -> part[nblk][2][C]
__global__ void bn_stats_partial_kernel(const float* __restrict__ x, const float* __restrict__ mask, long long rows,
                                        int rows_per_img, int C, float* __restrict__ part) {
  __shared__ float red1[kEwBlock], red2[kEwBlock];
  const int t = threadIdx.x;
  int lanes_c = 1;
  while (lanes_c < C) lanes_c <<= 1;
  const int lanes_r = kEwBlock / lanes_c;  // C <= 256 enforced by the host
  const int tc = t % lanes_c, tr = t / lanes_c;
  float s1 = 0.f, s2 = 0.f;
  if (tc < C)
    for (long long r = (long long)blockIdx.x * lanes_r + tr; r < rows; r += (long long)gridDim.x * lanes_r) {
      float v = x[r * C + tc];
      if (mask) v *= mask[(r / rows_per_img) * C + tc];
      s1 += v;
      s2 += v * v;
    }
  red1[t] = s1;
  red2[t] = s2;
  __syncthreads();
  for (int off = lanes_r / 2; off > 0; off >>= 1) {
    if (tr < off) {
      red1[t] += red1[t + off * lanes_c];
      red2[t] += red2[t + off * lanes_c];
    }
    __syncthreads();
  }
  if (tr == 0 && tc < C) {
    part[((long long)blockIdx.x * 2 + 0) * C + tc] = red1[t];
    part[((long long)blockIdx.x * 2 + 1) * C + tc] = red2[t];
  }
}
// partial (b, s, c) of group gi lives at part[gi*lay.g + b*lay.b + s*lay.s + c*lay.c]: [nblk][2][C] blocks per group for
// the reduction kernels above, [2][C][slots] (slots of a group contiguous) for partials left by a convolution epilogue
struct PartLayout { long long g, b, s, c; };
// fixed-order sum of a block's per-thread doubles (256 threads): thread 0 gets the total
__device__ __forceinline__ double block_sum_256(double v, double* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const double tot = ((red[0] + red[1]) + red[2]) + red[3];
  __syncthreads();
  return tot;
}
__global__ void __launch_bounds__(kEwBlock)
bn_stats_final_kernel(const float* __restrict__ part, PartLayout lay, int nblk, int C, long long count,
                      const float* __restrict__ gamma, const float* __restrict__ beta,
                      float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                      int training, float* __restrict__ mean_out, float* __restrict__ invstd_out, float* __restrict__ sc,
                      float* __restrict__ sh, int groups, long long stat_stride) {
  __shared__ double red[8];
  const int c = blockIdx.x;   // one block per channel; the groups (batched passes) update the running stats in order
  for (int gi = 0; gi < groups; ++gi) {
    const float* pg = part + (long long)gi * lay.g + (long long)c * lay.c;
    float mean, var;
    if (training) {
      double s1 = 0.0, s2 = 0.0;
      for (int b = threadIdx.x; b < nblk; b += kEwBlock) {
        s1 += (double)pg[(long long)b * lay.b];
        s2 += (double)pg[(long long)b * lay.b + lay.s];
      }
      s1 = block_sum_256(s1, red);
      s2 = block_sum_256(s2, red + 4);
      const double m = s1 / (double)count;
      double v = s2 / (double)count - m * m;
      if (v < 0.0) v = 0.0;
      mean = (float)m;
      var = (float)v;
      if (running_mean && threadIdx.x == 0) {
        const double unb = count > 1 ? v * (double)count / (double)(count - 1) : v;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    } else {
      mean = running_mean[c];
      var = running_var[c];
    }
    if (threadIdx.x == 0) {
      const float invstd = 1.f / sqrtf(var + eps);
      const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
      const long long o = (long long)gi * stat_stride + c;
      mean_out[o] = mean;
      invstd_out[o] = invstd;
      sc[o] = g * invstd;
      sh[o] = b - mean * g * invstd;
    }
  }
}
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                const float* __restrict__ mask_in, const float* __restrict__ mask_post,
                                float* __restrict__ out, long long n, int rows_per_img, int C) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  const long long per_img = (long long)rows_per_img * C;
  for (; i < n; i += step) {
    const int c = (int)(i % C);
    const long long img = i / per_img;
    float v = x[i];
    if (mask_in) v *= mask_in[img * C + c];
    v = v * sc[c] + sh[c];
    if (mask_post) v *= mask_post[img * C + c];
    out[i] = v;
  }
}
// backward stage 1: part[nblk][2][C] = (sum g~*xhat, sum g~)
__global__ void bn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                      const float* __restrict__ mask_in, const float* __restrict__ mask_pre,
                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                      long long rows, int rows_per_img, int C, float* __restrict__ part) {
  __shared__ float red1[kEwBlock], red2[kEwBlock];
  const int t = threadIdx.x;
  int lanes_c = 1;
  while (lanes_c < C) lanes_c <<= 1;
  const int lanes_r = kEwBlock / lanes_c;
  const int tc = t % lanes_c, tr = t / lanes_c;
  float s1 = 0.f, s2 = 0.f;
  if (tc < C) {
    const float mu = mean[tc], is = invstd[tc];
    for (long long r = (long long)blockIdx.x * lanes_r + tr; r < rows; r += (long long)gridDim.x * lanes_r) {
      const long long img = r / rows_per_img;
      float xv = x[r * C + tc];
      if (mask_in) xv *= mask_in[img * C + tc];
      float gv = g[r * C + tc];
      if (mask_pre) gv *= mask_pre[img * C + tc];
      s1 += gv * (xv - mu) * is;
      s2 += gv;
    }
  }
  red1[t] = s1;
  red2[t] = s2;
  __syncthreads();
  for (int off = lanes_r / 2; off > 0; off >>= 1) {
    if (tr < off) {
      red1[t] += red1[t + off * lanes_c];
      red2[t] += red2[t + off * lanes_c];
    }
    __syncthreads();
  }
  if (tr == 0 && tc < C) {
    part[((long long)blockIdx.x * 2 + 0) * C + tc] = red1[t];
    part[((long long)blockIdx.x * 2 + 1) * C + tc] = red2[t];
  }
}
__global__ void __launch_bounds__(kEwBlock)
bn_bwd_final_kernel(const float* __restrict__ part, PartLayout lay, int nblk, int C, float* __restrict__ dgamma,
                    float* __restrict__ dbeta) {
  __shared__ double red[8];
  const int c = blockIdx.x;   // one block per channel
  double s1 = 0.0, s2 = 0.0;
  const float* pg = part + (long long)c * lay.c;
  for (int b = threadIdx.x; b < nblk; b += kEwBlock) {
    s1 += (double)pg[(long long)b * lay.b];
    s2 += (double)pg[(long long)b * lay.b + lay.s];
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red + 4);
  if (threadIdx.x == 0) {
    dgamma[c] = (float)s1;
    dbeta[c] = (float)s2;
  }
}
__global__ void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                    const float* __restrict__ mask_in, const float* __restrict__ mask_pre,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta, long long n, int rows_per_img, int C,
                                    float inv_count, int batch_stats, float slope, float* __restrict__ gx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  const long long per_img = (long long)rows_per_img * C;
  for (; i < n; i += step) {
    const int c = (int)(i % C);
    const long long img = i / per_img;
    const float xraw = x[i];
    float xv = xraw;
    float mi = 1.f;
    if (mask_in) { mi = mask_in[img * C + c]; xv *= mi; }
    float gv = g[i];
    if (mask_pre) gv *= mask_pre[img * C + c];
    const float is = invstd[c];
    const float gm = gamma ? gamma[c] : 1.f;
    float r = gv;
    if (batch_stats) r -= (dbeta[c] + (xv - mean[c]) * is * dgamma[c]) * inv_count;
    r *= gm * is;
    r *= mi;
    if (slope >= 0.f) r *= (xraw > 0.f ? 1.f : slope);
    gx[i] = r;
  }
}

// ---- float4 variants of the BatchNorm / column-sum passes (C % 4 == 0, < 2^31 elements): 16-byte accesses, 32-bit
// index arithmetic, four rows in flight per thread.  Same partial layouts as the scalar kernels above.
// MODE 0: (sum x~, sum x~^2)   MODE 1: (sum g~*xhat, sum g~)   MODE 2: column sums of x (row stride ld)
template <int MODE>
__global__ void __launch_bounds__(kEwBlock)
rowreduce_vec_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ mask_in,
                     const float* __restrict__ mask_pre, const float* __restrict__ mean,
                     const float* __restrict__ invstd, unsigned rows, unsigned rows_per_img, unsigned C, unsigned ld,
                     float* __restrict__ part) {
  __shared__ float4 red1[kEwBlock], red2[kEwBlock];
  // blockIdx.y = group (independent passes batched along the sample axis: `rows` rows each, own partial block range)
  x += (size_t)blockIdx.y * rows * ld;
  if (mask_in) mask_in += (size_t)blockIdx.y * (rows / rows_per_img) * C;
  part += (size_t)blockIdx.y * gridDim.x * (MODE == 2 ? 1 : 2) * C;
  const unsigned t = threadIdx.x;
  const unsigned C4 = C >> 2, lanes_r = kEwBlock / C4;
  const unsigned tr = t / C4, tc = (t - tr * C4) * 4;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (tr < lanes_r) {
    float4 mu = s1, is = s1;
    if (MODE == 1) {
      mu = *reinterpret_cast<const float4*>(mean + tc);
      is = *reinterpret_cast<const float4*>(invstd + tc);
    }
    const unsigned stride = gridDim.x * lanes_r;
    for (unsigned r0 = blockIdx.x * lanes_r + tr; r0 < rows; r0 += 4 * stride) {
      float4 xv[4], gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned r = r0 + u * stride;
        xv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        gv[u] = xv[u];
        if (r < rows) {
          xv[u] = *reinterpret_cast<const float4*>(x + (size_t)r * ld + tc);
          if (MODE == 1) gv[u] = *reinterpret_cast<const float4*>(g + (size_t)r * ld + tc);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned r = r0 + u * stride;
        if (r >= rows) break;
        float4 a = xv[u];
        if (MODE != 2 && (mask_in || mask_pre)) {
          const unsigned img = r / rows_per_img;
          if (mask_in) {
            const float4 m = *reinterpret_cast<const float4*>(mask_in + (size_t)img * C + tc);
            a.x *= m.x; a.y *= m.y; a.z *= m.z; a.w *= m.w;
          }
          if (MODE == 1 && mask_pre) {
            const float4 m = *reinterpret_cast<const float4*>(mask_pre + (size_t)img * C + tc);
            gv[u].x *= m.x; gv[u].y *= m.y; gv[u].z *= m.z; gv[u].w *= m.w;
          }
        }
        if (MODE == 0) {
          s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
          s2.x += a.x * a.x; s2.y += a.y * a.y; s2.z += a.z * a.z; s2.w += a.w * a.w;
        } else if (MODE == 1) {
          const float4 b = gv[u];
          s1.x += b.x * (a.x - mu.x) * is.x; s1.y += b.y * (a.y - mu.y) * is.y;
          s1.z += b.z * (a.z - mu.z) * is.z; s1.w += b.w * (a.w - mu.w) * is.w;
          s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
        } else {
          s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
        }
      }
    }
  }
  red1[t] = s1;
  if (MODE != 2) red2[t] = s2;
  __syncthreads();
  if (tr == 0) {
    for (unsigned j = 1; j < lanes_r; ++j) {
      const float4 a = red1[j * C4 + t];
      s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
      if (MODE != 2) {
        const float4 b = red2[j * C4 + t];
        s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
      }
    }
    if (MODE == 2) {
      *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * C + tc) = s1;
    } else {
      *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 0) * C + tc) = s1;
      *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + 1) * C + tc) = s2;
    }
  }
}

// elementwise passes: thread (tr, tc) owns channels [4tc, 4tc+4) of rows tr, tr+stride, ...: the per-channel
// constants are loaded once, the only division left is row -> image for the Dropout2d masks
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ldc4(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }  // any alignment

__global__ void __launch_bounds__(kEwBlock)
bn_apply_vec_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                    const float* __restrict__ mask_in, const float* __restrict__ mask_post, float* __restrict__ out,
                    unsigned rows, unsigned rows_per_img, unsigned C, long long stat_stride) {
  const unsigned C4 = C >> 2, lanes_r = kEwBlock / C4;
  const unsigned tr = threadIdx.x / C4, c = (threadIdx.x - tr * C4) * 4;
  if (tr >= lanes_r) return;
  // blockIdx.y = group: its own rows, masks and (scale, shift)
  x += (size_t)blockIdx.y * rows * C;
  out += (size_t)blockIdx.y * rows * C;
  if (mask_in) mask_in += (size_t)blockIdx.y * (rows / rows_per_img) * C;
  if (mask_post) mask_post += (size_t)blockIdx.y * (rows / rows_per_img) * C;
  sc += blockIdx.y * stat_stride;
  sh += blockIdx.y * stat_stride;
  const float4 a = ldc4(sc + c), b = ldc4(sh + c);
  const unsigned stride = gridDim.x * lanes_r;
  for (unsigned r = blockIdx.x * lanes_r + tr; r < rows; r += stride) {
    float4 v = ld4(x + (size_t)r * C + c);
    const unsigned img = (mask_in || mask_post) ? r / rows_per_img : 0u;
    if (mask_in) v = mul4(v, ld4(mask_in + (size_t)img * C + c));
    v.x = v.x * a.x + b.x; v.y = v.y * a.y + b.y; v.z = v.z * a.z + b.z; v.w = v.w * a.w + b.w;
    if (mask_post) v = mul4(v, ld4(mask_post + (size_t)img * C + c));
    *reinterpret_cast<float4*>(out + (size_t)r * C + c) = v;
  }
}

__global__ void __launch_bounds__(kEwBlock)
bn_bwd_apply_vec_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ mask_in,
                        const float* __restrict__ mask_pre, const float* __restrict__ mean,
                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                        const float* __restrict__ dgamma, const float* __restrict__ dbeta, unsigned rows,
                        unsigned rows_per_img, unsigned C, float inv_count, int batch_stats, float slope,
                        float* __restrict__ gx) {
  const unsigned C4 = C >> 2, lanes_r = kEwBlock / C4;
  const unsigned tr = threadIdx.x / C4, c = (threadIdx.x - tr * C4) * 4;
  if (tr >= lanes_r) return;
  const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 is = ldc4(invstd + c), mu = ldc4(mean + c), gm = gamma ? ldc4(gamma + c) : one;
  const float4 dg = ldc4(dgamma + c), db = ldc4(dbeta + c);
  const float4 k = mul4(gm, is);
  const unsigned stride = gridDim.x * lanes_r;
  for (unsigned r = blockIdx.x * lanes_r + tr; r < rows; r += stride) {
    const float4 xr = ld4(x + (size_t)r * C + c);
    float4 gv = ld4(g + (size_t)r * C + c);
    const unsigned img = (mask_in || mask_pre) ? r / rows_per_img : 0u;
    const float4 mi = mask_in ? ld4(mask_in + (size_t)img * C + c) : one;
    if (mask_pre) gv = mul4(gv, ld4(mask_pre + (size_t)img * C + c));
    const float4 xv = mul4(xr, mi);
    float4 o = gv;
    if (batch_stats) {
      o.x -= (db.x + (xv.x - mu.x) * is.x * dg.x) * inv_count;
      o.y -= (db.y + (xv.y - mu.y) * is.y * dg.y) * inv_count;
      o.z -= (db.z + (xv.z - mu.z) * is.z * dg.z) * inv_count;
      o.w -= (db.w + (xv.w - mu.w) * is.w * dg.w) * inv_count;
    }
    o = mul4(mul4(o, k), mi);
    if (slope >= 0.f) {
      o.x *= xr.x > 0.f ? 1.f : slope; o.y *= xr.y > 0.f ? 1.f : slope;
      o.z *= xr.z > 0.f ? 1.f : slope; o.w *= xr.w > 0.f ? 1.f : slope;
    }
    *reinterpret_cast<float4*>(gx + (size_t)r * C + c) = o;
  }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static bool vec_ok(long long rows, int C) {
  return (C % 4) == 0 && C >= 4 && C <= 4 * kEwBlock && rows * C < (1LL << 31);
}
static int reduce_blocks_vec(long long rows, int C) {
  const int lanes_r = kEwBlock / (C / 4);
  long long nb = (rows + (long long)lanes_r * 8 - 1) / ((long long)lanes_r * 8);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// ---- embedding-table gradient of the conditioning planes: one block per table cell, threads over samples.
// Every thread first gathers the contribution of its (<= 8) samples -- all loads up front --, then the per-class sums
// are wave reductions in a fixed order: one barrier in total.
constexpr int kPtgMaxChunks = 8;   // B <= 2048
__global__ void __launch_bounds__(256)
plane_table_grad_kernel(const float* __restrict__ g, int g_ld, int g_ch, const float* __restrict__ x, int x_ld, int x_ch,
                        const int* __restrict__ idx, int idx_ld, int idx_col, int B, int H, int W, int n_rows,
                        float* __restrict__ out, const float* __restrict__ table) {
  __shared__ float red[64][4];
  const int cell = blockIdx.x, ch = cell >> 4, cw = cell & 15, t = threadIdx.x, wave = t >> 6;
  // pixels h with floor(h*16/H) == ch:  ceil(ch*H/16) <= h < ceil((ch+1)*H/16)
  const int h0 = (ch * H + 15) / 16, h1 = ((ch + 1) * H + 15) / 16;
  const int w0 = (cw * W + 15) / 16, w1 = ((cw + 1) * W + 15) / 16;
  float val[kPtgMaxChunks];
  int cls[kPtgMaxChunks];
#pragma unroll
  for (int c = 0; c < kPtgMaxChunks; ++c) {
    const int b = c * 256 + t;
    val[c] = 0.f;
    cls[c] = -1;
    if (b < B) {
      cls[c] = idx[(long long)b * idx_ld + idx_col];
      const float pt = table ? tanhf(table[cls[c] * 256 + cell]) : 0.f;   // (every pixel of the cell shows this entry)
      float acc = 0.f;
      for (int h = h0; h < h1; ++h)
        for (int w = w0; w < w1; ++w) {
          const long long pix = ((long long)b * H + h) * W + w;
          // the plane value: stored, or recomputed from the table (the stored planes may carry a Dropout2d mask)
          const float p = table ? pt : x[pix * x_ld + x_ch];
          acc += g[pix * g_ld + g_ch] * (1.f - p * p);
        }
      val[c] = acc;
    }
  }
  for (int n0 = 0; n0 < n_rows; n0 += 64) {
    const int nn = min(64, n_rows - n0);
    for (int n = 0; n < nn; ++n) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < kPtgMaxChunks; ++c) acc += cls[c] == n0 + n ? val[c] : 0.f;
      acc = wave_sum(acc);
      if ((t & 63) == 0) red[n][wave] = acc;
    }
    __syncthreads();
    if (t < nn) out[(n0 + t) * 256 + cell] = ((red[t][0] + red[t][1]) + red[t][2]) + red[t][3];
    __syncthreads();
  }
}

// ---- col2im for transposed convolutions in scatter form (see ali_hip.h): one thread per output pixel, NC channels
template <int NC>
__global__ void __launch_bounds__(kEwBlock)
col2im_kernel(const float* __restrict__ contrib, int ldc, const float* __restrict__ bias, float* __restrict__ out,
              int B, int H, int W, int Hout, int Wout, int ostride, int R, int S, int stride, int pad, int act,
              float slope) {
  const long long total = (long long)B * Hout * Wout;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wout);
    const long long t2 = i / Wout;
    const int oy = (int)(t2 % Hout), b = (int)(t2 / Hout);
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = bias ? bias[c] : 0.f;
    // taps r = r0, r0 + stride, ... are the ones with (oy + pad - r) divisible by stride
    const int r0 = (oy + pad) % stride, s0 = (ox + pad) % stride;
    for (int r = r0; r < R; r += stride) {
      const int ih = (oy + pad - r) / stride;
      if (oy + pad - r < 0 || ih >= H) continue;
      for (int s = s0; s < S; s += stride) {
        const int iw = (ox + pad - s) / stride;
        if (ox + pad - s < 0 || iw >= W) continue;
        const float* src = contrib + ((long long)(b * H + ih) * W + iw) * ldc + (r * S + s) * NC;
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] += src[c];
      }
    }
    float* o = out + i * ostride;
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = apply_act(acc[c], act, slope);
  }
}

// ---- spectrogram tail: power, log, optional standardise + clip, [B,T,2F] -> [B,F,T] through a 32x32 LDS tile
__global__ void __launch_bounds__(256)
spect_post_kernel(const float* __restrict__ y, int T, int F, const float* __restrict__ mean,
                  const float* __restrict__ stdv, float clip_k, float* __restrict__ out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {                        // rows = frames, columns = frequencies (contiguous in y)
    const int t = t0 + r, f = f0 + tx;
    float v = 0.f;
    if (t < T && f < F) {
      const float* row = y + ((long long)b * T + t) * (2 * F);
      const float re = row[f], im = row[F + f];
      v = logf(re * re + im * im + 1e-6f);
      if (mean) {
        v = (v - mean[t]) / (stdv[t] + 1e-6f);
        v = fminf(fmaxf(v, -clip_k), clip_k) / clip_k;
      }
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {                        // rows = frequencies, columns = frames (contiguous in out)
    const int f = f0 + r, t = t0 + tx;
    if (f < F && t < T) out[((long long)b * F + f) * T + t] = tile[tx][r];
  }
}

// ---- BCE with logits against a constant target (single block; B <= a few thousand)
__global__ void bce_logits_kernel(const float* __restrict__ logit, int B, float target, float gscale,
                                  float* __restrict__ out2, float* __restrict__ glogit) {
  __shared__ double r1[kEwBlock], r2[kEwBlock];
  const int t = threadIdx.x;
  double l = 0.0, sg = 0.0;
  for (int i = t; i < B; i += kEwBlock) {
    const float x = logit[i];
    // max(x,0) - x*t + log1p(exp(-|x|))   (torch's stable form)
    const float loss = fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x)));
    const float s = 1.f / (1.f + expf(-x));
    l += (double)loss;
    sg += (double)s;
    if (glogit) glogit[i] = gscale * (s - target) / (float)B;
  }
  r1[t] = l;
  r2[t] = sg;
  __syncthreads();
  for (int off = kEwBlock / 2; off > 0; off >>= 1) {
    if (t < off) { r1[t] += r1[t + off]; r2[t] += r2[t + off]; }
    __syncthreads();
  }
  if (t == 0 && out2) {
    out2[0] = (float)(r1[0] / B);
    out2[1] = (float)(r2[0] / B);
  }
}

// `arrive` != null: dev_step holds the number of COMPLETED steps; this launch is step dev_step[0] + 1 and its last block
// stores that back (every block has read dev_step before it arrives: no block can see the new value) -- the device-side
// step count of a captured graph advances without a launch of its own.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                            int step_host, int* __restrict__ dev_step, int* __restrict__ arrive, float grad_scale,
                            _Float16* __restrict__ p16) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long step = (long long)gridDim.x * blockDim.x;
  const int t = dev_step ? __hip_atomic_load(dev_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + (arrive ? 1 : 0) : step_host;
  const float bc1 = (float)(1.0 - pow((double)b1, (double)t));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)t));
  const float step_size = lr / bc1;
  for (; i < n; i += step) {
    const float gi = g[i] * grad_scale;
    const float w = 1.f - b1;                                    // exp_avg.lerp_(grad, 1-beta1): torch's two-branch lerp
    const float mi = w < 0.5f ? m[i] + w * (gi - m[i]) : gi - (gi - m[i]) * (1.f - w);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pn = p[i] - step_size * (mi / denom);
    p[i] = pn;
    if (p16) p16[i] = (_Float16)pn;           // fp16 twin of the (kernel-layout) master weights, fp16-MFMA path
  }
  if (arrive) {
    __syncthreads();                   // (every thread of the block has computed with t)
    if (threadIdx.x == 0) {
      const int a = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a == (int)gridDim.x - 1) {
        __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dev_step, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// several int64 counters += their increments in one launch (num_batches_tracked of every BatchNorm2d, the iteration counter)
constexpr int kAddJobs = 16;
struct AddJobs { int n; long long* ptr[kAddJobs]; long long inc[kAddJobs]; };
__global__ void add_i64_multi_kernel(const AddJobs jobs) {
  const int i = threadIdx.x;
  if (i < jobs.n) jobs.ptr[i][0] += jobs.inc[i];
}


// ---- attribute plumbing of an iteration (replaces a dozen tiny ATen launches): arg-max class of every categorical
// attribute + the continuous attributes gathered into one [B][n_cont] row (mnist.py:47-55,204-209; audio_mnist.py:203-210)
struct AttrPtrs { const void* cat[8]; int ncls[8]; int is_int[8]; const float* cont[4]; };
__device__ __forceinline__ float attr_val(const void* p, int is_int, long long i) {
  return is_int ? (float)reinterpret_cast<const int*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__global__ void attr_pack_kernel(AttrPtrs a, int n_cat, int n_cont, int B, int* __restrict__ idx, float* __restrict__ cont) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  for (int j = 0; j < n_cat; ++j) {     // torch.argmax: first maximal entry
    int best = 0;
    float bv = attr_val(a.cat[j], a.is_int[j], (long long)b * a.ncls[j]);
    for (int n = 1; n < a.ncls[j]; ++n) {
      const float v = attr_val(a.cat[j], a.is_int[j], (long long)b * a.ncls[j] + n);
      if (v > bv) { bv = v; best = n; }
    }
    idx[b * n_cat + j] = best;
  }
  for (int j = 0; j < n_cont; ++j) cont[b * n_cont + j] = a.cont[j][b];
}

// Generator input row (mnist.py:76-85, audio_mnist.py:250-256): [ z | onehot_j @ table_j (256 each) | cont | 0 pad ].
// The product stays a true sum over classes, so soft (non one-hot) attributes give what the reference's matmul gives.
struct GInPtrs { const void* oh[8]; const float* tab[8]; int ncls[8]; int is_int[8]; };
__global__ void g_input_kernel(const float* __restrict__ z, int zdim, GInPtrs p, int n_emb, const float* __restrict__ cont,
                               int n_cont, int B, int ld, float* __restrict__ out) {
  const int b = blockIdx.x;
  float* o = out + (long long)b * ld;
  for (int c = threadIdx.x; c < ld; c += blockDim.x) {
    float v = 0.f;
    if (c < zdim) v = z[(long long)b * zdim + c];
    else if (c < zdim + 256 * n_emb) {
      const int j = (c - zdim) >> 8, k = (c - zdim) & 255;
      for (int n = 0; n < p.ncls[j]; ++n) {
        const float w = attr_val(p.oh[j], p.is_int[j], (long long)b * p.ncls[j] + n);
        if (w != 0.f) v += w * p.tab[j][n * 256 + k];
      }
    } else if (c < zdim + 256 * n_emb + n_cont) v = cont[b * n_cont + (c - zdim - 256 * n_emb)];
    o[c] = v;
  }
}
// its table gradient: dT[n][k] = sum_b onehot[b][n] * g[b*ld + off + k].  Block = (class n, 16 columns k): 16 sample
// lanes x 16 columns, each lane sums its samples b = lane, lane+16, ... in order, the lanes are combined in order.
__global__ void __launch_bounds__(256)
g_input_table_grad_kernel(const void* __restrict__ oh, int is_int, int ncls, const float* __restrict__ g, int ld, int off,
                          int B, float* __restrict__ out) {
  __shared__ float red[16][17];
  const int n = blockIdx.x >> 4, k = ((blockIdx.x & 15) << 4) + (threadIdx.x & 15), sl = threadIdx.x >> 4;
  float acc = 0.f;
#pragma unroll 4
  for (int b = sl; b < B; b += 16)
    acc += attr_val(oh, is_int, (long long)b * ncls + n) * g[(long long)b * ld + off + k];
  red[sl][threadIdx.x & 15] = acc;
  __syncthreads();
  if (threadIdx.x < 16) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) v += red[j][threadIdx.x];
    out[n * 256 + ((blockIdx.x & 15) << 4) + threadIdx.x] = v;
  }
}

// BCE-with-logits of two passes batched along the rows ([0,B): target ta, [B,2B): target tb), e.g. the E+G loss
// (bce(D(x,E(x)),0) + bce(D(G(z),z),1)) / 2 of mnist.py:228 or the two scores of :245-248, in one launch:
// out[0] = (loss_a + loss_b) / 2, out[1] = mean sigmoid(a), out[2] = mean sigmoid(b); glogit = gscale*(sigmoid - t)/B
__global__ void bce_logits_pair_kernel(const float* __restrict__ logit, int B, float ta, float tb, float gscale,
                                       float* __restrict__ out3, float* __restrict__ glogit) {
  __shared__ double r1[kEwBlock], r2[kEwBlock], r3[kEwBlock], r4[kEwBlock];
  const int t = threadIdx.x;
  double la = 0.0, lb = 0.0, sa = 0.0, sb = 0.0;
  for (int i = t; i < 2 * B; i += kEwBlock) {
    const float x = logit[i];
    const float tg = i < B ? ta : tb;
    const float loss = fmaxf(x, 0.f) - x * tg + log1pf(expf(-fabsf(x)));
    const float s = 1.f / (1.f + expf(-x));
    if (i < B) { la += (double)loss; sa += (double)s; } else { lb += (double)loss; sb += (double)s; }
    if (glogit) glogit[i] = gscale * (s - tg) / (float)B;
  }
  r1[t] = la; r2[t] = lb; r3[t] = sa; r4[t] = sb;
  __syncthreads();
  for (int off = kEwBlock / 2; off > 0; off >>= 1) {
    if (t < off) { r1[t] += r1[t + off]; r2[t] += r2[t + off]; r3[t] += r3[t + off]; r4[t] += r4[t + off]; }
    __syncthreads();
  }
  if (t == 0) {
    // (a + b) / 2 on the two fp32 means, as the reference computes it
    out3[0] = ((float)(r1[0] / B) + (float)(r2[0] / B)) / 2.f;
    out3[1] = (float)(r3[0] / B);
    out3[2] = (float)(r4[0] / B);
  }
}

struct EmbPtrs { const float* t[8]; };
// One block = a run of pixels of ONE image (grid: chunks x B).  The embedding planes are 16 x 16 tables blown up to the
// image size (nearest), so their tanh is taken once per block into LDS (256 values per table) instead of once per pixel
// (512^2 maps: 1024x redundant, and with two 32-bit divisions per pixel on top the kernel was VALU bound: 1.6 TB/s).
__global__ void __launch_bounds__(256)
assemble_planes_kernel(const float* __restrict__ X, const int* __restrict__ idx, EmbPtrs emb, int n_emb,
                       const float* __restrict__ cont, int n_cont, float* __restrict__ out, int B, int H, int W, int Cpad,
                       const float* __restrict__ mask, int mask_ld, int chunk) {
  __shared__ float tab[8][256];
  const int t = threadIdx.x, b = blockIdx.y;
  for (int j = 0; j < n_emb; ++j) tab[j][t] = tanhf(emb.t[j][idx[b * n_emb + j] * 256 + t]);
  float cv[8], mk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cv[j] = j < n_cont ? cont[b * n_cont + j] : 0.f;
    mk[j] = (mask && j < Cpad) ? mask[(long long)b * mask_ld + j] : 1.f;
  }
  __syncthreads();
  const int HW = H * W;
  const int p1 = min(HW, (int)(blockIdx.x + 1) * chunk);
  const float* Xb = X + (long long)b * HW;
  float* ob = out + (long long)b * HW * Cpad;
  auto emit = [&](int p, int sh, int sw) {
    float v[8];
    v[0] = Xb[p];
    int c = 1;
    for (int j = 0; j < n_emb; ++j, ++c) v[c] = tab[j][sh * 16 + sw];
    for (int j = 0; j < n_cont; ++j, ++c) v[c] = cv[j];
    for (; c < 8; ++c) v[c] = 0.f;
    float* o = ob + (long long)p * Cpad;
    if (Cpad == 8 || Cpad == 4) {        // 5 planes in 8 channels (MNIST), 2-4 in 4 (whale / ESRF): 16-byte stores per pixel
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= mk[e];            // the Dropout2d in front of the consuming conv, per (sample, channel)
      *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
      if (Cpad == 8) *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      for (int e = 0; e < Cpad; ++e) o[e] = (e < 8 ? v[e] * mk[e] : 0.f);
    }
  };
  // nearest: src = floor(dst * 16 / size)  (torch 'nearest', probe-verified in SURVEY.md K8)
  const int p0 = blockIdx.x * chunk;
  if ((W & 255) == 0 && (chunk % W) == 0) {
    // wide maps, a run = whole rows: a thread keeps its columns, the row index is uniform -- no division per pixel
    // (three 32-bit divisions per pixel kept the kernel at 1.8 TB/s)
    const int per = W >> 8, row0 = p0 / W, nrows = (p1 - p0) / W;
    for (int k = 0; k < per; ++k) {
      const int w = t + 256 * k;
      const int sw = (int)((unsigned)(w * 16) / (unsigned)W);
      for (int r = 0; r < nrows; ++r) {
        const int h = row0 + r;
        emit(h * W + w, (int)((unsigned)(h * 16) / (unsigned)H), sw);
      }
    }
  } else {
    for (int p = p0 + t; p < p1; p += 256) {
      const int h = (int)((unsigned)p / (unsigned)W), w = p - h * W;
      emit(p, (int)((unsigned)(h * 16) / (unsigned)H), (int)((unsigned)(w * 16) / (unsigned)W));
    }
  }
}

// ---- the one-output head of the Discriminator (Conv2d(1024, 1, 1) on a 1x1 map, mnist.py:127): a GEMV and its
// weight gradient.  As GEMMs they use 1/64 of a tile and need four launches (GEMM + slab fold + two column-sum stages).
__global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ x, int ld, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int B, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per row
  if (row >= B) return;
  const float* xr = x + (long long)row * ld;
  float acc = 0.f;
  for (int c = lane * 4; c + 3 < C; c += 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(xr + c), b = *reinterpret_cast<const f32x4*>(w + c);
    acc += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
  }
  acc = wave_sum(acc);
  if (lane == 0) y[row] = acc + (bias ? bias[0] : 0.f);
}

// dw[c] = sum_b g[b] * x[b][c], db = sum_b g[b]: 32 columns x 8 row lanes per block, eight independent loads in flight per
// thread (the loop is latency bound: 2 MB of x behind a handful of blocks), the row lanes meet in LDS (fixed order)
__global__ void __launch_bounds__(256) head_wgrad_kernel(const float* __restrict__ x, int ld, const float* __restrict__ g,
                                                         float* __restrict__ dw, float* __restrict__ db, int B, int C) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float acc[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[u] = 0.f;
  float s0 = 0.f;
  const bool sum_g = blockIdx.x == 0 && cl == 0 && db != nullptr;
  if (c < C || sum_g) {
    const int cc = c < C ? c : 0;
    for (int b0 = rl; b0 < B; b0 += 64) {
      float gv[8], xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 8 * u;
        gv[u] = b < B ? g[b] : 0.f;
        xv[u] = b < B ? x[(long long)b * ld + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[u] += gv[u] * xv[u]; s0 += gv[u]; }
    }
  }
  red[rl][cl] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  if (cl == 0) red[rl][32] = s0;
  __syncthreads();
  if (rl == 0) {
    float v = red[0][cl], sg = red[0][32];
#pragma unroll
    for (int r = 1; r < 8; ++r) { v += red[r][cl]; sg += red[r][32]; }
    if (c < C) dw[c] = v;
    if (sum_g) db[0] = sg;
  }
}

// ---- several small device-to-device copies in one launch (the step's inputs into the captured graph's buffers) --------
constexpr int kCopyJobs = 8;
struct CopyJobs { int n; const unsigned* src[kCopyJobs]; unsigned* dst[kCopyJobs]; long long words[kCopyJobs]; int blk0[kCopyJobs + 1]; };
__global__ void __launch_bounds__(256) copy_multi_kernel(const CopyJobs jobs) {
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < jobs.n; ++i)
    if ((int)blockIdx.x >= jobs.blk0[i]) k = i;
  const long long i0 = ((long long)(blockIdx.x - jobs.blk0[k]) * 256 + threadIdx.x) * 4;
  const long long n = jobs.words[k];
  const unsigned* s = jobs.src[k];
  unsigned* d = jobs.dst[k];
  if (i0 + 4 <= n && ((((unsigned long long)s) | ((unsigned long long)d)) & 15ull) == 0) {
    *reinterpret_cast<uint4*>(d + i0) = *reinterpret_cast<const uint4*>(s + i0);
  } else {
    for (long long i = i0; i < n && i < i0 + 4; ++i) d[i] = s[i];
  }
}

}  // namespace ali

using namespace ali;
#define ST(s) ((hipStream_t)(s))

extern "C" int ali_pack_weights(const float* src, float* dst, int32_t N, int32_t T, int32_t C, int32_t Cpad,
                                int64_t s_n, int64_t s_tap, int64_t s_c, ali_stream_t stream) {
  if (!src || !dst || N <= 0 || T <= 0 || C <= 0 || Cpad < C) { set_error("ali_pack_weights: bad argument"); return ALI_ERR_BAD_ARG; }
  const long long total = (long long)N * T * Cpad;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(ew_grid(total)), dim3(kEwBlock), 0, ST(stream), src, dst, N, T, C, Cpad,
                     (long long)s_n, (long long)s_tap, (long long)s_c);
  return check_launch("pack_weights_kernel");
}

extern "C" int ali_pack_weights_multi(int32_t n_jobs, const float* const* src, float* const* dst, void* const* dst16,
                                      const int32_t* dims, const int64_t* strides, ali_stream_t stream) {
  if (n_jobs < 1 || n_jobs > 40 || !src || !dst || !dims || !strides) { set_error("ali_pack_weights_multi: bad argument"); return ALI_ERR_BAD_ARG; }
  PackJobs jobs;
  memset(&jobs, 0, sizeof(jobs));
  long long blk = 0;
  for (int i = 0; i < n_jobs; ++i) {
    PackJob& J = jobs.j[i];
    J.src = src[i]; J.dst = dst[i];
    J.dst16 = dst16 ? reinterpret_cast<_Float16*>(dst16[i]) : nullptr;
    J.N = dims[4 * i]; J.T = dims[4 * i + 1]; J.C = dims[4 * i + 2]; J.Cpad = dims[4 * i + 3];
    J.s_n = strides[3 * i]; J.s_tap = strides[3 * i + 1]; J.s_c = strides[3 * i + 2];
    if (!J.src || !J.dst || J.N <= 0 || J.T <= 0 || J.C <= 0 || J.Cpad < J.C) { set_error("ali_pack_weights_multi: bad job %d", i); return ALI_ERR_BAD_ARG; }
    J.blk0 = (int)blk;
    // dst[(n*T + t)*Cpad + c] = src[n*s_n + t*s_tap + c*s_c] as a batched transpose in[b][i = c][j] -> out[b][j][i]:
    //   src [N][C][T] (s_c = T, s_tap = 1, s_n = C*T): b = n, j = t                  (Conv2d forward pack, ...)
    //   src [C][N][T] (s_n = T, s_tap = 1, s_c = N*T): one batch, j = n*T + t        (data-gradient packs, ...)
    const long long NT = (long long)J.N * J.T;
    // ... for the big ones only: a 32 x 32 tile per block costs more in block turnover than the strided reads cost
    // while the source still sits in L2 / Infinity Cache (MorphoMNIST: 48 vs 107 us per launch; ESRF: 4.2 vs 1.1 ms)
    const bool big = NT * J.Cpad >= (8LL << 20);
    if (!big) {
    } else if (J.s_tap == 1 && J.s_c == J.T && J.s_n == (long long)J.C * J.T && (long long)J.C * J.T >= 64) {
      J.tiled = 1; J.batch = J.N; J.I = J.C; J.J = J.T; J.in_b = J.s_n; J.in_i = J.T; J.out_b = (long long)J.T * J.Cpad;
    } else if (J.T >= 1 && J.s_tap == 1 && J.s_n == J.T && J.s_c == NT && NT < (1LL << 31) && NT >= 64) {
      J.tiled = 1; J.batch = 1; J.I = J.C; J.J = (int)NT; J.in_b = 0; J.in_i = NT; J.out_b = 0;
    }
    // master weights in the forward pack's order (contiguous n, FlatGroup.layouts): per tap a [C] x [N] transpose
    const bool al16 = ((((unsigned long long)J.src) | ((unsigned long long)J.dst)) & 15ull) == 0 &&
                      (!J.dst16 || (((unsigned long long)J.dst16) & 7ull) == 0);
    if (!J.tiled && J.s_n == 1 && (J.N % 4) == 0 && (J.Cpad % 4) == 0 && (J.s_tap % 4) == 0 && (J.s_c % 4) == 0 && al16 &&
        NT * J.Cpad >= (1LL << 18) && J.N >= 32 && J.C >= 32) {
      J.tiled = 2; J.batch = J.T; J.I = J.C; J.J = J.N; J.Ipad = J.Cpad;
      J.in_b = J.s_tap; J.in_i = J.s_c; J.out_b = J.Cpad; J.out_j = (long long)J.T * J.Cpad;
      J.ti = (J.Ipad + 63) / 64; J.tj = (J.J + 63) / 64;
      blk += (long long)J.batch * J.ti * J.tj;
    } else if (J.tiled) {
      J.Ipad = J.Cpad;
      J.out_j = J.Ipad;
      J.ti = (J.Ipad + 31) / 32; J.tj = (J.J + 31) / 32;
      blk += (long long)J.batch * J.ti * J.tj;
    } else {
      long long nb = ((long long)J.N * J.T * J.Cpad + kEwBlock * 4 - 1) / (kEwBlock * 4);
      if (nb > 256) nb = 256;
      if (nb < 1) nb = 1;
      blk += nb;
    }
    if (blk >= (1LL << 31)) { set_error("ali_pack_weights_multi: too many tiles"); return ALI_ERR_BAD_ARG; }
  }
  jobs.n = n_jobs;
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((unsigned)blk), dim3(kEwBlock), 0, ST(stream), jobs);
  return check_launch("pack_weights_multi_kernel");
}

extern "C" int ali_head_fwd(const float* x, int32_t ld, const float* w, const float* bias, float* y, int32_t B, int32_t C,
                            ali_stream_t stream) {
  if (!x || !w || !y || B <= 0 || C <= 0 || (C % 4) || ld < C || (ld % 4)) { set_error("ali_head_fwd: bad argument"); return ALI_ERR_BAD_ARG; }
  hipLaunchKernelGGL(head_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, ST(stream), x, ld, w, bias, y, B, C);
  return check_launch("head_fwd_kernel");
}

extern "C" int ali_head_wgrad(const float* x, int32_t ld, const float* g, float* dw, float* db, int32_t B, int32_t C,
                              ali_stream_t stream) {
  if (!x || !g || !dw || B <= 0 || C <= 0 || ld < C) { set_error("ali_head_wgrad: bad argument"); return ALI_ERR_BAD_ARG; }
  hipLaunchKernelGGL(head_wgrad_kernel, dim3((C + 31) / 32), dim3(256), 0, ST(stream), x, ld, g, dw, db, B, C);
  return check_launch("head_wgrad_kernel");
}

extern "C" int ali_copy_multi(int32_t n, const void* const* src, void* const* dst, const int64_t* bytes, ali_stream_t stream) {
  if (n < 0 || (n > 0 && (!src || !dst || !bytes))) { set_error("ali_copy_multi: bad argument"); return ALI_ERR_BAD_ARG; }
  for (int j0 = 0; j0 < n; j0 += kCopyJobs) {
    CopyJobs cj;
    memset(&cj, 0, sizeof(cj));
    int blocks = 0;
    for (int i = j0; i < n && i < j0 + kCopyJobs; ++i) {
      if (!src[i] || !dst[i] || bytes[i] < 0 || (bytes[i] & 3) || ((unsigned long long)src[i] & 3) || ((unsigned long long)dst[i] & 3)) {
        set_error("ali_copy_multi: job %d: pointers and sizes must be multiples of 4 bytes", i);
        return ALI_ERR_BAD_ARG;
      }
      const int k = cj.n++;
      cj.src[k] = reinterpret_cast<const unsigned*>(src[i]);
      cj.dst[k] = reinterpret_cast<unsigned*>(dst[i]);
      cj.words[k] = bytes[i] / 4;
      cj.blk0[k] = blocks;
      const long long nb = (cj.words[k] + 1023) / 1024;
      if (blocks + nb > (1 << 30)) { set_error("ali_copy_multi: too large"); return ALI_ERR_BAD_ARG; }
      blocks += (int)nb;
    }
    cj.blk0[cj.n] = blocks;
    if (blocks > 0) hipLaunchKernelGGL(copy_multi_kernel, dim3(blocks), dim3(256), 0, ST(stream), cj);
    int rc = check_launch("copy_multi_kernel");
    if (rc) return rc;
  }
  return ALI_OK;
}

extern "C" int ali_act_bwd(const float* gy, const float* y, float* gpre, int64_t n, int32_t act, float slope,
                           ali_stream_t stream) {
  if (!gy || !y || !gpre || n < 0) { set_error("ali_act_bwd: bad argument"); return ALI_ERR_BAD_ARG; }
  if (n == 0) return ALI_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(kEwBlock), 0, ST(stream), gy, y, gpre, (long long)n, act, slope);
  return check_launch("act_bwd_kernel");
}

static int reduce_blocks(long long rows, int C) {
  int lanes_c = 1;
  const int cw = C < kEwBlock ? C : kEwBlock;
  while (lanes_c < cw) lanes_c <<= 1;
  const int lanes_r = kEwBlock / lanes_c;
  long long nb = (rows + (long long)lanes_r * 8 - 1) / ((long long)lanes_r * 8);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}

extern "C" int ali_colsum(const float* x, int64_t rows, int32_t C, int32_t ld, float* out, void* ws, size_t ws_bytes,
                          ali_stream_t stream) {
  ws = ws_payload(ws);
  ws_bytes = ws_payload_bytes(ws_bytes);
  if (!x || !out || rows <= 0 || C <= 0 || ld < C) { set_error("ali_colsum: bad argument"); return ALI_ERR_BAD_ARG; }
  const bool vec = vec_ok(rows, C) && (ld % 4) == 0 && (long long)rows * ld < (1LL << 31) && aligned16(x);
  const int nb = vec ? reduce_blocks_vec(rows, C) : reduce_blocks(rows, C);
  if (!ws || ws_bytes < (size_t)nb * C * sizeof(float)) { set_error("ali_colsum: workspace too small"); return ALI_ERR_WORKSPACE; }
  float* part = reinterpret_cast<float*>(ws);
  if (vec)
    hipLaunchKernelGGL(rowreduce_vec_kernel<2>, dim3(nb), dim3(kEwBlock), 0, ST(stream), x, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (unsigned)rows, 1u, (unsigned)C, (unsigned)ld, part);
  else
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(kEwBlock), 0, ST(stream), x, (long long)rows, C, ld, part);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(C), dim3(64), 0, ST(stream), part, nb, C, out);
  return check_launch("colsum");
}

extern "C" int ali_rowmask_mul(const float* x, const float* mask, float* out, int32_t B, int32_t rows_per_img,
                               int32_t C, ali_stream_t stream) {
  if (!x || !mask || !out || B <= 0 || rows_per_img <= 0 || C <= 0) { set_error("ali_rowmask_mul: bad argument"); return ALI_ERR_BAD_ARG; }
  const long long n = (long long)B * rows_per_img * C;
  hipLaunchKernelGGL(rowmask_mul_kernel, dim3(ew_grid(n)), dim3(kEwBlock), 0, ST(stream), x, mask, out, n, rows_per_img, C);
  return check_launch("rowmask_mul_kernel");
}

extern "C" int ali_dropout_mask(uint64_t seed, uint64_t offset, const int64_t* dev_counter, float p, float* out,
                                int64_t n, ali_stream_t stream) {
  if (!out || n <= 0 || !(p >= 0.f && p < 1.f)) { set_error("ali_dropout_mask: bad argument"); return ALI_ERR_BAD_ARG; }
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(ew_grid(n)), dim3(kEwBlock), 0, ST(stream), seed, offset,
                     reinterpret_cast<const long long*>(dev_counter), p, out, (long long)n);
  return check_launch("dropout_mask_kernel");
}

extern "C" int ali_dropout_mask_multi(uint64_t seed, const int64_t* dev_counter, const int64_t* seg_end,
                                      const float* seg_p, const int32_t* seg_clog, const int32_t* seg_cpad,
                                      int32_t n_seg, float* out, ali_stream_t stream) {
  if (!seg_end || !seg_p || !seg_clog || !seg_cpad || !out || n_seg < 1 || n_seg > 64) {
    set_error("ali_dropout_mask_multi: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  MaskSegs segs;
  long long prev = 0, draws = 0, longest = 0;
  for (int i = 0; i < n_seg; ++i) {
    const long long len = seg_end[i] - prev;
    if (len <= 0 || !(seg_p[i] >= 0.f && seg_p[i] < 1.f) || seg_clog[i] < 1 || seg_cpad[i] < seg_clog[i] ||
        len % seg_cpad[i] != 0) {
      set_error("ali_dropout_mask_multi: bad segment");
      return ALI_ERR_BAD_ARG;
    }
    segs.out_end[i] = prev = seg_end[i];
    segs.draw_start[i] = draws;
    draws += len / seg_cpad[i] * seg_clog[i];
    segs.p[i] = seg_p[i];
    segs.clog[i] = seg_clog[i];
    segs.cpad[i] = seg_cpad[i];
    if (len > longest) longest = len;
  }
  segs.n = n_seg;
  int gx = (int)((longest + kEwBlock * 4 - 1) / (kEwBlock * 4));
  if (gx > 128) gx = 128;
  hipLaunchKernelGGL(dropout_mask_multi_kernel, dim3(gx, n_seg), dim3(kEwBlock), 0, ST(stream), seed,
                     reinterpret_cast<const long long*>(dev_counter), segs, out);
  return check_launch("dropout_mask_multi_kernel");
}

extern "C" int ali_bn_stats(const float* x, const float* mask, int32_t B, int32_t rows_per_img, int32_t C,
                            const float* gamma, const float* beta, float* running_mean, float* running_var,
                            float momentum, float eps, int32_t training, float* mean, float* invstd, float* sc,
                            float* sh, int32_t groups, int64_t stat_stride, void* ws, size_t ws_bytes,
                            ali_stream_t stream) {
  ws = ws_payload(ws);
  ws_bytes = ws_payload_bytes(ws_bytes);
  if (!x || B <= 0 || rows_per_img <= 0 || C <= 0 || C > kEwBlock || !mean || !invstd || !sc || !sh ||
      (!training && (!running_mean || !running_var)) || groups < 1 || B % groups != 0 ||
      (groups > 1 && stat_stride < C)) {
    set_error("ali_bn_stats: bad argument (C must be <= 256, B a multiple of groups)");
    return ALI_ERR_BAD_ARG;
  }
  const int Bg = B / groups;
  const long long rows = (long long)Bg * rows_per_img;       // per group
  const bool vec = vec_ok(rows * groups, C) && aligned16(x) && (!mask || aligned16(mask));
  const int nb = vec ? reduce_blocks_vec(rows, C) : reduce_blocks(rows, C);
  if (!ws || ws_bytes < (size_t)groups * nb * 2 * C * sizeof(float)) { set_error("ali_bn_stats: workspace too small"); return ALI_ERR_WORKSPACE; }
  float* part = reinterpret_cast<float*>(ws);
  if (training && vec)
    hipLaunchKernelGGL(rowreduce_vec_kernel<0>, dim3(nb, groups), dim3(kEwBlock), 0, ST(stream), x, (const float*)nullptr, mask,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (unsigned)rows,
                       (unsigned)rows_per_img, (unsigned)C, (unsigned)C, part);
  else if (training)
    for (int gi = 0; gi < groups; ++gi)
      hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb), dim3(kEwBlock), 0, ST(stream), x + (size_t)gi * rows * C,
                         mask ? mask + (size_t)gi * Bg * C : nullptr, rows, rows_per_img, C, part + (size_t)gi * nb * 2 * C);
  const PartLayout lay = {(long long)nb * 2 * C, 2LL * C, (long long)C, 1LL};
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(kEwBlock), 0, ST(stream), part, lay, nb, C, rows, gamma, beta,
                     running_mean, running_var, momentum, eps, training, mean, invstd, sc, sh, groups,
                     (long long)stat_stride);
  return check_launch("bn_stats");
}

extern "C" int ali_bn_stats_from_partials(const float* part, int32_t slots, int32_t groups, int32_t C, int64_t count,
                                          const float* gamma, const float* beta, float* running_mean,
                                          float* running_var, float momentum, float eps, float* mean, float* invstd,
                                          float* sc, float* sh, int64_t stat_stride, ali_stream_t stream) {
  if (!part || slots <= 0 || groups < 1 || slots % groups != 0 || C <= 0 || count <= 0 || !mean || !invstd || !sc || !sh ||
      (groups > 1 && stat_stride < C)) {
    set_error("ali_bn_stats_from_partials: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  const int per = slots / groups;
  const PartLayout lay = {(long long)per, 1LL, (long long)C * slots, (long long)slots};
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(kEwBlock), 0, ST(stream), part, lay, per, C, (long long)count, gamma,
                     beta, running_mean, running_var, momentum, eps, 1, mean, invstd, sc, sh, groups,
                     (long long)stat_stride);
  return check_launch("bn_stats_from_partials");
}

extern "C" int ali_bn_apply(const float* x, const float* sc, const float* sh, const float* mask_in,
                            const float* mask_post, float* out, int32_t B, int32_t rows_per_img, int32_t C,
                            int32_t groups, int64_t stat_stride, ali_stream_t stream) {
  if (!x || !sc || !sh || !out || B <= 0 || rows_per_img <= 0 || C <= 0 || groups < 1 || B % groups != 0) {
    set_error("ali_bn_apply: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  const int Bg = B / groups;
  const long long rows = (long long)Bg * rows_per_img, n = rows * C;   // per group
  if (vec_ok(rows * groups, C) && aligned16(x) && aligned16(out) && (!mask_in || aligned16(mask_in)) &&
      (!mask_post || aligned16(mask_post)))
    hipLaunchKernelGGL(bn_apply_vec_kernel, dim3(ew_grid(n / 4), groups), dim3(kEwBlock), 0, ST(stream), x, sc, sh, mask_in,
                       mask_post, out, (unsigned)rows, (unsigned)rows_per_img, (unsigned)C, (long long)stat_stride);
  else
    for (int gi = 0; gi < groups; ++gi)
      hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(n)), dim3(kEwBlock), 0, ST(stream), x + (size_t)gi * n,
                         sc + gi * stat_stride, sh + gi * stat_stride, mask_in ? mask_in + (size_t)gi * Bg * C : nullptr,
                         mask_post ? mask_post + (size_t)gi * Bg * C : nullptr, out + (size_t)gi * n, n, rows_per_img, C);
  return check_launch("bn_apply_kernel");
}

static void launch_bn_bwd_apply(const float* x, const float* g, const float* mask_in, const float* mask_pre,
                                const float* mean, const float* invstd, const float* gamma, const float* dgamma,
                                const float* dbeta, long long rows, int rows_per_img, int C, int batch_stats,
                                float lrelu_slope, float* gx, bool vec, hipStream_t stream) {
  const long long n = rows * C;
  if (vec)
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(ew_grid(n / 4)), dim3(kEwBlock), 0, stream, x, g, mask_in, mask_pre,
                       mean, invstd, gamma, dgamma, dbeta, (unsigned)rows, (unsigned)rows_per_img, (unsigned)C,
                       1.f / (float)rows, batch_stats, lrelu_slope, gx);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n)), dim3(kEwBlock), 0, stream, x, g, mask_in, mask_pre, mean,
                       invstd, gamma, dgamma, dbeta, n, rows_per_img, C, 1.f / (float)rows, batch_stats, lrelu_slope, gx);
}

extern "C" int ali_bn_bwd(const float* x, const float* g, const float* mask_in, const float* mask_pre,
                          const float* mean, const float* invstd, const float* gamma, int32_t B,
                          int32_t rows_per_img, int32_t C, int32_t batch_stats, float lrelu_slope, float* dgamma,
                          float* dbeta, float* gx, void* ws, size_t ws_bytes, ali_stream_t stream) {
  ws = ws_payload(ws);
  ws_bytes = ws_payload_bytes(ws_bytes);
  if (!x || !g || !mean || !invstd || !dgamma || !dbeta || B <= 0 || rows_per_img <= 0 || C <= 0 || C > kEwBlock) {
    set_error("ali_bn_bwd: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  const long long rows = (long long)B * rows_per_img;
  const bool vec = vec_ok(rows, C) && aligned16(x) && aligned16(g) && (!gx || aligned16(gx)) &&
                   (!mask_in || aligned16(mask_in)) && (!mask_pre || aligned16(mask_pre)) && aligned16(mean) &&
                   aligned16(invstd);
  const int nb = vec ? reduce_blocks_vec(rows, C) : reduce_blocks(rows, C);
  if (!ws || ws_bytes < (size_t)nb * 2 * C * sizeof(float)) { set_error("ali_bn_bwd: workspace too small"); return ALI_ERR_WORKSPACE; }
  float* part = reinterpret_cast<float*>(ws);
  if (vec)
    hipLaunchKernelGGL(rowreduce_vec_kernel<1>, dim3(nb), dim3(kEwBlock), 0, ST(stream), x, g, mask_in, mask_pre, mean,
                       invstd, (unsigned)rows, (unsigned)rows_per_img, (unsigned)C, (unsigned)C, part);
  else
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nb), dim3(kEwBlock), 0, ST(stream), x, g, mask_in, mask_pre, mean, invstd,
                       rows, rows_per_img, C, part);
  const PartLayout lay = {0LL, 2LL * C, (long long)C, 1LL};
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(kEwBlock), 0, ST(stream), part, lay, nb, C, dgamma, dbeta);
  if (gx) launch_bn_bwd_apply(x, g, mask_in, mask_pre, mean, invstd, gamma, dgamma, dbeta, rows, rows_per_img, C,
                              batch_stats, lrelu_slope, gx, vec, ST(stream));
  return check_launch("bn_bwd");
}

extern "C" int ali_bn_bwd_from_partials(const float* part, int32_t slots, const float* x, const float* g,
                                        const float* mask_in, const float* mask_pre, const float* mean,
                                        const float* invstd, const float* gamma, int32_t B, int32_t rows_per_img,
                                        int32_t C, int32_t batch_stats, float lrelu_slope, float* dgamma, float* dbeta,
                                        float* gx, ali_stream_t stream) {
  if (!part || slots <= 0 || !x || !g || !mean || !invstd || !dgamma || !dbeta || B <= 0 || rows_per_img <= 0 || C <= 0) {
    set_error("ali_bn_bwd_from_partials: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  const long long rows = (long long)B * rows_per_img;
  const bool vec = vec_ok(rows, C) && aligned16(x) && aligned16(g) && (!gx || aligned16(gx)) &&
                   (!mask_in || aligned16(mask_in)) && (!mask_pre || aligned16(mask_pre)) && aligned16(mean) &&
                   aligned16(invstd);
  const PartLayout lay = {0LL, 1LL, (long long)C * slots, (long long)slots};
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(kEwBlock), 0, ST(stream), part, lay, slots, C, dgamma, dbeta);
  if (gx) launch_bn_bwd_apply(x, g, mask_in, mask_pre, mean, invstd, gamma, dgamma, dbeta, rows, rows_per_img, C,
                              batch_stats, lrelu_slope, gx, vec, ST(stream));
  return check_launch("bn_bwd_from_partials");
}

extern "C" int ali_plane_table_grad(const float* g, int32_t g_ld, int32_t g_ch, const float* x, int32_t x_ld, int32_t x_ch,
                                    const int32_t* idx, int32_t idx_ld, int32_t idx_col, int32_t B, int32_t H, int32_t W,
                                    int32_t n_rows, float* out, const float* table, ali_stream_t stream) {
  if (!g || (!x && !table) || !idx || !out || B <= 0 || B > 256 * kPtgMaxChunks || H <= 0 || W <= 0 || n_rows <= 0 || g_ch < 0 ||
      g_ch >= g_ld || (x && (x_ch < 0 || x_ch >= x_ld)) || idx_col < 0 || idx_col >= idx_ld) {
    set_error("ali_plane_table_grad: bad argument (B <= 2048)");
    return ALI_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(plane_table_grad_kernel, dim3(256), dim3(256), 0, ST(stream), g, g_ld, g_ch, x, x_ld, x_ch, idx, idx_ld,
                     idx_col, B, H, W, n_rows, out, table);
  return check_launch("plane_table_grad_kernel");
}

extern "C" int ali_col2im(const float* contrib, int32_t ldc, const float* bias, float* out, int32_t B, int32_t H,
                          int32_t W, int32_t Hout, int32_t Wout, int32_t NC, int32_t ostride, int32_t R, int32_t S,
                          int32_t stride, int32_t pad, int32_t act, float slope, ali_stream_t stream) {
  if (!contrib || !out || B <= 0 || H <= 0 || W <= 0 || Hout <= 0 || Wout <= 0 || NC < 1 || NC > 8 || ostride < NC ||
      R <= 0 || S <= 0 || stride <= 0 || pad < 0 || ldc < NC * R * S) {
    set_error("ali_col2im: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  const long long total = (long long)B * Hout * Wout;
  const dim3 grid(ew_grid(total)), block(kEwBlock);
#define C2I(NC_)                                                                                                      \
  hipLaunchKernelGGL(col2im_kernel<NC_>, grid, block, 0, ST(stream), contrib, ldc, bias, out, B, H, W, Hout, Wout, ostride, \
                     R, S, stride, pad, act, slope)
  switch (NC) {
    case 1: C2I(1); break;
    case 2: C2I(2); break;
    case 3: C2I(3); break;
    case 4: C2I(4); break;
    case 5: C2I(5); break;
    case 6: C2I(6); break;
    case 7: C2I(7); break;
    default: C2I(8); break;
  }
#undef C2I
  return check_launch("col2im_kernel");
}

extern "C" int ali_spect_post(const float* y, int32_t B, int32_t T, int32_t F, const float* mean, const float* stdv,
                              float clip_k, float* out, ali_stream_t stream) {
  if (!y || !out || B <= 0 || T <= 0 || F <= 0 || B > 65535 || (mean && (!stdv || !(clip_k > 0.f)))) {
    set_error("ali_spect_post: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(spect_post_kernel, dim3((T + 31) / 32, (F + 31) / 32, B), dim3(256), 0, ST(stream), y, T, F, mean, stdv,
                     clip_k, out);
  return check_launch("spect_post_kernel");
}

extern "C" int ali_bce_logits(const float* logit, int32_t B, float target, float gscale, float* out2, float* glogit,
                              ali_stream_t stream) {
  if (!logit || B <= 0) { set_error("ali_bce_logits: bad argument"); return ALI_ERR_BAD_ARG; }
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(kEwBlock), 0, ST(stream), logit, B, target, gscale, out2, glogit);
  return check_launch("bce_logits_kernel");
}

extern "C" int ali_attr_pack(const void* const* cat, const int32_t* n_classes, const int32_t* cat_is_int, int32_t n_cat,
                             const float* const* cont_in, int32_t n_cont, int32_t B, int32_t* idx, float* cont,
                             ali_stream_t stream) {
  if (n_cat < 0 || n_cat > 8 || n_cont < 0 || n_cont > 4 || B <= 0 || (n_cat && (!cat || !n_classes || !idx)) ||
      (n_cont && (!cont_in || !cont))) {
    set_error("ali_attr_pack: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  AttrPtrs a;
  memset(&a, 0, sizeof(a));
  for (int j = 0; j < n_cat; ++j) { a.cat[j] = cat[j]; a.ncls[j] = n_classes[j]; a.is_int[j] = cat_is_int ? cat_is_int[j] : 0; }
  for (int j = 0; j < n_cont; ++j) a.cont[j] = cont_in[j];
  hipLaunchKernelGGL(attr_pack_kernel, dim3((B + 255) / 256), dim3(256), 0, ST(stream), a, n_cat, n_cont, B, idx, cont);
  return check_launch("attr_pack_kernel");
}

extern "C" int ali_g_input(const float* z, int32_t zdim, const void* const* onehot, const int32_t* n_classes,
                           const int32_t* onehot_is_int, const float* const* tables, int32_t n_emb, const float* cont,
                           int32_t n_cont, int32_t B, int32_t ld, float* out, ali_stream_t stream) {
  if (!z || !out || zdim <= 0 || n_emb < 0 || n_emb > 8 || n_cont < 0 || B <= 0 || ld < zdim + 256 * n_emb + n_cont ||
      (n_emb && (!onehot || !n_classes || !tables)) || (n_cont && !cont)) {
    set_error("ali_g_input: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  GInPtrs p;
  memset(&p, 0, sizeof(p));
  for (int j = 0; j < n_emb; ++j) {
    p.oh[j] = onehot[j]; p.tab[j] = tables[j]; p.ncls[j] = n_classes[j]; p.is_int[j] = onehot_is_int ? onehot_is_int[j] : 0;
  }
  hipLaunchKernelGGL(g_input_kernel, dim3(B), dim3(256), 0, ST(stream), z, zdim, p, n_emb, cont, n_cont, B, ld, out);
  return check_launch("g_input_kernel");
}

extern "C" int ali_g_input_table_grad(const void* onehot, int32_t onehot_is_int, int32_t n_classes, const float* g,
                                      int32_t ld, int32_t off, int32_t B, float* out, ali_stream_t stream) {
  if (!onehot || !g || !out || n_classes <= 0 || B <= 0 || off < 0 || off + 256 > ld) {
    set_error("ali_g_input_table_grad: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(g_input_table_grad_kernel, dim3(n_classes * 16), dim3(256), 0, ST(stream), onehot, onehot_is_int,
                     n_classes, g, ld, off, B, out);
  return check_launch("g_input_table_grad_kernel");
}

extern "C" int ali_bce_logits_pair(const float* logit, int32_t B, float target_a, float target_b, float gscale,
                                   float* out3, float* glogit, ali_stream_t stream) {
  if (!logit || !out3 || B <= 0) { set_error("ali_bce_logits_pair: bad argument"); return ALI_ERR_BAD_ARG; }
  hipLaunchKernelGGL(bce_logits_pair_kernel, dim3(1), dim3(kEwBlock), 0, ST(stream), logit, B, target_a, target_b, gscale,
                     out3, glogit);
  return check_launch("bce_logits_pair_kernel");
}

extern "C" int ali_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, int32_t step, int32_t* dev_step, int32_t* arrive, float grad_scale, void* p16,
                        ali_stream_t stream) {
  if (!p || !g || !m || !v || n <= 0 || (!dev_step && step < 1) || (arrive && !dev_step)) {
    set_error("ali_adam: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n, 4)), dim3(kEwBlock), 0, ST(stream), p, g, m, v, (long long)n, lr, beta1, beta2,
                     eps, (int)step, reinterpret_cast<int*>(dev_step), reinterpret_cast<int*>(arrive), grad_scale,
                     reinterpret_cast<_Float16*>(p16));
  return check_launch("adam_kernel");
}

extern "C" int ali_add_i64_multi(int32_t n, int64_t* const* ptrs, const int64_t* incs, ali_stream_t stream) {
  if (n < 0 || (n > 0 && (!ptrs || !incs))) { set_error("ali_add_i64_multi: bad argument"); return ALI_ERR_BAD_ARG; }
  for (int j0 = 0; j0 < n; j0 += kAddJobs) {
    AddJobs aj;
    memset(&aj, 0, sizeof(aj));
    for (int i = j0; i < n && i < j0 + kAddJobs; ++i) {
      if (!ptrs[i]) { set_error("ali_add_i64_multi: null counter %d", i); return ALI_ERR_BAD_ARG; }
      aj.ptr[aj.n] = reinterpret_cast<long long*>(ptrs[i]);
      aj.inc[aj.n++] = incs[i];
    }
    hipLaunchKernelGGL(add_i64_multi_kernel, dim3(1), dim3(64), 0, ST(stream), aj);
    int rc = check_launch("add_i64_multi_kernel");
    if (rc) return rc;
  }
  return ALI_OK;
}

extern "C" int ali_assemble_planes(const float* X, const int32_t* idx, const float* const* emb_tables, int32_t n_emb,
                                   const float* cont, int32_t n_cont, float* out, int32_t B, int32_t H, int32_t W,
                                   int32_t Cpad, const float* mask, int32_t mask_ld, ali_stream_t stream) {
  if (!X || !out || B <= 0 || H <= 0 || W <= 0 || n_emb < 0 || n_emb > 8 || n_cont < 0 || 1 + n_emb + n_cont > Cpad ||
      1 + n_emb + n_cont > 8 ||
      (n_emb > 0 && (!idx || !emb_tables)) || (n_cont > 0 && !cont)) {
    set_error("ali_assemble_planes: bad argument");
    return ALI_ERR_BAD_ARG;
  }
  EmbPtrs e;
  for (int j = 0; j < 8; ++j) e.t[j] = j < n_emb ? emb_tables[j] : nullptr;
  if (B > 65535 || (long long)H * W >= (1LL << 27)) { set_error("ali_assemble_planes: map too large"); return ALI_ERR_BAD_ARG; }
  const int HW = H * W;
  // >= ~2048 blocks on large maps, one block per image on small ones; a run is a multiple of 256 pixels
  int chunks = (int)std::min<long long>((HW + 1023) / 1024, std::max<long long>(1, (2048 + B - 1) / B));
  int chunk = (((HW + chunks - 1) / chunks) + 255) & ~255;
  if ((W & 255) == 0) chunk = ((chunk + W - 1) / W) * W;          // whole rows (the kernel's division-free walk)
  chunks = (HW + chunk - 1) / chunk;
  hipLaunchKernelGGL(assemble_planes_kernel, dim3(chunks, B), dim3(256), 0, ST(stream), X, idx, e, n_emb, cont, n_cont,
                     out, B, H, W, Cpad, mask, mask_ld, chunk);
  return check_launch("assemble_planes_kernel");
}
