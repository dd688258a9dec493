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
// padded to 36 floats => conflict-free ds_read_b128).
//
// What shaped the main loop (measured on MI355X, scratch/ub/mfma_overlap.hip and
// rocprofv3 SQ counters): v_mfma_f32_32x32x2_f32 occupies the SIMD for 64 cycles and
// shares the vector issue path -- a VALU instruction is NEVER hidden behind it (each
// costs its full ~4 cycles, also across the two waves of a SIMD), while LDS, vector
// memory and scalar instructions issued right behind an MFMA are.  So the loop
//   * keeps VALU work per k-tile minimal: gathers are buffer loads (32-bit offset,
//     hardware range check instead of selects; a dead lane gets an out-of-range
//     offset and reads 0), tap validity is a per-row bit mask built once, the tap of a
//     k-tile is block-uniform (channel stride % 32 == 0) and its constants come from
//     LDS one iteration ahead: no division, no scalar-memory load, no branch;
//   * slices everything that CAN hide (LDS fragment reads, LDS tile writes, global
//     gathers) into single instructions placed behind individual MFMAs.
// Layers whose channel stride is not a multiple of 32 (first layers) take a simpler
// generic loop.
#include "ali_common.h"
#include <algorithm>
#include <vector>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

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
constexpr int kMaxGrid = 5;  // taps form a product grid nr x ns (<= 5 x 5)

// Taps of one phase are the product {ir} x {is}: input pixel = (qh*mult + dh[ir], qw*mult + dw[is]),
// weight tap = wr[ir]*S + ws[is]; tap index tp = ir*ns + is.
struct Phase {
  int Hq, Wq;           // extent of this phase's output sub-grid
  int oh0, ow0, ostep;  // output pixel = (oh0 + qh*ostep, ow0 + qw*ostep)
  int mult;
  int nr, ns, M, tile0, pixmajor;
  signed char dh[kMaxGrid], dw[kMaxGrid];
  unsigned char wr[kMaxGrid], ws[kMaxGrid];
};

struct GDesc {
  const float* in;
  const float* w;
  float* out;
  float* ws;
  int* ctr;            // split-K: one arrival counter per output tile (zero between launches)
  AliEpilogue ep;
  int B, Hin, Win, Cin;
  int Hout, Wout, Cout, ldo;
  int ldw, S;          // weight row stride (floats), kernel width (taps per kernel row)
  int nphase, splitk, kt_per_split;
  int ntile_m, ntile_n;     // m-tiles (all phases) and n-tiles of the launch
  int lin1d;                // 1-D grid: tail-split and/or cost-ordered launches, see the kernel's block-index decoding
  int xcd_chunk;            // > 0 (1-D grid): XCD-contiguous tile order, blocks per XCD (see the decoding)
  int tail_u0, tail_split;  // whole tiles first, then the left-over tiles cut tail_split ways along K
  const int* order;         // AliEpilogue.tile_order (M-tile ids, longest k-loop first) or null
  int ldi;                  // pixel pitch of the gathered operand (floats): Cin, or AliEpilogue.in_ld
  int f16;             // AliEpilogue.mfma_f16: fp16 operands on v_mfma_f32_32x32x16_f16 where the fast path applies
  const _Float16* in16;   // AliEpilogue.in16 / w16 / out16 (fp16 twins of in / w / out), or null
  const _Float16* w16;
  _Float16* out16;
  unsigned in_bytes, w_bytes;
  long long out_elems;
  Phase ph[4];
};

__device__ __attribute__((aligned(64))) float g_zero[16];  // generic path: dead lanes read zeros from here

// q = m / dv, r = m % dv for 0 <= m < 2^24 (float reciprocal + one correction step)
__device__ __forceinline__ void fast_divmod(int m, int dv, float rcp, int& q, int& r) {
  q = (int)((float)m * rcp);
  r = m - q * dv;
  if (r < 0) { --q; r += dv; }
  else if (r >= dv) { ++q; r -= dv; }
}

// MODE 0: scalar gathers (channel stride % 4 != 0); 1: 16-byte gathers, tap per thread (division); 2: uniform-tap fast
// path (channel stride % 32 == 0); 3: fast path for channel stride 4 / 8 / 16 (first layers): a k-tile holds 32/C whole
// taps, the tap of a thread's 16-byte chunk is a shift, its constants a per-thread LDS read
// F16 (MODE 2 only): the fp16-MFMA variant (BASELINE config 5, esrf_acoustic.py:134-260).  Operands stay fp32 in HBM and
// are rounded to fp16 (RNE) on their way into LDS; v_mfma_f32_32x32x16_f16 accumulates in fp32.  Same prologue, tap
// skipping, split-K and epilogues as the fp32 kernel; the k-loop is a plain double-buffered one (a 32-cycle MFMA leaves
// the vector issue free most of the time, unlike the 64-cycle fp32 one: no per-slot pinning needed).
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int LDH = 40;      // fp16 LDS rows: 32 halves + 8 pad = 80 B (16-B aligned, ds_read_b128 conflict-free)

// F16 == 2: both operands are read from fp16 copies in memory (AliEpilogue.in16 / w16: the shadow the producing launch
// left through out16, the fp16 twin of the packed weights): 16-byte gathers carry 8 k-values, a k-tile is 64 deep and
// occupies exactly the fp32 tile's LDS image (128-B rows + 16 B pad), no conversion work, half the bytes through L2.
// The kernel's body.  (bx_, by_, bz_) = the block's 3-D index in a plain launch; u_ = its linear index in a 1-D
// ("lin1d": tail-split / cost-ordered) launch.  A job of a multi-job launch (gconv_multi_kernel) passes the same values,
// decoded from its share of that launch's 1-D grid.
template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, int F16 = 0, int NT = 256>
__device__ __forceinline__ void gconv_body(const GDesc& d, const int bx_, const int by_, const int bz_, const int u_) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int RPP = NT / 8;                // operand rows a pass of the block's threads stages (8 threads per row)
  constexpr int AP = BM / RPP, BP = BN / RPP;  // 16-byte gathers per thread per k-tile
  constexpr bool DMA = F16 == 3;             // fp16 twins straight into LDS (buffer_load ... lds), see the k-loop
  constexpr int LDX = DMA ? 32 : LDK;        // LDS row pitch in floats (DMA: 128-byte rows, no pad: swizzled)
  constexpr int NL = AP + BP;
  constexpr int NMF = 16 * TM * TN;          // MFMAs per wave per k-tile
  static_assert(WAVES_M * WAVES_N * 64 == NT, "one wave per 64 threads");

  __shared__ __attribute__((aligned(1024))) float As[2][BM * LDX];
  __shared__ __attribute__((aligned(1024))) float Bs[2][BN * LDX];
  __shared__ int s_rowoff[BM];  // element offset of the row's output pixel (host guarantees < 2^31), -1 = none
  __shared__ int s_rowimg[BM];
  __shared__ int s_tap[kMaxTaps];                                  // dh | dw<<8 | wt<<16 (generic path)
  __shared__ __attribute__((aligned(16))) int s_live[kMaxTaps + 8][4]; // per (live) tap {doff, woff, tap bit, 0} (bytes)
  __shared__ unsigned s_tapmask;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // Which (m-tile bx, n-tile by, k-slice bz of nsplit) this block is.  Plain launches: the 3-D block index.  "Tail split"
  // launches (all-resident grids of 1-4 tiles per CU, see finalize_and_launch): a 1-D grid whose first tail_u0 blocks
  // take one whole tile each (multiples of the CU count: every CU the same number) and whose remaining blocks share
  // the left-over tiles tail_split ways along K -- so the left-over costs every CU 1/tail_split of a tile instead of
  // costing a few CUs a whole one while the rest idle.
  int bx = bx_, by = by_, bz = bz_, nsplit = d.splitk;
  if (d.lin1d) {
    int u = u_;
    bz = 0;
    nsplit = 1;
    if (u >= d.tail_u0) {
      const int r = u - d.tail_u0;
      u = d.tail_u0 + r / d.tail_split;
      bz = r - (r / d.tail_split) * d.tail_split;
      nsplit = d.tail_split;
    }
    if (d.xcd_chunk) {
      // Deep grids of the large-map layers (spectrogram models: rows ordered (image, pixel)): neighbouring M-tiles are
      // neighbouring output rows and share (R - stride) of their R input rows, the n-tiles of an M-tile share all of
      // them.  The dispatcher deals consecutive blocks to DIFFERENT XCDs, so in raster order every L2 fetches those
      // rows again (fp16 5x5 stride-2 layers: 60 % L2 hit rate, 3.9x the algorithmic bytes over the fabric at 4.2 TB/s --
      // memory bound, profiles/r03_*esrf_f16*).  Here XCD x (blocks u = x mod 8; which blocks share an XCD is all that
      // is assumed, and only for speed) walks the contiguous range [x * chunk, (x + 1) * chunk) of (M-tile, n-tile)
      // pairs, n fastest: what its resident blocks gather overlaps, and stays in its L2.
      const int id = (u & 7) * d.xcd_chunk + (u >> 3);
      if (id >= d.ntile_m * d.ntile_n) return;
      bx = id / d.ntile_n;
      by = id - bx * d.ntile_n;
    } else if (d.order) {
      // "Longest tile first" launches: dispatch slot u -> (rank in the cost-sorted M-tile list, n-tile), n fastest.
      // The dispatcher deals workgroups round-robin over the CUs (measured: CU c of an all-resident grid holds slots
      // c, c+256, c+512, ...), so every CU receives one tile of each cost quartile instead of e.g. four corner tiles.
      // Slots u, u+8, u+16, ... run on the same XCD (the dispatcher deals slots round-robin over the 8 XCDs): the n-tiles
      // of an M-tile are placed 8 slots apart, so its gathered rows are fetched into ONE L2 instead of ntile_n of them.
      const int full = (d.ntile_m >> 3) << 3;
      int rank;
      if (u < full * d.ntile_n) {
        const int grp = u / (8 * d.ntile_n), rem = u - grp * 8 * d.ntile_n;
        rank = grp * 8 + (rem & 7);
        by = rem >> 3;
      } else {
        const int r = u - full * d.ntile_n;
        rank = full + r / d.ntile_n;
        by = r - (r / d.ntile_n) * d.ntile_n;
      }
      bx = d.order[rank];
    } else {
      by = u / d.ntile_m;
      bx = u - by * d.ntile_m;
    }
  }
  int p = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < d.nphase && bx >= d.ph[i].tile0) p = i;
  const Phase& P = d.ph[p];
  const int m0 = (bx - P.tile0) * BM;
  const int n0 = by * BN;
  const int Cin = d.Cin, Hin = d.Hin, Win = d.Win;
  const int ntaps = P.nr * P.ns;

  if (t < kMaxTaps) {
    int v = 0;
    if (t < ntaps) {
      const int ir = t / P.ns, is = t - ir * P.ns;
      v = (P.dh[ir] & 0xff) | ((P.dw[is] & 0xff) << 8) | (((int)P.wr[ir] * d.S + (int)P.ws[is]) << 16);
    }
    s_tap[t] = v;
  }
  if (t == 0) s_tapmask = 0u;

  const float rW = 1.0f / (float)P.Wq, rH = 1.0f / (float)P.Hq, rB = 1.0f / (float)d.B;
  auto decode = [&](int m, int& img, int& qh, int& qw) {
    if (P.pixmajor) {
      int pix;
      fast_divmod(m, d.B, rB, pix, img);
      fast_divmod(pix, P.Wq, rW, qh, qw);
    } else {
      int t2;
      fast_divmod(m, P.Wq, rW, t2, qw);
      fast_divmod(t2, P.Hq, rH, img, qh);
    }
  };
  for (int r = t; r < BM; r += NT) {
    const int m = m0 + r;
    int off = -1, img = 0;
    if (m < P.M) {
      int qh, qw;
      decode(m, img, qh, qw);
      off = ((img * d.Hout + P.oh0 + qh * P.ostep) * d.Wout + (P.ow0 + qw * P.ostep)) * d.ldo;
    }
    s_rowoff[r] = off;
    s_rowimg[r] = img;
  }

  // ---- per-thread gather rows: element offset of the row's base pixel and a bit mask of the taps that hit the input
  const int c4 = t & 7;
  const int r0 = t >> 3;
  int aoff[AP], aih[AP], aiw[AP];
  unsigned amask[AP];
  unsigned blockmask = 0u;
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int m = m0 + r0 + RPP * i;
    const bool valid = m < P.M;
    int qw, qh, img;
    decode(valid ? m : 0, img, qh, qw);
    aih[i] = qh * P.mult;
    aiw[i] = qw * P.mult;
    aoff[i] = ((img * Hin + aih[i]) * Win + aiw[i]) * d.ldi;
    unsigned wbits = 0u, mk = 0u;
    for (int is = 0; is < P.ns; ++is)
      if ((unsigned)(aiw[i] + P.dw[is]) < (unsigned)Win) wbits |= 1u << is;
    for (int ir = 0; ir < P.nr; ++ir)
      if ((unsigned)(aih[i] + P.dh[ir]) < (unsigned)Hin) mk |= wbits << (ir * P.ns);
    amask[i] = valid ? mk : 0u;
    blockmask |= amask[i];
  }
  __syncthreads();
  if (c4 == 0 && blockmask) atomicOr(&s_tapmask, blockmask);
  __syncthreads();
  const unsigned tapmask = s_tapmask;

  // the epilogue's bias values, requested now: by the end of the k-loop they have long arrived (a dependent load in
  // the epilogue costs a full memory latency, which is a tenth of a small launch)
  float bias_pre[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 32 + (lane & 31);
    bias_pre[j] = (d.ep.bias && n < d.Cout) ? d.ep.bias[n] : 0.f;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[AP], rb[BP];     // staging registers of the tile being fetched (fast path: the even tiles)
  f32x4 ra1[AP], rb1[BP];   // fast path: the odd tiles -- two k-tiles of gathers are in flight
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AP; ++i) *reinterpret_cast<f32x4*>(&As[buf][(r0 + RPP * i) * LDK + c4 * 4]) = ra[i];
#pragma unroll
    for (int j = 0; j < BP; ++j) *reinterpret_cast<f32x4*>(&Bs[buf][(r0 + RPP * j) * LDK + c4 * 4]) = rb[j];
  };
  const int lrow = lane & 31, lh = lane >> 5;
  const float* Ab = &As[0][(wm * WM + lrow) * LDK + lh * 4];
  const float* Bb = &Bs[0][(wn * WN + lrow) * LDK + lh * 4];
  f32x4 fa[2][TM], fb[2][TN];

  if (MODE >= 2) {
    // ================= uniform-tap fast path (MODE 2) / small power-of-two channel stride (MODE 3) =================
    // live taps, compacted: {byte offset into the activation, byte offset into the weight row, tap bit}
    if (MODE == 3) {   // all taps in order (no compaction); entries past the last tap are dead
      if (t < kMaxTaps + 8) {
        const bool live = t < ntaps && ((tapmask >> t) & 1u);
        const int tv = s_tap[t < kMaxTaps ? t : 0];
        const int dh = (signed char)(tv & 0xff), dw = (signed char)((tv >> 8) & 0xff), wt = (tv >> 16) & 0xff;
        s_live[t][0] = ((dh * Win + dw) * d.ldi) * 4;
        s_live[t][1] = live ? (wt * Cin) * 4 : (int)0xFFFFFF00u;
        s_live[t][2] = live ? (int)(1u << t) : 0;
        s_live[t][3] = 0;
      }
    } else if (t < ntaps && ((tapmask >> t) & 1u)) {
      const int pos = __popc(tapmask & ((1u << t) - 1u));
      const int tv = s_tap[t];
      const int dh = (signed char)(tv & 0xff), dw = (signed char)((tv >> 8) & 0xff), wt = (tv >> 16) & 0xff;
      s_live[pos][0] = ((dh * Win + dw) * d.ldi) * 4;
      s_live[pos][1] = (wt * Cin) * 4;
      s_live[pos][2] = (int)(1u << t);
      s_live[pos][3] = 0;
    }
    __syncthreads();
    const int tpt = MODE == 3 ? BK / Cin : 1;                 // whole taps per k-tile (MODE 3)
    const int tsub = MODE == 3 ? (c4 * 4) / Cin : 0;          // this thread's tap inside the k-tile
    const int ccol = MODE == 3 ? (c4 * 4) % Cin : c4 * 4;     // this thread's channel offset inside the tap / chunk
    const int cpt = MODE == 3 ? 1 : Cin / BK;
    const int nlive = MODE == 3 ? (ntaps + tpt - 1) / tpt : __popc(tapmask);
    const int total = nlive * cpt;
    const int per = (total + nsplit - 1) / nsplit;
    const int qb = bz * per;
    const int qe = min(total, qb + per);

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)d.in, 0, d.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w, 0, d.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFF00u;  // >= num_records: the load returns 0
    unsigned aoffB[AP], woffB[BP];
#pragma unroll
    for (int i = 0; i < AP; ++i) aoffB[i] = (unsigned)(aoff[i] + ccol) * 4u;
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const int n = n0 + r0 + RPP * j;
      woffB[j] = n < d.Cout ? (unsigned)(((long long)n * d.ldw + ccol) * 4) : OOB;
    }
    struct Ctx { int doff, woff; unsigned bit; };
    auto tile_ctx = [&](int li, int ch, bool live) -> Ctx {   // MODE 3: a per-thread LDS read; MODE 2: a broadcast
      const int lc = MODE == 3 ? (li < nlive ? li * tpt + tsub : 0)
                               : (li < nlive ? li : (nlive > 0 ? nlive - 1 : 0));
      const int4 ti = *reinterpret_cast<const int4*>(&s_live[lc][0]);
      if (MODE == 3 && li >= nlive) live = false;
      Ctx cx;
      cx.doff = ti.x + ch * (BK * 4);
      cx.woff = live ? ti.y + ch * (BK * 4) : (int)OOB;
      cx.bit = live ? (unsigned)ti.z : 0u;
      return cx;
    };
    auto load_a = [&](const Ctx& cx, int i, auto SET) {
      const unsigned off = (amask[i] & cx.bit) ? aoffB[i] + (unsigned)cx.doff : OOB;
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)off, 0, 0));
      if (decltype(SET)::value) ra1[i] = v; else ra[i] = v;
    };
    auto load_b = [&](const Ctx& cx, int j, auto SET) {
      // an invalid row / dead tile keeps the offset out of range (weights are < 2 GiB, checked on the host)
      const unsigned off = (woffB[j] | (unsigned)cx.woff) >= OOB ? OOB : woffB[j] + (unsigned)cx.woff;
      const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, (int)off, 0, 0));
      if (decltype(SET)::value) rb1[j] = v; else rb[j] = v;
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    if constexpr (F16 == 3) {
      // ---- fp16 operands in memory, staged by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write.
      // The register-staged loop above spends 13 LDS cycles per ds_write_b128 and is bound by L2 -> CU bytes at 128 x 128
      // (DESIGN.md 3.4 / 3.6); here a 512-thread block owns 256 x 256 outputs (8 waves of 128 x 64: half the operand
      // bytes per FLOP) and the accumulators may take half the register file because nothing is staged through it.
      // A wave-instruction writes 1 KiB of LDS linearly (base + 16 * lane) = 8 rows x 128 B: lane (row r, slot s) fetches
      // the row's chunk s ^ ((r >> 1) & 7), the fragment reads apply the same XOR -- conflict-free ds_read_b128 without
      // padding.  Two stages: tile q+1 is in flight while tile q is multiplied; per k-tile one counted wait (vmcnt(0):
      // this wave's pieces of tile q have landed), one raw barrier (every wave's have, and every wave is done reading
      // the other stage), then the next tile's DMA goes into that other stage.  Same k order as the F16 == 2 loop:
      // bit-identical results.  (Host: only without split-K; out-of-range rows / padding taps fetch zeros.)
      constexpr int BK16 = 64;
      const int cpt16 = Cin / BK16;
      const int total16 = nlive * cpt16;
      const int per16 = (total16 + nsplit - 1) / nsplit;
      const int qb16 = bz * per16;
      const int qe16 = min(total16, qb16 + per16);
      const __amdgpu_buffer_rsrc_t rin16 = __builtin_amdgcn_make_buffer_rsrc((void*)d.in16, 0, d.in_bytes / 2, 0x00020000);
      const __amdgpu_buffer_rsrc_t rw16 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w16, 0, d.w_bytes / 2, 0x00020000);
      typedef __attribute__((address_space(3))) void* lds_ptr;
      const int csw = c4 ^ ((r0 >> 1) & 7);                      // (RPP is a multiple of 16: the same for every pass)
      unsigned aoffS[AP], woffS[BP];
#pragma unroll
      for (int i = 0; i < AP; ++i) aoffS[i] = (unsigned)(aoff[i] + csw * 8) * 2u;
#pragma unroll
      for (int j = 0; j < BP; ++j) {
        const int n = n0 + r0 + RPP * j;
        woffS[j] = n < d.Cout ? (unsigned)(((long long)n * d.ldw + csw * 8) * 2) : OOB;
      }
      auto issue = [&](int li, int ch, int stage) {
        const int4 ti = *reinterpret_cast<const int4*>(&s_live[li][0]);
        const unsigned doff = (unsigned)((ti.x >> 1) + ch * (BK16 * 2)), woff = (unsigned)((ti.y >> 1) + ch * (BK16 * 2));
        const unsigned bit = (unsigned)ti.z;
        float* ab = &As[stage][wave * 8 * 32];
        float* bb = &Bs[stage][wave * 8 * 32];
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          const unsigned off = (amask[i] & bit) ? aoffS[i] + doff : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rin16, (lds_ptr)(ab + i * RPP * 32), 16, (int)off, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
          const unsigned off = woffS[j] >= OOB ? OOB : woffS[j] + woff;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw16, (lds_ptr)(bb + j * RPP * 32), 16, (int)off, 0, 0, 0);
        }
      };
      if (qb16 < qe16) {
        int q = qb16;
        int li = q / cpt16, ch = q - li * cpt16;
        issue(li, ch, 0);
        int stage = 0;
        const int swz = (lrow >> 1) & 7;
        for (; q < qe16; ++q) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (++ch == cpt16) { ch = 0; ++li; }
          if (q + 1 < qe16) issue(li, ch, stage ^ 1);
          const float* Ac = &As[stage][(wm * WM + lrow) * 32];
          const float* Bc = &Bs[stage][(wn * WN + lrow) * 32];
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            const int pc = ((2 * st + lh) ^ swz) * 4;
            f16x8 ha[TM], hb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
              ha[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(Ac + i * 32 * 32 + pc));
#pragma unroll
            for (int j = 0; j < TN; ++j)
              hb[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(Bc + j * 32 * 32 + pc));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[i], hb[j], acc[i][j], 0, 0, 0);
          }
          stage ^= 1;
        }
      }
      __syncthreads();       // (the epilogues reuse the operand tiles as scratch)
    } else if (F16 == 2) {
      // ---- fp16 operands in memory: k-tile = 64 halves.  Offsets below are the fp32 path's halved (bytes of halves).
      constexpr int BK16 = 64;
      const int cpt16 = Cin / BK16;
      const int total16 = nlive * cpt16;
      const int per16 = (total16 + nsplit - 1) / nsplit;
      const int qb16 = bz * per16;
      const int qe16 = min(total16, qb16 + per16);
      const __amdgpu_buffer_rsrc_t rin16 = __builtin_amdgcn_make_buffer_rsrc((void*)d.in16, 0, d.in_bytes / 2, 0x00020000);
      const __amdgpu_buffer_rsrc_t rw16 = __builtin_amdgcn_make_buffer_rsrc((void*)d.w16, 0, d.w_bytes / 2, 0x00020000);
      unsigned aoffH[AP], woffH[BP];
#pragma unroll
      for (int i = 0; i < AP; ++i) aoffH[i] = (unsigned)(aoff[i] + c4 * 8) * 2u;
#pragma unroll
      for (int j = 0; j < BP; ++j) {
        const int n = n0 + r0 + RPP * j;
        woffH[j] = n < d.Cout ? (unsigned)(((long long)n * d.ldw + c4 * 8) * 2) : OOB;
      }
      auto ctx16 = [&](int li, int ch, bool live) -> Ctx {
        const int lc = li < nlive ? li : (nlive > 0 ? nlive - 1 : 0);
        const int4 ti = *reinterpret_cast<const int4*>(&s_live[lc][0]);
        Ctx cx;
        cx.doff = (ti.x >> 1) + ch * (BK16 * 2);              // ti.x, ti.y: byte offsets of fp32 elements (multiples of 4)
        cx.woff = live ? (ti.y >> 1) + ch * (BK16 * 2) : (int)OOB;
        cx.bit = live ? (unsigned)ti.z : 0u;
        return cx;
      };
      auto load16 = [&](const Ctx& cx, auto SET) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          const unsigned off = (amask[i] & cx.bit) ? aoffH[i] + (unsigned)cx.doff : OOB;
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin16, (int)off, 0, 0));
          if (decltype(SET)::value) ra1[i] = v; else ra[i] = v;
        }
#pragma unroll
        for (int j = 0; j < BP; ++j) {
          const unsigned off = (woffH[j] | (unsigned)cx.woff) >= OOB ? OOB : woffH[j] + (unsigned)cx.woff;
          const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw16, (int)off, 0, 0));
          if (decltype(SET)::value) rb1[j] = v; else rb[j] = v;
        }
      };
      auto store16 = [&](int buf, auto SET) {   // 16 bytes = 8 halves per lane: the fp32 tile's LDS image as it stands
#pragma unroll
        for (int i = 0; i < AP; ++i)
          *reinterpret_cast<f32x4*>(&As[buf][(r0 + RPP * i) * LDK + c4 * 4]) = decltype(SET)::value ? ra1[i] : ra[i];
#pragma unroll
        for (int j = 0; j < BP; ++j)
          *reinterpret_cast<f32x4*>(&Bs[buf][(r0 + RPP * j) * LDK + c4 * 4]) = decltype(SET)::value ? rb1[j] : rb[j];
      };
      if (qb16 < qe16) {
        int q = qb16;
        int li = q / cpt16, ch = q - li * cpt16;
        load16(ctx16(li, ch, true), Set0{});
        if (++ch == cpt16) { ch = 0; ++li; }
        load16(ctx16(li, ch, q + 1 < qe16), Set1{});
        store16(0, Set0{});
        __syncthreads();
        int buf = 0;
        auto iter16 = [&](auto FETCH, auto OTHER) {
          if (++ch == cpt16) { ch = 0; ++li; }
          load16(ctx16(li, ch, q + 2 < qe16), FETCH);
          // lane l: row l&31, halves [16*step + 8*(l>>5), +8) = bytes 32*step + 16*(l>>5) of its 128-byte row
          const float* Ac = Ab + buf * (BM * LDK);
          const float* Bc = Bb + buf * (BN * LDK);
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            f16x8 ha[TM], hb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
              ha[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(Ac + i * 32 * LDK + st * 8));
#pragma unroll
            for (int j = 0; j < TN; ++j)
              hb[j] = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDK + st * 8));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[i], hb[j], acc[i][j], 0, 0, 0);
          }
          store16(buf ^ 1, OTHER);
          __syncthreads();
          buf ^= 1;
          ++q;
        };
        while (q < qe16) {
          iter16(Set0{}, Set1{});
          if (q >= qe16) break;
          iter16(Set1{}, Set0{});
        }
      }
    } else if (F16 == 1) {
      if (qb < qe) {
        _Float16* Ah = reinterpret_cast<_Float16*>(&As[0][0]);     // [2][BM][LDH]
        _Float16* Bh = reinterpret_cast<_Float16*>(&Bs[0][0]);     // [2][BN][LDH]
        static_assert(LDH * 2 <= LDK * 4, "fp16 rows fit the fp32 tile buffers");
        // staging set 0 = (ra, rb), set 1 = (ra1, rb1): the gathers of TWO k-tiles are in flight (the loop is bound by
        // bytes in flight per CU, not by the 32-cycle MFMAs: measured 2x over one set on the ESRF layers)
        auto store_tile16 = [&](int buf, auto SET) {
          constexpr int set = decltype(SET)::value;
#pragma unroll
          for (int i = 0; i < AP; ++i) {
            const f32x4 s = set ? ra1[i] : ra[i];
            const f16x4 v = {(_Float16)s[0], (_Float16)s[1], (_Float16)s[2], (_Float16)s[3]};
            *reinterpret_cast<f16x4*>(&Ah[(buf * BM + r0 + RPP * i) * LDH + c4 * 4]) = v;
          }
#pragma unroll
          for (int j = 0; j < BP; ++j) {
            const f32x4 s = set ? rb1[j] : rb[j];
            const f16x4 v = {(_Float16)s[0], (_Float16)s[1], (_Float16)s[2], (_Float16)s[3]};
            *reinterpret_cast<f16x4*>(&Bh[(buf * BN + r0 + RPP * j) * LDH + c4 * 4]) = v;
          }
        };
        int q = qb;
        int li = q / cpt, ch = q - li * cpt;
        {
          const Ctx c0 = tile_ctx(li, ch, true);
#pragma unroll
          for (int i = 0; i < AP; ++i) load_a(c0, i, Set0{});
#pragma unroll
          for (int j = 0; j < BP; ++j) load_b(c0, j, Set0{});
          if (++ch == cpt) { ch = 0; ++li; }
          const Ctx c1 = tile_ctx(li, ch, q + 1 < qe);
#pragma unroll
          for (int i = 0; i < AP; ++i) load_a(c1, i, Set1{});
#pragma unroll
          for (int j = 0; j < BP; ++j) load_b(c1, j, Set1{});
        }
        store_tile16(0, Set0{});
        __syncthreads();
        int buf = 0;
        // one iteration: tile q is multiplied out of LDS[buf]; tile q+2 is fetched into set FETCH (free since the last
        // iteration wrote it to LDS); tile q+1, fetched an iteration ago into the other set, goes to LDS[buf^1]
        auto iteration16 = [&](auto FETCH, auto OTHER) {
          if (++ch == cpt) { ch = 0; ++li; }
          const Ctx cn = tile_ctx(li, ch, q + 2 < qe);       // past the end: every offset out of range, loads return 0
#pragma unroll
          for (int i = 0; i < AP; ++i) load_a(cn, i, FETCH);
#pragma unroll
          for (int j = 0; j < BP; ++j) load_b(cn, j, FETCH);
          // lane l: row l&31, halves [16*step + 8*(l>>5), +8) of its row -- A[m][k] and W[n][k] alike
          const _Float16* Ac = Ah + (buf * BM + wm * WM + lrow) * LDH + lh * 8;
          const _Float16* Bc = Bh + (buf * BN + wn * WN + lrow) * LDH + lh * 8;
#pragma unroll
          for (int st = 0; st < 2; ++st) {
            f16x8 ha[TM], hb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ha[i] = *reinterpret_cast<const f16x8*>(Ac + i * 32 * LDH + st * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) hb[j] = *reinterpret_cast<const f16x8*>(Bc + j * 32 * LDH + st * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[i], hb[j], acc[i][j], 0, 0, 0);
          }
          store_tile16(buf ^ 1, OTHER);
          __syncthreads();
          buf ^= 1;
          ++q;
        };
        while (q < qe) {
          iteration16(Set0{}, Set1{});      // tile q+2 -> set 0 (its tile q went to LDS before), tile q+1 sits in set 1
          if (q >= qe) break;
          iteration16(Set1{}, Set0{});
        }
      }
    } else if (qb < qe) {
      // Software pipeline, two k-tiles deep: while tile q is multiplied out of LDS, tile q+1 (gathers issued one
      // iteration ago) is written to the other LDS buffer and the gathers of tile q+2 are issued.  A lone block on
      // a CU (small layers, split-K tails) has ~1.5 iterations to cover the memory latency instead of ~0.5.
      int q = qb;
      int li = q / cpt, ch = q - li * cpt;
      {
        const Ctx c0 = tile_ctx(li, ch, true);
#pragma unroll
        for (int i = 0; i < AP; ++i) load_a(c0, i, Set0{});
#pragma unroll
        for (int j = 0; j < BP; ++j) load_b(c0, j, Set0{});
        if (++ch == cpt) { ch = 0; ++li; }
        const Ctx c1 = tile_ctx(li, ch, q + 1 < qe);
#pragma unroll
        for (int i = 0; i < AP; ++i) load_a(c1, i, Set1{});
#pragma unroll
        for (int j = 0; j < BP; ++j) load_b(c1, j, Set1{});
      }
      store_tile(0);
      if (++ch == cpt) { ch = 0; ++li; }
      Ctx cxn = tile_ctx(li, ch, q + 2 < qe);
      __syncthreads();
      int buf = 0;
      // one iteration; FETCH = staging set that receives tile q+2 (the other one holds tile q+1)
      auto iteration = [&](auto FETCH) {
        constexpr int fetch = decltype(FETCH)::value;
        const float* Ac = Ab + buf * (BM * LDK);
        const float* Bc = Bb + buf * (BN * LDK);
        float* Aw = &As[buf ^ 1][r0 * LDK + c4 * 4];
        float* Bw = &Bs[buf ^ 1][r0 * LDK + c4 * 4];
        Ctx cxn2 = cxn;
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDK);
        // One MFMA per slot; behind each, at most one memory instruction:
        //   first quarter  : the NL global gathers of tile q+2
        //   per k-group    : the TM+TN fragment reads of the next k-group
        //   last quarter   : the NL LDS writes of tile q+1 (its gathers are more than an iteration old)
#pragma unroll
        for (int s = 0; s < NMF; ++s) {
          const int kg = s / (4 * TM * TN), e = (s / (TM * TN)) & 3, i = (s / TN) % TM, j = s % TN;
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kg & 1][i][e], fb[kg & 1][j][e], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          {  // gathers: slot s issues gather x when x == s * NL / (NMF/4) over the first quarter
            constexpr int Q = NMF / 4;
            if (s < Q) {
#pragma unroll
              for (int x = 0; x < NL; ++x) {
                if ((x * Q) / NL != s) continue;
                if (x < AP) load_a(cxn, x, FETCH); else load_b(cxn, x - AP, FETCH);
              }
            }
          }
          {  // fragment reads of k-group kg+1, one per slot, starting at the 2nd slot of the group
            const int sg = s - kg * 4 * TM * TN - 1;
            if (kg < 3 && sg >= 0 && sg < TM + TN) {
              if (sg < TM)
                fa[(kg + 1) & 1][sg] = *reinterpret_cast<const f32x4*>(Ac + sg * 32 * LDK + (kg + 1) * 8);
              else
                fb[(kg + 1) & 1][sg - TM] = *reinterpret_cast<const f32x4*>(Bc + (sg - TM) * 32 * LDK + (kg + 1) * 8);
            }
          }
          if (s == NMF / 2) {  // constants of the tile fetched by the next iteration (LDS broadcast read)
            if (++ch == cpt) { ch = 0; ++li; }
            cxn2 = tile_ctx(li, ch, q + 3 < qe);
          }
          {  // LDS writes over the last quarter
            constexpr int Q = NMF / 4;
            const int sw = s - 3 * Q;
            if (sw >= 0) {
#pragma unroll
              for (int x = 0; x < NL; ++x) {
                if ((x * Q) / NL != sw) continue;
                if (x < AP) *reinterpret_cast<f32x4*>(Aw + RPP * x * LDK) = fetch ? ra[x] : ra1[x];
                else *reinterpret_cast<f32x4*>(Bw + RPP * (x - AP) * LDK) = fetch ? rb[x - AP] : rb1[x - AP];
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        cxn = cxn2;
        __syncthreads();
        buf ^= 1;
        ++q;
      };
      while (q < qe) {
        iteration(Set0{});        // tile q+2 -> set 0 (q - qb even), tile q+1 sits in set 1
        if (q >= qe) break;
        iteration(Set1{});
      }
    }
  } else {
    // ================= generic path (first layers: channel stride not a multiple of 32) =================
    const int Ktot = ntaps * Cin;
    const int nkt = (Ktot + BK - 1) / BK;
    const int kt_begin = bz * d.kt_per_split;
    const int kt_end = min(nkt, kt_begin + d.kt_per_split);
    const float* wptr[BP];
    bool bvalid[BP];
#pragma unroll
    for (int j = 0; j < BP; ++j) {
      const int n = n0 + r0 + RPP * j;
      bvalid[j] = n < d.Cout;
      wptr[j] = d.w + (long long)(bvalid[j] ? n : 0) * d.ldw;
    }
    auto tile_live = [&](int kt) -> bool {
      const int tlo = (kt * BK) / Cin;
      int thi = (kt * BK + BK - 1) / Cin;
      if (thi > ntaps - 1) thi = ntaps - 1;
      const unsigned span = (thi - tlo + 1) >= 32 ? 0xffffffffu : ((1u << (thi - tlo + 1)) - 1u);
      return ((tapmask >> tlo) & span) != 0u;
    };
    auto next_live = [&](int kt) -> int {
      while (kt < kt_end && !tile_live(kt)) ++kt;
      return kt;
    };
    auto load_tile = [&](int kt, bool live) {
      const int kflat = kt * BK + c4 * 4;
      if (MODE == 1) {
        const bool kvalid = live && kflat < Ktot;
        const int tap = kvalid ? kflat / Cin : 0;
        const int c = kflat - tap * Cin;
        const int tv = s_tap[tap];
        const int dh = (signed char)(tv & 0xff), dw = (signed char)((tv >> 8) & 0xff), wt = (tv >> 16) & 0xff;
        const int doff = (dh * Win + dw) * d.ldi + c;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          const bool ok = kvalid && ((amask[i] >> tap) & 1u);
          const float* pa = ok ? d.in + aoff[i] + doff : g_zero;
          ra[i] = *reinterpret_cast<const f32x4*>(pa);
        }
        const int woff = wt * Cin + c;
#pragma unroll
        for (int j = 0; j < BP; ++j) {
          const float* pb = (kvalid && bvalid[j]) ? wptr[j] + woff : g_zero;
          rb[j] = *reinterpret_cast<const f32x4*>(pb);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int kf = kflat + e;
          const bool kvalid = live && kf < Ktot;
          const int tap = kvalid ? kf / Cin : 0;
          const int c = kf - tap * Cin;
          const int tv = s_tap[tap];
          const int dh = (signed char)(tv & 0xff), dw = (signed char)((tv >> 8) & 0xff), wt = (tv >> 16) & 0xff;
          const int doff = (dh * Win + dw) * d.ldi + c;
#pragma unroll
          for (int i = 0; i < AP; ++i) {
            const bool ok = kvalid && ((amask[i] >> tap) & 1u);
            const float* pa = ok ? d.in + aoff[i] + doff : g_zero;
            ra[i][e] = *pa;
          }
          const int woff = wt * Cin + c;
#pragma unroll
          for (int j = 0; j < BP; ++j) {
            const float* pb = (kvalid && bvalid[j]) ? wptr[j] + woff : g_zero;
            rb[j][e] = *pb;
          }
        }
      }
    };
    int kt = next_live(kt_begin);
    if (kt < kt_end) {
      load_tile(kt, true);
      store_tile(0);
      __syncthreads();
      int buf = 0;
      while (kt < kt_end) {
        const int nk = next_live(kt + 1);
        load_tile(nk, nk < kt_end);
        const float* Ac = Ab + buf * (BM * LDK);
        const float* Bc = Bb + buf * (BN * LDK);
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(Ac + i * 32 * LDK + kg * 8);
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(Bc + j * 32 * LDK + kg * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][i][e], fb[0][j][e], acc[i][j], 0, 0, 0);
        }
        store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        kt = nk;
      }
    }
  }

  // epilogue: acc(row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31).  Row offsets are fetched from LDS in one
  // batch, the optional mask / act' operands in one batch of global loads (no branch, no wait per element).
  const AliEpilogue& ep = d.ep;
  // BNM (fused BatchNorm reductions, see AliEpilogue): 0 none; 1 column sums (v~, v~^2) of the stored value; 2 column
  // sums (g~ * xhat, g~) of a data gradient.  Accumulated per lane over its rows, combined across the half-waves,
  // then across the tile's row-waves through LDS in a fixed order: deterministic, no atomics.
  auto run_epilogue = [&](auto has_mask_t, auto has_dact_t, auto bn_t, const bool partial, float* outp) {
    constexpr bool HAS_MASK = decltype(has_mask_t)::value, HAS_DACT = decltype(has_dact_t)::value;
    constexpr int BNM = decltype(bn_t)::value;
    float bs0[TN], bs1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bs0[j] = bs1[j] = 0.f;
    const _Float16* dact16 = (HAS_DACT && d.f16) ? reinterpret_cast<const _Float16*>(ep.dact_y16) : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // 8 rows at a time keeps the register footprint small
        int roff[8], rimg[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = h * 8 + q;
          const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          roff[q] = s_rowoff[row];
          if (HAS_MASK || BNM != 0) rimg[q] = s_rowimg[row];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * WN + j * 32 + (lane & 31);
          const bool nok = n < d.Cout;
          const int nc = nok ? n : 0;
          const float bias = partial ? 0.f : bias_pre[j];
          float mk[8], dy[8], bx[8], bm1[8], bm2[8];
          float bmu = 0.f, bis = 0.f;
          if (HAS_MASK) {
#pragma unroll
            for (int q = 0; q < 8; ++q) mk[q] = ep.mask[(long long)rimg[q] * ep.mask_ld + nc];
          }
          if (HAS_DACT) {
            if (dact16) {                  // (fp16 path: the twin of the previous layer's output, half the bytes)
#pragma unroll
              for (int q = 0; q < 8; ++q) dy[q] = (float)dact16[(roff[q] < 0 ? 0 : roff[q]) + nc];
            } else {
#pragma unroll
              for (int q = 0; q < 8; ++q) dy[q] = ep.dact_y[(roff[q] < 0 ? 0 : roff[q]) + nc];
            }
          }
          if (BNM == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
              bm1[q] = ep.bn_stat_mask ? ep.bn_stat_mask[(long long)rimg[q] * ep.bn_mask_ld + nc] : 1.f;
          }
          if (BNM == 2) {
            bmu = ep.bn_mean[nc];
            bis = ep.bn_invstd[nc];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              bx[q] = ep.bn_x[(roff[q] < 0 ? 0 : roff[q]) + nc];
              bm1[q] = ep.bn_mask_in ? ep.bn_mask_in[(long long)rimg[q] * ep.bn_mask_ld + nc] : 1.f;
              bm2[q] = ep.bn_mask_pre ? ep.bn_mask_pre[(long long)rimg[q] * ep.bn_mask_ld + nc] : 1.f;
            }
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float v = acc[i][j][h * 8 + q];
            if (!partial) {
              v = apply_act(v + bias, ep.act, ep.slope);
              if (HAS_MASK) v *= mk[q];
              if (HAS_DACT) v *= act_grad_from_output(dy[q], ep.dact, ep.dslope);
            }
            if (nok && roff[q] >= 0) {
              if (partial) __hip_atomic_store(outp + roff[q] + n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else {
                outp[roff[q] + n] = v;
                if (d.out16) d.out16[roff[q] + n] = (_Float16)v;
              }
              if (BNM == 1) {
                const float vs = v * bm1[q];
                bs0[j] += vs;
                bs1[j] += vs * vs;
              }
              if (BNM == 2) {
                const float gs = v * bm2[q];
                bs0[j] += gs * ((bx[q] * bm1[q] - bmu) * bis);
                bs1[j] += gs;
              }
            }
          }
        }
      }
    }
    if (BNM != 0) {
      // the k-loop is over for every wave of the block (it ends in a barrier; the split-K winner passed two more):
      // the operand tiles in LDS are dead and serve as scratch [2][WAVES_M][BN]
      float* red = &As[0][0];
      static_assert(2 * WAVES_M * BN <= BM * LDK, "reduction scratch fits the A tile");
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bs0[j] += __shfl_xor(bs0[j], 32, 64);
        bs1[j] += __shfl_xor(bs1[j], 32, 64);
        if (lane < 32) {
          red[(0 * WAVES_M + wm) * BN + wn * WN + j * 32 + lane] = bs0[j];
          red[(1 * WAVES_M + wm) * BN + wn * WN + j * 32 + lane] = bs1[j];
        }
      }
      __syncthreads();
      if (t < BN && n0 + t < d.Cout) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES_M; ++w) {
          a += red[(0 * WAVES_M + w) * BN + t];
          b += red[(1 * WAVES_M + w) * BN + t];
        }
        int slot = bx;
        const int nslots = d.ntile_m;
        if (ep.bn_groups > 1 && P.pixmajor) {   // tiles of one pixel position: images [0,B) in order, passes back to back
          const int tpp = d.B / BM, tpg = tpp / ep.bn_groups;
          const int pix = bx / tpp, tin = bx - pix * tpp;
          const int grp = tin / tpg;
          slot = grp * (nslots / ep.bn_groups) + pix * tpg + (tin - grp * tpg);
        }
        ep.bn_part[((long long)0 * d.Cout + n0 + t) * nslots + slot] = a;
        ep.bn_part[((long long)1 * d.Cout + n0 + t) * nslots + slot] = b;
      }
    }
  };
  using BN0 = std::integral_constant<int, 0>;
  using BN1 = std::integral_constant<int, 1>;
  using BN2 = std::integral_constant<int, 2>;
  using T_ = std::true_type;
  using F_ = std::false_type;
  if (!DMA && nsplit > 1) {          // (the LDS-DMA kernel is launched without split-K: no slab code in its register budget)
    // split-K: every block stores its raw partial tile in its slab; the block that arrives last at the tile's
    // counter sums the slabs in slab order (deterministic whichever block that is) and runs the real epilogue.
    // Slab stores / loads are device-scope (sc1) accesses -- written through to memory, never served from another
    // XCD's stale L2 line -- so no cache-wide release / acquire fence is needed, only "my stores have completed".
    //
    // Why the last block sees every slab (hardware-level argument; tests/test_gpu_kernels.py stress-tests it):
    //  1. a slab element is written by a relaxed agent-scope atomic store = global_store ... sc1: the write goes
    //     through this XCD's L2 to the device-coherent level and is acknowledged from there;
    //  2. s_waitcnt(0) retires only when all of this thread's stores are acknowledged, and the barrier behind it
    //     (a workgroup fence for the compiler: no memory access is moved across it) means "every store of this
    //     block is complete at device scope" before thread 0 issues the counter atomic;
    //  3. the counter is an agent-scope RMW executed at that same coherent level; the block that reads S-1 from it
    //     therefore runs after all S-1 other RMWs, each of which was issued after its block's step 2;
    //  4. the winner's slab loads are issued after the barrier that publishes s_last (so after the RMW returned) and
    //     are buffer_load ... sc1: device-scope loads, never served from a line this XCD's L2 kept from an earlier
    //     launch.  The counter is reset by the winner alone, after all arrivals: the next launch (stream order)
    //     finds it at zero.
    // Nothing else is shared between the blocks of a tile, so no cache-wide fence is required.
    run_epilogue(F_{}, F_{}, BN0{}, true, d.ws + (long long)bz * d.out_elems);
    __shared__ int s_last;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (t == 0) {
      int* c = d.ctr + (by * d.ntile_m + bx);
      const int arrived = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = arrived == nsplit - 1;
      if (s_last) __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // all blocks have arrived
    }
    __syncthreads();
    if (!s_last) return;
    // all 16 rows x 4 slabs of a sub-tile are requested before the first one is consumed (the loads bypass L2:
    // one memory latency per 4 slabs instead of one per slab).  Buffer loads: 32-bit offsets, invalid rows -> 0.
    const __amdgpu_buffer_rsrc_t rws =
        __builtin_amdgcn_make_buffer_rsrc((void*)d.ws, 0, (unsigned)(nsplit * d.out_elems * 4), 0x00020000);
    constexpr int kSc1 = 16;          // cache policy: device scope
    const unsigned slab_bytes = (unsigned)(d.out_elems * 4);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + (lane & 31);
        unsigned voff[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = s_rowoff[wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
          voff[r] = (n < d.Cout && ro >= 0) ? (unsigned)(ro + n) * 4u : 0xFFFFFF00u;
        }
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 0.f;
        // (the bounds check of a buffer load covers voffset only: every soffset used here is a real slab)
        int sl0 = 0;
        for (; sl0 + 4 <= nsplit; sl0 += 4) {
          float tmp[4][16];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned soff = (unsigned)(sl0 + u) * slab_bytes;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              tmp[u][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rws, (int)voff[r], (int)soff, kSc1));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] += tmp[u][r];
        }
        for (; sl0 < nsplit; ++sl0) {
          const unsigned soff = (unsigned)sl0 * slab_bytes;
          float tmp[16];
#pragma unroll
          for (int r = 0; r < 16; ++r)
            tmp[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rws, (int)voff[r], (int)soff, kSc1));
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] += tmp[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = v[r];
      }
    }
  }
  const bool hm = ep.mask != nullptr, hd = ep.dact_y != nullptr;
  if (ep.bn_part != nullptr) {   // host: mode 1 never comes with dact_y, mode 2 with neither mask nor dact_y
    if (ep.bn_mode == 2) run_epilogue(F_{}, F_{}, BN2{}, false, d.out);
    else if (hm) run_epilogue(T_{}, F_{}, BN1{}, false, d.out);
    else run_epilogue(F_{}, F_{}, BN1{}, false, d.out);
  }
  else if (hm && hd) run_epilogue(T_{}, T_{}, BN0{}, false, d.out);
  else if (hm) run_epilogue(T_{}, F_{}, BN0{}, false, d.out);
  else if (hd) run_epilogue(F_{}, T_{}, BN0{}, false, d.out);
  else run_epilogue(F_{}, F_{}, BN0{}, false, d.out);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, int F16 = 0>
__global__ __launch_bounds__(256, 2) void gconv_kernel(const GDesc d) {
  gconv_body<BM, BN, WAVES_M, WAVES_N, MODE, F16>(d, blockIdx.x, blockIdx.y, blockIdx.z, blockIdx.x);
}

// 512-thread variant (the LDS-DMA fp16 loop: 8 waves, one block per CU)
template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, int F16>
__global__ __launch_bounds__(512, 1) void gconv_kernel512(const GDesc d) {
  gconv_body<BM, BN, WAVES_M, WAVES_N, MODE, F16, 512>(d, blockIdx.x, blockIdx.y, blockIdx.z, blockIdx.x);
}

// Several independent GEMM launches of the same kernel variant in ONE launch (ali_gemm_launch_multi): the Encoder and the
// Generator of an ALI iteration (mnist.py:224-226), the two branches of its backward pass, the dz / dx stacks of the
// Discriminator are independent chains, so their layers go out pairwise -- half the launch boundaries, and the small
// tail GEMMs (M <= 1024: 512 x 512 x 512 at 22 TF/s alone on 64-256 CUs) run beside a large layer's blocks instead of
// alone.  Jobs are laid out one after the other in a 1-D grid, longest k-loop first; every job keeps its own workspace
// (split-K slabs and arrival counters), epilogue and dispatch-order table.
constexpr int kGJobs = 4;
struct GJobs { int n, pad_; int blk0[kGJobs]; int gx[kGJobs], gy[kGJobs]; GDesc j[kGJobs]; };
static_assert(sizeof(GJobs) <= 4000, "kernel argument segment");
template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, int F16 = 0>
__global__ __launch_bounds__(256, 2) void gconv_multi_kernel(const GJobs jobs) {
  int k = 0;
#pragma unroll 1
  for (int i = 1; i < jobs.n; ++i)
    if ((int)blockIdx.x >= jobs.blk0[i]) k = i;
  const int u = blockIdx.x - jobs.blk0[k];
  const int gx = jobs.gx[k], gy = jobs.gy[k];
  const int t2 = u / gx;
  gconv_body<BM, BN, WAVES_M, WAVES_N, MODE, F16>(jobs.j[k], u - t2 * gx, t2 % gy, t2 / gy, u);
}

struct TileCfg { int bm, bn; };

// fp32 MFMA cannot overlap with VALU work of the same SIMD, and a lone wave per SIMD cannot hide its barrier / memory
// waits: prefer the largest tile that still gives every CU two resident blocks (>= 512 blocks), measured best on the
// MorphoMNIST layer shapes at bs=512 (scratch/mb3.py).
static TileCfg pick_tile(long long M, int N, bool f16, int cin) {
  TileCfg best = {64, 64};
  if (N <= 32) { best.bm = 128; best.bn = 32; return best; }
  auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * (long long)((N + bn - 1) / bn); };
  if (f16) {
    // the fp16 loop is bound by the bytes its blocks pull through L2 (a 32-cycle MFMA is 16x the fp32 rate; operands
    // are still fp32 in memory): 128x128 tiles halve them per FLOP -- measured 149 -> 290 TF/s on the 1024 -> 2048
    // layer of the ESRF stacks -- as soon as they still give every other CU a block.
    // Round 3, measured and rejected on the ESRF iteration (62.1 ms of GEMM time with this rule): a 256 x 256 tile with
    // 8 waves (128 accumulator registers per lane leave room for one staged k-tile: 66.5 ms) or with 4 waves and the
    // accumulators in AGPRs (one wave per SIMD hides nothing: 194 ms); three k-tiles of gathers in flight instead of
    // two (62.5 ms: the loop is not waiting for global memory); XCD-contiguous tile order (kept, neutral).  Per k-tile
    // a CU issues 64 ds_write_b128 (13 LDS cycles each) beside its 128 ds_read_b128: 1344 LDS cycles for 1024 MFMA
    // cycles -- the register-staged LDS fill is the limit; the fix is LDS-DMA staging, not tile area.
    if (N > 64 && blocks(128, 128) >= kNumCU / 2) { best.bm = 128; best.bn = 128; return best; }
    if (N > 64 && blocks(64, 128) >= kNumCU / 2) { best.bm = 64; best.bn = 128; return best; }
    // (64-channel outputs, e.g. the data gradient of the stacks' second conv, 1.6 M rows x 64 <- 128 x 25 at 330 TF/s:
    // a 192 x 64 tile -- 256 staged rows per k-tile like 128 x 128, two blocks per CU -- won 10 % stand-alone
    // (scratch/mb_dgrad16.py) and lost 17 % inside the iteration, 256 x 64 with one block per CU lost 50 %: dropped)
    return best;
  }
  // measured (scratch/mb3.py): 64x64 (4 resident blocks per CU) wins or ties up to a few thousand blocks;
  // larger tiles only pay through lower L2/HBM traffic once the grid is many waves deep
  if (blocks(64, 64) <= 16 * kNumCU || N <= 64) return best;
  if (blocks(64, 128) <= 16 * kNumCU) { best.bm = 64; best.bn = 128; return best; }
  best.bm = 128; best.bn = 128;
  return best;
}

// tile shape, row order and M-tile count of a launch (everything the grid's x extent depends on)
// dma_ok: the launch may use the 256 x 256 LDS-DMA fp16 kernel (both fp16 twins present, see finalize_and_launch); the
// host-side queries (ali_conv_mtiles / ali_conv_tile_order) never say so -- such launches carry no fused BatchNorm, and a
// dispatch-order table of another tile count is ignored by the launch
static int plan_tiles(GDesc& d, bool vec, TileCfg& tc, int& max_taps, bool dma_ok = false) {
  long long Mtot = 0;
  max_taps = 0;
  for (int i = 0; i < d.nphase; ++i) {
    Mtot += d.ph[i].M;
    if (d.ph[i].nr * d.ph[i].ns > max_taps) max_taps = d.ph[i].nr * d.ph[i].ns;
    if (d.ph[i].M >= (1 << 24)) { set_error("gconv: more than 2^24 rows in one phase"); return -1; }
  }
  const bool f16_loop = d.f16 && vec && (d.Cin % BK) == 0;
  const long long Mpick = Mtot * (tuning().tile_m_scale > 0 ? tuning().tile_m_scale : 1);
  tc = pick_tile(Mpick, d.Cout, f16_loop, d.Cin);
  if (tuning().bm > 0 && tuning().bn > 0) { tc.bm = tuning().bm; tc.bn = tuning().bn; }
  if (tc.bm > 128 && !dma_ok) tc = pick_tile(Mpick, d.Cout, f16_loop, d.Cin);      // (a forced 256 x 256 applies to DMA launches only)
  if (dma_ok && tuning().bm == 0) {
    // 256 x 256 as soon as it fills the chip once (one block per CU).  Stand-alone (scratch/ub/dma_gemm.hip, a 1-D
    // "convolution" with the layers' reuse): 881-1008 TF/s against 700-840 of the 128 x 128 register-staged loop.  Inside
    // the ESRF iteration the forward convolutions 128 -> 256 ... 512 -> 1024 gain 5-8 % (697 -> 732, 781 -> 842, 783 ->
    // 849 TF/s); launches in sub-pixel phases (data gradients: 786 -> 591) and tiny pixel-major maps (7 x 7: 911 -> 810)
    // LOSE -- a 256-row tile spans two pixel positions, so fewer padding taps are skipped tile-wide, and there is no
    // cost-ordered dispatch table for this tile: they keep the 128 x 128 loop.
    const long long b256 = ((Mpick + 255) / 256) * (long long)((d.Cout + 255) / 256);
    const bool plain = d.nphase == 1 && (long long)d.ph[0].Hq * d.ph[0].Wq >= 200;
    if (plain && d.Cout >= 256 && b256 >= (kNumCU * 3) / 4) { tc.bm = 256; tc.bn = 256; }
  }
  if (!vec && tc.bn == 128) tc.bn = 64;
  int tiles = 0;
  for (int i = 0; i < d.nphase; ++i) {
    // small maps with a large batch: order rows (pixel, image) so that every M-tile sees one pixel position
    // and the taps falling outside the input (padding, 1x1 -> 3x3 transposed convs) are skipped tile-wide
    d.ph[i].pixmajor = (d.ph[i].Hq * d.ph[i].Wq <= 1024 && d.B >= 64) ? 1 : 0;
    d.ph[i].tile0 = tiles;
    tiles += (d.ph[i].M + tc.bm - 1) / tc.bm;
  }
  return tiles;
}

// k-loop length (live taps) of every M-tile of a launch whose rows are ordered (pixel, image): the taps that fall
// outside the input are skipped tile-wide, so edge tiles of padded / transposed convolutions are short.  Mirrors the
// kernel's amask / tapmask computation.  Returns false when some phase is not pixel-major (tiles mix positions).
static bool tile_costs(const GDesc& d, int bm, std::vector<int>& cost) {
  cost.clear();
  for (int p = 0; p < d.nphase; ++p) {
    const Phase& P = d.ph[p];
    if (!P.pixmajor) return false;
    const int nt = (P.M + bm - 1) / bm;
    for (int ti = 0; ti < nt; ++ti) {
      const int m_lo = ti * bm, m_hi = std::min(P.M, m_lo + bm) - 1;
      unsigned mask = 0u;
      for (int pix = m_lo / d.B; pix <= m_hi / d.B; ++pix) {
        const int qh = pix / P.Wq, qw = pix - qh * P.Wq;
        unsigned wbits = 0u;
        for (int is = 0; is < P.ns; ++is)
          if ((unsigned)(qw * P.mult + P.dw[is]) < (unsigned)d.Win) wbits |= 1u << is;
        for (int ir = 0; ir < P.nr; ++ir)
          if ((unsigned)(qh * P.mult + P.dh[ir]) < (unsigned)d.Hin) mask |= wbits << (ir * P.ns);
      }
      cost.push_back(__builtin_popcount(mask));
    }
  }
  return true;
}

// A recorded launch (AliGemmJob): everything finalize_and_launch decided, for ali_gemm_launch_multi
struct GJobRec {
  uint64_t state;          // 1 = recorded
  int variant;             // 0: <64,64,..,MODE 2> fp32, 1: <128,32,..,MODE 2> fp32 (the variants a multi-job kernel exists for)
  int gx, gy, gz;          // its own grid
  int cost;                // k-tiles per block (order of the jobs inside a combined launch)
  GDesc d;
};
static_assert(sizeof(GJobRec) <= sizeof(AliGemmJob), "AliGemmJob too small");

static int finalize_and_launch(GDesc& d, void* ws, size_t ws_bytes, hipStream_t stream, bool vec, bool dense_k,
                               AliGemmJob* job = nullptr) {
  TileCfg tc;
  int max_taps = 0;
  const bool uni = vec && (d.Cin % BK) == 0;
  // (fp16 launches are never recorded as jobs of a combined launch: a job pointer does not matter here)
  const bool dma_ok = d.f16 && uni && d.in16 && d.w16 && (d.Cin % 64) == 0 && !d.ep.bn_part && d.ep.in_ld <= 0 &&
                      tuning().no_dma16 == 0 && tuning().splitk == 0;
  const int tiles = plan_tiles(d, vec, tc, max_taps, dma_ok);
  if (tiles < 0) return ALI_ERR_BAD_ARG;
  const bool dma = tc.bm == 256;
  const int max_nkt = (max_taps * d.Cin + BK - 1) / BK;
  const bool pow2 = vec && (d.Cin == 4 || d.Cin == 8 || d.Cin == 16) && max_taps <= kMaxTaps;
  if (d.ep.bn_part) {
    const AliEpilogue& e = d.ep;
    bool ok = (e.bn_mode == 1 || e.bn_mode == 2) && !e.dact_y;
    if (e.bn_mode == 2) ok = ok && !e.mask && e.bn_x && e.bn_mean && e.bn_invstd && e.bn_groups <= 1;
    if (e.bn_mode == 1 && e.bn_groups > 1) {
      ok = ok && d.nphase == 1 && d.B % e.bn_groups == 0;
      const long long rows_g = (long long)(d.B / e.bn_groups) * (d.ph[0].pixmajor ? 1 : d.ph[0].Hq * d.ph[0].Wq);
      ok = ok && rows_g % tc.bm == 0 && (!d.ph[0].pixmajor || d.B % tc.bm == 0);
    }
    if (!ok) { set_error("gconv: bad fused BatchNorm epilogue (see AliEpilogue / ali_conv_mtiles)"); return ALI_ERR_BAD_ARG; }
    if (e.bn_slots > 0 && e.bn_slots != tiles) {
      set_error("gconv: bn_part was sized for %d slots, this launch has %d M-tiles (stale ali_conv_mtiles result?)", e.bn_slots, tiles);
      return ALI_ERR_BAD_ARG;
    }
  }
  const int ntile_n = (d.Cout + tc.bn - 1) / tc.bn;
  d.ldi = d.Cin;
  if (d.ep.in_ld > 0 || d.ep.out_ld > 0) {     // operands that are column ranges of wider row-major buffers
    if ((d.ep.in_ld > 0 && (d.ep.in_ld < d.Cin || (vec && d.ep.in_ld % 4))) ||
        (d.ep.out_ld > 0 && (d.ep.out_ld < d.Cout || d.ep.bn_part))) {
      set_error("gconv: bad in_ld / out_ld (>= the channel count, in_ld % 4 == 0 for vector gathers, no out_ld with fused BatchNorm)");
      return ALI_ERR_BAD_ARG;
    }
    if (d.ep.in_ld > 0) d.ldi = d.ep.in_ld;
    if (d.ep.out_ld > 0) d.ldo = d.ep.out_ld;
  }
  d.out_elems = (long long)d.B * d.Hout * d.Wout * d.ldo;
  const long long in_elems = ((long long)d.B * d.Hin * d.Win - 1) * d.ldi + d.Cin;
  const long long w_elems = (long long)d.Cout * d.ldw;
  if (d.out_elems >= (1LL << 31) || in_elems >= (1LL << 30) || w_elems >= (1LL << 29)) {
    set_error("gconv: tensor too large for 32-bit byte offsets");
    return ALI_ERR_BAD_ARG;
  }
  d.in_bytes = (unsigned)(in_elems * 4);
  d.w_bytes = (unsigned)(w_elems * 4);
  // split-K when the grid cannot give every CU two blocks.  All blocks of such a grid are resident at once, so the
  // launch lasts as long as the fullest CU: n = ceil(blocks*S / CUs) blocks of nkt/S k-tiles each, a little slower
  // per block when the CU holds fewer than 4, plus about half a k-tile of slab traffic per slab
  // (scratch/sweep_splitk.py).  Data-gradient launches skip dead taps tile by tile, so their k-depth is not known
  // here: they keep the plain "two blocks per CU" rule.
  int S = 1;
  const long long blocks = (long long)tiles * ntile_n;
  if (blocks < 2 * kNumCU && max_nkt >= 8 && dense_k) {
    static const double eff[5] = {1.0, 0.82, 0.96, 0.99, 1.0};
    double best = 1e30;
    for (int sc = 1; sc <= 8 && max_nkt / sc >= 4; ++sc) {
      if (sc > 1 && (size_t)sc * d.out_elems * sizeof(float) > ws_payload_bytes(ws_bytes)) break;
      const long long n = (blocks * sc + kNumCU - 1) / kNumCU;
      const double cost = (double)n / sc / eff[n < 4 ? n : 4] * max_nkt + (sc > 1 ? 0.5 * sc : 0.0);
      if (cost < 0.97 * best) { best = cost; S = sc; }
    }
  } else if (blocks < 2 * kNumCU && max_nkt >= 8) {
    S = (int)((2 * kNumCU + blocks - 1) / blocks);
    if (S > max_nkt / 4) S = max_nkt / 4;
    if (S > 32) S = 32;
    while (S > 1 && (size_t)S * d.out_elems * sizeof(float) > ws_payload_bytes(ws_bytes)) --S;
    if (S < 1) S = 1;
  }
  if (tuning().splitk > 0 && (size_t)tuning().splitk * d.out_elems * sizeof(float) <= ws_payload_bytes(ws_bytes))
    S = tuning().splitk;
  if (dma) S = 1;                      // (the LDS-DMA kernel runs whole k-loops only)
  if (blocks > (long long)(kWsReserved / sizeof(int)) || !ws) S = 1;   // one arrival counter per tile
  while (S > 1 && (long long)S * d.out_elems * 4 >= 0xFF000000LL) --S;  // slabs addressed with 32-bit byte offsets
  d.splitk = S;
  d.kt_per_split = (max_nkt + S - 1) / S;
  if (d.kt_per_split < 1) d.kt_per_split = 1;
  d.ntile_m = tiles;
  d.ntile_n = ntile_n;
  // Tail split.  A grid of 1-4 tiles per CU is resident all at once and lasts as long as the fullest CU: 576 tiles on
  // 256 CUs cost 3 tile-times although the chip only has 2.25 to do.  The first floor(blocks / CUs) * CUs tiles run
  // whole; the R left-over tiles are cut Sr ways along K (Sr * R <= CUs: one piece per CU), folded in the kernel
  // like any split-K tile.  Only those R tiles pay slab traffic.  (uniform-tap loops only; fp32 and fp16 alike)
  d.tail_split = 1;
  d.tail_u0 = 0;
  if (S == 1 && uni && ws && blocks > kNumCU && blocks < 4 * kNumCU && blocks <= (long long)(kWsReserved / sizeof(int))
      && tuning().splitk == 0 && !dma) {
    const int R = (int)(blocks % kNumCU);
    int Sr = 1;
    while (R > 0 && Sr * 2 <= 8 && Sr * 2 * R <= kNumCU && max_nkt / (Sr * 2) >= 4) Sr *= 2;
    if (Sr > 1 && (size_t)Sr * d.out_elems * sizeof(float) <= ws_payload_bytes(ws_bytes) &&
        (long long)Sr * d.out_elems * 4 < 0xFF000000LL) {
      d.tail_split = Sr;
      d.tail_u0 = (int)(blocks - R);
    }
  }
  // cost-ordered dispatch (AliEpilogue.tile_order / ali_conv_tile_order): only launches whose blocks each own a whole
  // k-loop or a tail-split share of one
  d.order = nullptr;
  if (d.ep.tile_order && uni && S == 1 && d.ep.tile_order_n == tiles && tuning().no_order == 0)
    d.order = reinterpret_cast<const int*>(d.ep.tile_order);
  d.lin1d = (d.tail_split > 1 || d.order) ? 1 : 0;
  if (d.tail_split == 1) d.tail_u0 = (int)blocks;
  // XCD-contiguous tile order for deep grids of (image, pixel)-ordered rows (see the kernel's decoding)
  d.xcd_chunk = 0;
  bool img_major = d.nphase >= 1;
  for (int i = 0; i < d.nphase; ++i) img_major = img_major && !d.ph[i].pixmajor;
  if (!d.lin1d && S == 1 && !job && img_major && blocks >= 8LL * kNumCU && blocks < (1LL << 30) && tuning().no_xcd == 0) {
    d.xcd_chunk = (int)((blocks + 7) / 8);
    d.lin1d = 1;
  }
  d.ctr = reinterpret_cast<int*>(ws);
  d.ws = reinterpret_cast<float*>(ws_payload(ws));
  dim3 grid(tiles, ntile_n, S), block(256);
  if (d.lin1d) grid = dim3(d.tail_u0 + (int)(blocks - d.tail_u0) * d.tail_split, 1, 1);
  if (d.xcd_chunk) grid = dim3(8 * d.xcd_chunk, 1, 1);
  if (tiles == 0 || d.out_elems == 0) return ALI_OK;
  const bool f16 = d.f16 && uni;
  const bool op16 = f16 && d.in16 && d.w16 && (d.Cin % 64) == 0;
  if (!d.f16) d.out16 = nullptr;    // twins are left by mfma_f16 launches only -- also by those of them that keep fp32
                                    // arithmetic (first layers): their consumers then read fp16 operands from memory
  if (job && uni && !f16 && ((tc.bm == 64 && tc.bn == 64) || (tc.bm == 128 && tc.bn == 32))) {
    GJobRec r;
    memset(&r, 0, sizeof(r));
    r.state = 1;
    r.variant = tc.bm == 64 ? 0 : 1;
    r.gx = (int)grid.x; r.gy = (int)grid.y; r.gz = (int)grid.z;
    r.cost = (max_nkt + S - 1) / S;
    r.d = d;
    memcpy(job, &r, sizeof(r));
    return ALI_OK;
  }
#define LAUNCH(BM_, BN_, WMM, WNN)                                                                    \
  do {                                                                                                  \
    if (op16) hipLaunchKernelGGL((gconv_kernel<BM_, BN_, WMM, WNN, 2, 2>), grid, block, 0, stream, d);  \
    else if (f16) hipLaunchKernelGGL((gconv_kernel<BM_, BN_, WMM, WNN, 2, 1>), grid, block, 0, stream, d); \
    else if (uni) hipLaunchKernelGGL((gconv_kernel<BM_, BN_, WMM, WNN, 2>), grid, block, 0, stream, d); \
    else if (pow2) hipLaunchKernelGGL((gconv_kernel<BM_, BN_, WMM, WNN, 3>), grid, block, 0, stream, d);  \
    else if (vec) hipLaunchKernelGGL((gconv_kernel<BM_, BN_, WMM, WNN, 1>), grid, block, 0, stream, d); \
    else hipLaunchKernelGGL((gconv_kernel<BM_, BN_, WMM, WNN, 0>), grid, block, 0, stream, d);          \
  } while (0)
  if (dma) {
    if (tc.bn != 256 || !op16) { set_error("gconv: no kernel for this tile"); return ALI_ERR_BAD_ARG; }
    block = dim3(512);
    hipLaunchKernelGGL((gconv_kernel512<256, 256, 2, 4, 2, 3>), grid, block, 0, stream, d);
    return check_launch("gconv_kernel512");
  }
  if (tc.bm > 128 || tc.bn > 128) { set_error("gconv: no kernel for this tile"); return ALI_ERR_BAD_ARG; }
  else if (tc.bm == 128 && tc.bn == 128) LAUNCH(128, 128, 2, 2);
  else if (tc.bm == 128 && tc.bn == 64) LAUNCH(128, 64, 2, 2);
  else if (tc.bm == 128 && tc.bn == 32) LAUNCH(128, 32, 4, 1);
  else if (tc.bm == 64 && tc.bn == 128) LAUNCH(64, 128, 2, 2);
  else LAUNCH(64, 64, 2, 2);
#undef LAUNCH
  return check_launch("gconv_kernel");
}

static bool geom_ok(const AliConvGeom* g) {
  if (!g) return false;
  if (g->B <= 0 || g->H <= 0 || g->W <= 0 || g->C <= 0 || g->P <= 0 || g->Q <= 0 || g->K <= 0) return false;
  if (g->R <= 0 || g->S <= 0 || g->R > kMaxGrid || g->S > kMaxGrid || g->stride <= 0 || g->pad < 0) return false;
  if (g->pad > 100) return false;  // taps are stored as signed bytes
  return true;
}

static void fill_epilogue(GDesc& d, const AliEpilogue* ep) {
  if (ep) d.ep = *ep;
  else memset(&d.ep, 0, sizeof(d.ep));
  d.f16 = d.ep.mfma_f16 != 0;
  d.in16 = reinterpret_cast<const _Float16*>(d.ep.in16);
  d.w16 = reinterpret_cast<const _Float16*>(d.ep.w16);
  d.out16 = reinterpret_cast<_Float16*>(d.ep.out16);
}

}  // namespace ali

using namespace ali;

extern "C" const char* ali_last_error(void) { return ali::get_error(); }
extern "C" int ali_version(void) { return 1; }
extern "C" void ali_reload_tuning(void) { ali::tuning_slot() = ali::read_tuning(); }

extern "C" size_t ali_conv_workspace_bytes(const AliConvGeom* g, int32_t which) {
  if (!geom_ok(g)) return 0;
  if (which == 0) return kWsReserved + (size_t)32 * g->B * g->P * g->Q * g->K * sizeof(float);
  if (which == 1) return kWsReserved + (size_t)32 * g->B * g->H * g->W * g->C * sizeof(float);
  return kWsReserved + (size_t)64 * g->R * g->S * g->C * g->K * sizeof(float);
}


// ---------------------------------------------------------------------------------------------------------------
// First Conv2d of the MorphoMNIST Discriminator (mnist.py:108: 5 -> 32 channels, 5x5, stride 1, no padding, on the
// 8-channel NHWC plane tensor): K = 200, N = 32.  In the implicit-GEMM kernel it is neither MFMA nor HBM bound (7
// k-tiles per block: prologue + epilogue dominate, 65 TF/s; the layer moves 50 MB for 2.4 GFLOP).  Here ONE block owns
// ONE image: the 28x28x8 image (25 KB) and the 32x200 weights are staged in LDS once, every wave keeps all 25 weight
// fragments in registers and walks its share of the 18 32-pixel M-tiles: per tap one ds_read_b128 of the image and 4
// MFMAs (a lane's 4 consecutive channels feed 4 k-steps; A and B use the same permutation).  Same arithmetic as the
// GEMM kernel (fp32 MFMA, k ascending tap by tap), same epilogue semantics (bias, LeakyReLU, fused BatchNorm
// statistics: one slot per image).
constexpr int CF_C = 8, CF_K = 32;
constexpr int CF_PIXLD = 12;   // LDS pixel stride (dwords): 12 mod 64 -> the image's ds_read_b128 are conflict-free

struct CFDesc {
  const float* in; const float* w; float* out;
  AliEpilogue ep;
  int B, H, W, P, Q, R, S;
};

// CL = number of leading input channels that carry data (AliEpilogue.in_ch_live; 8 = all).  CL < 8: the reduction runs
// over the TAPS*CL live (tap, channel) pairs only, two per MFMA -- MNIST's 5 planes (mnist.py:108: image, digit plane,
// three attributes) padded to 8 would spend 3/8 of the MFMAs on zeros.
template <int TAPS, int CL = 8>
__global__ __launch_bounds__(256) void conv_first_kernel(const CFDesc d) {
  extern __shared__ __attribute__((aligned(16))) float cf_smem[];
  float* img = cf_smem;                                   // [H*W][CF_PIXLD]
  float* red = cf_smem + d.H * d.W * CF_PIXLD;            // [2][4][32] BatchNorm partials of the 4 waves
  // (the 32 x TAPS x 8 weights go straight from memory -- 25.6 KB, L2-resident -- into every lane's registers: staging
  // them in LDS first cost 26 of the block's 64 KB, i.e. two resident blocks per CU instead of four, and a barrier)
  const float* wl = d.w;                                  // packed [32][TAPS][8]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int b = blockIdx.x;
  const int HW = d.H * d.W, PQ = d.P * d.Q;
  const float* src = d.in + (long long)b * HW * CF_C;
  for (int i = t; i < HW * 2; i += 256) {                 // 2 float4 per pixel
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
    *reinterpret_cast<f32x4*>(img + (i >> 1) * CF_PIXLD + (i & 1) * 4) = v;
  }
  const int lrow = lane & 31, lh = lane >> 5;
  constexpr int CF_WLD = TAPS * CF_C;                      // row pitch of the packed weights
  constexpr int KL = TAPS * CL, NS = (KL + 1) / 2;         // live reduction length, MFMA steps (k = 2*step + lh)
  f32x4 wf[CL == 8 ? TAPS : 1];                            // CL == 8: B fragments of every tap (row n = lane&31, channels 4*lh..+3)
  float wb[CL == 8 ? 1 : NS];                              // CL < 8: B value and A offset of this lane's k of every step
  int aoff[CL == 8 ? 1 : NS];
  if constexpr (CL == 8) {
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) wf[tp] = *reinterpret_cast<const f32x4*>(wl + lrow * CF_WLD + tp * CF_C + lh * 4);
  } else {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const int k = 2 * st + lh;
      const bool live = k < KL;
      const int kk = live ? k : KL - 1;
      const int tp = kk / CL, c = kk - tp * CL;
      const int r = tp / d.S, sx = tp - r * d.S;
      wb[st] = live ? wl[lrow * CF_WLD + tp * CF_C + c] : 0.f;
      aoff[st] = (r * d.W + sx) * CF_PIXLD + c;
    }
  }
  __syncthreads();                                         // the image is in LDS
  const AliEpilogue& ep = d.ep;
  const float bias = ep.bias ? ep.bias[lrow] : 0.f;
  const float smask = (ep.bn_part && ep.bn_stat_mask) ? ep.bn_stat_mask[(long long)b * ep.bn_mask_ld + lrow] : 1.f;
  float bs0 = 0.f, bs1 = 0.f;
  const int ntile = (PQ + 31) / 32;
  float* outb = d.out + (long long)b * PQ * CF_K;
  for (int mt = wave; mt < ntile; mt += 4) {
    const int m = mt * 32 + lrow;
    const int mc = m < PQ ? m : PQ - 1;                   // clamp: rows past the end compute garbage, never stored
    const int p = mc / d.Q, q = mc - p * d.Q;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (CL == 8) {
      const float* a0 = img + (p * d.W + q) * CF_PIXLD + lh * 4;
#pragma unroll
      for (int tp = 0; tp < TAPS; ++tp) {
        const int r = tp / d.S, sx = tp - r * d.S;
        const f32x4 av = *reinterpret_cast<const f32x4*>(a0 + (r * d.W + sx) * CF_PIXLD);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], wf[tp][e], acc, 0, 0, 0);
      }
    } else {
      const float* a0 = img + (p * d.W + q) * CF_PIXLD;
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        float av = a0[aoff[st]];
        if ((KL & 1) && st == NS - 1 && lh) av = 0.f;     // the odd reduction's last step has one live k
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wb[st], acc, 0, 0, 0);
      }
    }
    // acc(row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < PQ) {
        const float v = apply_act(acc[r] + bias, ep.act, ep.slope);
        outb[(long long)row * CF_K + lrow] = v;
        const float vs = v * smask;
        bs0 += vs;
        bs1 += vs * vs;
      }
    }
  }
  if (ep.bn_part) {                                       // bn_mode 1: slot = image (passes of a batched launch: contiguous)
    bs0 += __shfl_xor(bs0, 32, 64);
    bs1 += __shfl_xor(bs1, 32, 64);
    if (lane < 32) { red[(0 * 4 + wave) * 32 + lane] = bs0; red[(1 * 4 + wave) * 32 + lane] = bs1; }
    __syncthreads();
    if (t < 32) {
      const float a = ((red[t] + red[32 + t]) + red[64 + t]) + red[96 + t];
      const float c = ((red[128 + t] + red[160 + t]) + red[192 + t]) + red[224 + t];
      ep.bn_part[((long long)0 * CF_K + t) * d.B + b] = a;
      ep.bn_part[((long long)1 * CF_K + t) * d.B + b] = c;
    }
  }
}

// geometry / epilogue the per-image kernel covers
static bool conv_first_ok(const AliConvGeom* g, const AliEpilogue* ep, bool f16) {
  if (g->C != CF_C || g->K != CF_K || g->stride != 1 || g->pad != 0 || g->R != g->S || (g->R != 5 && g->R != 3)) return false;
  if (g->H > 32 || g->W > 32 || g->B < 64 || f16) return false;
  if (g->P != g->H - g->R + 1 || g->Q != g->W - g->S + 1) return false;
  if (ep && (ep->mask || ep->dact_y || (ep->bn_part && ep->bn_mode != 1))) return false;
  if (ep && ep->bn_part && ep->bn_groups > 1 && g->B % ep->bn_groups != 0) return false;
  return true;
}

static int conv_first_launch(const AliConvGeom* g, const float* x, const float* w, float* y, const AliEpilogue* ep,
                             hipStream_t stream) {
  CFDesc d;
  memset(&d, 0, sizeof(d));
  d.in = x; d.w = w; d.out = y;
  if (ep) d.ep = *ep;
  d.B = g->B; d.H = g->H; d.W = g->W; d.P = g->P; d.Q = g->Q; d.R = g->R; d.S = g->S;
  const size_t lds = ((size_t)g->H * g->W * CF_PIXLD + 256) * sizeof(float);
  const int live = ep ? ep->in_ch_live : 0;
  if (g->R == 5 && live == 5) hipLaunchKernelGGL((conv_first_kernel<25, 5>), dim3(g->B), dim3(256), lds, stream, d);
  else if (g->R == 5) hipLaunchKernelGGL((conv_first_kernel<25>), dim3(g->B), dim3(256), lds, stream, d);
  else hipLaunchKernelGGL((conv_first_kernel<9>), dim3(g->B), dim3(256), lds, stream, d);
  return check_launch("conv_first_kernel");
}

// ---------------------------------------------------------------------------------------------------------------
// First Conv2d of the spectrogram stacks (audio_mnist.py:186, whalecalls.py / esrf_acoustic.py copies: 1 + n attribute
// planes, padded to 4 or 8 channels, -> 64 channels, 5x5, stride 2, pad 1, on 128^2 ... 512^2 maps).  As an implicit GEMM
// its k-loop is 4-7 tiles long and 65 k blocks spend their time in prologue and epilogue: 1.05 ms per ESRF launch
// (50 TF/s, 1.3 TB/s) against 0.34 ms of fp32 MFMA work and 0.3 ms of output traffic.  Here a wave owns 32 output pixels of
// one row x all 64 channels and walks along the row; the 64 x 25 x C weights sit in LDS once per block (4 rows x the
// whole row length), the input is gathered straight from memory into MFMA operand registers (buffer loads, hardware range
// check for the padding): lane (m, h) loads channels [4h', 4h'+4) of tap t for pixel m -- one 16-byte load = the k-slots of
// four v_mfma_f32_32x32x2_f32 steps (C = 4: the half-waves take taps 2j / 2j+1; C = 8: the two halves of tap j).  Same
// arithmetic as the GEMM kernel (exact fp32 fma chain, k ascending), bias + activation epilogue, optional fp16 twin.
struct S2Desc {
  const float* in; const float* w; const float* bias; float* out; _Float16* out16;
  int B, H, W, P, Q, pad, act; float slope;
  unsigned in_bytes;
};

template <int C, bool F16>
__global__ __launch_bounds__(256) void conv_s2_first_kernel(const S2Desc d) {
  constexpr int T = 25, K = 64;
  // fp32: k-slot pairs -- C = 4 (tap 2j | tap 2j+1), C = 8 (lo | hi half of tap j), one v_mfma_f32_32x32x2_f32 per channel
  // F16 (precision "f16" launches): groups of 16 k-slots = one v_mfma_f32_32x32x16_f16 -- C = 4: taps 4j .. 4j+3, the
  //      half-wave h holds taps 4j+2h, 4j+2h+1; C = 8: taps 2j, 2j+1, half-wave h holds tap 2j+h.  Operands rounded to
  //      fp16 (RNE) in registers / on their way into LDS, fp32 accumulation: the arithmetic of the fp16 GEMM loop.
  constexpr int NP = F16 ? (C == 4 ? 7 : 13) : (C == 4 ? 13 : 25);
  constexpr int NBATCH = F16 ? NP : 13;                 // gathers in flight per lane: F16 2 per group, fp32 1 per pair
  constexpr int TPAD = F16 ? (C == 4 ? 28 : 26) : T;    // taps per weight row in LDS (dead taps hold zeros)
  // LDS row pitch: ds_read_b128 of 16 consecutive n conflict-free (fp32: 100 / 204 floats; fp16: 120 / 216 halves)
  constexpr int LDW = F16 ? TPAD * C + 8 : T * C + (C == 8 ? 4 : 0);
  extern __shared__ __attribute__((aligned(16))) float s2_w[];   // fp32 [64][LDW] floats, F16 [64][LDW] halves
  _Float16* s2_h = reinterpret_cast<_Float16*>(s2_w);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (F16) {
    for (int i = t; i < K * TPAD * C / 4; i += 256) {
      const int n = i / (TPAD * C / 4), rem = i - n * (TPAD * C / 4);
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rem * 4 < T * C) v = *reinterpret_cast<const f32x4*>(d.w + (long long)n * T * C + rem * 4);
      f16x4 hv;
#pragma unroll
      for (int e = 0; e < 4; ++e) hv[e] = (_Float16)v[e];
      *reinterpret_cast<f16x4*>(s2_h + ((n >> 1) + 32 * (n & 1)) * LDW + rem * 4) = hv;
    }
  } else {
    for (int i = t; i < K * T * C / 4; i += 256) {
      const int n = i / (T * C / 4), rem = i - n * (T * C / 4);
      *reinterpret_cast<f32x4*>(s2_w + ((n >> 1) + 32 * (n & 1)) * LDW + rem * 4) =
          *reinterpret_cast<const f32x4*>(d.w + (long long)n * T * C + rem * 4);
    }
  }
  // (LDS row m holds channel 2m, row 32 + m channel 2m + 1: a lane then owns two NEIGHBOURING channels of its pixels and
  // a store instruction writes whole 256-byte pixel rows -- 128-byte runs of the fp16 twin)
  __syncthreads();
  const int m = lane & 31, h = lane >> 5;
  const int prow = blockIdx.x * 4 + wave;               // this wave's output row (all waves stay for the barrier above)
  const int b = blockIdx.y;
  if (prow >= d.P) return;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)d.in, 0, d.in_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  const float bias0 = d.bias ? d.bias[2 * m] : 0.f, bias1 = d.bias ? d.bias[2 * m + 1] : 0.f;
  const int ih0 = prow * 2 - d.pad;
  float* orow = d.out + ((long long)(b * d.P + prow) * d.Q) * K;
  _Float16* orow16 = d.out16 ? d.out16 + ((long long)(b * d.P + prow) * d.Q) * K : nullptr;
  for (int q0 = 0; q0 < d.Q; q0 += 32) {
    const int q = q0 + m;
    const int iw0 = q * 2 - d.pad;
    int wsel = 0;
    asm volatile("" : "+v"(wsel));     // (keeps the weight fragments' LDS reads inside the loop: hoisted, they cost 100-200 VGPRs)
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    auto gather = [&](int tap, int choff) -> f32x4 {
      const int r = tap / 5, sx = tap - r * 5;
      const int ih = ih0 + r, iw = iw0 + sx;
      const bool ok = tap < T && q < d.Q && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
      const unsigned off = ok ? (unsigned)((((b * d.H + ih) * d.W + iw) * C + choff) * 4) : OOB;
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (int)off, 0, 0));
    };
    if constexpr (F16) {
      const _Float16* wl = s2_h + m * LDW + wsel;
      f32x4 a[NP][2];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        if (C == 4) { a[j][0] = gather(4 * j + 2 * h, 0); a[j][1] = gather(4 * j + 2 * h + 1, 0); }
        else { a[j][0] = gather(2 * j + h, 0); a[j][1] = gather(2 * j + h, 4); }
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        f16x8 ha;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ha[e] = (_Float16)a[j][0][e]; ha[4 + e] = (_Float16)a[j][1][e]; }
        const f16x8 w0 = *reinterpret_cast<const f16x8*>(wl + 16 * j + 8 * h);
        const f16x8 w1 = *reinterpret_cast<const f16x8*>(wl + 32 * LDW + 16 * j + 8 * h);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, w0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, w1, acc1, 0, 0, 0);
      }
    } else {
      const float* wl = s2_w + m * LDW + wsel;
      // two batches of pairs: at most 13 gathers in flight per lane
#pragma unroll
      for (int j0 = 0; j0 < NP; j0 += NBATCH) {
        f32x4 a[NBATCH];
#pragma unroll
        for (int jj = 0; jj < NBATCH; ++jj) {
          const int j = j0 + jj;
          if (j >= NP) break;
          a[jj] = gather(C == 4 ? 2 * j + h : j, C == 8 ? 4 * h : 0);
        }
#pragma unroll
        for (int jj = 0; jj < NBATCH; ++jj) {
          const int j = j0 + jj;
          if (j >= NP) break;
          const int tap = C == 4 ? 2 * j + h : j;
          const int woff = (tap < T ? tap : 0) * C + (C == 8 ? 4 * h : 0);
          f32x4 w0 = *reinterpret_cast<const f32x4*>(wl + woff);
          f32x4 w1 = *reinterpret_cast<const f32x4*>(wl + 32 * LDW + woff);
          if (tap >= T) { w0 = f32x4{0.f, 0.f, 0.f, 0.f}; w1 = w0; }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jj][e], w0[e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[jj][e], w1[e], acc1, 0, 0, 0);
          }
        }
      }
    }
    // acc0 / acc1 (row = pixel (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31) = channels 2m / 2m+1 of that pixel
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (qq < d.Q) {
        const float v0 = apply_act(acc0[r] + bias0, d.act, d.slope), v1 = apply_act(acc1[r] + bias1, d.act, d.slope);
        *reinterpret_cast<float2*>(orow + (long long)qq * K + 2 * m) = float2{v0, v1};
        if (orow16) {
          using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
          *reinterpret_cast<f16x2*>(orow16 + (long long)qq * K + 2 * m) = f16x2{(_Float16)v0, (_Float16)v1};
        }
      }
    }
  }
}

static bool conv_s2_first_ok(const AliConvGeom* g, const AliEpilogue* ep) {
  if (tuning().no_s2_first) return false;
  if ((g->C != 4 && g->C != 8) || g->K != 64 || g->R != 5 || g->S != 5 || g->stride != 2 || g->pad > 2) return false;
  if (g->P != (g->H + 2 * g->pad - 5) / 2 + 1 || g->Q != (g->W + 2 * g->pad - 5) / 2 + 1 || g->Q < 32 || g->B > 65535) return false;
  if ((long long)g->B * g->H * g->W * g->C >= (1LL << 29)) return false;       // 32-bit byte offsets
  if (ep && (ep->mask || ep->dact_y || ep->bn_part || ep->in_ld || ep->out_ld || ep->tile_order_n < 0)) return false;
  return true;
}

static int conv_s2_first_launch(const AliConvGeom* g, const float* x, const float* w, float* y, const AliEpilogue* ep,
                                hipStream_t stream) {
  S2Desc d;
  memset(&d, 0, sizeof(d));
  d.in = x; d.w = w; d.out = y;
  if (ep) { d.bias = ep->bias; d.act = ep->act; d.slope = ep->slope; if (ep->mfma_f16) d.out16 = reinterpret_cast<_Float16*>(ep->out16); }
  d.B = g->B; d.H = g->H; d.W = g->W; d.P = g->P; d.Q = g->Q; d.pad = g->pad;
  d.in_bytes = (unsigned)((long long)g->B * g->H * g->W * g->C * 4);
  dim3 grid((g->P + 3) / 4, g->B);
  const bool f16 = ep && ep->mfma_f16;
  if (g->C == 4 && f16) hipLaunchKernelGGL((conv_s2_first_kernel<4, true>), grid, dim3(256), (size_t)64 * 120 * sizeof(_Float16), stream, d);
  else if (g->C == 4) hipLaunchKernelGGL((conv_s2_first_kernel<4, false>), grid, dim3(256), (size_t)64 * 100 * sizeof(float), stream, d);
  else if (f16) hipLaunchKernelGGL((conv_s2_first_kernel<8, true>), grid, dim3(256), (size_t)64 * 216 * sizeof(_Float16), stream, d);
  else hipLaunchKernelGGL((conv_s2_first_kernel<8, false>), grid, dim3(256), (size_t)64 * 204 * sizeof(float), stream, d);
  return check_launch("conv_s2_first_kernel");
}

static void setup_fwd(const AliConvGeom* g, GDesc& d) {
  d.B = g->B; d.Hin = g->H; d.Win = g->W; d.Cin = g->C;
  d.Hout = g->P; d.Wout = g->Q; d.Cout = g->K; d.ldo = g->K;
  d.ldw = g->R * g->S * g->C;
  d.S = g->S;
  d.nphase = 1;
  Phase& P = d.ph[0];
  P.Hq = g->P; P.Wq = g->Q; P.oh0 = 0; P.ow0 = 0; P.ostep = 1; P.mult = g->stride;
  P.M = g->B * g->P * g->Q;
  P.nr = g->R; P.ns = g->S;
  for (int r = 0; r < g->R; ++r) { P.dh[r] = (signed char)(r - g->pad); P.wr[r] = (unsigned char)r; }
  for (int s = 0; s < g->S; ++s) { P.dw[s] = (signed char)(s - g->pad); P.ws[s] = (unsigned char)s; }
}

static bool setup_bwd_data(const AliConvGeom* g, GDesc& d) {
  d.B = g->B; d.Hin = g->P; d.Win = g->Q; d.Cin = g->K;
  d.Hout = g->H; d.Wout = g->W; d.Cout = g->C; d.ldo = g->C;
  d.ldw = g->R * g->S * g->K;
  d.S = g->S;
  const int st = g->stride;
  if (st > 2) { set_error("ali_conv_bwd_data: stride %d unsupported", st); return false; }
  d.nphase = 0;
  for (int ph = 0; ph < st; ++ph)
    for (int pw = 0; pw < st; ++pw) {
      const int Hq = (g->H - ph + st - 1) / st, Wq = (g->W - pw + st - 1) / st;
      if (Hq <= 0 || Wq <= 0) continue;
      Phase& P = d.ph[d.nphase++];
      P.Hq = Hq; P.Wq = Wq; P.oh0 = ph; P.ow0 = pw; P.ostep = st; P.mult = 1;
      P.M = g->B * Hq * Wq;
      P.nr = P.ns = 0;
      for (int r = 0; r < g->R; ++r) {
        const int nh = ph + g->pad - r;
        if (((nh % st) + st) % st) continue;
        P.dh[P.nr] = (signed char)(nh / st);  // exact: nh is a multiple of st
        P.wr[P.nr] = (unsigned char)r;
        ++P.nr;
      }
      for (int s = 0; s < g->S; ++s) {
        const int nw = pw + g->pad - s;
        if (((nw % st) + st) % st) continue;
        P.dw[P.ns] = (signed char)(nw / st);
        P.ws[P.ns] = (unsigned char)s;
        ++P.ns;
      }
    }
  return true;
}

extern "C" int32_t ali_conv_mtiles(const AliConvGeom* g, int32_t which, int32_t mfma_f16, int32_t* tile_rows,
                                   int32_t* pixel_major) {
  if (!geom_ok(g) || (which != 0 && which != 1)) return 0;
  if (which == 0 && conv_first_ok(g, nullptr, mfma_f16 != 0)) {      // per-image kernel: one slot per image
    if (tile_rows) *tile_rows = g->P * g->Q;
    if (pixel_major) *pixel_major = 0;
    return g->B;
  }
  GDesc d;
  memset(&d, 0, sizeof(d));
  d.f16 = mfma_f16 != 0;
  bool vec;
  if (which == 0) { setup_fwd(g, d); vec = (g->C % 4) == 0; }
  else { if (!setup_bwd_data(g, d)) return 0; vec = (g->K % 4) == 0; }
  TileCfg tc;
  int max_taps = 0;
  const int tiles = plan_tiles(d, vec, tc, max_taps);
  if (tiles < 0) return 0;
  if (tile_rows) *tile_rows = tc.bm;
  if (pixel_major) *pixel_major = d.ph[0].pixmajor;
  return tiles;
}

extern "C" int32_t ali_conv_tile_order(const AliConvGeom* g, int32_t which, int32_t mfma_f16, int32_t* order,
                                       int32_t cap) {
  if (!geom_ok(g) || (which != 0 && which != 1) || !order) return 0;
  if (which == 0 && conv_first_ok(g, nullptr, mfma_f16 != 0)) return 0;
  GDesc d;
  memset(&d, 0, sizeof(d));
  d.f16 = mfma_f16 != 0;
  bool vec;
  if (which == 0) { setup_fwd(g, d); vec = (g->C % 4) == 0; }
  else { if (!setup_bwd_data(g, d)) return 0; vec = (g->K % 4) == 0; }
  if (!vec || (d.Cin % BK) != 0) return 0;          // only the uniform-tap loop skips dead taps tile-wide
  TileCfg tc;
  int max_taps = 0;
  const int tiles = plan_tiles(d, vec, tc, max_taps);
  if (tiles <= 0 || tiles > cap) return 0;
  std::vector<int> cost;
  if (!tile_costs(d, tc.bm, cost) || (int)cost.size() != tiles) return 0;
  if (*std::max_element(cost.begin(), cost.end()) == *std::min_element(cost.begin(), cost.end())) return 0;
  std::vector<int> idx(tiles);
  for (int i = 0; i < tiles; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cost[a] > cost[b]; });
  // The blocks that are resident when the launch starts (the first 4 * CUs slots) stay where the dispatcher puts them:
  // slot u on CU u % CUs.  Handing the sorted tiles out in slot order gives CU c the c-th heaviest tile of EVERY
  // quartile, i.e. CU 0 the heaviest four (34 taps against a mean of 26 on D's 4x4 stride-1 data gradient).  Instead:
  // longest-processing-time-first onto the CUs -- each tile, heaviest first, goes to the free rank position whose
  // CUs carry the least so far.  Tiles beyond the resident window follow in sorted order (they are dispatched as
  // blocks retire).  Grids whose left-over tiles are split along K keep the plain order (their last slots are special).
  const int nt = (d.Cout + tc.bn - 1) / tc.bn;
  const long long blocks = (long long)tiles * nt;
  bool tail_split = false;
  if (blocks > kNumCU && blocks < 4 * kNumCU) {
    const int R = (int)(blocks % kNumCU);
    const int max_nkt = (max_taps * d.Cin + BK - 1) / BK;
    tail_split = R > 0 && 2 * R <= kNumCU && max_nkt / 2 >= 4;
  }
  const int W = std::min(tiles, 4 * kNumCU / nt);
  if (tail_split || W < 2) {
    for (int i = 0; i < tiles; ++i) order[i] = idx[i];
    return tiles;
  }
  const int full = (tiles / 8) * 8;
  auto slot_of = [&](int r, int by) { return r < full ? (r / 8) * 8 * nt + by * 8 + (r % 8) : full * nt + (r - full) * nt + by; };
  std::vector<long long> load(kNumCU, 0);
  std::vector<char> used(W, 0);
  for (int i = 0; i < W; ++i) {
    const int tile = idx[i];
    int best = -1;
    long long best_load = 0;
    for (int p = 0; p < W; ++p) {
      if (used[p]) continue;
      long long l = 0;
      for (int by = 0; by < nt; ++by) l = std::max(l, load[slot_of(p, by) % kNumCU]);
      if (best < 0 || l < best_load) { best = p; best_load = l; }
    }
    used[best] = 1;
    order[best] = tile;
    for (int by = 0; by < nt; ++by) load[slot_of(best, by) % kNumCU] += cost[tile];
  }
  for (int i = W; i < tiles; ++i) order[i] = idx[i];
  return tiles;
}

extern "C" int32_t ali_conv_writes_out16(const AliConvGeom* g, int32_t which) {
  (void)which;
  return geom_ok(g) ? 1 : 0;       // every mfma_f16 launch of the GEMM kernel does, whatever arithmetic its loop uses
}

extern "C" int32_t ali_conv_uses_f16(const AliConvGeom* g, int32_t which, const AliEpilogue* ep) {
  if (!geom_ok(g) || !ep || !ep->mfma_f16) return 0;
  if (which == 0 && conv_s2_first_ok(g, ep)) return 1;          // row-walking first-layer kernel, fp16 variant
  return ((which == 0 ? g->C : g->K) % 32) == 0 ? 1 : 0;        // uniform-tap GEMM loop
}

static int conv_fwd_impl(const AliConvGeom* g, const float* x, const float* w, float* y, const AliEpilogue* ep,
                         void* ws, size_t ws_bytes, ali_stream_t stream, AliGemmJob* job) {
  if (!geom_ok(g) || !x || !w || !y) { set_error("ali_conv_fwd: bad argument"); return ALI_ERR_BAD_ARG; }
  if (conv_first_ok(g, nullptr, ep && ep->mfma_f16)) {
    // (the slot count ali_conv_mtiles reports depends on the geometry alone: an epilogue the per-image kernel cannot
    // serve is an error here rather than a silent change of the partial layout)
    if (conv_first_ok(g, ep, false)) {
      if (ep && ep->bn_part && ep->bn_slots > 0 && ep->bn_slots != g->B) {
        set_error("ali_conv_fwd: bn_part was sized for %d slots, the per-image kernel leaves %d", ep->bn_slots, g->B);
        return ALI_ERR_BAD_ARG;
      }
      return conv_first_launch(g, x, w, y, ep, (hipStream_t)stream);
    }
    if (ep && ep->bn_part) { set_error("ali_conv_fwd: epilogue not supported for this first-layer geometry"); return ALI_ERR_BAD_ARG; }
    // (no partials requested: the GEMM kernel serves the epilogue the per-image kernel cannot)
  }
  if (conv_s2_first_ok(g, ep)) return conv_s2_first_launch(g, x, w, y, ep, (hipStream_t)stream);
  GDesc d;
  memset(&d, 0, sizeof(d));
  d.in = x; d.w = w; d.out = y;
  fill_epilogue(d, ep);
  setup_fwd(g, d);
  return finalize_and_launch(d, ws, ws_bytes, (hipStream_t)stream, (g->C % 4) == 0, g->pad == 0, job);
}

static int conv_bwd_data_impl(const AliConvGeom* g, const float* dy, const float* w, float* dx, const AliEpilogue* ep,
                              void* ws, size_t ws_bytes, ali_stream_t stream, AliGemmJob* job) {
  if (!geom_ok(g) || !dy || !w || !dx) { set_error("ali_conv_bwd_data: bad argument"); return ALI_ERR_BAD_ARG; }
  GDesc d;
  memset(&d, 0, sizeof(d));
  d.in = dy; d.w = w; d.out = dx;
  fill_epilogue(d, ep);
  if (!setup_bwd_data(g, d)) return ALI_ERR_BAD_ARG;
  return finalize_and_launch(d, ws, ws_bytes, (hipStream_t)stream, (g->K % 4) == 0, false, job);
}

extern "C" int ali_conv_fwd(const AliConvGeom* g, const float* x, const float* w, float* y, const AliEpilogue* ep,
                            void* ws, size_t ws_bytes, ali_stream_t stream) {
  return conv_fwd_impl(g, x, w, y, ep, ws, ws_bytes, stream, nullptr);
}
extern "C" int ali_conv_bwd_data(const AliConvGeom* g, const float* dy, const float* w, float* dx,
                                 const AliEpilogue* ep, void* ws, size_t ws_bytes, ali_stream_t stream) {
  return conv_bwd_data_impl(g, dy, w, dx, ep, ws, ws_bytes, stream, nullptr);
}
extern "C" int ali_conv_fwd_job(const AliConvGeom* g, const float* x, const float* w, float* y, const AliEpilogue* ep,
                                void* ws, size_t ws_bytes, AliGemmJob* job, ali_stream_t stream) {
  if (!job) { set_error("ali_conv_fwd_job: job is NULL"); return ALI_ERR_BAD_ARG; }
  job->opaque[0] = 0;
  return conv_fwd_impl(g, x, w, y, ep, ws, ws_bytes, stream, job);
}
extern "C" int ali_conv_bwd_data_job(const AliConvGeom* g, const float* dy, const float* w, float* dx,
                                     const AliEpilogue* ep, void* ws, size_t ws_bytes, AliGemmJob* job,
                                     ali_stream_t stream) {
  if (!job) { set_error("ali_conv_bwd_data_job: job is NULL"); return ALI_ERR_BAD_ARG; }
  job->opaque[0] = 0;
  return conv_bwd_data_impl(g, dy, w, dx, ep, ws, ws_bytes, stream, job);
}

extern "C" int ali_gemm_launch_multi(int32_t n, const AliGemmJob* jobs, ali_stream_t stream_) {
  if (n < 0 || (n > 0 && !jobs)) { set_error("ali_gemm_launch_multi: bad argument"); return ALI_ERR_BAD_ARG; }
  hipStream_t stream = (hipStream_t)stream_;
  std::vector<GJobRec> rec[2];
  for (int i = 0; i < n; ++i) {
    GJobRec r;
    memcpy(&r, &jobs[i], sizeof(r));
    if (r.state != 1 || r.variant < 0 || r.variant > 1) { set_error("ali_gemm_launch_multi: job %d was not recorded", i); return ALI_ERR_BAD_ARG; }
    rec[r.variant].push_back(r);
  }
  for (int v = 0; v < 2; ++v) {
    std::vector<GJobRec>& all = rec[v];
    // longest blocks first (the dispatcher hands blocks out in order: the short ones fill the tail)
    std::stable_sort(all.begin(), all.end(), [](const GJobRec& a, const GJobRec& b) { return a.cost > b.cost; });
    for (size_t j0 = 0; j0 < all.size(); j0 += kGJobs) {
      const size_t cnt = std::min(all.size() - j0, (size_t)kGJobs);
      if (cnt == 1) {
        const GJobRec& r = all[j0];
        dim3 grid(r.gx, r.gy, r.gz);
        if (v == 0) hipLaunchKernelGGL((gconv_kernel<64, 64, 2, 2, 2>), grid, dim3(256), 0, stream, r.d);
        else hipLaunchKernelGGL((gconv_kernel<128, 32, 4, 1, 2>), grid, dim3(256), 0, stream, r.d);
      } else {
        GJobs gj;
        memset(&gj, 0, sizeof(gj));
        long long blocks = 0;
        for (size_t i = 0; i < cnt; ++i) {
          const GJobRec& r = all[j0 + i];
          gj.j[i] = r.d;
          gj.gx[i] = r.gx; gj.gy[i] = r.gy;
          gj.blk0[i] = (int)blocks;
          blocks += (long long)r.gx * r.gy * r.gz;
        }
        gj.n = (int)cnt;
        if (blocks > (1LL << 30)) { set_error("ali_gemm_launch_multi: grid too large"); return ALI_ERR_BAD_ARG; }
        if (v == 0) hipLaunchKernelGGL((gconv_multi_kernel<64, 64, 2, 2, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, gj);
        else hipLaunchKernelGGL((gconv_multi_kernel<128, 32, 4, 1, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, gj);
      }
      int rc = check_launch("gconv_multi_kernel");
      if (rc) return rc;
    }
  }
  return ALI_OK;
}
