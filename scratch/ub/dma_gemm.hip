// r03 prototype: fp16 GEMM C[M][N] = A[M][K] . B[N][K]^T, fp32 accumulate -- the staging structure an LDS-DMA version of
// the fp16 gconv loop would have: 256 x 128 tile, 8 waves (4M x 2N, 64 x 64 per wave, v_mfma_f32_32x32x16_f16), k-tile 64,
// operands by buffer_load_dwordx4 ... lds into THREE stages of [rows][128 B] (XOR-swizzled through the source address),
// counted vmcnt + raw barrier: one barrier per k-tile, one tile in flight across it.
// build: hipcc -O3 --offload-arch=gfx950 scratch/ub/dma_gemm.hip -o scratch/ub/dma_gemm ; run: ./dma_gemm [M N K]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
#ifndef BM_
#define BM_ 256
#define BN_ 128
#define WAVES_M_ 4
#define NST_ 3
#endif
constexpr int BM = BM_, BN = BN_, BK = 64, NT = 512, NST = NST_;
constexpr int WAVES_M = WAVES_M_, WAVES_N = 8 / WAVES_M, WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
constexpr int AP = BM / 64, BP = BN / 64;
constexpr int STAGE = (BM + BN) * 128;
typedef __attribute__((address_space(3))) void* lds_ptr;

__global__ __launch_bounds__(NT, 1) void dma_gemm(const _Float16* __restrict__ A, const _Float16* __restrict__ B,
                                                  float* __restrict__ C, int M, int N, int K, int Cin) {
  // A is a 1-D 'convolution' view of X[M + taps][Cin]: A[m][tap * Cin + c] = X[m + tap][c] (the L2 reuse of an implicit GEMM)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ntn = (N + BN - 1) / BN;
  const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((long long)(M + K / Cin) * Cin * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, (unsigned)((long long)N * K * 2), 0x00020000);
  constexpr unsigned OOB = 0xFFFFFF00u;
  unsigned voa[AP], vob[BP];
  const int prow = 8 * w + (lane >> 3), pc = lane & 7;
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int r = prow + 64 * i, c = pc ^ ((r >> 1) & 7);
    voa[i] = (m0 + r < M) ? (unsigned)(((long long)(m0 + r) * Cin + c * 8) * 2) : OOB;
  }
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int r = prow + 64 * j, c = pc ^ ((r >> 1) & 7);
    vob[j] = (n0 + r < N) ? (unsigned)(((long long)(n0 + r) * K + c * 8) * 2) : OOB;
  }
  const int cpt = Cin / BK;
  auto issue = [&](int kt, int stage) {
    char* base = smem + stage * STAGE + 8 * w * 128;
    const int tap = kt / cpt, ch = kt - tap * cpt;
    const int soa = (tap * Cin + ch * BK) * 2;
#pragma unroll
    for (int i = 0; i < AP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(base + 64 * i * 128), 16, (int)voa[i], soa, 0, 0);
#pragma unroll
    for (int j = 0; j < BP; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(base + BM * 128 + 64 * j * 128), 16, (int)vob[j], kt * 128, 0, 0);
  };
  const int wm = w / WAVES_N, wn = w % WAVES_N;
  const int l31 = lane & 31, lh = lane >> 5, sw = (l31 >> 1) & 7;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = K / BK;
  issue(0, 0);
  if (NST == 3 && nk > 1) issue(1, 1);
  int stage = 0;
  for (int q = 0; q < nk; ++q) {
    if (NST == 3 && q + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(AP + BP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifndef SPREAD
    if (q + NST - 1 < nk) issue(q + NST - 1, stage == 0 ? NST - 1 : stage - 1);
#endif
    const char* ab = smem + stage * STAGE + (wm * WM + l31) * 128;
    const char* bb = smem + stage * STAGE + BM * 128 + (wn * WN + l31) * 128;
#ifdef SPREAD
    const int nkt = q + NST - 1, nstage = stage == 0 ? NST - 1 : stage - 1;
    const bool more = nkt < nk;
    char* nbase = smem + nstage * STAGE + 8 * w * 128;
    const int ntap = nkt / cpt, nch = nkt - ntap * cpt;
    const int nsoa = (ntap * Cin + nch * BK) * 2;
#endif
#pragma unroll
    for (int st = 0; st < 4; ++st) {
#ifdef SPREAD
      if (more) {
#pragma unroll
        for (int i = st * AP / 4; i < (st + 1) * AP / 4; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(nbase + 64 * i * 128), 16, (int)voa[i], nsoa, 0, 0);
#pragma unroll
        for (int j = st * BP / 4; j < (st + 1) * BP / 4; ++j)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(nbase + BM * 128 + 64 * j * 128), 16, (int)vob[j], nkt * 128, 0, 0);
      }
#endif
      const int pcx = ((2 * st + lh) ^ sw) * 16;
      h8 ha[TM], hb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) ha[i] = *reinterpret_cast<const h8*>(ab + i * 32 * 128 + pcx);
#pragma unroll
      for (int j = 0; j < TN; ++j) hb[j] = *reinterpret_cast<const h8*>(bb + j * 32 * 128 + pcx);
#ifdef PRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[i], hb[j], acc[i][j], 0, 0, 0);
#ifdef PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    stage = stage == NST - 1 ? 0 : stage + 1;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = n0 + wn * WN + j * 32 + l31;
        if (row < M && col < N) C[(long long)row * N + col] = acc[i][j][r];
      }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 262144, N = argc > 2 ? atoi(argv[2]) : 256, Cin = argc > 3 ? atoi(argv[3]) : 128;
  const int K = 25 * Cin;
  std::vector<_Float16> hA((size_t)(M + 25) * Cin), hB((size_t)N * K);
  unsigned s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hA) v = (_Float16)rnd();
  for (auto& v : hB) v = (_Float16)(rnd() * 0.05f);
  _Float16 *dA, *dB; float* dC;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * N * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)dma_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE));
  dim3 grid(((M + BM - 1) / BM) * ((N + BN - 1) / BN));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(dma_gemm, grid, dim3(NT), NST * STAGE, 0, dA, dB, dC, M, N, K, Cin);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(dma_gemm, grid, dim3(NT), NST * STAGE, 0, dA, dB, dC, M, N, K, Cin);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("M %d N %d K %d (Cin %d): %.1f us  %.1f TF/s\n", M, N, K, Cin, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  // check a few rows against the host
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  const int rows[] = {0, 1, 255, 256, 300, M / 2 + 7, M - 1};
  for (int r : rows) {
    if (r >= M) continue;
    for (int c = 0; c < N; ++c) {
      double a = 0, mag = 0;
      for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)(r + k / Cin) * Cin + k % Cin] * (double)hB[(size_t)c * K + k]; a += p; mag += fabs(p); }
      worst = fmax(worst, fabs(a - hC[(size_t)r * N + c]) / (mag + 1e-30));
    }
  }
  printf("max err / sum|products| over %zu rows: %.2e %s\n", sizeof(rows) / sizeof(int), worst, worst < 1e-5 ? "OK" : "MISMATCH");
  return 0;
}
