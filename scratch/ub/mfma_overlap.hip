// How many VALU / LDS instructions hide behind one v_mfma_f32_32x32x2_f32 issued by the same wave?
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NV, int NL>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0;
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  float l = 0.f;
  int idx = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[q & 7] = v[q & 7] * 1.0001f + 0.5f;
#pragma unroll
      for (int q = 0; q < NL; ++q) { l += lds[(idx + q * 64 + m * 7) & 4095]; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = l;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int q = 0; q < 8; ++q) s += v[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int NL>
void run(const char* name, int blocks) {
  float* out; hipMalloc(&out, 4 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 2000;
  hipLaunchKernelGGL((k<NV, NL>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NL>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mf = (double)iters * 16;
  printf("%-10s blocks %4d: %.1f ns per MFMA per wave, %.1f TF\n", name, blocks, ms * 1e6 / mf, blocks * 4.0 * mf * 4096 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int blocks : {256, 512}) {
    run<0, 0>("v0 l0", blocks); run<4, 0>("v4 l0", blocks); run<8, 0>("v8 l0", blocks); run<12, 0>("v12 l0", blocks);
    run<16, 0>("v16 l0", blocks); run<24, 0>("v24 l0", blocks); run<0, 1>("v0 l1", blocks); run<0, 2>("v0 l2", blocks); run<8, 1>("v8 l1", blocks);
  }
  return 0;
}
