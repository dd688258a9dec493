// Where does the dispatcher put the workgroups of a 1024-block, 4-blocks-per-CU launch?  (scratch experiment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
  __shared__ float pad[9400];   // ~37.6 KB like gconv<64,64>: 4 blocks per CU
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long t0 = __builtin_readcyclecounter();
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  pad[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    out[4 * lin + 0] = hw;
    out[4 * lin + 1] = xcc;
    out[4 * lin + 2] = (unsigned)(t0 & 0xffffffffu);
    out[4 * lin + 3] = (unsigned)pad[5];
  }
}
int main(int argc, char** argv) {
  int gx = argc > 1 ? atoi(argv[1]) : 512, gy = argc > 2 ? atoi(argv[2]) : 2, spin = argc > 3 ? atoi(argv[3]) : 20000;
  int n = gx * gy;
  unsigned* d; hipMalloc(&d, n * 16);
  hipLaunchKernelGGL(probe, dim3(gx, gy), dim3(256), 0, 0, d, spin);
  hipLaunchKernelGGL(probe, dim3(gx, gy), dim3(256), 0, 0, d, spin);
  hipDeviceSynchronize();
  std::vector<unsigned> h(4 * n); hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
  // HW_ID (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...
  std::map<unsigned, std::vector<int>> cu;
  for (int i = 0; i < n; ++i) {
    unsigned hw = h[4 * i], x = h[4 * i + 1] & 0xf;
    unsigned key = (x << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 7) | ((hw >> 8) & 0xf);
    cu[key].push_back(i);
  }
  printf("grid %dx%d: %zu distinct CUs\n", gx, gy, cu.size());
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ < 24) {
      printf("xcc %u se %u sh %u cu %2u :", kv.first >> 16, (kv.first >> 8) & 7, (kv.first >> 7) & 1, kv.first & 0xf);
      for (int i : kv.second) printf(" %d", i);
      printf("\n");
    }
  }
  std::map<size_t, int> hist;
  for (auto& kv : cu) hist[kv.second.size()]++;
  for (auto& kv : hist) printf("  %d CUs hold %zu blocks\n", kv.second, kv.first);
  return 0;
}
