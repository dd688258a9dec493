// micro-check of ds_read_b64_tr_b16 lane/element mapping (cdna_hip_programming.md T10)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(const _Float16* in, _Float16* out) {
  __shared__ _Float16 lds[4 * 64];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  const int grp = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  const _Float16* a = &lds[q * 64 + grp * 16 + 4 * p];
  s4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)a);
  h4 v = __builtin_bit_cast(h4, r);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
  std::vector<_Float16> h(256), o(256);
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 64; ++c) h[r * 64 + c] = (_Float16)(r * 64 + c);
  _Float16 *di, *dout;
  hipMalloc(&di, 512); hipMalloc(&dout, 512);
  hipMemcpy(di, h.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    const int expect = e * 64 + 16 * (l >> 4) + (l & 15);
    if ((int)(float)o[l * 4 + e] != expect) { if (bad < 8) printf("lane %d elem %d: got %d expect %d\n", l, e, (int)(float)o[l*4+e], expect); ++bad; }
  }
  printf("tr16 mapping: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as documented", bad);
  for (int l = 0; l < 4; ++l) printf("lane %d: %d %d %d %d\n", l, (int)(float)o[l*4], (int)(float)o[l*4+1], (int)(float)o[l*4+2], (int)(float)o[l*4+3]);
  return 0;
}
