// Shared internals of libali_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ali_hip.h"

namespace ali {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return ALI_ERR_LAUNCH;
  }
  return ALI_OK;
}

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kMaxTaps = 28;   // 5x5 kernels are the largest on the path
constexpr int kNumCU = 256;

// The first kWsReserved bytes of every caller-provided workspace hold the split-K arrival counters of the GEMM
// kernels (zero-filled by the caller once, left zero by every launch); all scratch data lives behind them.
constexpr size_t kWsReserved = ALI_WS_RESERVED;
inline void* ws_payload(void* ws) { return ws ? static_cast<char*>(ws) + kWsReserved : nullptr; }
inline size_t ws_payload_bytes(size_t bytes) { return bytes > kWsReserved ? bytes - kWsReserved : 0; }

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  if (act == ALI_ACT_LEAKY) return v > 0.f ? v : v * slope;
  if (act == ALI_ACT_TANH) return tanhf(v);
  return v;
}
__device__ __forceinline__ float act_grad_from_output(float y, int act, float slope) {
  if (act == ALI_ACT_LEAKY) return y > 0.f ? 1.f : slope;
  if (act == ALI_ACT_TANH) return 1.f - y * y;
  return 1.f;
}

}  // namespace ali
