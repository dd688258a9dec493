// Shared internals of libali_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
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

// Developer tuning knobs (scratch/ sweeps, tests), read from the environment once per process and again on
// ali_reload_tuning(); 0 = use the built-in rule.
struct Tuning {
  int bm, bn, splitk;                       // ALI_BM / ALI_BN / ALI_SPLITK: force the gconv tile / split
  long long wgrad_small;                    // ALI_WGRAD_SMALL
  int wgrad_blocks, wgrad_scap;             // ALI_WGRAD_BLOCKS / ALI_WGRAD_SCAP
  int no_first_wgrad;                       // ALI_NO_FIRST_WGRAD=1: the first conv's weight gradient stays a GEMM (A/B)
  int no_order;                             // ALI_NO_ORDER=1: ignore AliEpilogue.tile_order (A/B measurements)
  int tile_m_scale;                         // ALI_TILE_M_SCALE=n: choose the gconv / wgrad tiles as if the batch were n
                                            // times larger (tests: a small batch runs the tiles of the bench batch)
  int wbm, wbn;                             // ALI_WBM / ALI_WBN: force the weight-gradient tile (one of its variants)
  int no_xcd;                               // ALI_NO_XCD=1: raster tile order on deep grids instead of XCD-contiguous chunks (A/B)
  int no_s2_first;                          // ALI_NO_S2_FIRST=1: the spectrogram stacks' first conv stays an implicit GEMM (A/B)
  int no_dma16;                             // ALI_NO_DMA16=1: fp16 twins are always staged through registers (A/B of the LDS-DMA loop)
  int no_t1_mfma;                           // ALI_NO_T1_MFMA=1: the VALU gather forms of the direct one-channel kernels (A/B)
};
inline Tuning read_tuning() {
  {
    auto num = [](const char* name) -> long long { const char* e = getenv(name); return e ? atoll(e) : 0; };
    Tuning v;
    v.bm = (int)num("ALI_BM"); v.bn = (int)num("ALI_BN"); v.splitk = (int)num("ALI_SPLITK");
    v.wgrad_small = getenv("ALI_WGRAD_SMALL") ? num("ALI_WGRAD_SMALL") : -1;
    v.wgrad_blocks = (int)num("ALI_WGRAD_BLOCKS"); v.wgrad_scap = (int)num("ALI_WGRAD_SCAP");
    v.no_order = (int)num("ALI_NO_ORDER");
    v.no_first_wgrad = (int)num("ALI_NO_FIRST_WGRAD");
    v.tile_m_scale = (int)num("ALI_TILE_M_SCALE");
    v.wbm = (int)num("ALI_WBM"); v.wbn = (int)num("ALI_WBN");
    v.no_t1_mfma = (int)num("ALI_NO_T1_MFMA");
    v.no_xcd = (int)num("ALI_NO_XCD");
    v.no_s2_first = (int)num("ALI_NO_S2_FIRST");
    v.no_dma16 = (int)num("ALI_NO_DMA16");
    return v;
  }
}
inline Tuning& tuning_slot() {
  static Tuning t = read_tuning();
  return t;
}
inline const Tuning& tuning() { return tuning_slot(); }

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
