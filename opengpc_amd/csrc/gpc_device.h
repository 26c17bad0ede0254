// gpc_device.h -- shared device-side definitions for the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GPC_R 13                    // patch radius == candidate margin (inference.hpp:322)
#define GPC_NOCAND 0xFFFFFFFFu      // code-image value of a non-candidate pixel (valid codes have bit 31 clear)
#define GPC_WAVE 64

// per-image statistics block (int32 x 4): [0] candidates, [1] last row holding a candidate
#define GPC_STAT_STRIDE 4
#define GPC_STAT_NCAND 0
#define GPC_STAT_LASTROW 1

// hash kernel tile: 64 x 16 outputs, smooth staged with a 16-byte aligned 16-pixel apron
#define HT_X 64
#define HT_Y 16
#define HT_APRON 16
#define HT_STRIDE (HT_X + 2 * HT_APRON)  // 96 bytes per LDS row
#define HT_ROWS (HT_Y + 2 * GPC_R)       // 42 rows

struct GpcForestDev {
  int16_t off_a[32];  // LDS byte offset of tap i:  iy * HT_STRIDE + ix
  int16_t off_b[32];  // LDS byte offset of tap j
  int32_t tau[32];    // (int8_t) tau, sign-extended
  int32_t num_tests;
  int32_t type;
};

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}
