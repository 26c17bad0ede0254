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
#define GPC_STAT_CODEOR 2  // OR of every code k_hash computed for the image: how many low bits its codes really use

// hash kernel tile: 256 x HT_Y outputs (4 pixels per lane, 4 rows per wave), smooth staged with a 16-byte aligned 16-pixel apron
#define HT_X 256
// (tile height, workgroup size) measured on MI355X, k_hash per 64 images 1024x436:
// (16,256) 123 us, (16,512) 100 us, (32,512) 93 us, (64,1024) 94 us, (32,256) 129 us, (8,256) 151 us
#ifndef HT_Y
#define HT_Y 32
#endif
#ifndef HT_THREADS
#define HT_THREADS 512
#endif
// Tiles per workgroup (next window prefetched into registers) are chosen per launch by the host:
// measured on 64 images 1024x436: 1 -> 88 us, 2 -> 82, 4 -> 89 (14 tiles do not divide), 7 -> 79.
#define HT_APRON 16
#define HT_STRIDE (HT_X + 2 * HT_APRON)  // 288 bytes per LDS row
#define HT_ROWS (HT_Y + 2 * GPC_R)       // 58 rows
#define HT_COPY (HT_ROWS * HT_STRIDE)    // bytes of one (shifted) copy of the window
#define HT_Y_TALL 40                     // the taller tile (k_hash<..., TY>): 66 window rows, 76 KB of LDS, still two workgroups per CU

// Lives in device memory (one copy per arithmetic, gpc_hip_set_forest); the hash kernel reads the
// fields with scalar loads, so a tap address is `lane base + SGPR`.
struct GpcForestDev {
  int32_t off[32];    // LDS DWORD offsets of a test's two taps, packed (off_a & 0xFFFF) | (off_b << 16);
                      // byte offset = (dx & 3) * HT_COPY + dy * HT_STRIDE + (dx - (dx & 3))
  int32_t boff[64];   // the same as BYTE offsets, tap a of test t in [2t], tap b in [2t+1] (k_hash.h, HT_BYTE_OFFS: one plain add
                      // per tap address instead of shift + mask + shift-add)
  int32_t tau[32];    // (int8_t) tau, sign-extended (SSE arithmetic) / the int as given (Naive arithmetic)
  int32_t num_tests;
  int32_t type;
  int32_t tauk[32];   // SSE arithmetic: the minuend of k_hash's complemented saturating subtract in both 16-bit halves -- 0 for a tau of
                      // 0 (plain compare), else (tau - 1) * 256 + 255, or tau * 256 in a forest that holds a tau of -128
  int32_t tau_m128;   // one of these bytes is 0x80 (tau = -128): k_hash takes the complemented subtract that holds for every tau
};

__device__ __forceinline__ unsigned lane_id() { return threadIdx.x & 63u; }

// Pixel index k = y*W + x  ->  (x, y) without an integer division: for k < 2^31,
// k / W == umulhi(k, ceil(2^(31+p) / W)) >> (p - 1) with p = ceil(log2 W)  (Granlund & Montgomery).
struct GpcDivW {
  int W;
  uint32_t magic;
  int sh;
};
#ifdef __HIPCC__
__device__ __forceinline__ int divw(uint32_t k, const GpcDivW& d) { return (int)(__umulhi(k, d.magic) >> d.sh); }
#endif
inline GpcDivW make_divw(int W) {
  int p = 1;
  while ((1 << p) < W) ++p;
  GpcDivW d;
  d.W = W;
  d.magic = (uint32_t)(((1ull << (31 + p)) + (unsigned long long)W - 1ull) / (unsigned long long)W);
  d.sh = p - 1;
  return d;
}

// ndb::Hashmatch's bucket of a 64-bit state (y << 32 | code) % 214673 (hashmatch.hpp:252-263) without 64-bit division:
// 2^32 mod 214673 = 4585, and x mod 214673 for a 32-bit x by a multiply-high whose quotient is at most one short
// (checked on the host for every 32-bit x and for 5e7 random states).
#define HM_BUCKETS 214673u
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t hm_mod(uint32_t x) {
  const uint32_t q = __umulhi(x, 2622360303u) >> 17;  // floor(2^49 / 214673)
  uint32_t r = x - q * HM_BUCKETS;
  if (r >= HM_BUCKETS) r -= HM_BUCKETS;
  return r;
}
__device__ __forceinline__ uint32_t hm_bucket(uint32_t code, uint32_t y) {  // y < 2^30 / 16
  uint32_t t = hm_mod((y < HM_BUCKETS ? y : hm_mod(y)) * 4585u) + hm_mod(code);
  if (t >= HM_BUCKETS) t -= HM_BUCKETS;
  return t;
}
#endif

__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

// Inclusive prefix sum across the 64 lanes of a wave with DPP only (no LDS traffic):
// Kogge-Stone inside each row of 16 (row_shr 1,2,4,8), then row_bcast15 / row_bcast31.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true);  // row_bcast15 -> rows 1,3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true);  // row_bcast31 -> rows 2,3
  return v;
}

// Maximum over the 64 lanes of a wave with the same DPP steps (zeros shift in: the identity of an unsigned maximum);
// every lane gets the result.  Minimum: ~wave_max_u32(~v).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true));  // row_shr:1
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true));  // row_shr:2
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true));  // row_shr:4
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true));  // row_shr:8
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, true));  // row_bcast15 -> rows 1,3
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, true));  // row_bcast31 -> rows 2,3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Exclusive scan of arr[0 .. NT*SPT) in place by an NT-thread workgroup (thread t owns SPT
// consecutive counters); arr[NT*SPT] receives the total (TOTAL).  s_w: NT/64 words of scratch.
// Ends with a barrier.
template <int SPT, int NT = 256, bool TOTAL = true>
__device__ __forceinline__ void block_exscan(uint32_t* __restrict__ arr, uint32_t* __restrict__ s_w, int tid) {
  // (opaque: inside a row loop the addresses derived from tid are loop-invariant; hoisted at 64 VGPRs they are spilled to
  // scratch, and a scratch reload waits for every vector-memory operation the wave has in flight)
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t v[SPT];
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < SPT; ++i) {
    v[i] = arr[tid * SPT + i];
    sum += v[i];
  }
  const uint32_t incl = wave_incl_scan(sum);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = incl - sum;
#pragma unroll
  for (int w = 0; w < NT / 64 - 1; ++w)
    if (w < wave) base += s_w[w];
#pragma unroll
  for (int i = 0; i < SPT; ++i) {
    arr[tid * SPT + i] = base;
    base += v[i];
  }
  if (TOTAL && tid == NT - 1) arr[NT * SPT] = base;
  __syncthreads();
}
