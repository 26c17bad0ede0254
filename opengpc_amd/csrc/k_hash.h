// k_hash.h -- fern hash codes: T <= 32 pixel-pair tests in a 27x27 window of `smooth`.
//
// Replaces ndb::gpcFilter / ndb::gpcFilterTau (filter.hpp:547-606, 619-683) plus the
// zero-filled code buffer and descriptor gather of Forest::evalFastMaskOnSubsetSSE
// (inference.hpp:274-290).
//
// Layout of the work (gfx950):
//   * a 512-thread workgroup owns a 256 x 32 output tile; the (256+32) x (32+26) smooth
//     window is staged once into LDS with 16-byte coalesced loads;
//   * a LANE owns 4 horizontally adjacent pixels, a WAVE one 256-pixel row segment in 4 consecutive rows.
//     The window is kept in LDS FOUR times, copy s shifted left by s bytes, so a tap for 4 pixels is ONE ALIGNED
//     dword of copy (dx & 3) at `lane base + scalar offset + row immediate` (unaligned LDS dwords work on gfx950 but
//     measured ~30x slower), and the compiler pairs the rows: two rows of a tap, 72 dwords apart, per ds_read2_b32 --
//     the cheapest LDS read form on this chip for this pattern (profiles/r03_ubench2_issue_rates.txt: a test's taps +
//     its 24 VALU take 8.7 ns per CU this way, 9.2-9.4 ns as aligned ds_read_b64 of row-interleaved storage, 11.4 ns as
//     eight ds_read_b32).  64 lanes read 256 contiguous bytes: conflict-free;
//   * the four unsigned byte compares of a test are done SWAR in 2 VALU ops (v_not, v_lerp_u8: swar_ge below) and shifted
//     into byte planes exactly like the reference's out[0..3] registers (2 more: 4 ops per test and 4 pixels,
//     + 2 address adds per test); the planes are transposed into 4 codes with v_perm_b32 at the end;
//   * the tests (packed LDS offsets, tau) are READ FROM DEVICE MEMORY with scalar loads, eight at a time (as a
//     by-value kernel argument the 64 words stayed live in SGPRs for the whole kernel and the allocator spilled 59 of
//     them); the test loop is fully unrolled, the taps of test t+1 are requested before test t is evaluated;
//   * what bounds the kernel: during the tests LDS reads (2 dwords per test and 4 pixels) and the 4 VALU per test add
//     up rather than overlap (52-56 % of a wave's time); the rest is waiting for the other waves at the two barriers
//     per tile, staging, candidate flags and stores (profiles/r03_a_phase_stamps.txt).
// No MFMA: this is gather/compare.
//
// Output is a dense code image (u32 per pixel): the code for candidates, GPC_NOCAND for
// everything else (DENSE=false), or exactly the reference's gpcstates buffer (DENSE=true,
// used by gpc_hip_hash_codes for parity checks).
#pragma once
#include "gpc_device.h"

namespace gpc {

// Diagnostic build only (-DGPC_STAMPS, tools/stamp_profile.py): s_memtime at the phase boundaries of a tile,
// summed per phase into a debug buffer nothing else reads.  No stamp executes in the product build.
#ifdef GPC_STAMPS
__device__ unsigned long long g_ht_stamps[16];
__device__ unsigned long long g_ht_wg[3 * 8192];  // per workgroup (flat block id < 8192): start, end (s_memrealtime), hardware id
#define HT_STAMP(i)                                                                         \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    ht_acc[i] += t_ - ht_t0;                                                                \
    ht_t0 = t_;                                                                             \
  } while (0)
#define HT_STAMP_FLUSH()                                                                    \
  if (threadIdx.x == 0) { /* every workgroup: earliest start, latest start, latest end of the launch (s_memrealtime) */ \
    unsigned long long rt2_;                                                                  \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt2_)::"memory");         \
    atomicMax(&g_ht_stamps[8], (1ull << 62) - ht_rt0);                                        \
    atomicMax(&g_ht_stamps[9], ht_rt0);                                                       \
    atomicMax(&g_ht_stamps[10], rt2_);                                                        \
    const unsigned fb_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);      \
    if (fb_ < 8192u) {                                                                        \
      unsigned hw_, xcc_;                                                                     \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                       \
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                     \
      g_ht_wg[3 * fb_] = ht_rt0;                                                              \
      g_ht_wg[3 * fb_ + 1] = rt2_;                                                            \
      g_ht_wg[3 * fb_ + 2] = ((unsigned long long)xcc_ << 32) | hw_;                          \
    }                                                                                         \
  }                                                                                           \
  if (threadIdx.x == 0 && ((blockIdx.x + blockIdx.y + blockIdx.z) & 31) == 5) {             \
    unsigned long long rt1_;                                                                  \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1_)::"memory");         \
    ht_acc[6] = rt1_ - ht_rt0; /* slot 6: the workgroup's life in s_memrealtime ticks (100 MHz) */ \
    ht_acc[7] = 1; /* slot 7: the number of workgroups that reported */                       \
    for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_ht_stamps[i_], ht_acc[i_]);                   \
  }
#define HT_STAMP_INIT()                                                                     \
  unsigned long long ht_t0;                                                                 \
  unsigned long long ht_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                  \
  unsigned long long ht_rt0;                                                                \
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ht_rt0)::"memory");        \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ht_t0)::"memory")
#else
#define HT_STAMP(i)
#define HT_STAMP_INIT()
#define HT_STAMP_FLUSH()
#endif

#define HT_FAR 0x40000000   // a byte offset beyond every image (check_dims: at most 2^30 pixels)
#define SW_H 0x80808080u
#define SW_M 0x7F7F7F7Fu

// bit 7 of every byte: (b_byte >= a_byte), unsigned; the other bits are garbage
__device__ __forceinline__ uint32_t swar_ge(uint32_t a, uint32_t b) {
#ifdef HT_SWAR_SUB
  const uint32_t d = (b | SW_H) - (a & SW_M);  // per byte b_lo + 128 - a_lo: no borrow crosses bytes
  const uint32_t x = a ^ b;
  return (x & b) | (~x & d);                   // top bits differ -> b's top bit decides (v_bitop3_b32)
#else
  // v_lerp_u8 is a per-byte (x + y + (z & 1)) >> 1 with a 9-bit sum: (b + (255 - a) + 1) >> 1 = (b - a + 256) >> 1 has
  // bit 7 set exactly when b >= a.  Two instructions (v_not, v_lerp_u8 -- the latter issues at 1.6 times the cost of a
  // plain op, profiles/r03_ubench2_issue_rates.txt) where the subtract form takes four (or, and, sub, bitop3):
  // k_hash 432 -> 412-417 us per 256 pairs on one box.
  return __builtin_amdgcn_lerp(b, ~a, 0x01010101u);
#endif
}

// _mm_subs_epi8(b, tau) on 4 packed bytes: signed saturating subtract (filter.hpp:649-651).
// An int8 value placed in the HIGH byte of a 16-bit lane saturates under a saturating 16-bit
// subtract exactly when the int8 subtraction would (v_pk_sub_i16 with clamp), whatever the low
// byte holds: the lane is s * 256 + g with 0 <= g <= 255, minus tau * 256 it leaves [-32768, 32767]
// exactly when s - tau leaves [-128, 127], and a saturated lane (0x7FFF / 0x8000) has the saturated
// int8 in its high byte.  So the odd bytes are subtracted where they lie (the even byte below each is
// the garbage g), the even bytes after one shift (b1 below b2), and one permute picks the four high
// bytes: shl, pk_sub, pk_sub, perm = 4 VALU per 4 pixels (round 3 isolated both byte sets first: 5, two of
// them half-rate permutes).  tau_hi = (tau & 0xFF) << 8 in both 16-bit halves.
typedef short gpc_short2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t subs_epi8x4(uint32_t b, uint32_t tau_hi) {
  const uint32_t xe = b << 8;  // lanes [b0 : 0], [b2 : b1]
  gpc_short2 t, e, o;
  __builtin_memcpy(&t, &tau_hi, 4);
  __builtin_memcpy(&e, &xe, 4);
  __builtin_memcpy(&o, &b, 4);   // lanes [b1 : b0], [b3 : b2]
  e = __builtin_elementwise_sub_sat(e, t);
  o = __builtin_elementwise_sub_sat(o, t);
  uint32_t re, ro;
  __builtin_memcpy(&re, &e, 4);
  __builtin_memcpy(&ro, &o, 4);
  return __builtin_amdgcn_perm(ro, re, 0x07030501u);  // high bytes back in place: [e.1, o.1, e.3, o.3]
}

// ~_mm_subs_epi8(b, tau): the bytewise complement of the saturated difference.  ~s = -s - 1 maps [-128, 127] onto itself in
// reverse order, so ~clamp(s - tau) = clamp((tau - 1) - s): the same two packed subtracts with the constant as the minuend.
//   FORCE = false (forests without a tau of -128): the minuend is (tau - 1) * 256 + 255 in both halves and the byte below
//     the int8 stays as it lies (g): ((tau - 1) * 256 + 255) - (s * 256 + g) = (tau - 1 - s) * 256 + (255 - g) leaves
//     [-32768, 32767] exactly when tau - 1 - s leaves [-128, 127], and a saturated lane has the saturated int8 in its high
//     byte: shl, pk_sub, pk_sub, perm.  tau = -128 has no such minuend in 16 bits ((tau - 1) = -129).
//   FORCE = true (EVERY tau): the byte below the int8 is forced to 255 first (one OR; for the even bytes it rides in the
//     shift: v_lshl_or_b32) and the minuend is tau * 256: (tau * 256) - (s * 256 + 255) = (tau - 1 - s) * 256 + 1 -- two
//     operations more per test and row, taken only by forests that hold a tau of -128 (GpcForestDev::tau_m128; deciding it
//     per test cost every test of the kernel a three-way branch, ~20 scalar instructions).
// k = the minuend in both 16-bit halves.
template <bool FORCE>
__device__ __forceinline__ uint32_t subs_epi8x4_not(uint32_t b, uint32_t k) {
  const uint32_t xe = FORCE ? ((b << 8) | 0x00FF00FFu) : (b << 8);  // lanes [b0 : 255 / 0], [b2 : 255 / b1]
  const uint32_t xo = FORCE ? (b | 0x00FF00FFu) : b;                // lanes [b1 : 255 / b0], [b3 : 255 / b2]
  gpc_short2 t, e, o;
  __builtin_memcpy(&t, &k, 4);
  __builtin_memcpy(&e, &xe, 4);
  __builtin_memcpy(&o, &xo, 4);
  e = __builtin_elementwise_sub_sat(t, e);
  o = __builtin_elementwise_sub_sat(t, o);
  uint32_t re, ro;
  __builtin_memcpy(&re, &e, 4);
  __builtin_memcpy(&ro, &o, 4);
  return __builtin_amdgcn_perm(ro, re, 0x07030501u);
}

// unsigned saturating add of a uniform constant to 4 packed bytes (same high-byte argument: the lane x * 256 + g plus
// t * 256 passes 65535 exactly when x + t passes 255; v_pk_add_u16 clamp)
typedef unsigned short gpc_ushort2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t uaddsat_x4(uint32_t x, uint32_t t_hi) {
  const uint32_t xe = x << 8;
  gpc_ushort2 t, e, o;
  __builtin_memcpy(&t, &t_hi, 4);
  __builtin_memcpy(&e, &xe, 4);
  __builtin_memcpy(&o, &x, 4);
  e = __builtin_elementwise_add_sat(e, t);
  o = __builtin_elementwise_add_sat(o, t);
  uint32_t re, ro;
  __builtin_memcpy(&re, &e, 4);
  __builtin_memcpy(&ro, &o, 4);
  return __builtin_amdgcn_perm(ro, re, 0x07030501u);
}

// One test for RPW rows: shift the 4 compare bits of every row into its byte plane
// (new bit enters at bit 7, so the first test of a plane ends up on bit 0 after 8 steps).
// NAIVE (the reference's SSE=OFF build): the tau predicate is the plain integer a > b - tau
// (filter.hpp:276):  tau >= 1:  sat(a + tau - 1) >= b ;  tau <= 0:  a > sat(b - tau).
template <bool TAU, bool NAIVE, int RPW>
__device__ __forceinline__ void fern_test(const uint8_t* __restrict__ tile, int lanebase, int packed, int tau,
                                          uint32_t (&plane)[RPW]) {
  // packed: dword offsets (copy select + row + column) of the two taps
  const uint32_t* pa = reinterpret_cast<const uint32_t*>(tile + lanebase) + (int)(int16_t)(packed & 0xFFFF);
  const uint32_t* pb = reinterpret_cast<const uint32_t*>(tile + lanebase) + (packed >> 16);
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    uint32_t a = pa[r * (HT_STRIDE / 4)];
    uint32_t b = pb[r * (HT_STRIDE / 4)];
    uint32_t ge;  // bit 7 of each byte = NOT(code bit)
    if (TAU && NAIVE) {
      if (tau >= 1) {  // wave-uniform
        const uint32_t c = uaddsat_x4(a, (uint32_t)min(tau - 1, 255) * 0x01000100u);
        ge = ~swar_ge(b, c);  // code bit = (c >= b)
      } else {
        ge = swar_ge(a, uaddsat_x4(b, (uint32_t)min(-tau, 255) * 0x01000100u));
      }
    } else {
      if (TAU) b = subs_epi8x4(b, (uint32_t)(tau & 0xFF) * 0x01000100u);
      ge = swar_ge(a, b);
    }
    // bit 7 of every byte from ge, the rest from plane >> 1: one v_bitop3 (full rate; v_bfi / v_and_or
    // issue at half rate on gfx950, profiles/r02_ubench2_issue_rates.txt)
    plane[r] = __builtin_amdgcn_bitop3_b32(ge, plane[r] >> 1, SW_H, 0xE4);
  }
}

// N consecutive tests (slots t0 .. t0+N-1, the first `cnt` of them real) into one plane, software-pipelined: the taps
// of test i + HT_PIPE are requested before test i is evaluated, so HT_PIPE tests' LDS reads are in flight behind the
// 24 VALU operations of one.  (Left to itself the compiler requests one test ahead.)
#ifndef HT_PIPE
#define HT_PIPE 1   // measured per 256 pairs on one box: 0 (compiler's own order) 448 us, 1 -> 439, 2 -> 443, 3 -> 447
#endif
template <bool TAU, bool NAIVE, int RPW, int N, bool M128 = true>
__device__ __forceinline__ void fern_group(const uint8_t* __restrict__ tile, int lanebase,
                                           const GpcForestDev* __restrict__ fp, int t0, int cnt, uint32_t (&plane)[RPW]) {
#if HT_PIPE == 0
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < cnt) fern_test<TAU, NAIVE, RPW>(tile, lanebase, fp->off[t0 + i], TAU ? fp->tau[t0 + i] : 0, plane);
#else
  constexpr int D = HT_PIPE + 1;
  uint32_t a[D][RPW], b[D][RPW];
  const uint32_t* base = reinterpret_cast<const uint32_t*>(tile + lanebase);
#ifndef HT_PACKED_OFFS
  // the group's tap offsets (and, TAU, its tests' minuends) in one scalar load ahead of the tests: fetched where
  // each test needs them, the branches a TAU test takes keep the compiler from merging the loads, and every test then
  // waits out a scalar-cache round trip (s_load_dwordx2 + s_waitcnt lgkmcnt(0)) in front of its LDS reads.  boff[] has 64
  // entries and tauk[] 32 whatever T is, so slots past `cnt` are readable.
  uint32_t goff[2 * N];
#pragma unroll
  for (int i = 0; i < 2 * N; ++i) goff[i] = fp->boff[2 * t0 + i];
  // (SSE arithmetic: the minuend of each test's complemented subtract, 0 for a tau of 0 -- built by the host, gpc_hip_set_forest)
  uint32_t gk[N];
  if (TAU && !NAIVE) {
#pragma unroll
    for (int i = 0; i < N; ++i) gk[i] = (uint32_t)fp->tauk[t0 + i];
  }
#endif
  auto load = [&](int i, int slot) {
#ifndef HT_PACKED_OFFS  // (byte offsets, two words per test: 404-405 vs 408-414 us per 256 pairs against the packed dword offsets)
    const uint32_t* pa = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(base) + goff[2 * i]);
    const uint32_t* pb = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(base) + goff[2 * i + 1]);
#else
    const int packed = fp->off[t0 + i];
    const uint32_t* pa = base + (int)(int16_t)(packed & 0xFFFF);
    const uint32_t* pb = base + (packed >> 16);
#endif
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      a[slot][r] = pa[r * (HT_STRIDE / 4)];
      b[slot][r] = pb[r * (HT_STRIDE / 4)];
    }
  };
#pragma unroll
  for (int i = 0; i < HT_PIPE && i < N; ++i)
    if (i < cnt) load(i, i % D);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (i + HT_PIPE < N && i + HT_PIPE < cnt) load(i + HT_PIPE, (i + HT_PIPE) % D);
    __builtin_amdgcn_sched_barrier(0);
    if (i < cnt) {
      // SSE arithmetic: the test's minuend; Naive: the int
#ifndef HT_PACKED_OFFS
      const uint32_t k = (TAU && !NAIVE) ? gk[i] : 0u;
#else
      const uint32_t k = (TAU && !NAIVE) ? (uint32_t)fp->tauk[t0 + i] : 0u;
#endif
      const int tau = (TAU && NAIVE) ? fp->tau[t0 + i] : 0;
      if (TAU && !NAIVE && k != 0u) {
        // k is wave-uniform (a scalar load) and the loop is unrolled, so this is a scalar branch per test (a test whose tau
        // is 0 -- _mm_subs_epi8(b, 0) = b, 7 of the 30 tests of defaultTauForest.txt -- takes the plain compare below).
        // The compare needs one operand complemented: here the saturating subtract delivers ~b' itself (no v_not),
        // v_lerp_u8(a, ~b', 0) has bit 7 = (a + 255 - b' >= 256) = (a > b') = the code bit, and the plane takes its
        // complement (the planes hold NOT(code bit), complemented once per row at the end).
        {
#pragma unroll
          for (int r = 0; r < RPW; ++r) {
            const uint32_t gt = __builtin_amdgcn_lerp(a[i % D][r], subs_epi8x4_not<M128>(b[i % D][r], k), 0u);
            // (the first test of a plane: see below)
            plane[r] = i == 0 ? ~gt : __builtin_amdgcn_bitop3_b32(gt, plane[r] >> 1, SW_H, 0x4E);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
          uint32_t av = a[i % D][r], bv = b[i % D][r];
          uint32_t ge;
          if (TAU && NAIVE) {
            if (tau >= 1) {
              const uint32_t c = uaddsat_x4(av, (uint32_t)min(tau - 1, 255) * 0x01000100u);
              ge = ~swar_ge(bv, c);
            } else {
              ge = swar_ge(av, uaddsat_x4(bv, (uint32_t)min(-tau, 255) * 0x01000100u));
            }
          } else {
            ge = swar_ge(av, bv);
          }
          // The first test of a plane takes the compare word as it is (bits 0 .. 6 of its bytes are garbage): whatever lies
          // below bit 7 after the first step has left the byte after seven more, and the planes with fewer tests are read
          // through masks that keep only the tests' bits (q0: bit 7 of P8's bytes; q3: the n3 bits m3 selects after the shift
          // by 8 - n3) -- a shift and a v_bitop3 less per plane and row.
          plane[r] = i == 0 ? ge : __builtin_amdgcn_bitop3_b32(ge, plane[r] >> 1, SW_H, 0xE4);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#endif
}

#ifndef HT_NO_SETPRIO
#define HT_PRIO_STEP(l) ht_set_prio(min((l), prio_cap))
#else
#define HT_PRIO_STEP(l) do { } while (0)
#endif
// s_setprio takes an immediate: one scalar branch per level (lvl is wave-uniform)
__device__ __forceinline__ void ht_set_prio(int lvl) {
  if (lvl >= 3) __builtin_amdgcn_s_setprio(3);
  else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
  else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// bit 7 of every byte of x that is not zero (SWAR)
__device__ __forceinline__ uint32_t swar_nonzero(uint32_t x) { return (((x & SW_M) + SW_M) | x) & SW_H; }

// smooth, grad, candmap: [nimg][H][W]; codes: [nimg][H][W] u32
// candmap == nullptr: candidate <=> grad != 0 inside the margin (preprocessImage's mask).
// NAIVE: gpcFilterNaive / gpcFilterTauNaive (filter.hpp:237-281) -- code bits MSB-first
// (test t on bit T-1-t), every candidate row hashed, no 16-pixel group skip.  The host passes the
// tests in reverse order (slot u = test T-1-u) so that slot u lands on bit u.
//
// Tiles start at row 13: rows above it (and from H-13 on) hold no candidate, the matchers never read
// them, and 410 candidate rows of a 436-row image are 13 tiles of 32 rows where the whole image is 14.
// (DENSE, the parity entry point: the host zero-fills the code image first.)
// The tests are read from memory (fp, 264 bytes every wave shares) with scalar loads, eight at a time: as
// a by-value kernel argument the 64 words stayed live in SGPRs for the whole kernel and the allocator spilled
// 59 of them to VGPR lanes (~100 v_readlane / v_writelane per tile).
#ifdef HT_WAVES_PER_EU   // tuning builds: a register budget for more workgroups per CU than the launch bounds alone give
#define HT_OCC __attribute__((amdgpu_waves_per_eu(HT_WAVES_PER_EU, HT_WAVES_PER_EU)))
#else
#define HT_OCC
#endif
// GBITS: `grad` is k_preprocess<..., BITS>'s bit image (one bit per pixel); a lane fetches the 16 bits of its 16-pixel group:
// "any gradient in the group" is that word != 0 (where the byte image needs two DPP permutes), its own nibble the candidates.
// TY: rows of a tile (HT_Y = 32; 40 for launches that fit ONE round of resident workgroups with the taller tile and two
// with the lower: a single 1920x1080 pair is 528 tiles of 32 rows for 512 slots, 432 of 40 rows).  The taps' LDS offsets
// depend on it (a shifted copy of the window is (TY + 26) rows): the host keeps one GpcForestDev per height.
template <bool TAU, bool DENSE, bool NAIVE, bool GBITS = false, int TY = HT_Y>
__global__ __launch_bounds__(HT_THREADS) HT_OCC void k_hash(const uint8_t* __restrict__ smooth,
                                              const uint8_t* __restrict__ grad,
                                              const uint8_t* __restrict__ candmap,
                                              uint32_t* __restrict__ codes, int W, int H,
                                              const GpcForestDev* __restrict__ fp, int32_t* __restrict__ img_stats,
                                              int tpw, int last_round_from) {
  static_assert(TY % (HT_THREADS / 64) == 0, "a wave owns TY / 8 rows of the tile");
  constexpr int RPW = TY / (HT_THREADS / 64);
  constexpr int T_ROWS = TY + 2 * GPC_R, T_COPY = T_ROWS * HT_STRIDE;  // window rows; bytes of one (shifted) copy of the window
  __shared__ __attribute__((aligned(16))) uint8_t tile[4 * T_COPY];
  __shared__ int s_cnt, s_last, s_or;

  // XCD-aware tile order: workgroups go round-robin to the 8 XCDs (each with its own L2) in launch
  // order, so launch-order neighbours never share an L2.  Remapped, XCD k works through its own
  // contiguous eighth of the (x, y, image) tile list: the tiles that share a window apron (left /
  // right, above / below) meet in one L2.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#ifndef HT_NO_XCD_REMAP
  {
    const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
    if ((nwg & 7u) == 0u) {
      const unsigned flat = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const unsigned logical = (flat & 7u) * (nwg >> 3) + (flat >> 3);
      bx = logical % gridDim.x;
      by = (logical / gridDim.x) % gridDim.y;
      bz = logical / (gridDim.x * gridDim.y);
    }
  }
#endif
  const int img = bz;
  const long n = (long)W * H;
  const uint8_t* sm = smooth + (long)img * n;
  static_assert(!GBITS || (!DENSE && !NAIVE), "the bit image exists in the batched SSE pipelines only");
  const uint8_t* gr = grad + (long)img * (GBITS ? n / 8 : n);
  // (the bit image's launches never bring a candidate map -- run_hash: gbits requires d_cand == nullptr)
  const uint8_t* cm = (!GBITS && candmap) ? candmap + (long)img * n : nullptr;
  uint32_t* out = codes + (long)img * n;
  const int tx0 = bx * HT_X;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int x0 = tx0 + 4 * lane;

  if (tid == 0) { s_cnt = 0; s_last = -1; s_or = 0; }

  // bit 7 of byte j: pixel x0 + j lies inside the image and the 13-pixel margin (constant per lane)
  uint32_t xmask = 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (x0 + j >= GPC_R && x0 + j < W - GPC_R) xmask |= 0x80u << (8 * j);

  // A workgroup walks `tpw` vertically adjacent tiles.  The window of the NEXT tile is fetched
  // into registers (16-byte coalesced loads) before the current tile's tests run, so the global
  // latency hides behind ~7 us of VALU/LDS work; it is written to LDS (as 4 byte-shifted copies)
  // once the current tile is done.
  constexpr int QPR = HT_STRIDE / 16;                          // 16-byte chunks per window row
  constexpr int NCHUNK = T_ROWS * QPR;
  constexpr int CPT = (NCHUNK + HT_THREADS - 1) / HT_THREADS;  // chunks per thread
  // chunk -> (window row, chunk in row), byte offset inside a copy: the same for every tile
  int crow[CPT], cnxt[CPT], cdst[CPT];
  uint32_t cflag[CPT];  // bit 0: chunk exists, bit 1: it has a right neighbour inside the window
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = tid + i * HT_THREADS;
    const int r = c / QPR, q = c - r * QPR;
    // offset of the chunk from the window's first row; a chunk this thread does not have lies 2^30 bytes out: beyond any image
    crow[i] = c < NCHUNK ? r * W + q * 16 - HT_APRON + tx0 : HT_FAR;
    cnxt[i] = (c < NCHUNK && q + 1 < QPR) ? crow[i] + 16 : HT_FAR;  // the dword behind the chunk -- the row's last chunk has none inside the window
    cdst[i] = r * HT_STRIDE + q * 16;
    cflag[i] = (c < NCHUNK ? 1u : 0u) | (q + 1 < QPR ? 2u : 0u);
  }
  uint4 pv[CPT];
  uint32_t pn[CPT];
  uint32_t pg[RPW];  // gradient bytes of this thread's 4 pixels in its RPW rows of the fetched tile
  // The image's bytes and its gradient image as BUFFER resources (base, size, no stride): a buffer load outside [0, size)
  // returns 0 by itself -- the reference's "bytes outside the image read as 0" (its unaligned loads reach above row 0 and
  // below row H - 1) is the hardware's range check.  Chunks are 16-byte aligned and so is the size: none straddles the end.
  // (As flat loads behind compares the fetch of a tile was ~150 instructions of EXEC regions and zero moves; it is 25.)
  const uint32_t nbytes = (uint32_t)n;  // an image has at most 2^30 pixels (check_dims): 32-bit offsets
  const __amdgpu_buffer_rsrc_t rs_sm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sm), 0, (int)nbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(gr), 0, (int)(GBITS ? nbytes / 8u : nbytes), 0x00020000);
  // a lane beyond the image's width asks for a gradient word 2^30 bytes out
  const uint32_t gcol = x0 < W ? (GBITS ? (uint32_t)(x0 & ~15) >> 3 : (uint32_t)x0) : (uint32_t)HT_FAR;
  auto fetch = [&](int ty0) {
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int y = ty0 + wave * RPW + r;  // (a row below the image: beyond the gradient image's size)
      if (GBITS)  // the group's 16 bits (2-byte aligned: W is a multiple of 16)
        pg[r] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs_gr, ((uint32_t)(y * W) >> 3) + gcol, 0, 0);
      else
        pg[r] = __builtin_amdgcn_raw_buffer_load_b32(rs_gr, (uint32_t)(y * W) + gcol, 0, 0);
    }
    const int base = (ty0 - GPC_R) * W;  // linear addressing like the reference's unaligned loads
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const uint32_t k = (uint32_t)(base + crow[i]);  // (negative above the image's first byte: wraps beyond its size)
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_sm, k, 0, 0);
      // (nothing here may touch what the loads return: the wait for them belongs in front of the staging, a tile later)
      pn[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_sm, (uint32_t)(base + cnxt[i]), 0, 0);
      pv[i] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  auto stage = [&]() {  // copy s holds the window shifted left by s bytes (v_alignbyte of neighbouring dwords)
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      if (cflag[i] & 1u) {
        uint8_t* dst = tile + cdst[i];
        const uint4 v = pv[i];
        *reinterpret_cast<uint4*>(dst) = v;
#pragma unroll
        for (int sft = 1; sft < 4; ++sft) {
          uint4 w;
          w.x = __builtin_amdgcn_alignbyte(v.y, v.x, sft);
          w.y = __builtin_amdgcn_alignbyte(v.z, v.y, sft);
          w.z = __builtin_amdgcn_alignbyte(v.w, v.z, sft);
          w.w = __builtin_amdgcn_alignbyte(pn[i], v.w, sft);
          *reinterpret_cast<uint4*>(dst + sft * T_COPY) = w;
        }
      }
    }
  };

  constexpr bool INV = !DENSE && !NAIVE;
  constexpr int CBIT = GBITS ? 0 : 7;   // where cand8 keeps a pixel's candidate flag inside its byte
  const int tile0 = by * tpw;
  const int ntiles = (H - 2 * GPC_R + TY - 1) / TY;
  // workgroups are dispatched in the order of their flat index: those from `last_round_from` on are the last the places take
  const bool last_round = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >= last_round_from;
  HT_STAMP_INIT();
  fetch(GPC_R + tile0 * TY);
  int cnt = 0, last = -1;
  uint32_t cor = 0u;  // OR of the codes computed here (candidates or not: a superset costs the join nothing)
  uint32_t cor3 = 0u; // INV: OR of the last plane's complemented bytes (the codes' bits 24 .. 30), folded into cor at the end
  const int T = fp->num_tests;
  const bool m128 = TAU && !NAIVE && fp->tau_m128 != 0;  // the forest holds a tau of -128 (wave-uniform, read once)
  int lanebase = (wave * RPW + GPC_R) * HT_STRIDE + 4 * lane + HT_APRON;
  // keep the constant part (13 rows + apron = 3760 bytes) inside the register: left to the compiler it
  // becomes an immediate that no longer fits the 8-bit dword offsets of ds_read2_b32, and every pair of
  // row reads then needs its own address add (6 adds per test instead of 2)
  asm volatile("" : "+v"(lanebase));
  const int n3 = max(1, min(T, 32) - 25);                     // tests that went into the last plane (T <= 25: plane empty, ~p3 = 0)
  const uint32_t m3 = 0x01010101u * ((1u << n3) - 1u);       // n3 = 7: 0x7F7F7F7F
  // test 8 is OR-ed into bit 0 unless x % 8 == 0 (64-bit-lane carry of bitMask += bitMask)
  const uint32_t m8 = (x0 & 4) ? 0x01010101u : 0x01010100u;
#pragma unroll 1
  for (int tt = 0; tt < tpw && tile0 + tt < ntiles; ++tt) {
  const int ty0 = GPC_R + (tile0 + tt) * TY;
#ifndef HT_NO_SETPRIO
  // Wave priorities (s_setprio, 0 .. 3; a CU's arbiter serves the higher one first, then the older wave).
  //  * Inside a tile the priority steps down with the test groups (3 until test 8, then 2, 1, and 0 from test 25 through the
  //    stores): a wave that is behind its workgroup is served before one that is ahead, and the eight waves reach the tile's
  //    barrier together (they waited there for 24 % of the kernel): 318-321 -> 310-315 us per 256 pairs (two levels, 3 then
  //    0: 315-317; the steps ascending: 321-327).
  //  * The two workgroups of a CU share its issue slots oldest wave first: the older one runs ahead, ends early, and the
  //    younger works its last tiles alone at half the CU's occupancy (a 32-pair launch: a CU's 13 tiles in 50 us where 41
  //    would do, tools/exp/hash_wg_lives.py).  In the launch's LAST round of workgroups the steps are capped by the tiles a
  //    workgroup has left -- 6 and more: 3, 4-5: 2, 2-3: 1, the last: 0 -- so whoever is behind is served first and both
  //    reach their last tile together: k_hash 51.5 -> 49.0 us at 32 pairs, 89.6 -> 83.5 at 64, 324.5 -> 318.7 at 256
  //    (caps 3 / 2 / 1 over the last three tiles, by quarters of the workgroup's tiles, 8 / 5 / 3: 0-3 us behind, and
  //    within 1 % of each other over the BASELINE configurations, tools/exp/ab_configs.sh; a cap in every round, where a
  //    place is refilled when a workgroup ends, cost 324 -> 328 us per 256 pairs).
  int prio_cap = 3;
  if (last_round) {
    const int left = min(tpw - tt, ntiles - tile0 - tt);
    prio_cap = left >= 6 ? 3 : left >= 4 ? 2 : left >= 2 ? 1 : 0;
  }
  ht_set_prio(prio_cap);
#endif
  if (tt) __syncthreads();  // every wave has finished reading the previous window
  HT_STAMP(0);   // wait for the other waves' tests
  stage();
  uint32_t gq[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) gq[r] = pg[r];
  __syncthreads();
  HT_STAMP(1);   // window arrives (vmcnt), shifted copies written, barrier
  if (tt + 1 < tpw && tile0 + tt + 1 < ntiles) fetch(ty0 + TY);
  HT_STAMP(2);   // next window's loads issued

  const int yw = ty0 + wave * RPW;  // first row of this wave (>= 13)

  // ---- per row: candidate flags (bit 7 of byte j = pixel x0+j), group-of-16 activity
  uint32_t cand8[RPW];
  bool rowdo[RPW];
  bool any = false;
  if (GBITS) {
    // straight-line for the bit image: the fetched word is 0 for a row below the image and for a lane beyond its width
    // (buffer loads), so what is left of the row conditions is "above the last 13 rows" for the candidates and "above
    // the last 15" for the rows that are hashed -- two compares per row, no EXEC region
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int y = yw + r;
      const uint32_t g4 = gq[r];
      // own nibble -> BIT 0 of byte j (CBIT): nib * 0x204081 puts bit k of the nibble at 7j + k for j = 0 .. 3, i.e. bit j at
      // bit 0 of byte j (the other products fall on bits 1 .. 3 and are masked with the margin mask) -- a 24-bit multiply at
      // full rate where nib * 0x10204080 (bit 7 of byte j) was a v_mul_lo_u32 at a quarter of it
      const uint32_t nib = __builtin_amdgcn_ubfe(g4, (uint32_t)(x0 & 15), 4u);
      const uint32_t cb = __umul24(nib, 0x204081u) & (xmask >> 7) & (y < H - GPC_R ? ~0u : 0u);
      cand8[r] = cb;
      rowdo[r] = (y < H - 15) & (g4 != 0u);  // gpcFilterSegment(13, height-15) :602; groups without a gradient byte are skipped :566
      any = any | ((cb != 0u) & rowdo[r]);
    }
  } else
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int y = yw + r;
    const uint32_t g4 = gq[r];  // 0 outside the image
    uint32_t c4 = g4;
    if (cm) c4 = (x0 < W && y < H) ? *reinterpret_cast<const uint32_t*>(cm + (uint32_t)(y * W + x0)) : 0u;
    uint32_t cb;
    if (GBITS && !cm) {  // own nibble -> bit 7 of byte j (n * 0x10204080: bit j lands on 7, 15, 23, 31; the other products fall elsewhere)
      const uint32_t nib = (g4 >> (x0 & 15)) & 0xFu;
      cb = (y < H - GPC_R) ? ((nib * 0x10204080u) & SW_H & xmask) : 0u;
    } else {
      cb = (y < H - GPC_R) ? (swar_nonzero(c4) & xmask) : 0u;
    }
    // the reference skips 16-pixel groups (4 lanes here) without any gradient byte (filter.hpp:566):
    // OR over the quad of lanes with two DPP quad permutes (GBITS: the fetched word is the group's)
    uint32_t gany = g4;
    if (!NAIVE && !GBITS) {
      gany |= (uint32_t)__builtin_amdgcn_mov_dpp((int)gany, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
      gany |= (uint32_t)__builtin_amdgcn_mov_dpp((int)gany, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    }
    const bool rows_ok = y < (NAIVE ? H - GPC_R : H - 15);  // gpcFilterSegment(13, height-15) :602
    cand8[r] = cb;
    rowdo[r] = (x0 < W) && rows_ok && (NAIVE || gany != 0u);
    any = any || ((DENSE && !NAIVE) ? rowdo[r] : (cb != 0u && rowdo[r]));
  }

  // ---- the tests, in the reference's byte planes: P0 = tests 0..7, (test 8), P1 = 9..16,
  //      P2 = 17..24, P3 = 25..31.  Tests >= T are padded with equal taps (compare false).
  // The planes hold NOT(code bit).  The batched SSE instantiations (INV) keep the codes complemented through the transposes
  // and take the complement inside the store phase's v_bitop3 (a truth table costs nothing): four v_not per row less.
  uint32_t code[RPW][4];   // INV: the complemented codes
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int j = 0; j < 4; ++j) code[r][j] = INV ? ~0u : 0u;
  HT_STAMP(3);   // candidate flags, group activity

#ifdef HT_EXP_NOCOMPUTE
  if (W < 0) {
#else
  if (__ballot(any)) {  // wave-uniform: skip segments with nothing to hash
#endif
    uint32_t p0[RPW], p1[RPW], p2[RPW], p3[RPW], p8[RPW];
#pragma unroll
    for (int r = 0; r < RPW; ++r) p0[r] = p1[r] = p2[r] = p3[r] = p8[r] = ~0u;  // "ge" planes: all-ones = no bit
    if (NAIVE) {
      // slot u -> bit u: four full byte planes, no special test 8 (slots >= T are padded with equal taps)
      if (T > 0) fern_group<TAU, true, RPW, 8>(tile, lanebase, fp, 0, 8, p0);
      HT_PRIO_STEP(2);
      if (T > 8) fern_group<TAU, true, RPW, 8>(tile, lanebase, fp, 8, 8, p1);
      HT_PRIO_STEP(1);
      if (T > 16) fern_group<TAU, true, RPW, 8>(tile, lanebase, fp, 16, 8, p2);
      HT_PRIO_STEP(0);
      if (T > 24) fern_group<TAU, true, RPW, 8>(tile, lanebase, fp, 24, 8, p3);
    } else if (TAU && m128) {
      // (a forest with a tau of -128: the complemented subtract that holds for every tau, two operations more per test and row)
      if (T > 0) fern_group<TAU, false, RPW, 8, true>(tile, lanebase, fp, 0, 8, p0);
      if (T > 8) fern_group<TAU, false, RPW, 1, true>(tile, lanebase, fp, 8, 1, p8);
      HT_PRIO_STEP(2);
      if (T > 9) fern_group<TAU, false, RPW, 8, true>(tile, lanebase, fp, 9, 8, p1);
      HT_PRIO_STEP(1);
      if (T > 17) fern_group<TAU, false, RPW, 8, true>(tile, lanebase, fp, 17, 8, p2);
      HT_PRIO_STEP(0);
      if (T > 25) fern_group<TAU, false, RPW, 7, true>(tile, lanebase, fp, 25, min(T, 32) - 25, p3);
    } else {
      if (T > 0) fern_group<TAU, false, RPW, 8, false>(tile, lanebase, fp, 0, 8, p0);
      if (T > 8) fern_group<TAU, false, RPW, 1, false>(tile, lanebase, fp, 8, 1, p8);
      HT_PRIO_STEP(2);
      if (T > 9) fern_group<TAU, false, RPW, 8, false>(tile, lanebase, fp, 9, 8, p1);
      HT_PRIO_STEP(1);
      if (T > 17) fern_group<TAU, false, RPW, 8, false>(tile, lanebase, fp, 17, 8, p2);
      HT_PRIO_STEP(0);
      // the last plane holds tests 25 .. min(T, 32) - 1: no padded tests here (T = 30: 5, not 7)
      if (T > 25) fern_group<TAU, false, RPW, 7, false>(tile, lanebase, fp, 25, min(T, 32) - 25, p3);
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      // planes hold "b >= a"; the code bit is its complement.  P3 saw 7 tests: one more shift.
      // (INV: the same with every term complemented -- ~(~p0 | ((~p8 >> 7) & m8)) = p0 & ((p8 >> 7) | ~m8) on the bits m8
      //  selects, ~((~p3 >> s) & m3) = (p3 >> s) | ~m3 on the bits m3 selects: neither shift crosses into a selected bit)
      const uint32_t q0 = INV ? (p0[r] & ((p8[r] >> 7) | ~m8)) : (NAIVE ? ~p0[r] : (~p0[r] | ((~p8[r] >> 7) & m8)));
      const uint32_t q1 = INV ? p1[r] : ~p1[r];
      const uint32_t q2 = INV ? p2[r] : ~p2[r];
      // P3 saw n3 tests (first one now n3 - 1 places below bit 7): bring the first down to bit 0
      const uint32_t q3 = INV ? ((p3[r] >> (8 - n3)) | ~m3) : (NAIVE ? ~p3[r] : ((~p3[r] >> (8 - n3)) & m3));
      // INV: the joins want the highest bit any code of the image has set (GPC_STAT_CODEOR: its leading zeros size their rank
      // buckets).  Bits 24 .. 30 of a code are its pixel's byte of the last plane, so the OR of that plane's bytes over the
      // rows that are hashed says which of them occur; below bit 24 the statistic is "every bit the forest can set" (a
      // superset is all the joins need).  One operation per row: cor3 | (~q3 & do).
      if (INV) cor3 = __builtin_amdgcn_bitop3_b32(q3, rowdo[r] ? ~0u : 0u, cor3, 0xAE);
      // transpose 4 planes x 4 pixels -> 4 codes (byte k of code j = plane k, byte j)
      const uint32_t lo01 = __builtin_amdgcn_perm(q1, q0, 0x05010400u);
      const uint32_t hi01 = __builtin_amdgcn_perm(q1, q0, 0x07030602u);
      const uint32_t lo23 = __builtin_amdgcn_perm(q3, q2, 0x05010400u);
      const uint32_t hi23 = __builtin_amdgcn_perm(q3, q2, 0x07030602u);
      code[r][0] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
      code[r][1] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
      code[r][2] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
      code[r][3] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
    }
  }

  HT_STAMP(4);   // tests + plane transposes
  // ---- store (16 bytes per lane and row, 1 KiB per wave and row) + statistics
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int y = yw + r;
    if ((x0 < W) && (y < H)) {
      uint4 o;
      uint32_t* op = reinterpret_cast<uint32_t*>(&o);
      if (DENSE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t c = rowdo[r] ? code[r][j] : 0u;
          cor |= c;
          const bool is_cand = (cand8[r] >> (8 * j + 7)) & 1u;
          op[j] = (NAIVE && !is_cand) ? 0u : c;
        }
      } else {
        // a candidate gets its code (0 where the row's 16-pixel group was skipped), anything else GPC_NOCAND (all ones):
        // (code & do & cand) | ~cand with the candidate bit spread over the word by a signed bit-field extract -- one
        // v_bfe_i32 + one v_bitop3 per pixel (select by select it was two v_cndmask, an and and a compare)
        const uint32_t dom = rowdo[r] ? ~0u : 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t cm4 = (uint32_t)__builtin_amdgcn_sbfe((int)cand8[r], 8 * j + CBIT, 1);  // all ones for a candidate
          if (INV) {
            // (the OR of the codes is kept on the planes: cor3 below -- one operation per row instead of one per pixel)
            op[j] = __builtin_amdgcn_bitop3_b32(code[r][j], dom, cm4, 0x5D);  // cand ? (~ncode & do) : all ones
          } else {
            cor |= code[r][j] & dom;
            op[j] = __builtin_amdgcn_bitop3_b32(code[r][j], dom, cm4, 0xD5);  // cand ? (code & do) : all ones
          }
        }
      }
#if defined(HT_EXP_NOSTORE)   // experiment: how much of the kernel is the code image's write stream?
      if (o.x == 0x12345678u && W < 0) *reinterpret_cast<uint4*>(out + (uint32_t)(y * W + x0)) = o;
#elif defined(HT_EXP_HALFSTORE)
      if ((lane & 1) == 0) *reinterpret_cast<uint4*>(out + (uint32_t)(y * W + x0)) = o;
#else
      *reinterpret_cast<uint4*>(out + (uint32_t)(y * W + x0)) = o;
#endif
    }
    if (cand8[r]) { cnt += __popc(cand8[r]); last = y; }
  }
  HT_STAMP(5);   // code stores issued
  }  // tiles of this workgroup
  HT_STAMP_FLUSH();
  if (INV) {
    // bits 24 .. 30 from the last plane's bytes (only the n3 bits m3 selects are code bits); every lower bit the forest
    // has: tests 0 .. 7 -> bits 0 .. 7, test 8 -> bit 0, tests 9 .. 24 -> bits 8 .. 23
    uint32_t t = cor3 & m3;
    t |= t >> 16;
    t |= t >> 8;
    const int lowbits = T <= 8 ? T : (T - 1 < 24 ? T - 1 : 24);
    cor = ((t & 0x7Fu) << 24) | ((1u << lowbits) - 1u);
  }
  if (!DENSE) {
    for (int o = 32; o > 0; o >>= 1) {
      cnt += __shfl_xor(cnt, o);
      last = max(last, __shfl_xor(last, o));
      cor |= (uint32_t)__shfl_xor((int)cor, o);
    }
    if (lane == 0 && cnt) { atomicAdd(&s_cnt, cnt); atomicMax(&s_last, last); atomicOr(&s_or, (int)cor); }
    __syncthreads();
    if (tid == 0 && s_cnt) {
      atomicAdd(&img_stats[img * GPC_STAT_STRIDE + GPC_STAT_NCAND], s_cnt);
      atomicMax(&img_stats[img * GPC_STAT_STRIDE + GPC_STAT_LASTROW], s_last);
      atomicOr(&img_stats[img * GPC_STAT_STRIDE + GPC_STAT_CODEOR], s_or);
    }
  }
}

// candmap[img][k] = 1 for every k of the caller's mask list (inside the margin)
__global__ void k_scatter_mask(const int32_t* __restrict__ mask, int n_mask, uint8_t* __restrict__ candmap,
                               int W, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_mask) return;
  const int k = mask[i];
  if (k < 0 || k >= W * H) return;
  const int x = k % W, y = k / W;
  if (x >= GPC_R && x < W - GPC_R && y >= GPC_R && y < H - GPC_R) candmap[k] = 1;
}

}  // namespace gpc
