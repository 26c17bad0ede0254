// k_preprocess.h -- raw image -> smooth (3x3 box + clearBoundary) and grad (binary Sobel).
//
// Replaces ndb::box (filter.hpp:293-392), Buffer::clearBoundary (buffer.hpp:630-654) and
// ndb::sobel (filter.hpp:404-519) as called by Forest::preprocessImage (inference.hpp:306-313).
//
// HBM-bound streaming kernel: 1 byte read, 2 bytes written per pixel.  One thread owns a
// PP_PX (8)-pixel-wide column strip (one 16-byte load per row that includes both neighbour bytes, 8-byte stores) and marches
// down ROWS (14; 6 or 2 for smaller launches) rows with a rolling 3-row window in registers, so every raw
// row is read (ROWS+2)/ROWS times.  An 8-pixel group is the natural unit of the reference's Sobel
// lane-duplication quirk (4 decisions shown twice per 8 pixels).
#pragma once
#include "gpc_device.h"

#ifndef PP_PX
#define PP_PX 8        // pixels per thread along x (8 or 16); measured on MI355X: 8 -> 29.5 us, 16 -> 32.1 us per 64 images
#endif
#define PP_ROWS 14     // rows per thread: a strip of r rows reads r + 2 (8 -> 14: 112 -> 109 us per 256 pairs; 28: the same)
#define PP_ROWS_MID 6  // launches that would leave workgroup slots empty with the tall strip ...
#define PP_ROWS_SMALL 2   // ... and with the middle one
#define PP_TX 64       // threads along x per block
#define PP_TY 4        // row strips per block

namespace gpc {

// (24-bit multiplies, said so: the sums are at most 3 * 255 / 4 * 255, but where the compiler cannot see that it takes
// v_mul_lo_u32, which issues at a quarter of the rate -- 44 of them per thread were a tenth of the kernel's vector time)
__device__ __forceinline__ int third(int s) { return (int)(__umul24((unsigned)s, 21846u) >> 16); }  // mulhi_epi16(s,21846), s >= 0
__device__ __forceinline__ int ninth(int s) { return (int)(__umul24((unsigned)s, 7282u) >> 16); }   // mulhi_epi16(s,7282), s >= 0

struct PreRow {
  int h[PP_PX];      // SSE: third(p[x-1]+p[x]+p[x+1]) for the strip's pixels; NAIVE: the plain 3-sum
  int a[PP_PX + 2];  // raw pixels x0-1 .. x0+PP_PX
  int g[PP_PX / 2];  // SSE: ninth(l + 2c + r) of this row at the strip's Sobel decisions (a row is the "below" of the output row
                     // above it and the "above" of the one below it: computed once per row, where it was once per use)
};
__device__ __forceinline__ void pre_row_sobel_h(PreRow& o) {
#pragma unroll
  for (int g8 = 0; g8 < PP_PX / 8; ++g8)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = g8 * 8 + j;
      o.g[g8 * 4 + j] = ninth(o.a[q] + o.a[q + 2] + 2 * o.a[q + 1]);
    }
}

// Linear addressing as in the reference: the byte left of column 0 is the previous row's
// last byte, the byte right of column W-1 the next row's first (filter.hpp:325-327); bytes outside the image read as 0.
// The image is a BUFFER resource (base, size n, no stride) and a row is ONE 16-byte buffer load at byte r * W + x0 - 4
// (4-byte aligned): the strip, its left neighbour in the top byte of the first dword, its right one in the low byte of the
// last.  A buffer load that does not lie inside [0, n) returns 0 -- all of it, also the part that does (measured: gfx950
// checks the access, not its dwords) -- so a load that is not wholly inside is made again in four parts (neighbour byte,
// two dwords, neighbour byte), each inside or outside as a whole: the rows above and below the image (all zero -- but for
// the byte "left of" column 0 of row H, which is the image's last byte: what the naive filters' window at (H-1) * W
// reaches), the first strip of row 0 (nothing to its left) and the last strip of row H-1 (nothing to its right).  One
// compare and a branch few lanes take, where the flat loads had a branch per case and a clamp.
template <bool NAIVE>
__device__ __forceinline__ void pre_load_row(const __amdgpu_buffer_rsrc_t raw, uint32_t nbytes, int W, int r, int x0, PreRow& o) {
  static_assert(PP_PX == 8, "16 bytes hold an 8-pixel strip and its neighbours");
  int p[PP_PX + 2];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t k = (uint32_t)(r * W + x0);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(raw, k - 4u, 0, 0);
  if (k - 4u > nbytes - 16u) {  // (unsigned: a negative offset is a huge one)
    v.x = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(raw, k - 1u, 0, 0) << 24;
    v.y = __builtin_amdgcn_raw_buffer_load_b32(raw, k, 0, 0);
    v.z = __builtin_amdgcn_raw_buffer_load_b32(raw, k + 4u, 0, 0);
    v.w = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(raw, k + 8u, 0, 0);
  }
  p[0] = (int)(v.x >> 24);
  p[PP_PX + 1] = (int)(v.w & 0xFFu);
  // (taking these two neighbour bytes from the adjacent lanes with wave_shr / wave_shl DPP moves instead, memory
  // only at the wave's ends, measured SLOWER on the same box: 230 vs 191 us per 256 pairs)
#pragma unroll
  for (int i = 0; i < PP_PX; ++i) p[1 + i] = ((i < 4 ? v.y : v.z) >> (8 * (i % 4))) & 0xFF;
#pragma unroll
  for (int i = 0; i < PP_PX; ++i) o.h[i] = NAIVE ? (p[i] + p[i + 1] + p[i + 2]) : third(p[i] + p[i + 1] + p[i + 2]);
#pragma unroll
  for (int i = 0; i < PP_PX + 2; ++i) o.a[i] = p[i];
  if (!NAIVE) pre_row_sobel_h(o);
}

// The same in two steps (PP_PX == 8): the row's 16 bytes k-4 .. k+11 as they lie in memory -- left neighbour in the top byte
// of .x, the strip in .y / .z, the right neighbour in the low byte of .w; zeros where the reference reads nothing -- and their
// unpacking.  EXPERIMENT (-DPP_PREFETCH, -DPP_DEPTH=n): a thread requests rows ahead of the one it unpacks instead of one load
// per loop iteration, used at once.  Slower at every depth (all rows: 183 us, 88 VGPRs; 1 / 2 / 3 ahead: 150 / 152 / 162 us
// against 140-147): eight waves per SIMD hide the round trips better than the registers of a deeper pipeline do.
__device__ __forceinline__ uint4 pre_fetch_row(const uint8_t* __restrict__ raw, int n, int W, int H, int r, int x0) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (r < 0 || r >= H) {
    if (r == H && x0 == 0) v.x = (uint32_t)raw[n - 1] << 24;
    return v;
  }
  const int k = r * W + x0;
  if (k >= 4 && k + 12 <= n) return *reinterpret_cast<const uint4*>(raw + k - 4);
  const uint2 w = *reinterpret_cast<const uint2*>(raw + k);  // the image's first and last strip
  v.y = w.x;
  v.z = w.y;
  v.x = (k - 1 >= 0) ? ((uint32_t)raw[k - 1] << 24) : 0u;
  v.w = (k + 8 < n) ? (uint32_t)raw[k + 8] : 0u;
  return v;
}
template <bool NAIVE>
__device__ __forceinline__ void pre_unpack_row(const uint4 v, PreRow& o) {
  static_assert(PP_PX == 8, "16 bytes hold an 8-pixel strip and its neighbours");
  int p[PP_PX + 2];
  p[0] = (int)(v.x >> 24);
  p[PP_PX + 1] = (int)(v.w & 0xFFu);
#pragma unroll
  for (int i = 0; i < PP_PX; ++i) p[1 + i] = ((i < 4 ? v.y : v.z) >> (8 * (i % 4))) & 0xFF;
#pragma unroll
  for (int i = 0; i < PP_PX; ++i) o.h[i] = NAIVE ? (p[i] + p[i + 1] + p[i + 2]) : third(p[i] + p[i + 1] + p[i + 2]);
#pragma unroll
  for (int i = 0; i < PP_PX + 2; ++i) o.a[i] = p[i];
  if (!NAIVE) pre_row_sobel_h(o);
}

// raw0/raw1: [npairs][H][W] for side 0 / 1 (raw1 unused when sides == 1)
// smooth/grad: [npairs*sides][H][W]
// NAIVE = the reference built with SSE=OFF: boxNaive (sum/9) and sobelNaive (C integer division,
// no lane duplication), both over output positions W+1 .. (H-1)*W (filter.hpp:157-223).
// BITS (SSE arithmetic only): the gradient image is binary (0 / 255), and inside the batched pipelines its only reader is the
// hash kernel's candidate test -- so it leaves as ONE BIT per pixel (grad[img][(y * W + x) / 8], bit x % 8; W % 16 == 0):
// 2.125 instead of 3 bytes of traffic per pixel in a kernel that runs at the device's copy rate.  The entry points that hand
// the gradient image to the host (gpc_hip_preprocess) keep the byte image.
#ifdef PP_WAVES_PER_EU
#define PP_OCC __attribute__((amdgpu_waves_per_eu(PP_WAVES_PER_EU, PP_WAVES_PER_EU)))
#else
#define PP_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))
#endif
template <bool NAIVE, int ROWS, bool BITS = false>
__global__ __launch_bounds__(PP_TX * PP_TY) PP_OCC void k_preprocess(
    const uint8_t* __restrict__ raw0, const uint8_t* __restrict__ raw1, uint8_t* __restrict__ smooth,
    uint8_t* __restrict__ grad, int W, int H, int sides, int thr_sq, int32_t* __restrict__ img_stats) {
  // XCD-aware block order (see k_hash.h): launch-order neighbours sit on different XCDs; remapped, each
  // XCD streams through its own contiguous eighth of the blocks, so the row above / below a strip that
  // the neighbouring block also reads is found in the same L2.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#ifndef PP_NO_XCD_REMAP
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
  const int pair = img / sides, side = img - pair * sides;
  const long n = (long)W * H;
  const uint8_t* raw = (side ? raw1 : raw0) + (long)pair * n;
  uint8_t* sm = smooth + (long)img * n;
  uint8_t* gr = grad + (long)img * (BITS ? n / 8 : n);

  if (bx == 0 && by == 0 && threadIdx.x == 0) {
    img_stats[img * GPC_STAT_STRIDE + GPC_STAT_NCAND] = 0;
    img_stats[img * GPC_STAT_STRIDE + GPC_STAT_LASTROW] = -1;
    img_stats[img * GPC_STAT_STRIDE + GPC_STAT_CODEOR] = 0;
  }

  // a wave is one row strip (PP_TX == 64): its index, and with it every row condition below, is wave-uniform -- said so, the
  // conditions are scalar branches; left as threadIdx.x / 64 in a VGPR they were exec-masked regions
  static_assert(PP_TX == 64, "one wave per row strip");
  const int tx = threadIdx.x % PP_TX, ty = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / PP_TX));
  const int x0 = (bx * PP_TX + tx) * PP_PX;
  const int ys = (by * PP_TY + ty) * ROWS;
  if (x0 >= W || ys >= H) return;

  // last row the box filter writes: rows come in pairs from y=1 while y < H-3 (filter.hpp:307);
  // boxNaive writes up to row H-2, which clearBoundary then zeroes
  const int box_last = NAIVE ? H - 3 : ((H & 1) ? H - 3 : H - 4);

  // columns 0, 1 and W-1 of smooth are cleared (buffer.hpp:637-652): only the image's first and last strip have a byte to
  // clear -- a branch those two lanes take (kept as masks the two words cost two registers the kernel does not have)
  const bool strip_first = x0 == 0, strip_last = x0 + PP_PX == W;
  const __amdgpu_buffer_rsrc_t rs_raw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(raw), 0, (int)n, 0x00020000);
  PreRow rows[3];
#if PP_PX == 8 && defined(PP_PREFETCH)  // (experiment: measured slower at every depth, docs/HISTORY.md 7)
#ifndef PP_DEPTH
#define PP_DEPTH 2   // rows requested ahead of the one being unpacked
#endif
  constexpr int PD = PP_DEPTH < ROWS ? PP_DEPTH : ROWS;
  uint4 rv[ROWS + 2];  // the rows of the strip this thread touches; row i + 2 + PD is requested while row i + 2 is unpacked
#pragma unroll
  for (int i = 0; i < 2 + PD; ++i) rv[i] = pre_fetch_row(raw, (int)n, W, H, ys - 1 + i, x0);
  pre_unpack_row<NAIVE>(rv[0], rows[0]);
  pre_unpack_row<NAIVE>(rv[1], rows[1]);
#else
  pre_load_row<NAIVE>(rs_raw, (uint32_t)n, W, ys - 1, x0, rows[0]);
  pre_load_row<NAIVE>(rs_raw, (uint32_t)n, W, ys, x0, rows[1]);
#endif
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int y = ys + i;
    if (y >= H) break;
    const bool row_in = true;
    PreRow& up = rows[i % 3];
    PreRow& mid = rows[(i + 1) % 3];
    PreRow& dn = rows[(i + 2) % 3];
#if PP_PX == 8 && defined(PP_PREFETCH)  // (experiment: measured slower at every depth, docs/HISTORY.md 7)
    if (i + 2 + PD < ROWS + 2) rv[i + 2 + PD] = pre_fetch_row(raw, (int)n, W, H, ys + 1 + i + PD, x0);
    asm volatile("" ::: "memory");  // (keeps the requests where they are: hoisted to the top they cost the kernel its occupancy)
    pre_unpack_row<NAIVE>(rv[i + 2], dn);
#else
    pre_load_row<NAIVE>(rs_raw, (uint32_t)n, W, y + 1, x0, dn);
#endif

    // ---- box + clearBoundary
    uint32_t sw[PP_PX / 4];
#pragma unroll
    for (int q = 0; q < PP_PX / 4; ++q) sw[q] = 0;
    if (y >= 1 && y <= box_last) {
      if (!NAIVE) {
        // third(s) = (s * 21846) >> 16 with s <= 765: the product is below 2^24, the quotient is its byte 2.  Four
        // products become a word of four quotients with three byte permutes (shift, mask and OR per pixel were twelve).
        uint32_t pr[PP_PX];
#pragma unroll
        for (int j = 0; j < PP_PX; ++j) pr[j] = __umul24((uint32_t)(up.h[j] + mid.h[j] + dn.h[j]), 21846u);
#pragma unroll
        for (int q = 0; q < PP_PX / 4; ++q) {
          const uint32_t lo = __builtin_amdgcn_perm(pr[4 * q + 1], pr[4 * q], 0x0C0C0602u);      // [p0.b2, p1.b2, 0, 0]
          const uint32_t hi = __builtin_amdgcn_perm(pr[4 * q + 3], pr[4 * q + 2], 0x06020C0Cu);  // [0, 0, p2.b2, p3.b2]
          sw[q] = lo | hi;
        }
      } else {
#pragma unroll
        for (int j = 0; j < PP_PX; ++j) {
          const int v = (up.h[j] + mid.h[j] + dn.h[j]) / 9;
          sw[j / 4] |= (uint32_t)v << (8 * (j % 4));
        }
      }
      if (strip_first) sw[0] &= 0xFFFF0000u;               // columns 0 and 1
      if (strip_last) sw[PP_PX / 4 - 1] &= 0x00FFFFFFu;    // column W-1
    }
    if (!row_in) {
    } else if (PP_PX == 16)
      *reinterpret_cast<uint4*>(sm + (uint32_t)(y * W + x0)) = make_uint4(sw[0], sw[1], sw[PP_PX / 4 - 2], sw[PP_PX / 4 - 1]);
    else
      *reinterpret_cast<uint2*>(sm + (uint32_t)(y * W + x0)) = make_uint2(sw[0], sw[1]);

    // ---- sobel
    uint32_t gw[PP_PX / 4];
    uint32_t gm = 0u;  // BITS: the strip's eight gradient bits
#pragma unroll
    for (int q = 0; q < PP_PX / 4; ++q) gw[q] = 0;
    if (NAIVE) {
      // sobelNaive: every pixel decides for itself; positions W+1 .. (H-1)*W
#pragma unroll
      for (int j = 0; j < PP_PX; ++j) {
        const long o = (long)y * W + x0 + j;
        if (o >= W + 1 && o <= (long)(H - 1) * W) {
          const int p11 = up.a[j], p12 = up.a[j + 1], p13 = up.a[j + 2];
          const int p21 = mid.a[j], p23 = mid.a[j + 2];
          const int p31 = dn.a[j], p32 = dn.a[j + 1], p33 = dn.a[j + 2];
          const int sx = (p11 + p31 + 2 * p21 - p13 - 2 * p23 - p33) / 9;
          const int sy = (p11 + p13 + 2 * p12 - p31 - 2 * p32 - p33) / 9;
          const uint32_t e = (sx * sx + sy * sy > thr_sq) ? 0xFFu : 0u;
          gw[j / 4] |= e << (8 * (j % 4));
        }
      }
    } else if (y >= 1 && y <= H - 4) {
      // per 8-pixel group: decisions at its first 4 pixels, each shown twice (filter.hpp:504-507)
#pragma unroll
      for (int g8 = 0; g8 < PP_PX / 8; ++g8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = g8 * 8 + j;
          const int l0 = up.a[q], r0 = up.a[q + 2];
          const int l1 = mid.a[q], r1 = mid.a[q + 2];
          const int l2 = dn.a[q], r2 = dn.a[q + 2];
          const int gx = ninth(l0 + l2 + 2 * l1) - ninth(r0 + r2 + 2 * r1);
          const int gy = up.g[g8 * 4 + j] - dn.g[g8 * 4 + j];
          const bool edge = gx * gx + gy * gy > thr_sq;
          if (BITS) gm |= edge ? (3u << (2 * j)) : 0u;  // decision j shows on pixels 2j and 2j + 1
          const uint32_t e = edge ? 0xFFFFu : 0u;
          if (!BITS) gw[g8 * 2 + j / 2] |= e << (16 * (j % 2));
        }
      }
    }
    if (!row_in) {
    } else if (BITS) {  // (PP_PX == 8: one byte of the bit image per thread and row; bit i = pixel x0 + i has gradient)
      static_assert(!BITS || (PP_PX == 8 && !NAIVE), "the bit image is written by 8-pixel strips of the SSE arithmetic");
      gr[(uint32_t)(y * W + x0) >> 3] = (uint8_t)gm;
    } else if (PP_PX == 16)
      *reinterpret_cast<uint4*>(gr + (uint32_t)(y * W + x0)) = make_uint4(gw[0], gw[1], gw[PP_PX / 4 - 2], gw[PP_PX / 4 - 1]);
    else
      *reinterpret_cast<uint2*>(gr + (uint32_t)(y * W + x0)) = make_uint2(gw[0], gw[1]);
  }
}

}  // namespace gpc
