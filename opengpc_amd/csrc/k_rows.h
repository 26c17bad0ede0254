// k_rows.h -- row-ordered output helpers shared by the matchers: prefix over per-row counts,
// expansion of the join's staged rows into gpc_support / gpc_correspondence arrays
// (Forest::rectifiedMatch's output loop, inference.hpp:384-391), and the candidate index list of
// Forest::preprocessImage (arr2ind + margin, filter.hpp:60-75, inference.hpp:316-330).
#pragma once
#include "gpc_device.h"

namespace gpc {

#define RM_THREADS 256

// Sum of cnt[first .. upto-1], all threads get the result.  blockDim.x == 256.
__device__ int block_prefix_rows(const int32_t* __restrict__ cnt, int first, int upto) {
  __shared__ int s_part[RM_THREADS / 64];
  int v = 0;
  for (int r = first + (int)threadIdx.x; r < upto; r += RM_THREADS) v += cnt[r];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  int s = 0;
  for (int w = 0; w < RM_THREADS / 64; ++w) s += s_part[w];
  __syncthreads();
  return s;
}

// Two page-locked host arrays (device views) -> device memory, 16 bytes per lane: the upload of the single-pair host
// path.  One launch where two hipMemcpyAsync cost two copy-engine submissions (14 us each for 446 KB, 7-8 us apart).
// n16: 16-byte words per array.  grid: (ceil(n16 / 256), 2)
__global__ __launch_bounds__(256) void k_upload2(const uint4* __restrict__ src0, const uint4* __restrict__ src1,
                                                 uint4* __restrict__ dst0, uint4* __restrict__ dst1, unsigned n16) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n16) return;
  if (blockIdx.y == 0) dst0[i] = src0[i];
  else dst1[i] = src1[i];
}

// totals[pair] = supports of the pair (sum of its row counts).  grid: (npairs)
__global__ __launch_bounds__(RM_THREADS) void k_pair_totals(const int32_t* __restrict__ rowcnt, int H,
                                                           int32_t* __restrict__ totals) {
  const int s = block_prefix_rows(rowcnt + (long)blockIdx.x * H, GPC_R, H - GPC_R);
  if (threadIdx.x == 0) totals[blockIdx.x] = s;
}

// Expands the staged rows into the caller's array, rows in ascending order.
// mode 0: gpc_support {x, y, float(xL-xR)}; mode 1: gpc_correspondence {xL, y, xR, y};
// mode 2: PACKED -- the staged words (xL | xR << 16) themselves, made contiguous, plus a copy of the pair's
//         row counts in rows_out[pair * rows_stride + y]: 4 bytes per support instead of 12 for results that leave over PCIe
//         (gpc_hip_match_batch expands them on the host, gpc_hip_expand_packed).
// grid: (ceil((H - 26) / rows_per_wg), npairs).  A workgroup expands rows_per_wg (GR_ROWS; 1 for launches too
// small to fill the device otherwise) consecutive rows: one block-wide
// sum of the earlier rows' counts for the first of them, a running offset for the rest (one row per
// workgroup spent most of its time on that sum: 105 k workgroups of ~2 us each at 256 pairs).
#ifndef GR_ROWS
#define GR_ROWS 4   // measured at 256 pairs: 1 row per workgroup 226 us, 2 -> 189, 4 -> 161, 8 -> 170, 16 -> 178
#endif
__global__ __launch_bounds__(RM_THREADS) void k_gather_rows(
    const uint32_t* __restrict__ staged, const int32_t* __restrict__ rowcnt, int W, int H, int mode,
    void* __restrict__ out, int cap, int32_t* __restrict__ counts, const int32_t* __restrict__ img_stats,
    int32_t* __restrict__ ncand, int rows_per_wg, int32_t* __restrict__ rows_out, long packed_stride, long rows_stride,
    const int32_t* __restrict__ totals) {
  const int y0 = GPC_R + blockIdx.x * rows_per_wg, pair = blockIdx.y;
  // packed mode with `totals`: the pairs' records follow one another without gaps (every pair cut at `cap`), so that
  // a chunk's results leave in ONE copy; packed_stride is ignored
  long cbase = -1;
  if (mode == 2 && totals) {
    cbase = 0;
    for (int q = 0; q < pair; ++q) cbase += min(totals[q], cap);
  }
  const int32_t* rc = rowcnt + (long)pair * H;
  int off = block_prefix_rows(rc, GPC_R, y0);
  const int yend = min(y0 + rows_per_wg, H - GPC_R);
  if (mode == 2 && (int)threadIdx.x < yend - y0) rows_out[pair * rows_stride + y0 + threadIdx.x] = rc[y0 + threadIdx.x];
  for (int y = y0; y < yend; ++y) {
    const int cnt = rc[y];
    const uint32_t* src = staged + ((long)pair * H + y) * W;
    if (mode == 0) {
      uint32_t* o = reinterpret_cast<uint32_t*>(out) + (long)pair * cap * 3;
      for (int i = threadIdx.x; i < cnt; i += RM_THREADS) {
        const int pos = off + i;
        if (pos >= cap) break;
        const uint32_t v = src[i];
        const int xl = v & 0xFFFF, xr = v >> 16;
        o[pos * 3 + 0] = xl;
        o[pos * 3 + 1] = y;
        o[pos * 3 + 2] = __float_as_uint((float)(xl - xr));
      }
    } else if (mode == 2) {
      uint32_t* o = reinterpret_cast<uint32_t*>(out) + (cbase >= 0 ? cbase : pair * packed_stride);  // packed_stride: words between pairs
      for (int i = threadIdx.x; i < cnt; i += RM_THREADS) {
        const int pos = off + i;
        if (pos >= cap) break;
        o[pos] = src[i];
      }
    } else {
      int4* o = reinterpret_cast<int4*>(out) + (long)pair * cap;
      for (int i = threadIdx.x; i < cnt; i += RM_THREADS) {
        const int pos = off + i;
        if (pos >= cap) break;
        const uint32_t v = src[i];
        o[pos] = make_int4(v & 0xFFFF, y, v >> 16, y);
      }
    }
    off += cnt;
  }
  if (yend == H - GPC_R && threadIdx.x == 0) {
    counts[pair] = off;
    if (ncand) {
      ncand[pair * 2 + 0] = img_stats[(pair * 2 + 0) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
      ncand[pair * 2 + 1] = img_stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
    }
  }
}

// ---- candidate index list (`mask`) of Forest::preprocessImage: arr2ind + margin filter
//      (filter.hpp:60-75, inference.hpp:316-330).  Two passes over grad, rows in order.
// grid: (H - 26, nimg)
__global__ __launch_bounds__(RM_THREADS) void k_mask_count(const uint8_t* __restrict__ grad, int W, int H,
                                                           int32_t* __restrict__ rowcnt) {
  const int y = GPC_R + blockIdx.x, img = blockIdx.y;
  const uint8_t* row = grad + ((long)img * H + y) * W;
  int v = 0;
  for (int x = GPC_R + threadIdx.x; x < W - GPC_R; x += RM_THREADS) v += row[x] != 0;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __shared__ int s_part[RM_THREADS / 64];
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < RM_THREADS / 64; ++w) s += s_part[w];
    rowcnt[(long)img * H + y] = s;
  }
}

__global__ __launch_bounds__(RM_THREADS) void k_mask_write(const uint8_t* __restrict__ grad, int W, int H,
                                                           const int32_t* __restrict__ rowcnt,
                                                           int32_t* __restrict__ mask, int cap,
                                                           int32_t* __restrict__ counts) {
  __shared__ int s_wcnt[RM_THREADS / 64];
  const int y = GPC_R + blockIdx.x, img = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t* rc = rowcnt + (long)img * H;
  int total = block_prefix_rows(rc, GPC_R, y);
  const uint8_t* row = grad + ((long)img * H + y) * W;
  int32_t* dst = mask + (long)img * cap;
  for (int x0 = 0; x0 < W; x0 += RM_THREADS) {
    const int x = x0 + threadIdx.x;
    const bool c = x >= GPC_R && x < W - GPC_R && row[x] != 0;
    const unsigned long long m = __ballot(c);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = total;
    for (int w = 0; w < wave; ++w) off += s_wcnt[w];
    const int pos = off + __popcll(m & lanemask_lt());
    if (c && pos < cap) dst[pos] = y * W + x;
    for (int w = 0; w < RM_THREADS / 64; ++w) total += s_wcnt[w];
    __syncthreads();
  }
  if (y == H - GPC_R - 1 && threadIdx.x == 0) counts[img] = total;
}

}  // namespace gpc
