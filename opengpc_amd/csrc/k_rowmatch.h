// k_rowmatch.h -- epipolar-mode collision matching, one image row per workgroup.
//
// Replaces, for settings.epipolarMode_ == true, the descriptor build + `state |= y<<32`
// (inference.hpp:189-197), Forest::findCorrespondences (std::sort x2 + merge scan,
// inference.hpp:227-254) and the disparity filter of Forest::rectifiedMatch
// (inference.hpp:384-391).
//
// With the row index in the upper 32 state bits, sorting all descriptors by state is a
// sort by code *within each row*, and a source code can only meet target codes of the
// same row.  So the global sort becomes H-26 independent small problems that live
// entirely in LDS: a workgroup loads row y of the left and right code images (coalesced),
// appends the candidates of both as 64-bit keys  code<<32 | side<<16 | x  (left side 0),
// bitonic-sorts them in LDS, and reads matches off neighbouring keys:
//     [.. c' < c][L c][R c][c'' > c ..]   <=>  c unique in L row, unique in R row.
// Matches leave in ascending (y, code) order == the reference's output order.
//
// Tail quirks of the reference's merge scan concern only the group of
// the globally last target state = largest code of the last row that has right-image
// candidates: there the rule is "L once and R exactly twice -> match (first R)", and
// "R once -> no match".
//
// Supports are staged per row as (xL | xR<<16) and put in place by k_gather_rows.
#pragma once
#include "gpc_device.h"

namespace gpc {

#define RM_THREADS 256

// Bitonic sort of P2 (power of two) 64-bit keys in LDS; RM_THREADS threads.
// Three consecutive strides are handled per LDS round trip in registers.
__device__ __forceinline__ void cmpxchg(unsigned long long& a, unsigned long long& b, bool asc) {
  const bool sw = (a > b) == asc;
  const unsigned long long lo = sw ? b : a, hi = sw ? a : b;
  a = lo;
  b = hi;
}

__device__ void bitonic_sort_lds(unsigned long long* __restrict__ keys, int P2, int tid) {
  for (int k = 2; k <= P2; k <<= 1) {
    int j = k >> 1;
    while (j > 0) {
      if (j >= 4 && P2 >= 8) {
        // strides j, j/2, j/4 on 8 keys  base + b*(j/4), b = 0..7
        const int q = j >> 2;
        for (int t = tid; t < (P2 >> 3); t += RM_THREADS) {
          // insert three zero bits above bit log2(q)
          const int low = t & (q - 1);
          const int base = ((t - low) << 3) | low;
          const bool asc = (base & k) == 0;
          unsigned long long e[8];
#pragma unroll
          for (int b = 0; b < 8; ++b) e[b] = keys[base + b * q];
#pragma unroll
          for (int b = 0; b < 4; ++b) cmpxchg(e[b], e[b + 4], asc);
#pragma unroll
          for (int b = 0; b < 8; b += 4) { cmpxchg(e[b], e[b + 2], asc); cmpxchg(e[b + 1], e[b + 3], asc); }
#pragma unroll
          for (int b = 0; b < 8; b += 2) cmpxchg(e[b], e[b + 1], asc);
#pragma unroll
          for (int b = 0; b < 8; ++b) keys[base + b * q] = e[b];
        }
        j >>= 3;
      } else if (j >= 2) {
        const int q = j >> 1;
        for (int t = tid; t < (P2 >> 2); t += RM_THREADS) {
          const int low = t & (q - 1);
          const int base = ((t - low) << 2) | low;
          const bool asc = (base & k) == 0;
          unsigned long long e0 = keys[base], e1 = keys[base + q], e2 = keys[base + 2 * q], e3 = keys[base + 3 * q];
          cmpxchg(e0, e2, asc); cmpxchg(e1, e3, asc);
          cmpxchg(e0, e1, asc); cmpxchg(e2, e3, asc);
          keys[base] = e0; keys[base + q] = e1; keys[base + 2 * q] = e2; keys[base + 3 * q] = e3;
        }
        j >>= 2;
      } else {
        for (int t = tid; t < (P2 >> 1); t += RM_THREADS) {
          const int base = t << 1;
          const bool asc = (base & k) == 0;
          unsigned long long e0 = keys[base], e1 = keys[base + 1];
          cmpxchg(e0, e1, asc);
          keys[base] = e0; keys[base + 1] = e1;
        }
        j >>= 1;
      }
      __syncthreads();
    }
  }
}

// codes:   [npairs*2][H][W]   (image 2p = left, 2p+1 = right)
// staged:  [npairs][H][W]     packed (xL | xR<<16), first rowcnt entries of each row valid
// rowcnt:  [npairs][H]
// grid: (H - 26, npairs); dynamic LDS: 8 * nmax bytes, nmax = pow2 >= 2*(W-26)
__global__ __launch_bounds__(RM_THREADS) void k_row_match(
    const uint32_t* __restrict__ codes, int W, int H, int disp_high, int apply_filter,
    const int32_t* __restrict__ img_stats, uint32_t* __restrict__ staged, int32_t* __restrict__ rowcnt) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ int s_n, s_max_r, s_wcnt[RM_THREADS / 64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y = GPC_R + blockIdx.x;
  const int pair = blockIdx.y;
  if (tid == 0) { s_n = 0; s_max_r = -1; }
  __syncthreads();

  // ---- append the candidates of both rows (order is irrelevant: keys are unique)
  int max_r = -1;
  for (int side = 0; side < 2; ++side) {
    const uint32_t* row = codes + ((long)(pair * 2 + side) * H + y) * W;
    for (int x0 = 0; x0 < W; x0 += RM_THREADS) {
      const int x = x0 + tid;
      const uint32_t c = (x < W) ? row[x] : GPC_NOCAND;
      const bool valid = c != GPC_NOCAND;
      const unsigned long long m = __ballot(valid);
      if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_n, __popcll(m));
        base = __shfl(base, 0);
        if (valid) {
          keys[base + __popcll(m & lanemask_lt())] =
              ((unsigned long long)c << 32) | ((unsigned long long)side << 16) | (unsigned)x;
          if (side) max_r = max(max_r, (int)c);
        }
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) max_r = max(max_r, __shfl_xor(max_r, o));
  if (lane == 0 && max_r >= 0) atomicMax(&s_max_r, max_r);
  __syncthreads();

  const int n = s_n;
  const long rowbase = (long)pair * H + y;
  if (n == 0) {
    if (tid == 0) rowcnt[rowbase] = 0;
    return;
  }
  int P2 = 2;
  while (P2 < n) P2 <<= 1;
  for (int i = n + tid; i < P2; i += RM_THREADS) keys[i] = ~0ull;
  __syncthreads();

  bitonic_sort_lds(keys, P2, tid);

  // ---- read matches off the sorted keys, keep them in order
  const bool tail_row = (y == img_stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_LASTROW]);
  const uint32_t tail_code = (uint32_t)s_max_r;
  uint32_t* dst = staged + rowbase * W;
  int total = 0;
  for (int i0 = 0; i0 < n; i0 += RM_THREADS) {
    const int i = i0 + tid;
    bool match = false;
    uint32_t packed = 0;
    if (i < n) {
      const unsigned long long k0 = keys[i];
      const uint32_t code = (uint32_t)(k0 >> 32);
      const bool is_l = ((k0 >> 16) & 1ull) == 0;
      if (is_l) {
        const bool prev_same = (i > 0) && ((uint32_t)(keys[i - 1] >> 32) == code);
        const unsigned long long k1 = (i + 1 < n) ? keys[i + 1] : ~0ull;
        const unsigned long long k2 = (i + 2 < n) ? keys[i + 2] : ~0ull;
        const unsigned long long k3 = (i + 3 < n) ? keys[i + 3] : ~0ull;
        const bool n1r = ((uint32_t)(k1 >> 32) == code) && ((k1 >> 16) & 1ull);
        const bool n2 = (uint32_t)(k2 >> 32) == code;
        const bool n3 = (uint32_t)(k3 >> 32) == code;
        const bool tail = tail_row && code == tail_code;
        match = !prev_same && n1r && (tail ? (n2 && !n3) : !n2);
        const int xl = (int)(k0 & 0xFFFFull), xr = (int)(k1 & 0xFFFFull);
        if (apply_filter && abs(xl - xr) > disp_high) match = false;
        packed = (uint32_t)xl | ((uint32_t)xr << 16);
      }
    }
    const unsigned long long m = __ballot(match);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = total;
    for (int w = 0; w < wave; ++w) off += s_wcnt[w];
    if (match) dst[off + __popcll(m & lanemask_lt())] = packed;
    for (int w = 0; w < RM_THREADS / 64; ++w) total += s_wcnt[w];
    __syncthreads();
  }
  if (tid == 0) rowcnt[rowbase] = total;
}

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

// Expands the staged rows into the caller's array, rows in ascending order.
// mode 0: gpc_support {x, y, float(xL-xR)}; mode 1: gpc_correspondence {xL, y, xR, y}
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
    int32_t* __restrict__ ncand, int rows_per_wg) {
  const int y0 = GPC_R + blockIdx.x * rows_per_wg, pair = blockIdx.y;
  const int32_t* rc = rowcnt + (long)pair * H;
  int off = block_prefix_rows(rc, GPC_R, y0);
  const int yend = min(y0 + rows_per_wg, H - GPC_R);
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
