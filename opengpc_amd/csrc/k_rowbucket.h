// k_rowbucket.h -- epipolar-mode collision matching, third generation: per-row counting
// (radix-bucket) join.  No compare-and-swap loops, no sorting network.
//
// Same contract as k_row_match / k_row_join: replaces, for epipolarMode_ == true, the
// descriptor build + `state |= y<<32`, Forest::findCorrespondences and the disparity filter of
// rectifiedMatch (inference.hpp:189-197, 227-254, 384-391) -- one image row per workgroup.
//
// A row has at most W-26 left and W-26 right candidates with 31-bit codes.  With NB = 256*SPT
// >= W buckets on the TOP bits of the code, a bucket holds ~1.4 records on average, and the
// order of buckets IS the order of codes.  So:
//   1. count: every record does one returning ds_add on its bucket counter (its arrival slot);
//   2. an exclusive scan over the NB counters gives every bucket's start;
//   3. scatter: records (code<<1 | side, x) are written bucket-contiguous into LDS;
//   4. decide: a left record reads its own (tiny) bucket: #left and #right records with its
//      code -> match iff 1 and 1 (tail quirk: right count 2 for the largest right code of the
//      last populated right row); matched records mark themselves and count per bucket;
//   5. a second scan over the matched counters + "matched records with a smaller code in my
//      bucket" is the record's rank in ascending code order == its output position; the
//      thread still holds xL and xR, so it writes the packed support directly.
// All LDS atomics are independent single ds_add's; the bucket walks read the first 4 records
// of a bucket in straight-line code and loop only for larger buckets.
#pragma once
#include "gpc_device.h"

namespace gpc {

#define RB_THREADS 256
#define RB_NONE 0xFFFFFFFFu
#define RB_MATCHED 0x8000u  // flag in the 16-bit x of a left record

// Exclusive scan of arr[0 .. 256*SPT) in place (thread t owns SPT consecutive counters);
// arr[256*SPT] receives the total.  s_w: RB_THREADS/64 words of scratch.  Ends with a barrier.
template <int SPT>
__device__ __forceinline__ void rb_exscan(uint32_t* __restrict__ arr, uint32_t* __restrict__ s_w, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t v[SPT];
  uint32_t sum = 0;
#pragma unroll
  for (int i = 0; i < SPT; ++i) {
    v[i] = arr[tid * SPT + i];
    sum += v[i];
  }
  uint32_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = incl - sum;
#pragma unroll
  for (int w = 0; w < RB_THREADS / 64; ++w)
    if (w < wave) base += s_w[w];
#pragma unroll
  for (int i = 0; i < SPT; ++i) {
    arr[tid * SPT + i] = base;
    base += v[i];
  }
  if (tid == RB_THREADS - 1) arr[RB_THREADS * SPT] = base;
  __syncthreads();
}

// codes:   [npairs*2][H][W]   (image 2p = left, 2p+1 = right)
// staged:  [npairs][H][W]     packed (xL | xR<<16), first rowcnt entries of each row valid
// rowcnt:  [npairs][H]
// grid: (H - 26, npairs); P = 256*SPT >= W, W <= 16384
// dynamic LDS (bytes): 4*(P+2) + 4*(P+2) + 4*2P + 2*2P  = 20*P + 16
template <int SPT>
__global__ __launch_bounds__(RB_THREADS) void k_row_bucket(
    const uint32_t* __restrict__ codes, int W, int H, int disp_high, int apply_filter,
    const int32_t* __restrict__ img_stats, uint32_t* __restrict__ staged, int32_t* __restrict__ rowcnt) {
  constexpr int P = RB_THREADS * SPT;   // buckets; also the number of pixel slots per side
  constexpr int LOG2P = 8 + (SPT == 1 ? 0 : SPT == 2 ? 1 : SPT == 4 ? 2 : SPT == 8 ? 3 : SPT == 16 ? 4 : SPT == 32 ? 5 : 6);
  constexpr int SHIFT = 31 - LOG2P;     // bucket = top LOG2P bits of the 31-bit code
  extern __shared__ __attribute__((aligned(16))) uint32_t rb_lds[];
  __shared__ int s_max_r, s_tail_cnt;
  __shared__ unsigned s_tail_minx;
  __shared__ uint32_t s_w[RB_THREADS / 64];
  uint32_t* b_cnt = rb_lds;                  // [P+2]: counts -> starts; [P] = total; [P+1] = spare bucket
  uint32_t* b_cm = rb_lds + (P + 2);         // [P+2]: matched counts -> starts
  uint32_t* e_key = rb_lds + 2 * (P + 2);    // [2P]: code<<1 | side, bucket-contiguous
  uint16_t* e_x = reinterpret_cast<uint16_t*>(rb_lds + 2 * (P + 2) + 2 * P);  // [2P]

  const int tid = threadIdx.x, lane = tid & 63;
  const int y = GPC_R + blockIdx.x;
  const int pair = blockIdx.y;

  // ---- 0. row loads first, then clear the counters
  const uint32_t* rowl = codes + ((long)(pair * 2) * H + y) * W;
  const uint32_t* rowr = rowl + (long)H * W;
  uint32_t cl[SPT], cr[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    const int x = j * RB_THREADS + tid;
    cl[j] = (x < W) ? rowl[x] : RB_NONE;
    cr[j] = (x < W) ? rowr[x] : RB_NONE;
  }
  for (int i = tid; i < P + 2; i += RB_THREADS) {
    b_cnt[i] = 0u;
    b_cm[i] = 0u;
  }
  if (tid == 0) {
    s_max_r = -1;
    s_tail_cnt = 0;
    s_tail_minx = 0xFFFFFFFFu;
  }
  __syncthreads();

  // ---- 1. count: arrival slot of every record inside its bucket (empty pixel slots use the spare bucket)
  uint32_t bl[SPT], br[SPT], sl[SPT], sr[SPT];
  int max_r = -1;
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    bl[j] = (cl[j] != RB_NONE) ? (cl[j] >> SHIFT) : (uint32_t)(P + 1);
    br[j] = (cr[j] != RB_NONE) ? (cr[j] >> SHIFT) : (uint32_t)(P + 1);
  }
#pragma unroll
  for (int j = 0; j < SPT; ++j) sl[j] = atomicAdd(&b_cnt[bl[j]], 1u);
#pragma unroll
  for (int j = 0; j < SPT; ++j) sr[j] = atomicAdd(&b_cnt[br[j]], 1u);
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if (cr[j] != RB_NONE) max_r = max(max_r, (int)cr[j]);
  for (int o = 32; o > 0; o >>= 1) max_r = max(max_r, __shfl_xor(max_r, o));
  if (lane == 0 && max_r >= 0) atomicMax(&s_max_r, max_r);
  __syncthreads();

  // ---- 2. bucket starts
  rb_exscan<SPT>(b_cnt, s_w, tid);

  // ---- 3. scatter the records
  uint32_t pl[SPT];  // index of my left records
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    pl[j] = 0;
    if (cl[j] != RB_NONE) {
      pl[j] = b_cnt[bl[j]] + sl[j];
      e_key[pl[j]] = cl[j] << 1;
      e_x[pl[j]] = (uint16_t)(j * RB_THREADS + tid);
    }
    if (cr[j] != RB_NONE) {
      const uint32_t p = b_cnt[br[j]] + sr[j];
      e_key[p] = (cr[j] << 1) | 1u;
      e_x[p] = (uint16_t)(j * RB_THREADS + tid);
    }
  }
  // tail quirk group: the largest right code of the last populated right row
  // matches iff it occurs exactly twice on the right (then with the first in mask order)
  const bool tail_row = (y == img_stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_LASTROW]);
  const uint32_t tail_code = (uint32_t)s_max_r;
  if (tail_row) {  // block-uniform
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (cr[j] == tail_code) {
        atomicAdd(&s_tail_cnt, 1);
        atomicMin(&s_tail_minx, (unsigned)(j * RB_THREADS + tid));
      }
  }
  __syncthreads();

  // ---- 4. decide every left record from its own bucket
  uint32_t xr[SPT];
  bool ok[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    ok[j] = false;
    xr[j] = 0;
    if (cl[j] != RB_NONE) {
      const uint32_t s = b_cnt[bl[j]], e = b_cnt[bl[j] + 1];
      const uint32_t kl = cl[j] << 1, kr = kl | 1u;
      uint32_t nl = 0, nr = 0, ir = s;
      uint32_t k[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) k[q] = e_key[min(s + q, (uint32_t)(2 * P - 1))];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (s + q < e) {
          nl += (k[q] == kl);
          if (k[q] == kr) { ++nr; ir = s + q; }
        }
      for (uint32_t i = s + 4; i < e && nl < 2; ++i) {
        const uint32_t kk = e_key[i];
        nl += (kk == kl);
        if (kk == kr) { ++nr; ir = i; }
      }
      const bool tail = tail_row && cl[j] == tail_code;
      bool good = (nl == 1u) && (tail ? (s_tail_cnt == 2) : (nr == 1u));
      if (good) {
        xr[j] = tail ? s_tail_minx : (uint32_t)e_x[ir];
        if (apply_filter && abs((int)(j * RB_THREADS + tid) - (int)xr[j]) > disp_high) good = false;
      }
      ok[j] = good;
      if (good) {
        atomicAdd(&b_cm[bl[j]], 1u);
        e_x[pl[j]] = (uint16_t)((j * RB_THREADS + tid) | RB_MATCHED);
      }
    }
  }
  __syncthreads();

  // ---- 5. rank of every match in ascending code order == output position
  rb_exscan<SPT>(b_cm, s_w, tid);
  const long rowbase = (long)pair * H + y;
  uint32_t* dst = staged + rowbase * W;
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if (ok[j]) {
      const uint32_t s = b_cnt[bl[j]], e = b_cnt[bl[j] + 1];
      const uint32_t kl = cl[j] << 1;
      uint32_t rank = b_cm[bl[j]];
      for (uint32_t i = s; i < e; ++i) {
        const uint32_t kk = e_key[i];
        // left record (even key) with a smaller code that matched
        if (!(kk & 1u) && kk < kl && (e_x[i] & RB_MATCHED)) ++rank;
      }
      dst[rank] = (uint32_t)(j * RB_THREADS + tid) | (xr[j] << 16);
    }
  if (tid == 0) rowcnt[rowbase] = (int32_t)b_cm[P];
}

}  // namespace gpc
