// k_global.h -- non-epipolar ("global") collision matching: device-wide radix sort + scan.
//
// Replaces, for settings.epipolarMode_ == false, Forest::findCorrespondences
// (std::sort x2 + merge scan, inference.hpp:227-254) and the filter of
// Forest::rectifiedMatch (inference.hpp:384-391).
//
// Without the row in the state, a code must be unique over the whole left image and over
// the whole right image.  Both descriptor sets go into ONE array of (code, side<<31 | k)
// records -- left records first, each side in mask (row-major) order -- which a stable LSD
// radix sort (4 passes x 8 bits: all 32 code bits) orders by (code, side, k).  Matches
// are then read off neighbouring records exactly as in the per-row kernel, including the
// tail quirks of the reference's merge scan for the group of the largest right-image code.
//
// All sizes stay on the device (candidate counts are never read back); grids are sized for
// the worst case N = 2*(W-26)*(H-26) and surplus workgroups exit.
#pragma once
#include "gpc_device.h"
#include "k_rows.h"

namespace gpc {

#ifndef GS_THREADS
#define GS_THREADS 512
#endif
#ifndef GS_ITEMS
#define GS_ITEMS 8
#endif
#define GS_TILE (GS_THREADS * GS_ITEMS)  // records per workgroup and pass
#define GS_WAVES (GS_THREADS / 64)
#define GS_CHUNK (GS_TILE / GS_WAVES)    // records per wave

// gmisc layout (int32, GM_STRIDE per pair): [0] number of records N, [1] largest right-image code (unsigned;
// meaningful when N > N_L), [2] N_L
#define GM_N 0
#define GM_MAXR 1
#define GM_NL 2
#define GM_STRIDE 16

// All kernels of this file (and k_hashtable.h) take the pair from blockIdx.y / .z and find their
// per-pair slices with these strides, so one launch serves a whole batch.
struct GpcBatchStrides {
  long codes;   // u32 words between the code images of consecutive pairs (2*W*H)
  long recs;    // records (u32 words) per pair in the key / value arrays (nmax)
  long hist;    // histogram words per pair (256 * nblk)
  long blk;     // match-block counters per pair (nmblk)
  long out;     // BYTES between the output arrays of consecutive pairs (cap * element size)
  int rows;     // rowcnt words per pair (2 * H)
};

// ---- build the record array from the two code images of one pair --------------------
// grid: (H - 26, 2, npairs)
// A pixel of the code image is a record: its code is not the sentinel, or -- 32-bit codes only (cand != nullptr,
// see k_rowjoin.h WIDE) -- it is the code 0xFFFFFFFF of a candidate (the hash kernel's rule: byte set, inside the margin).
__device__ __forceinline__ bool g_is_record(uint32_t c, const uint8_t* __restrict__ cand_row, int x, int W) {
  return c != GPC_NOCAND || (cand_row != nullptr && x >= GPC_R && x < W - GPC_R && cand_row[x] != 0);
}

__global__ __launch_bounds__(RM_THREADS) void k_g_rowcount(const uint32_t* __restrict__ codes,
                                                           const uint8_t* __restrict__ cand, int W, int H,
                                                           int32_t* __restrict__ rowcnt,
                                                           const int32_t* __restrict__ stats,
                                                           int32_t* __restrict__ gmisc, GpcBatchStrides bs) {
  const int y = GPC_R + blockIdx.x, side = blockIdx.y, pair = blockIdx.z;
  codes += pair * bs.codes;
  rowcnt += pair * bs.rows;
  stats += pair * 2 * GPC_STAT_STRIDE;
  gmisc += pair * GM_STRIDE;
  const uint32_t* row = codes + ((long)side * H + y) * W;
  const uint8_t* crow = cand ? cand + pair * bs.codes + ((long)side * H + y) * W : nullptr;
  int v = 0;
  for (int x = threadIdx.x; x < W; x += RM_THREADS) v += g_is_record(row[x], crow, x, W);
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __shared__ int s_part[RM_THREADS / 64];
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < RM_THREADS / 64; ++w) s += s_part[w];
    rowcnt[side * H + y] = s;
    if (blockIdx.x == 0 && side == 0) {
      const int nl = stats[GPC_STAT_NCAND], nr = stats[GPC_STAT_STRIDE + GPC_STAT_NCAND];
      gmisc[GM_N] = nl + nr;
      gmisc[GM_NL] = nl;
      gmisc[GM_MAXR] = 0;
    }
  }
}

// grid: (H - 26, 2, npairs); runs after k_g_rowcount
__global__ __launch_bounds__(RM_THREADS) void k_g_build(const uint32_t* __restrict__ codes,
                                                        const uint8_t* __restrict__ cand, int W, int H,
                                                        const int32_t* __restrict__ rowcnt,
                                                        const int32_t* __restrict__ stats,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        int32_t* __restrict__ gmisc, GpcBatchStrides bs) {
  __shared__ int s_wcnt[RM_THREADS / 64];
  const int y = GPC_R + blockIdx.x, side = blockIdx.y, pair = blockIdx.z;
  codes += pair * bs.codes;
  rowcnt += pair * bs.rows;
  stats += pair * 2 * GPC_STAT_STRIDE;
  keys += pair * bs.recs;
  vals += pair * bs.recs;
  gmisc += pair * GM_STRIDE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int total = block_prefix_rows(rowcnt + side * H, GPC_R, y);
  if (side) total += stats[GPC_STAT_NCAND];  // right records follow all left records
  const uint32_t* row = codes + ((long)side * H + y) * W;
  const uint8_t* crow = cand ? cand + pair * bs.codes + ((long)side * H + y) * W : nullptr;
  uint32_t max_r = 0u;
  for (int x0 = 0; x0 < W; x0 += RM_THREADS) {
    const int x = x0 + threadIdx.x;
    const uint32_t c = (x < W) ? row[x] : GPC_NOCAND;
    const bool valid = x < W && g_is_record(c, crow, x, W);
    const unsigned long long m = __ballot(valid);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = total;
    for (int w = 0; w < wave; ++w) off += s_wcnt[w];
    if (valid) {
      const int pos = off + __popcll(m & lanemask_lt());
      keys[pos] = c;
      vals[pos] = ((uint32_t)side << 31) | (uint32_t)(y * W + x);
      if (side) max_r = max(max_r, c);
    }
    for (int w = 0; w < RM_THREADS / 64; ++w) total += s_wcnt[w];
    __syncthreads();
  }
  if (side) {  // unsigned: 32-bit codes use bit 31
    for (int o = 32; o > 0; o >>= 1) max_r = max(max_r, (uint32_t)__shfl_xor((int)max_r, o));
    if (lane == 0 && max_r) atomicMax(reinterpret_cast<uint32_t*>(&gmisc[GM_MAXR]), max_r);
  }
}

// ---- one LSD radix pass ---------------------------------------------------------------
// Lanes of a wave that hold the same 8-bit digit.
__device__ __forceinline__ unsigned long long digit_peers(unsigned digit, bool valid) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const unsigned long long bal = __ballot((digit >> b) & 1u);
    peers &= ((digit >> b) & 1u) ? bal : ~bal;
  }
  return peers;
}

// hist[d * nblk + blk] = number of records of tile blk whose digit is d
__global__ __launch_bounds__(GS_THREADS) void k_g_hist(const uint32_t* __restrict__ keys,
                                                       const int32_t* __restrict__ gmisc, int shift,
                                                       int32_t* __restrict__ hist, int nblk, GpcBatchStrides bs) {
  __shared__ int s_hist[256];
  keys += blockIdx.y * bs.recs;
  hist += blockIdx.y * bs.hist;
  gmisc += blockIdx.y * GM_STRIDE;
  const int N = gmisc[GM_N];
  const int blk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 256) s_hist[tid] = 0;
  __syncthreads();
  const int base = blk * GS_TILE + wave * GS_CHUNK;
  if (blk * GS_TILE < N) {
    uint32_t k[GS_ITEMS];
#pragma unroll
    for (int r = 0; r < GS_ITEMS; ++r) {
      const int i = base + r * 64 + lane;
      k[r] = (i < N) ? keys[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < GS_ITEMS; ++r) {
      const int i = base + r * 64 + lane;
      const bool valid = i < N;
      const unsigned digit = (k[r] >> shift) & 0xFFu;
      const unsigned long long peers = digit_peers(digit, valid);
      if (valid && (peers & lanemask_lt()) == 0) atomicAdd(&s_hist[digit], __popcll(peers));
    }
  }
  __syncthreads();
  if (tid < 256) hist[tid * nblk + blk] = s_hist[tid];
}

// exclusive scan of `total` ints in place; one workgroup of 1024 threads = 16 waves, each wave
// owning a contiguous slice: pass 1 sums the slice (coalesced loads, no cross-lane work), the 16
// slice totals are combined through LDS, pass 2 rescans the slice with a DPP wave scan and a
// scalar carry.  (~100 KB of counters per radix pass: latency-, not bandwidth-bound.)
__global__ __launch_bounds__(1024) void k_g_scan(int32_t* __restrict__ data, int total, long pair_stride) {
  __shared__ int s_tot[16];
  data += blockIdx.y * pair_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (((total + 15) / 16) + 63) / 64 * 64;
  const int beg = wave * per;
  const int end = min(beg + per, total);
  // 16-byte accesses where the table allows them (its length and the pairs' stride multiples of four words): four times
  // fewer memory round trips in a kernel that is nothing but round trips (16 workgroups per batch of 8 pairs)
  const bool vec = ((total & 3) == 0) && ((reinterpret_cast<uintptr_t>(data) & 15) == 0);
  int acc = 0;
  if (vec) {
    for (int i = beg + 4 * lane; i < end; i += 256 * 4) {
      int4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = (i + 256 * u < end) ? *reinterpret_cast<const int4*>(data + i + 256 * u) : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += q[u].x + q[u].y + q[u].z + q[u].w;
    }
  } else {
    for (int i = beg + lane; i < end; i += 64) acc += data[i];
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) s_tot[wave] = acc;
  __syncthreads();
  int carry = 0;
  for (int w = 0; w < wave; ++w) carry += s_tot[w];
  if (vec) {
    constexpr int U = 4;
    for (int i0 = beg; i0 < end; i0 += 256 * U) {
      int4 q[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 256 * u + 4 * lane;
        q[u] = (i < end) ? *reinterpret_cast<const int4*>(data + i) : make_int4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + 256 * u + 4 * lane;
        const int sum = q[u].x + q[u].y + q[u].z + q[u].w;
        const int incl = (int)wave_incl_scan((uint32_t)sum);
        const int b0 = carry + incl - sum;
        if (i < end) *reinterpret_cast<int4*>(data + i) = make_int4(b0, b0 + q[u].x, b0 + q[u].x + q[u].y, b0 + q[u].x + q[u].y + q[u].z);
        carry += __builtin_amdgcn_readlane(incl, 63);
      }
    }
    return;
  }
  // eight rounds' values are requested together: the scan is in place, so a round's load may not pass the store of the
  // round before it
  constexpr int U = 8;
  for (int i0 = beg; i0 < end; i0 += 64 * U) {
    int v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 64 * u + lane;
      v[u] = (i < end) ? data[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 64 * u + lane;
      const int incl = (int)wave_incl_scan((uint32_t)v[u]);
      if (i < end) data[i] = carry + incl - v[u];
      carry += __builtin_amdgcn_readlane(incl, 63);
    }
  }
}

// One stable scatter pass.  The tile's records are first put in digit order in LDS (stable: wave,
// round, lane order = record order), then written out by consecutive threads, so that every run of
// a digit leaves the workgroup as one contiguous, coalesced piece.
__global__ __launch_bounds__(GS_THREADS) void k_g_scatter(const uint32_t* __restrict__ keys_in,
                                                          const uint32_t* __restrict__ vals_in,
                                                          uint32_t* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out,
                                                          const int32_t* __restrict__ gmisc, int shift,
                                                          const int32_t* __restrict__ hist, int nblk,
                                                          GpcBatchStrides bs) {
  __shared__ uint32_t s_k[GS_TILE], s_v[GS_TILE];
  __shared__ int s_run[GS_WAVES][256];  // counts per (wave, digit) -> tile-local start of that wave's run
  __shared__ int s_gofs[256];           // global position of the tile's first record of digit d, minus its local start
  __shared__ uint32_t s_w[4];
  keys_in += blockIdx.y * bs.recs;
  vals_in += blockIdx.y * bs.recs;
  keys_out += blockIdx.y * bs.recs;
  vals_out += blockIdx.y * bs.recs;
  hist += blockIdx.y * bs.hist;
  gmisc += blockIdx.y * GM_STRIDE;
  const int N = gmisc[GM_N];
  const int blk = blockIdx.x;
  if (blk * GS_TILE >= N) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < GS_WAVES * 256; i += GS_THREADS) (&s_run[0][0])[i] = 0;
  __syncthreads();

  // pass A: every wave counts the digits of its own contiguous chunk (records stay in registers)
  uint32_t k[GS_ITEMS], v[GS_ITEMS];
  const int base = blk * GS_TILE + wave * GS_CHUNK;
#pragma unroll
  for (int r = 0; r < GS_ITEMS; ++r) {
    const int i = base + r * 64 + lane;
    const bool valid = i < N;
    k[r] = valid ? keys_in[i] : 0u;
    v[r] = valid ? vals_in[i] : 0u;
  }
#pragma unroll
  for (int r = 0; r < GS_ITEMS; ++r) {
    const int i = base + r * 64 + lane;
    const bool valid = i < N;
    const unsigned digit = (k[r] >> shift) & 0xFFu;
    const unsigned long long peers = digit_peers(digit, valid);
    if (valid && (peers & lanemask_lt()) == 0) s_run[wave][digit] += __popcll(peers);
  }
  __syncthreads();
  // digit d (thread d): runs of the waves one behind the other; then the digits one behind the other
  uint32_t dcount = 0;
  if (tid < 256) {
    int acc = 0;
    for (int w = 0; w < GS_WAVES; ++w) {
      const int c = s_run[w][tid];
      s_run[w][tid] = acc;
      acc += c;
    }
    dcount = (uint32_t)acc;
    const uint32_t incl = wave_incl_scan(dcount);
    if (lane == 63) s_w[wave] = incl;
    dcount = incl - dcount;  // exclusive within the wave
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t lstart = dcount;
    for (int w = 0; w < wave; ++w) lstart += s_w[w];
    s_gofs[tid] = hist[tid * nblk + blk] - (int)lstart;
    for (int w = 0; w < GS_WAVES; ++w) s_run[w][tid] += (int)lstart;
  }
  __syncthreads();
  // pass B: stable tile-local ranks in (wave, round, lane) order == record order
#pragma unroll
  for (int r = 0; r < GS_ITEMS; ++r) {
    const int i = base + r * 64 + lane;
    const bool valid = i < N;
    const unsigned digit = (k[r] >> shift) & 0xFFu;
    const unsigned long long peers = digit_peers(digit, valid);
    int lpos = 0;
    if (valid) lpos = s_run[wave][digit] + __popcll(peers & lanemask_lt());
    // all lanes of the wave have read s_run before the leaders update it
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      s_k[lpos] = k[r];
      s_v[lpos] = v[r];
      if ((peers & lanemask_lt()) == 0) s_run[wave][digit] += __popcll(peers);
    }
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  const int cnt = min(GS_TILE, N - blk * GS_TILE);
#pragma unroll
  for (int r = 0; r < GS_ITEMS; ++r) {
    const int i = r * GS_THREADS + tid;
    if (i < cnt) {
      const uint32_t kk = s_k[i];
      const int pos = s_gofs[(kk >> shift) & 0xFFu] + i;
      keys_out[pos] = kk;
      vals_out[pos] = s_v[i];
    }
  }
}

// ---- matches off the sorted records -----------------------------------------------------
#define GMT_RPT 4                          // sorted records per thread
#define GMT_TILE (RM_THREADS * GMT_RPT)    // sorted records per workgroup

// Record j = jm1 + q (staged in LDS: s_k[q] / s_v[q] hold record jm1 + q, jm1 = j0 - 1) is a left record whose
// code occurs once on the left and once on the right (tail quirk: once + twice for the largest right code).
// Neighbours are told apart from "beyond the last record" by their index, not by a sentinel: every 32-bit
// value can be a code (k_rowjoin.h, WIDE).
__device__ __forceinline__ bool g_match_at(const uint32_t* s_k, const uint32_t* s_v, int q, int jm1, int N,
                                           const GpcDivW& wd, bool have_tail, uint32_t tail_code, int disp_high,
                                           int vtol, int apply_filter, int4& m) {
  const int j = jm1 + q;
  if (j >= N) return false;
  const uint32_t code = s_k[q];
  const uint32_t v0 = s_v[q];
  if (v0 >> 31) return false;  // right record
  const bool prev_same = j > 0 && s_k[q - 1] == code;
  const bool n1r = j + 1 < N && s_k[q + 1] == code && (s_v[q + 1] >> 31);
  if (prev_same || !n1r) return false;
  const bool n2 = j + 2 < N && s_k[q + 2] == code;
  const bool n3 = j + 3 < N && s_k[q + 3] == code;
  const bool ok = (have_tail && code == tail_code) ? (n2 && !n3) : !n2;
  if (!ok) return false;
  const int kl = (int)(v0 & 0x7FFFFFFFu), kr = (int)(s_v[q + 1] & 0x7FFFFFFFu);
  const int yl = divw((uint32_t)kl, wd), yr = divw((uint32_t)kr, wd);
  m = make_int4(kl - yl * wd.W, yl, kr - yr * wd.W, yr);
  if (apply_filter && (abs(m.y - m.w) > vtol || abs(m.x - m.z) > disp_high)) return false;
  return true;
}

// pass A (!WRITE): matches per workgroup -> blkcnt (then scanned exclusively by k_g_scan);
// pass B (WRITE): the matches themselves, in sorted order.  A workgroup stages GMT_TILE records
// (+1 before, +3 after) in LDS with coalesced loads; a thread owns GMT_RPT consecutive records.
template <bool WRITE>
__global__ __launch_bounds__(RM_THREADS) void k_g_match(const uint32_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ vals,
                                                        const int32_t* __restrict__ gmisc, GpcDivW wd,
                                                        int disp_high, int vtol, int apply_filter,
                                                        int32_t* __restrict__ blkcnt, int mode,
                                                        void* __restrict__ out, int cap,
                                                        int32_t* __restrict__ count_out,
                                                        const int32_t* __restrict__ stats,
                                                        int32_t* __restrict__ ncand_out, GpcBatchStrides bs) {
  __shared__ uint32_t s_k[GMT_TILE + 4], s_v[GMT_TILE + 4];
  __shared__ uint32_t s_w[RM_THREADS / 64];
  keys += blockIdx.y * bs.recs;
  vals += blockIdx.y * bs.recs;
  blkcnt += blockIdx.y * bs.blk;
  gmisc += blockIdx.y * GM_STRIDE;
  const int N = gmisc[GM_N];
  const uint32_t tail_code = (uint32_t)gmisc[GM_MAXR];
  const bool have_tail = N > gmisc[GM_NL];  // there are right records
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * GMT_TILE;
  for (int q = tid; q < GMT_TILE + 4; q += RM_THREADS) {
    const int j = j0 - 1 + q;
    const bool in = j >= 0 && j < N;
    s_k[q] = in ? keys[j] : 0u;
    s_v[q] = in ? vals[j] : 0u;
  }
  __syncthreads();
  uint32_t hits = 0u;
#pragma unroll
  for (int k = 0; k < GMT_RPT; ++k) {
    int4 m;
    if (g_match_at(s_k, s_v, 1 + tid * GMT_RPT + k, j0 - 1, N, wd, have_tail, tail_code, disp_high, vtol, apply_filter, m))
      hits |= 1u << k;
  }
  const uint32_t cnt = (uint32_t)__popc(hits);
  const uint32_t incl = wave_incl_scan(cnt);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = incl - cnt, total = 0;
  for (int w = 0; w < RM_THREADS / 64; ++w) {
    if (w < wave) base += s_w[w];
    total += s_w[w];
  }
  if (!WRITE) {
    if (tid == 0) blkcnt[blockIdx.x] = (int32_t)total;
    return;
  }
  stats += blockIdx.y * 2 * GPC_STAT_STRIDE;
  out = reinterpret_cast<char*>(out) + blockIdx.y * bs.out;
  const int off = blkcnt[blockIdx.x];  // exclusive prefix over the pair's workgroups
  int pos = off + (int)base;
#pragma unroll
  for (int k = 0; k < GMT_RPT; ++k)
    if ((hits >> k) & 1u) {
      if (pos < cap) {
        int4 m;
        (void)g_match_at(s_k, s_v, 1 + tid * GMT_RPT + k, j0 - 1, N, wd, have_tail, tail_code, disp_high, vtol, apply_filter, m);
        if (mode == 0) {
          uint32_t* o = reinterpret_cast<uint32_t*>(out) + (long)pos * 3;
          o[0] = m.x;
          o[1] = m.y;
          o[2] = __float_as_uint((float)(m.x - m.z));
        } else {
          reinterpret_cast<int4*>(out)[pos] = m;
        }
      }
      ++pos;
    }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    count_out[blockIdx.y] = off + (int)total;
    if (ncand_out) {
      ncand_out[2 * blockIdx.y + 0] = stats[GPC_STAT_NCAND];
      ncand_out[2 * blockIdx.y + 1] = stats[GPC_STAT_STRIDE + GPC_STAT_NCAND];
    }
  }
}

}  // namespace gpc
