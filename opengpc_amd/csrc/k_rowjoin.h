// k_rowjoin.h -- epipolar-mode collision matching, second generation: per-row LDS hash join
// + register-resident bitonic sort of the matched 32-bit codes.
//
// Same contract as k_row_match (k_rowmatch.h): replaces, for epipolarMode_ == true, the
// descriptor build + `state |= y<<32`, Forest::findCorrespondences and the disparity filter
// (inference.hpp:189-197, 227-254, 384-391) -- one image row per workgroup.
//
// What the reference's sort+merge decides for a row is, per code c:  cntL(c) == 1 and
// cntR(c) == 1  (with the tail-quirk variant cntR == 2 for the largest right code of the
// last populated right row), and what it needs the sort for is only the ORDER of the output
// (ascending code).  So instead of sorting 2(W-26) 64-bit (code, side, x) keys:
//   1. both rows are inserted into an open-addressing hash table in LDS keyed by code:
//      per slot  key | cntL,cntR (packed u32, ds_add) | xL | min xR (ds_min);
//   2. every left candidate reads its slot and keeps its code if it is a match, else ~0;
//   3. the <= W kept 32-bit codes (SPT per thread, already in registers) are bitonic-sorted
//      without touching LDS for strides below 64*SPT: in-register v_min/v_max, DPP
//      (quad_perm / row_half_mirror / row_mirror / row_ror), ds_swizzle and ds_bpermute for
//      the lane exchanges; only the two wave-level strides go through a 4*P-byte LDS buffer;
//   4. sorted position == output position: each kept code looks its (xL, xR) up again and
//      writes the packed support to its slot of the row's staging area.
#pragma once
#include "gpc_device.h"

namespace gpc {

#define RJ_THREADS 256
#define RJ_EMPTY 0xFFFFFFFFu

// ---------------------------------------------------------------- lane exchanges
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
  // every lane has a valid source for the controls used here, so `old` is never selected
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}

// value of `v` held by lane (lane ^ LM); LM in {1,2,3,4,7,8,15,16,31,32,63}
template <int LM>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
  if constexpr (LM == 1) return dpp_mov<0xB1>(v);         // quad_perm [1,0,3,2]
  else if constexpr (LM == 2) return dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  else if constexpr (LM == 3) return dpp_mov<0x1B>(v);    // quad_perm [3,2,1,0]
  else if constexpr (LM == 7) return dpp_mov<0x141>(v);   // row_half_mirror
  else if constexpr (LM == 15) return dpp_mov<0x140>(v);  // row_mirror
  else if constexpr (LM == 8) return dpp_mov<0x128>(v);   // row_ror:8
  else if constexpr (LM == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);   // xor 4
  else if constexpr (LM == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);  // xor 16
  else if constexpr (LM == 31) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x7C1F);  // xor 31
  else return (uint32_t)__shfl_xor((int)v, LM);           // 32, 63: ds_bpermute
}

constexpr int highest_bit(int m) {
  int b = 1;
  while ((b << 1) <= m) b <<= 1;
  return b;
}

// One compare-exchange stage of the sorting network: element i meets element i ^ M, the
// smaller key stays at the smaller index.  i = tid * SPT + reg.
template <int SPT, int M>
__device__ __forceinline__ void sort_stage(uint32_t (&key)[SPT], uint32_t* __restrict__ xbuf, int tid) {
  constexpr int MREG = M & (SPT - 1);
  constexpr int MLANE = (M / SPT) & 63;
  constexpr int MWAVE = M / (SPT * 64);
  if constexpr (MLANE == 0 && MWAVE == 0) {
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const int q = r ^ MREG;
      if (q > r) {
        const uint32_t lo = min(key[r], key[q]), hi = max(key[r], key[q]);
        key[r] = lo;
        key[q] = hi;
      }
    }
  } else if constexpr (MWAVE == 0) {
    uint32_t p[SPT];
#pragma unroll
    for (int r = 0; r < SPT; ++r) p[r] = lane_xor<MLANE>(key[r ^ MREG]);
    const bool lower = ((tid & 63) & highest_bit(MLANE)) == 0;
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      // lower index keeps the smaller key: one compare, the mask XOR is scalar, one select
      const bool take = (p[r] < key[r]) == lower;
      key[r] = take ? p[r] : key[r];
    }
  } else {
    const int base = tid * SPT;
#pragma unroll
    for (int r = 0; r < SPT; ++r) xbuf[base + r] = key[r];
    __syncthreads();
    const bool lower = (base & (highest_bit(MWAVE) * SPT * 64)) == 0;
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      const uint32_t p = xbuf[(base + r) ^ M];
      const bool take = (p < key[r]) == lower;
      key[r] = take ? p : key[r];
    }
    __syncthreads();
  }
}

template <int SPT, int J>
__device__ __forceinline__ void sort_substages(uint32_t (&key)[SPT], uint32_t* __restrict__ xbuf, int tid) {
  if constexpr (J >= 1) {
    sort_stage<SPT, J>(key, xbuf, tid);
    sort_substages<SPT, J / 2>(key, xbuf, tid);
  }
}

// merges of size K, 2K, ... up to P  (flip stage i ^ (K-1), then strides K/4 .. 1)
template <int SPT, int K, int P>
__device__ __forceinline__ void sort_merges(uint32_t (&key)[SPT], uint32_t* __restrict__ xbuf, int tid) {
  if constexpr (K <= P) {
    sort_stage<SPT, K - 1>(key, xbuf, tid);
    sort_substages<SPT, K / 4>(key, xbuf, tid);
    sort_merges<SPT, K * 2, P>(key, xbuf, tid);
  }
}

__device__ __forceinline__ uint32_t rj_hash(uint32_t code, int shift) { return (code * 0x9E3779B1u) >> shift; }

// codes:   [npairs*2][H][W]   (image 2p = left, 2p+1 = right)
// staged:  [npairs][H][W]     packed (xL | xR<<16), first rowcnt entries of each row valid
// rowcnt:  [npairs][H]
// grid: (H - 26, npairs); SPT*256 >= W; table of S = 1 << log2s slots, S > 2*(W-26)
// dynamic LDS: 16*S + 4*256*SPT bytes
template <int SPT>
__global__ __launch_bounds__(RJ_THREADS) void k_row_join(
    const uint32_t* __restrict__ codes, int W, int H, int disp_high, int apply_filter,
    const int32_t* __restrict__ img_stats, uint32_t* __restrict__ staged, int32_t* __restrict__ rowcnt,
    int log2s) {
  extern __shared__ __attribute__((aligned(16))) uint32_t rj_lds[];
  __shared__ int s_max_r, s_wcnt[RJ_THREADS / 64];
  const int S = 1 << log2s;
  uint32_t* t_key = rj_lds;
  uint32_t* t_info = rj_lds + S;
  uint32_t* t_xl = rj_lds + 2 * S;
  uint32_t* t_xr = rj_lds + 3 * S;
  uint32_t* xbuf = rj_lds + 4 * S;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y = GPC_R + blockIdx.x;
  const int pair = blockIdx.y;
  const int hshift = 32 - log2s;
  const uint32_t smask = (uint32_t)S - 1u;

  for (int i = tid; i < S; i += RJ_THREADS) {
    t_key[i] = RJ_EMPTY;
    t_info[i] = 0u;
    t_xr[i] = 0xFFFFFFFFu;
  }
  if (tid == 0) s_max_r = -1;
  __syncthreads();

  // ---- 1. insert both rows
  const uint32_t* rowl = codes + ((long)(pair * 2) * H + y) * W;
  const uint32_t* rowr = rowl + (long)H * W;
  uint32_t cl[SPT], cr[SPT];
  uint32_t hl[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {  // all global loads first, so their latency overlaps
    const int x = j * RJ_THREADS + tid;
    cl[j] = (x < W) ? rowl[x] : RJ_EMPTY;
    cr[j] = (x < W) ? rowr[x] : RJ_EMPTY;
  }
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    const int x = j * RJ_THREADS + tid;
    const uint32_t c = cl[j];
    hl[j] = 0;
    if (c != RJ_EMPTY) {
      uint32_t h = rj_hash(c, hshift);
      while (true) {
        const uint32_t old = atomicCAS(&t_key[h], RJ_EMPTY, c);
        if (old == RJ_EMPTY || old == c) break;
        h = (h + 1) & smask;
      }
      atomicAdd(&t_info[h], 1u);
      t_xl[h] = (uint32_t)x;  // only read back when cntL == 1
      hl[j] = h;
    }
  }
  int max_r = -1;
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    const int x = j * RJ_THREADS + tid;
    const uint32_t c = cr[j];
    if (c != RJ_EMPTY) {
      uint32_t h = rj_hash(c, hshift);
      while (true) {
        const uint32_t old = atomicCAS(&t_key[h], RJ_EMPTY, c);
        if (old == RJ_EMPTY || old == c) break;
        h = (h + 1) & smask;
      }
      atomicAdd(&t_info[h], 0x10000u);
      atomicMin(&t_xr[h], (uint32_t)x);  // Q2 tie-break: the first right record in mask order
      max_r = max(max_r, (int)c);
    }
  }
  for (int o = 32; o > 0; o >>= 1) max_r = max(max_r, __shfl_xor(max_r, o));
  if (lane == 0 && max_r >= 0) atomicMax(&s_max_r, max_r);
  __syncthreads();

  // ---- 2. decide every left candidate
  const bool tail_row = (y == img_stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_LASTROW]);
  const uint32_t tail_code = (uint32_t)s_max_r;
  uint32_t key[SPT];
  int nmatch = 0;
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    key[j] = RJ_EMPTY;
    if (cl[j] != RJ_EMPTY) {
      const uint32_t inf = t_info[hl[j]];
      const uint32_t nl = inf & 0xFFFFu, nr = inf >> 16;
      const bool tail = tail_row && cl[j] == tail_code;
      bool ok = (nl == 1u) && (tail ? (nr == 2u) : (nr == 1u));
      if (ok && apply_filter) {
        const int xl = j * RJ_THREADS + tid, xr = (int)t_xr[hl[j]];
        ok = abs(xl - xr) <= disp_high;
      }
      if (ok) {
        key[j] = cl[j];
        ++nmatch;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) nmatch += __shfl_xor(nmatch, o);
  if (lane == 0) s_wcnt[wave] = nmatch;

  // ---- 3. sort the kept codes (sentinels go last); element index = tid * SPT + reg
  sort_merges<SPT, 2, SPT * RJ_THREADS>(key, xbuf, tid);
  __syncthreads();

  // ---- 4. sorted position == output position
  const long rowbase = (long)pair * H + y;
  uint32_t* dst = staged + rowbase * W;
#pragma unroll
  for (int r = 0; r < SPT; ++r) {
    const uint32_t c = key[r];
    if (c != RJ_EMPTY) {
      uint32_t h = rj_hash(c, hshift);
      while (t_key[h] != c) h = (h + 1) & smask;
      dst[tid * SPT + r] = t_xl[h] | (t_xr[h] << 16);
    }
  }
  if (tid == 0) {
    int total = 0;
    for (int w = 0; w < RJ_THREADS / 64; ++w) total += s_wcnt[w];
    rowcnt[rowbase] = total;
  }
}

}  // namespace gpc
