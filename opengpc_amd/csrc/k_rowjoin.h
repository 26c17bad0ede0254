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
//   1. the left row's codes are inserted into an ordered open-addressing table in LDS with
//      ds_max_rtn (no CAS: measured ~9x cheaper); both rows then look their code up (reads)
//      and add (1<<16)+x into a per-slot left / right accumulator with ds_add;
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

// Diagnostic build only (-DGPC_STAMPS, tools/stamp_profile.py): s_memtime at phase boundaries,
// summed per phase into a debug buffer nothing else reads.  No stamp executes in the product build.
#ifdef GPC_STAMPS
__device__ unsigned long long g_rj_stamps[16];
#define RJ_STAMP(i)                                                                         \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    rj_acc[i] = t_ - rj_t0;                                                                 \
    rj_t0 = t_;                                                                             \
  } while (0)
// one workgroup in 64 reports (uncontended atomics, issued after the last stamp)
#define RJ_STAMP_FLUSH()                                                                    \
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 5)                                           \
    for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_rj_stamps[i_], rj_acc[i_])
#define RJ_STAMP_INIT()                                                                     \
  unsigned long long rj_t0;                                                                 \
  unsigned long long rj_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                  \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rj_t0)::"memory")
#else
#define RJ_STAMP(i)
#define RJ_STAMP_INIT()
#define RJ_STAMP_FLUSH()
#endif

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

// ---------------------------------------------------------------- the join table
// Measured on MI355X (profiles/r01_ubench_lds_valu_calibration.txt): a wave-level LDS
// compare-and-swap costs ~72 cycles of the CU's LDS pipe, a returning ds_max/ds_add ~8, a plain
// random ds_read ~3.  So the table is built WITHOUT CAS: ordered linear probing (Amble & Knuth)
// with ds_max_rtn -- a probe writes max(slot, key); if it displaced a smaller key it carries
// that key on to the next slot.  Within one insert phase this converges to the unique ordered
// table whatever the interleaving; lookups (after the barrier) are read-only and stop at the
// first slot holding a smaller key.  Stored key = code + 1 (0 = empty slot).
// Only LEFT codes are inserted; right records merely look their code up.  Per slot two
// accumulators collect (count << 16) + x with non-returning ds_add: while count == 1 the low
// half is that record's x, and a count field can never read 1 for count >= 2.

// insert SPT keys per thread (0 = none); first probe of all slots in straight-line code
template <int SPT>
__device__ __forceinline__ void rj_insert_ordered(uint32_t* __restrict__ t_key, const uint32_t (&k)[SPT],
                                                  int hshift, uint32_t smask) {
  uint32_t h[SPT], old[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) h[j] = k[j] ? rj_hash(k[j], hshift) : smask + 1u;  // spare slot S absorbs "none"
#pragma unroll
  for (int j = 0; j < SPT; ++j) old[j] = atomicMax(&t_key[h[j]], k[j]);
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    uint32_t cur = k[j], o = old[j];
    // o == 0: slot was empty (created) ; o == cur: already present ; o < cur: displaced o ; o > cur: keep cur
    while (o != 0u && o != cur) {
      if (o < cur) cur = o;
      h[j] = (h[j] + 1) & smask;
      o = atomicMax(&t_key[h[j]], cur);
    }
  }
}

// slot of key k (k != 0) or ~0u when absent; read-only
__device__ __forceinline__ uint32_t rj_find(const uint32_t* __restrict__ t_key, uint32_t k, uint32_t first,
                                            uint32_t h, uint32_t smask) {
  uint32_t kk = first;
  while (kk > k) {  // larger keys sit in front of k on its probe path
    h = (h + 1) & smask;
    kk = t_key[h];
  }
  return kk == k ? h : 0xFFFFFFFFu;
}

// codes:   [npairs*2][H][W]   (image 2p = left, 2p+1 = right)
// staged:  [npairs][H][W]     packed (xL | xR<<16), first rowcnt entries of each row valid
// rowcnt:  [npairs][H]
// grid: (H - 26, npairs); SPT*256 >= W; table of S = 1 << log2s slots, S >= 2*(W-26)
// dynamic LDS: 12*(S+1) + 4*256*SPT bytes  (28 KiB for W = 1024)
template <int SPT>
__global__ __launch_bounds__(RJ_THREADS) void k_row_join(
    const uint32_t* __restrict__ codes, int W, int H, int disp_high, int apply_filter,
    const int32_t* __restrict__ img_stats, uint32_t* __restrict__ staged, int32_t* __restrict__ rowcnt,
    int log2s) {
  extern __shared__ __attribute__((aligned(16))) uint32_t rj_lds[];
  __shared__ int s_max_r, s_tail_cnt, s_wcnt[RJ_THREADS / 64];
  __shared__ unsigned s_tail_minx;
  const int S = 1 << log2s;
  uint32_t* t_key = rj_lds;                 // S slots + 1 spare (index S) that absorbs no-op probes
  uint32_t* t_wl = rj_lds + (S + 1);        // left  accumulators (count << 16) + x
  uint32_t* t_wr = rj_lds + 2 * (S + 1);    // right accumulators
  uint32_t* xbuf = rj_lds + 3 * (S + 1);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y = GPC_R + blockIdx.x;
  const int pair = blockIdx.y;
  const int hshift = 32 - log2s;
  const uint32_t smask = (uint32_t)S - 1u;

  RJ_STAMP_INIT();
  // ---- 0. both rows' loads first (their latency hides behind the table init)
  const uint32_t* rowl = codes + ((long)(pair * 2) * H + y) * W;
  const uint32_t* rowr = rowl + (long)H * W;
  uint32_t cl[SPT], cr[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    const int x = j * RJ_THREADS + tid;
    cl[j] = (x < W) ? rowl[x] : RJ_EMPTY;
    cr[j] = (x < W) ? rowr[x] : RJ_EMPTY;
  }
  for (int i = tid; i < 3 * (S + 1); i += RJ_THREADS) rj_lds[i] = 0u;
  if (tid == 0) {
    s_max_r = -1;
    s_tail_cnt = 0;
    s_tail_minx = 0xFFFFFFFFu;
  }
  __syncthreads();
  RJ_STAMP(0);
#ifdef GPC_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RJ_STAMP(1);

  // ---- 1. build the ordered table from the left codes (stored key = code + 1)
  uint32_t kl[SPT], kr[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    kl[j] = (cl[j] != RJ_EMPTY) ? cl[j] + 1u : 0u;
    kr[j] = (cr[j] != RJ_EMPTY) ? cr[j] + 1u : 0u;
  }
  rj_insert_ordered<SPT>(t_key, kl, hshift, smask);
  int max_r = -1;
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if (cr[j] != RJ_EMPTY) max_r = max(max_r, (int)cr[j]);
  for (int o = 32; o > 0; o >>= 1) max_r = max(max_r, __shfl_xor(max_r, o));
  if (lane == 0 && max_r >= 0) atomicMax(&s_max_r, max_r);
  __syncthreads();
  RJ_STAMP(2);

  // ---- 2. every record finds its code's slot (read-only) and adds (1 << 16) + x to its side
  uint32_t hl[SPT];
  {
    uint32_t h0l[SPT], h0r[SPT], f0l[SPT], f0r[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      h0l[j] = kl[j] ? rj_hash(kl[j], hshift) : smask + 1u;
      h0r[j] = kr[j] ? rj_hash(kr[j], hshift) : smask + 1u;
    }
#pragma unroll
    for (int j = 0; j < SPT; ++j) {  // first probes of all records together
      f0l[j] = t_key[h0l[j]];
      f0r[j] = t_key[h0r[j]];
    }
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const uint32_t x = (uint32_t)(j * RJ_THREADS + tid);
      hl[j] = kl[j] ? rj_find(t_key, kl[j], f0l[j], h0l[j], smask) : smask + 1u;  // always found
      const uint32_t hr = kr[j] ? rj_find(t_key, kr[j], f0r[j], h0r[j], smask) : 0xFFFFFFFFu;
      atomicAdd(&t_wl[hl[j]], kl[j] ? ((1u << 16) + x) : 0u);
      if (hr != 0xFFFFFFFFu) atomicAdd(&t_wr[hr], (1u << 16) + x);
    }
  }
  // Tail quirks of the reference's merge scan concern only the largest right
  // code of the last right row that has candidates: it matches iff it occurs exactly TWICE on
  // the right (then with the first of the two in mask order) and once on the left.
  const bool tail_row = (y == img_stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_LASTROW]);
  const uint32_t tail_code = (uint32_t)s_max_r;
  if (tail_row) {  // block-uniform
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (cr[j] == tail_code) {
        atomicAdd(&s_tail_cnt, 1);
        atomicMin(&s_tail_minx, (unsigned)(j * RJ_THREADS + tid));
      }
  }
  __syncthreads();
  RJ_STAMP(3);

  // ---- 3. decide every left candidate
  uint32_t key[SPT];
  int nmatch = 0;
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    key[j] = RJ_EMPTY;
    const uint32_t wl = t_wl[hl[j]], wr = t_wr[hl[j]];
    if (cl[j] != RJ_EMPTY) {
      const bool tail = tail_row && cl[j] == tail_code;
      bool ok = ((wl >> 16) == 1u) && (tail ? (s_tail_cnt == 2) : ((wr >> 16) == 1u));
      if (ok && apply_filter) {
        const int xl = j * RJ_THREADS + tid;
        const int xr = tail ? (int)s_tail_minx : (int)(wr & 0xFFFFu);
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
  RJ_STAMP(4);

  // ---- 4. sort the kept codes (sentinels go last); element index = tid * SPT + reg
  sort_merges<SPT, 2, SPT * RJ_THREADS>(key, xbuf, tid);
  __syncthreads();
  RJ_STAMP(5);

  // ---- 5. sorted position == output position; (xL, xR) come from the code's slot
  const long rowbase = (long)pair * H + y;
  uint32_t* dst = staged + rowbase * W;
  {
    uint32_t h0[SPT], f0[SPT], h[SPT];
#pragma unroll
    for (int r = 0; r < SPT; ++r) h0[r] = (key[r] != RJ_EMPTY) ? rj_hash(key[r] + 1u, hshift) : smask + 1u;
#pragma unroll
    for (int r = 0; r < SPT; ++r) f0[r] = t_key[h0[r]];
#pragma unroll
    for (int r = 0; r < SPT; ++r)
      h[r] = (key[r] != RJ_EMPTY) ? rj_find(t_key, key[r] + 1u, f0[r], h0[r], smask) : smask + 1u;
    uint32_t wl[SPT], wr[SPT];
#pragma unroll
    for (int r = 0; r < SPT; ++r) {
      wl[r] = t_wl[h[r]];
      wr[r] = t_wr[h[r]];
    }
#pragma unroll
    for (int r = 0; r < SPT; ++r)
      if (key[r] != RJ_EMPTY) {
        const uint32_t xr = (tail_row && key[r] == tail_code) ? s_tail_minx : (wr[r] & 0xFFFFu);
        dst[tid * SPT + r] = (wl[r] & 0xFFFFu) | (xr << 16);
      }
  }
#ifdef GPC_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RJ_STAMP(6);
  RJ_STAMP_FLUSH();
  if (tid == 0) {
    int total = 0;
    for (int w = 0; w < RJ_THREADS / 64; ++w) total += s_wcnt[w];
    rowcnt[rowbase] = total;
  }
}

}  // namespace gpc
