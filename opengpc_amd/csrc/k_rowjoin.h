// k_rowjoin.h -- epipolar-mode collision matching: per-row LDS hash join + counting rank.
//
// Replaces, for settings.epipolarMode_ == true, the descriptor build + `state |= y<<32`
// (inference.hpp:189-197), Forest::findCorrespondences (std::sort x2 + merge scan,
// inference.hpp:227-254) and the disparity filter of rectifiedMatch (inference.hpp:384-391).
// One image row per workgroup (see k_rowmatch.h for why rows are independent).
//
// What the reference's sort+merge decides for a row is, per code c:  cntL(c) == 1 and
// cntR(c) == 1  (with the tail-quirk variant cntR == 2 for the largest right code of the
// last populated right row); the sort is needed only for the ORDER of the output (ascending
// code).  So, entirely in LDS and without any compare-and-swap or sorting network:
//   1. the left row's codes are inserted into an ordered open-addressing table with
//      ds_max_rtn (Amble-Knuth ordered linear probing; a wave-level LDS CAS measured ~72
//      cycles of LDS pipe on MI355X, a returning ds_max ~8);
//   2. both rows look their code up (plain reads) and mark the slot: a returning ds_or sets the
//      side's SEEN flag, and a record that finds it already set adds the side's DUP flag; a right
//      record also leaves its x in the low half of the slot's word (plain 16-bit store: if the
//      right code is unique there was one writer, otherwise the value is not used);
//   3. every left record reads its slot: match iff neither side is DUP and the right side was
//      SEEN (+ disparity filter);
//   4. output position = rank of the code among the row's matches, by COUNTING on the top
//      bits of the code: one returning ds_add per match, an exclusive scan over 256*SPT bucket
//      counters (DPP wave scan), and a look at the < 1 other matches sharing the bucket.
//      The thread still holds xL and xR, so it writes the packed support straight to its place.
#pragma once
#include "gpc_device.h"

namespace gpc {

#define RJ_THREADS 256
#define RJ_EMPTY 0xFFFFFFFFu
// flags of a table slot (high half of its word; the low half holds a right record's x)
#define RJ_LSEEN 0x00010000u
#define RJ_LDUP 0x00020000u
#define RJ_RSEEN 0x00040000u
#define RJ_RDUP 0x00080000u

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

__device__ __forceinline__ uint32_t rj_hash(uint32_t code, int shift) { return (code * 0x9E3779B1u) >> shift; }

// ---------------------------------------------------------------- the join table
// Measured on MI355X (profiles/r01_ubench_lds_valu_calibration.txt): a wave-level LDS
// compare-and-swap costs ~72 cycles of the CU's LDS pipe, a returning ds_max/ds_add ~8, a plain
// random ds_read ~3.  So the table is built WITHOUT CAS: ordered linear probing (Amble & Knuth)
// with ds_max_rtn -- a probe writes max(slot, key); if it displaced a smaller key it carries
// that key on to the next slot.  Within one insert phase this converges to the unique ordered
// table whatever the interleaving; lookups (after the barrier) are read-only and stop at the
// first slot holding a smaller key.  Stored key = code + 1 (0 = empty slot).
// Only LEFT codes are inserted; right records merely look their code up.  A slot costs 8 bytes
// (key + one word of flags and x): 16 KiB per 1024-px row and 64 VGPRs, so EIGHT workgroups = 32
// waves share a CU.  Earlier layouts, per 256 pairs: 12 bytes (count << 16 + x summed into two
// 32-bit accumulators per slot), six workgroups: 686 us; 10 bytes (two 16-bit counts + 16-bit x),
// seven: 628 us; this one: 595 us.

// insert SPT keys per thread (0 = none); the first probes of all slots are issued together
template <int SPT>
__device__ __forceinline__ void rj_insert_ordered(uint32_t* __restrict__ t_key, const uint32_t (&k)[SPT],
                                                  int hshift, uint32_t smask) {
  uint32_t h[SPT], old[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) h[j] = rj_hash(k[j], hshift);
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    old[j] = 0u;
    if (k[j]) old[j] = atomicMax(&t_key[h[j]], k[j]);
  }
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
// grid: (H - 26, npairs); NT threads, NB = NT*SPT >= W; table of S = 1 << log2s slots,
//       S >= max(2*(W-26), NB)   (only left codes are inserted: load factor <= 0.5)
// dynamic LDS: 8*(S+1) bytes  (16 KiB for W = 1024: 8 workgroups per CU = 32 waves, 64 VGPRs)
// Wide rows use more threads per row instead of more pixel slots per thread, so that the one
// or two workgroups that fit a CU (98 KiB of table at W = 3840) still fill its SIMDs.
template <int SPT, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_row_join(
    const uint32_t* __restrict__ codes, int W, int H, int disp_high, int apply_filter,
    const int32_t* __restrict__ img_stats, uint32_t* __restrict__ staged, int32_t* __restrict__ rowcnt,
    int log2s, int rpw) {
  constexpr int NB = NT * SPT;
  constexpr int RSHIFT = 31 - ((NT == 256 ? 8 : NT == 512 ? 9 : 10) + (SPT == 1 ? 0 : SPT == 2 ? 1 : SPT == 4 ? 2 : SPT == 8 ? 3 : 4));
  extern __shared__ __attribute__((aligned(16))) uint32_t rj_lds[];
  __shared__ int s_max_r, s_tail_cnt;
  __shared__ unsigned s_tail_minx;
  __shared__ uint32_t s_w[NT / 64];
  const int S = 1 << log2s;
  uint32_t* t_key = rj_lds;               // [S]   stored key = code + 1, 0 = empty
  uint32_t* t_w = rj_lds + (S + 1);       // [S]   per slot: seen / duplicate flags of either side (RJ_*), x of a right record in the low half
  uint32_t* r_cnt = t_key;                // [NB+1] bucket counters -> starts   (reuses t_key, dead after step 2)
  uint32_t* r_key = t_w;                  // [NB]   matched codes, bucket-contiguous (reuses t_w, dead after step 3)

  const int tid = threadIdx.x, lane = tid & 63;
  const int pair = blockIdx.y;
  const int hshift = 32 - log2s;
  const uint32_t smask = (uint32_t)S - 1u;

  // A workgroup handles `rpw` consecutive rows; the NEXT row's codes are fetched into registers
  // while the current row is joined, so only the first row's load latency is exposed.
  const int row0 = GPC_R + blockIdx.x * rpw;
  uint32_t ncl[SPT], ncr[SPT];
  auto fetch_row = [&](int yy) {
    const uint32_t* rl = codes + ((long)(pair * 2) * H + yy) * W;
    const uint32_t* rr_ = rl + (long)H * W;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const int x = j * NT + tid;
      ncl[j] = (x < W) ? rl[x] : RJ_EMPTY;
      ncr[j] = (x < W) ? rr_[x] : RJ_EMPTY;
    }
  };
  fetch_row(row0);
#pragma unroll 1
  for (int ri = 0; ri < rpw && row0 + ri < H - GPC_R; ++ri) {
  const int y = row0 + ri;
  RJ_STAMP_INIT();
  // ---- 0. this row's codes (already in flight), table clear, next row's loads
  uint32_t cl[SPT], kl[SPT], kr[SPT];
  {
    uint32_t cr[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      cl[j] = ncl[j];
      cr[j] = ncr[j];
    }
    if (ri + 1 < rpw && y + 1 < H - GPC_R) fetch_row(y + 1);
    {  // 16-byte stores; the host rounds the allocation up to a multiple of 16 bytes
      uint4* z = reinterpret_cast<uint4*>(rj_lds);
      for (int i = tid; i < (8 * (S + 1) + 15) / 16; i += NT) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid == 0) {
      s_max_r = -1;
      s_tail_cnt = 0;
      s_tail_minx = 0xFFFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < SPT; ++j) {  // stored key = code + 1 (0 = no record in this pixel slot)
      kl[j] = cl[j] + 1u;
      kr[j] = cr[j] + 1u;
    }
  }
  __syncthreads();
  RJ_STAMP(0);

  // ---- 1. build the ordered table from the left codes
  rj_insert_ordered<SPT>(t_key, kl, hshift, smask);
  uint32_t max_k = 0;
#pragma unroll
  for (int j = 0; j < SPT; ++j) max_k = max(max_k, kr[j]);
  for (int o = 32; o > 0; o >>= 1) max_k = max(max_k, (uint32_t)__shfl_xor((int)max_k, o));
  if (lane == 0 && max_k) atomicMax(&s_max_r, (int)(max_k - 1u));
  __syncthreads();
  RJ_STAMP(1);

  // ---- 2. every record finds its code's slot (read-only) and adds (1 << 16) + x to its side
  uint32_t hl[SPT];
  {
    uint32_t h0l[SPT], h0r[SPT], f0l[SPT], f0r[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      h0l[j] = rj_hash(kl[j], hshift);
      h0r[j] = rj_hash(kr[j], hshift);
    }
#pragma unroll
    for (int j = 0; j < SPT; ++j) {  // first probes of all records together
      f0l[j] = 0u;
      f0r[j] = 0u;
      if (kl[j]) f0l[j] = t_key[h0l[j]];
      if (kr[j]) f0r[j] = t_key[h0r[j]];
    }
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const uint32_t x = (uint32_t)(j * NT + tid);
      hl[j] = 0u;
      if (kl[j]) {
        hl[j] = rj_find(t_key, kl[j], f0l[j], h0l[j], smask);  // a left code is always found
        if (atomicOr(&t_w[hl[j]], RJ_LSEEN) & RJ_LSEEN) atomicOr(&t_w[hl[j]], RJ_LDUP);  // a second left record of this code
      }
      if (kr[j]) {
        const uint32_t hr = rj_find(t_key, kr[j], f0r[j], h0r[j], smask);
        if (hr != 0xFFFFFFFFu) {
          if (atomicOr(&t_w[hr], RJ_RSEEN) & RJ_RSEEN) atomicOr(&t_w[hr], RJ_RDUP);
          // plain 16-bit store beside the flags: several writers only when the code is not unique, and then x is not used
          reinterpret_cast<uint16_t*>(t_w)[2 * hr] = (uint16_t)x;
        }
      }
    }
  }
  // Tail quirks of the reference's merge scan concern only the largest right
  // code of the last right row that has candidates: it matches iff it occurs exactly TWICE on
  // the right (then with the first of the two in mask order) and once on the left.
  const bool tail_row = (y == img_stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_LASTROW]);
  const uint32_t tail_key = (uint32_t)s_max_r + 1u;
  if (tail_row) {  // block-uniform
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (kr[j] && kr[j] == tail_key) {
        atomicAdd(&s_tail_cnt, 1);
        atomicMin(&s_tail_minx, (unsigned)(j * NT + tid));
      }
  }
  __syncthreads();
  RJ_STAMP(2);

  // ---- 3. decide every left candidate; the key table is dead already: it becomes the rank counters
  for (int i = tid; i <= NB; i += NT) r_cnt[i] = 0u;
  bool ok[SPT];
  uint32_t xr[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    ok[j] = false;
    xr[j] = 0u;
    if (kl[j]) {
      const uint32_t w = t_w[hl[j]];
      const bool tail = tail_row && kl[j] == tail_key;
      bool good = !(w & RJ_LDUP) && (tail ? (s_tail_cnt == 2) : ((w & (RJ_RSEEN | RJ_RDUP)) == RJ_RSEEN));
      xr[j] = tail ? s_tail_minx : (w & 0xFFFFu);
      if (good && apply_filter) good = abs((int)(j * NT + tid) - (int)xr[j]) <= disp_high;
      ok[j] = good;
    }
  }
  __syncthreads();  // the accumulators are dead from here on: their LDS is reused
  RJ_STAMP(3);

  // ---- 4. output position = rank of the code among the row's matches (counting rank)
  uint32_t rb[SPT], rs[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    rb[j] = cl[j] >> RSHIFT;
    rs[j] = 0u;
    if (ok[j]) rs[j] = atomicAdd(&r_cnt[rb[j]], 1u);
  }
  __syncthreads();
  RJ_STAMP(4);
  block_exscan<SPT, NT>(r_cnt, s_w, tid);  // r_cnt[b] = first rank of bucket b, r_cnt[NB] = number of matches
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if (ok[j]) r_key[r_cnt[rb[j]] + rs[j]] = cl[j];
  __syncthreads();
  RJ_STAMP(5);
  const long rowbase = (long)pair * H + y;
  uint32_t* dst = staged + rowbase * W;
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if (ok[j]) {
      const uint32_t s0 = r_cnt[rb[j]], e0 = r_cnt[rb[j] + 1];
      uint32_t rank = s0;
      for (uint32_t i = s0; i < e0; ++i) rank += (r_key[i] < cl[j]);
      dst[rank] = (uint32_t)(j * NT + tid) | (xr[j] << 16);
    }
#ifdef GPC_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RJ_STAMP(6);
  RJ_STAMP_FLUSH();
  if (tid == 0) rowcnt[rowbase] = (int32_t)r_cnt[NB];
  if (ri + 1 < rpw && y + 1 < H - GPC_R) __syncthreads();  // the table is cleared again for the next row
  }  // rows of this workgroup
}

}  // namespace gpc
