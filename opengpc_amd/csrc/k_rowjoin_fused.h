// k_rowjoin_fused.h -- epipolar-mode collision matching, join + output in ONE persistent launch.
//
// The same per-row LDS hash join as k_rowjoin.h (descriptor build + `state |= y<<32`, inference.hpp:189-197;
// Forest::findCorrespondences, inference.hpp:227-254; the filter of rectifiedMatch, inference.hpp:384-391) -- see that
// header for the join itself: ordered ds_max table, SEEN/DUP flag marks, counting rank -- but the kernel also does what
// k_gather_rows did: it writes the row's supports straight to their final place in the caller's array, in row order.
// The place of a row is the number of supports of the rows before it, which is only known while the launch runs, so:
//   * rows are HANDED OUT, not mapped to block indices: the grid is as many workgroups as the device holds at once
//     (persistent), and a workgroup draws its next row from a ticket counter (one per shard of pairs; agent-scope
//     atomic add, requested most of a row ahead so that its latency is never waited for).  Ticket g of a shard of Ps
//     pairs is row g / Ps of the shard's pair g % Ps: every earlier row of a pair has a smaller ticket, so whoever holds
//     it is running (or done) -- waiting for its count cannot deadlock, whatever order the hardware starts workgroups in;
//   * a row publishes its support count as soon as it knows it, and its inclusive prefix as soon as it knows that
//     (decoupled look-back, Merrill & Garland): one 8-byte granule {epoch << 2 | state, value} per row, written by one
//     agent-scope store and polled with agent-scope loads by ONE wave (MI355X_MICROARCH.md: the data is the flag).
//     The look-back is bounded: a poll that never matches stores the launch's epoch into a host-visible error word and
//     the row goes on (the host then reports GPC_E_HIP) -- every wave of the grid reaches its exit whatever happens;
//   * matches go to their rank in an LDS array (D) first and leave as whole 12-byte records, consecutive lanes writing
//     consecutive records;
//   * the output of a row is DEFERRED by one row of its workgroup: the rows of a pair that are in flight together
//     (workgroups per shard / Ps: 8 at 256 pairs of 1024x436) reach their counts within a microsecond or two of each
//     other, and a poll of another workgroup's granule costs 2-3 us under load -- resolving the look-back inside the
//     row put that wait on every row's critical path.  So row n asks for its predecessors' granules after row n+1's
//     insert phase, looks at the answer after row n+1's lookup phase and writes row n's records while row n+1 ranks.
//
// Round 4: a kernel of its own (it was the FUSE instantiation of k_row_join).  What binds it (docs/HISTORY.md 3, "what binds
// the fused join"): the ~11 us a row spends inside its workgroup times the eight rows a CU holds -- all 32 wave slots and all
// 160 KB of LDS -- not instruction count, LDS throughput, HBM latency or barriers (each was varied by itself).  What changed:
//   * ONE by-value parameter (RjfArgs).  What every wave needs on its critical path (W, H, code image, statistics, filter
//     settings) is used by value; the rest is loaded from the kernel-argument segment (rjf_args()) by the one wave that
//     uses it.  As ~40 ordinary parameters they stayed live in SGPRs for the whole persistent loop and, at the 80 SGPRs
//     eight waves per SIMD leave a wave, 42 of them were spilled to VGPR lanes; now 0-11, and no scratch anywhere;
//   * the table size is a compile-time constant (S = 2 * NT * SPT): masks, shifts and LDS offsets are immediates;
//   * lane predicates (tid == 0, x < W, ...) are recomputed from an opaque thread index where they are used instead of
//     being kept as 64-bit masks across the row;
//   * ticket -> (pair, row) by a host-made multiply-high instead of a scalar division sequence per row;
//   * LDS regions no longer alias across phases that a barrier had to separate: the rank counters have a region of
//     their own (cleared with the key table at the top of the row), the matched codes of shared buckets live in the flag
//     words (dead after the decide phase, cleared during the NEXT row's insert phase), so the decide phase runs straight
//     into the rank-count atomics and the row's last phase straight into the next row's clear: 7 barriers per row, not 9
//     (the LDS of a 1024-pixel row is 20 384 + 48 bytes: exactly the sixteen 1280-byte granules an eighth of a CU holds);
//   * a code that occurs twice among the LEFT records is noticed at insert time, by the lane whose carried key meets its
//     copy: left records need no returning mark (4 LDS atomics and a dependent round trip per row less);
//   * rank phase: a match reads its bucket's COUNT before the scan turns the counters into starts -- four matches in
//     five are alone in their bucket and need neither a place among the bucket's codes nor the walk over them; the walk
//     itself is a hand-written v_cmpx / s_cbranch_execnz loop like the probe loops;
//   * the host deals the pairs over 3 .. 13 ticket counters, never a multiple of 8: workgroup b serves shard b % shards
//     and runs on XCD b % 8, and a shard pinned to one XCD made the XCDs finish apart (gpc_hip.hip: join_shards).
#pragma once
#include <cstddef>

#include "k_rowjoin.h"

namespace gpc {

#define RJ_SHARDS 64               // at most; the host takes 16 (GPC_HIP_FUSE_SHARDS)
#define RJ_TICKET_STRIDE 32        // words between the shards' counters: one 128-byte line each
#define RJ_ST_AGG 1u               // granule holds the row's own support count
#define RJ_ST_PREFIX 2u            // granule holds the supports of this row and all rows before it
#define RJ_SPIN_LIMIT (1 << 18)    // polls of one look-back window before the row gives up (~0.3 s)

// The ONE parameter of k_row_join_fused.  The kernel never touches its by-value copy: every field is loaded from the
// kernel-argument segment where it is used (HIP lays a by-value struct out at offset 0 of the segment as C++ lays it
// out in memory, so rjf_args() simply types the segment pointer with this very struct).
struct RjfArgs {
  const uint32_t* codes;        // [npairs*2][H][W]   (image 2p = left, 2p+1 = right), allocated with a pixel-slot row of slack
  const uint8_t* cand;          // [npairs*2][H][W]   candidate bytes: WIDE only
  const int32_t* img_stats;     // [npairs*2][GPC_STAT_STRIDE]
  uint32_t* tickets;            // [RJ_SHARDS * RJ_TICKET_STRIDE]: draw counters, zero between launches (the last draw resets)
  unsigned long long* status;   // [npairs][H - 26]: look-back granules
  int32_t* err;                 // host-visible word: the epoch of a launch one of whose look-backs timed out (0: none)
  void* out;
  int32_t* counts;              // [npairs]
  int32_t* ncand;               // [npairs][2] or null
  int32_t* rows_out;            // mode 2: [npairs][rows_stride]
  long packed_stride, rows_stride;
  int W, H, disp_high, apply_filter;
  uint32_t epoch;               // tag of this launch's granules (never 0; the host counts launches)
  int npairs, nshards;
  int mode;                     // 0: gpc_support, 1: gpc_correspondence, 2: packed words + row counts (k_rows.h)
  int cap;
  // pairs per shard and the multiply-high that divides a ticket by it (GpcDivW's form: g / ps == umulhi(g, magic) >> sh
  // for g < 2^31; ps == 1 divides by nothing): shards below n_hi hold ps[0] pairs, the others ps[1]
  int n_hi;
  int ps[2];
  uint32_t ps_magic[2];
  int ps_sh[2];
};

typedef const RjfArgs __attribute__((address_space(4))) RjfK;
// (opaque: a load through the returned pointer is made where it is used, not hoisted to the kernel's top and kept)
__device__ __forceinline__ RjfK* rjf_args() {
  RjfK* p = (RjfK*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}
#define RJF_OPAQUE_V(v) asm volatile("" : "+v"(v))

__device__ __forceinline__ unsigned long long rjf_granule(uint32_t epoch, uint32_t state, uint32_t value) {
  return ((unsigned long long)((epoch << 2) | state) << 32) | value;
}

// First window of the look-back of row t (st = its granule): lane l asks for row t-1-l; "rows" before the first hold
// a prefix of 0.  Executed by one whole wave; the answer is consumed by rjf_lookback.
__device__ __forceinline__ unsigned long long rjf_lookback_ask(const unsigned long long* st, int t, int lane, uint32_t epoch) {
  unsigned long long g = rjf_granule(epoch, RJ_ST_PREFIX, 0u);
#ifdef RJ_DBG_NOLB
  return g;
#endif
  // row0 = granule of the pair's first row (uniform): one scalar base + a 32-bit lane offset
  const unsigned long long* row0 = st - t;
  if (t - 1 - lane >= 0)
    g = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(reinterpret_cast<const char*>(row0) + (uint32_t)(t - 1 - lane) * 8u),
                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return g;
}

// Supports of the pair's rows before row t (one whole wave; every lane gets the result).  g = rjf_lookback_ask's answer.
// Counts are added up to the nearest published prefix; a window in which a row in front of that prefix has not
// published yet is asked for again (bounded: then *err receives the launch's epoch -- a plain system-scope store: the word
// lives in page-locked host memory, and a device ATOMIC over PCIe is dropped on links without PCIe atomics -- and the sum
// so far is returned).  Beyond the first window -- few pairs in flight: hundreds of rows of one pair run together and the
// nearest prefix is that far back -- RJ_LB_WINDOWS windows are asked for at a time, so that a round trip covers 256 rows.
#define RJ_LB_WINDOWS 4
__device__ __forceinline__ uint32_t rjf_lookback(const unsigned long long* st, int t, int lane, uint32_t epoch,
                                                 unsigned long long g, int32_t* err) {
  uint32_t base = 0u;
#ifdef RJ_DBG_NOLB
  return 0u;
#endif
  int pos = t - 1, spin = 0;  // wave-uniform: the row lane 0 of the next window looks at
#ifdef RJ_DBG_COUNT
  if (lane == 0) atomicAdd(err + 1, 1);  // [1] look-backs
#endif
  // one window: adds what it can; returns 1 when the prefix was reached, 0 when the window was all counts, -1 when a
  // row in front of the nearest prefix has not published yet
  auto window = [&](unsigned long long gv) -> int {
    const uint32_t tag = (uint32_t)(gv >> 32);
    const bool ready = (tag >> 2) == epoch;
    const unsigned long long notyet = __ballot(!ready);
    const unsigned long long pfx = __ballot(ready && (tag & 3u) == RJ_ST_PREFIX);
    const int first_n = notyet ? __ffsll((long long)notyet) - 1 : 64;
    const int first_p = pfx ? __ffsll((long long)pfx) - 1 : 64;
    if (first_n < first_p) return -1;
    uint32_t c = (lane <= first_p) ? (uint32_t)gv : 0u;  // counts of the rows in front of the prefix, and the prefix
    c = wave_incl_scan(c);
    base += (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
    return first_p < 64 ? 1 : 0;
  };
  int r = window(g);
  if (r == 0) pos -= 64;
  while (r != 1) {
    if (r < 0) {
#ifdef RJ_DBG_COUNT
      if (lane == 0) atomicAdd(err + 2, 1);  // [2] windows with a row that had not published
#endif
      if (++spin > RJ_SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(err, (int32_t)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
    unsigned long long gw[RJ_LB_WINDOWS];
#pragma unroll
    for (int k = 0; k < RJ_LB_WINDOWS; ++k) gw[k] = rjf_lookback_ask(st - (t - 1 - (pos - 64 * k)), pos - 64 * k + 1, lane, epoch);
#pragma unroll
    for (int k = 0; k < RJ_LB_WINDOWS; ++k) {
      r = window(gw[k]);
      if (r != 0) break;  // done, or a row of this window has to be waited for
      pos -= 64;
    }
  }
  return base;
}

// The cnt ranked words (xL | xR << 16) of row y of `pair` leave as records base .. base + cnt - 1 of the pair's array.
template <int NT>
__device__ __forceinline__ void rjf_emit_row(const uint32_t* __restrict__ words, uint32_t cnt, uint32_t base, int pair, int y, int tid) {
#ifdef RJ_DBG_NOEMIT
  return;
#endif
  RjfK* a = rjf_args();
  const int mode = a->mode, cap = a->cap;
  if (mode == 0) {
    struct __attribute__((packed, aligned(4))) Rec3 { uint32_t x, y, d; };
    Rec3* o = reinterpret_cast<Rec3*>(a->out) + (long)pair * cap;
    for (uint32_t i = tid; i < cnt; i += NT) {
      const uint32_t w = words[i], p = base + i;
      const int xl = (int)(w & 0xFFFFu), xr = (int)(w >> 16);
      if (p < (uint32_t)cap) o[p] = Rec3{(uint32_t)xl, (uint32_t)y, __float_as_uint((float)(xl - xr))};
    }
  } else if (mode == 2) {
    uint32_t* o = reinterpret_cast<uint32_t*>(a->out) + pair * a->packed_stride;
    for (uint32_t i = tid; i < cnt; i += NT) {
      const uint32_t p = base + i;
      if (p < (uint32_t)cap) o[p] = words[i];
    }
    if (tid == 0) a->rows_out[pair * a->rows_stride + y] = (int32_t)cnt;
  } else {
    int4* o = reinterpret_cast<int4*>(a->out) + (long)pair * cap;
    for (uint32_t i = tid; i < cnt; i += NT) {
      const uint32_t w = words[i], p = base + i;
      if (p < (uint32_t)cap) o[p] = make_int4((int)(w & 0xFFFFu), y, (int)(w >> 16), y);
    }
  }
}

// the pair's last row knows the total: counts[pair], and the candidate counts of its two images
__device__ __forceinline__ void rjf_finish_pair(int pair, uint32_t total) {
  RjfK* a = rjf_args();
  a->counts[pair] = (int32_t)total;
  int32_t* nc = a->ncand;
  if (nc) {
    const int32_t* st = a->img_stats;
    nc[pair * 2 + 0] = st[(pair * 2 + 0) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
    nc[pair * 2 + 1] = st[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
  }
}

// The hand-written loops narrow EXEC and put it back at their end.  Every call site in this kernel runs with all 64 lanes of
// the wave active (the workgroup is a whole number of waves, the row loop and everything the loops sit in is wave-uniform), so
// "back" is all ones: no copy of EXEC saved in front of each of the 16 loops of a row (one instruction each, and a scalar
// register pair the kernel does not have to spare).  -DRJF_CHECK_EXEC traps where that is not so.
#ifdef RJF_CHECK_EXEC
#define RJF_EXEC_ALL "s_mov_b64 exec, -1\n\t"
#define RJF_ASSERT_EXEC_ALL() do { if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap(); } while (0)
#else
#define RJF_EXEC_ALL "s_mov_b64 exec, -1"
#define RJF_ASSERT_EXEC_ALL() do { } while (0)
#endif

// rank of code cj among the `cnt` codes keys[s0 ..] of its bucket (cnt > 1: the bucket is shared; lanes with cnt <= 1 keep
// rank = s0).  The lanes still walking narrow EXEC with v_cmpx and the loop ends on s_cbranch_execnz: one scalar
// instruction per round where the compiler's structurised divergent loop spent more scalar than vector instructions.
// keys_lds = LDS byte offset of the code array.
__device__ __forceinline__ uint32_t rjf_walk(uint32_t keys_lds, uint32_t s0, uint32_t cnt, uint32_t cj) {
  uint32_t rank = s0, addr, k;
  asm volatile(
      "v_cmpx_lt_u32_e32 vcc, 1, %[cnt]\n\t"
      "s_cbranch_execz 2f\n\t"
      "v_lshl_add_u32 %[addr], %[s0], 2, %[base]\n"
      "1:\n\t"
      "ds_read_b32 %[k], %[addr]\n\t"
      "v_add_u32_e32 %[addr], 4, %[addr]\n\t"
      "v_add_u32_e32 %[cnt], -1, %[cnt]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_lt_u32_e32 vcc, %[k], %[cj]\n\t"
      "v_addc_co_u32_e32 %[rank], vcc, 0, %[rank], vcc\n\t"
      "v_cmpx_ne_u32_e32 vcc, 0, %[cnt]\n\t"
      "s_cbranch_execnz 1b\n"
      "2:\n\t"
      RJF_EXEC_ALL
      : [rank] "+v"(rank), [cnt] "+v"(cnt), [addr] "=&v"(addr), [k] "=&v"(k)
      : [s0] "v"(s0), [cj] "v"(cj), [base] "s"(keys_lds)
      : "vcc", "memory");
  return rank;
}

// rj_insert_chain with its end state handed back: cur = the key the lane carried last, o = what its last probe returned.
// o == cur (and cur != 0) says the carried key met a copy of itself: that key occurs twice among the left records.
__device__ __forceinline__ void rjf_insert_chain(uint32_t keys_lds, uint32_t& cur, uint32_t& o, uint32_t h, uint32_t smask) {
  uint32_t addr;
  asm volatile(
      "v_cmpx_ne_u32_e32 vcc, 0, %[cur]\n\t"       // lanes without a record never probe on
      "v_cmpx_ne_u32_e32 vcc, 0, %[o]\n\t"
      "v_cmpx_ne_u32_e32 vcc, %[o], %[cur]\n\t"
      "s_cbranch_execz 2f\n"
      "1:\n\t"
      "v_min_u32_e32 %[cur], %[o], %[cur]\n\t"
      "v_add_u32_e32 %[h], 1, %[h]\n\t"
      "v_and_b32_e32 %[h], %[smask], %[h]\n\t"
      "v_lshl_add_u32 %[addr], %[h], 2, %[base]\n\t"
      "ds_max_rtn_u32 %[o], %[addr], %[cur]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmpx_ne_u32_e32 vcc, 0, %[o]\n\t"
      "v_cmpx_ne_u32_e32 vcc, %[o], %[cur]\n\t"
      "s_cbranch_execnz 1b\n"
      "2:\n\t"
      RJF_EXEC_ALL
      : [cur] "+v"(cur), [o] "+v"(o), [h] "+v"(h), [addr] "=&v"(addr)
      : [smask] "s"(smask), [base] "s"(keys_lds)
      : "vcc", "memory");
}

// rj_find_chain (k_rowjoin.h) for this kernel's call sites, where every lane of the wave is active
__device__ __forceinline__ uint32_t rjf_find_chain(uint32_t keys_lds, uint32_t k, uint32_t& kk, uint32_t h, uint32_t smask) {
  uint32_t addr;
  asm volatile(
      "v_cmpx_gt_u32_e32 vcc, %[kk], %[k]\n\t"
      "s_cbranch_execz 2f\n"
      "1:\n\t"
      "v_add_u32_e32 %[h], 1, %[h]\n\t"
      "v_and_b32_e32 %[h], %[smask], %[h]\n\t"
      "v_lshl_add_u32 %[addr], %[h], 2, %[base]\n\t"
      "ds_read_b32 %[kk], %[addr]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmpx_gt_u32_e32 vcc, %[kk], %[k]\n\t"
      "s_cbranch_execnz 1b\n"
      "2:\n\t"
      RJF_EXEC_ALL
      : [kk] "+v"(kk), [h] "+v"(h), [addr] "=&v"(addr)
      : [k] "v"(k), [smask] "s"(smask), [base] "s"(keys_lds)
      : "vcc", "memory");
  return h;
}

// Home slot of key k: its hash -- or, for a pixel slot without a record (k == 0), a slot of the lane's own (idx): the
// no-op atomics and reads of such slots then never share an address.  Without a select the compiler turns into a branch
// (six scalar instructions of EXEC bookkeeping per slot): codes below 2^31 (every arithmetic but 32-test SSE=OFF) make
// k - 1 negative exactly for k == 0, and the hash of 0 is 0, so the own slot is OR-ed in under that sign mask.
__device__ __forceinline__ uint32_t rjf_home(uint32_t k, uint32_t idx, int shift, uint32_t smask, bool wide) {
  if (wide) return k ? rj_hash(k, shift) : (idx & smask);
  const uint32_t none = (uint32_t)((int32_t)(k - 1u) >> 31);
  return rj_hash(k, shift) | (idx & smask & none);
}

constexpr int rjf_log2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// grid: (workgroups the device holds at once); NT threads, NB = NT * SPT pixel slots >= W, NB <= 4096 (12 bits of x beside
//       the flags); the host guarantees npairs * (H - 26) < 2^31 - 65536.
// dynamic LDS (RJF_LDS_BYTES): keys [S] | 16-bit flag words [S/2 words] | rank counters [NB] | D: the pending row's
//       ranked words [NB - 24 >= W - 26]
#define RJF_LDS_BYTES(NB) ((size_t)6 * pow2_at_least(2 * (NB)) + (size_t)4 * (NB) + (size_t)4 * ((NB) - 24))
// Wave priority by the phase of the row (s_setprio at the phase boundaries; a CU's arbiter serves the higher priority first,
// then the older wave).  Left alone, the eight workgroups of a CU are served oldest first whatever they are doing; with
// the priority RISING through the row -- 0 while the codes arrive, 1 for the insert, 2 for the lookups, 3 from the decide
// phase to the end -- a row that is further along is finished first: 522 -> 502 us per 256 pairs, 89.5 -> 85.1 at 32
// (tools/exp/ab_join_prio.sh; seven hex digits, phase 0 leftmost: 0012233 / 0012333 506, 0011223 506, 0112233 509,
// 0133333 / 0233333 / 1233333 505, 0123222 / 0123210 505 (83.9 at 32 pairs), 0001233 513, 0000333 517, 0333333 513; falling:
// 3321100 511 (83.6 at 32 pairs), 3333210 514).
#ifndef RJF_PRIO_PAT
#define RJF_PRIO_PAT 0x0123333
#endif
#ifndef RJF_NO_PRIO
#define RJF_PRIO(ph) __builtin_amdgcn_s_setprio((RJF_PRIO_PAT >> (4 * (6 - (ph)))) & 3)
#else
#define RJF_PRIO(ph) do { } while (0)
#endif
#ifdef GPC_WGLIFE
// diagnostic build only (tools/exp/join_wg_lives.py): per workgroup start, end (s_memrealtime, 100 MHz), rows | HW_ID | XCC_ID
__device__ unsigned long long g_rjf_wg[3 * 4096];
#endif
template <int SPT, int NT, bool WIDE>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_row_join_fused(RjfArgs ka) {
#ifdef GPC_WGLIFE
  unsigned long long wl_t0;
  unsigned wl_rows = 0u;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wl_t0)::"memory");
#endif
  static_assert(SPT <= 4 && NT * SPT <= 4096, "16-bit flag words hold 12 bits of x");
  static_assert(NT >= 128, "the second wave draws the tickets");
  constexpr int NB = NT * SPT, S = 1 << rjf_log2(2 * NB);  // (a power of two: 2 * NB but for three slots per thread)
  constexpr uint32_t smask = (uint32_t)S - 1u;
  constexpr int hshift = 32 - rjf_log2(S);
  // flags of a table slot: halfwords, two slots per word, x of a right record in the low 12 bits
  constexpr uint32_t F_LDUP = 0x2000u, F_RSEEN = 0x4000u, F_RDUP = 0x8000u, F_XMASK = 0x0FFFu;  // (0x1000: unused since round 4 -- left records need no SEEN flag)
  extern __shared__ __attribute__((aligned(16))) uint32_t rjf_lds[];
  __shared__ uint32_t s_max_key;
  __shared__ int s_tail_cnt;
  __shared__ unsigned s_tail_minx;
  __shared__ int s_sp_l, s_sp_r;   // WIDE: left / right candidates of this row whose code is 0xFFFFFFFF
  __shared__ unsigned s_sp_minx;   //       smallest x among the right ones
  __shared__ uint32_t s_w[NT / 64];
  __shared__ uint32_t s_ticket, s_base;    // the row drawn for this workgroup; supports of the pair's earlier rows
  __shared__ uint32_t s_cnt;               // matches of the row, complete after the rank-count barrier
  uint32_t* const t_key = rjf_lds;                  // [S]    stored key = code + 1, 0 = empty
  uint32_t* const t_w = rjf_lds + S;                // [S/2]  two slots per word: seen / duplicate flags of either side, x of a right record
  uint32_t* const r_cnt = rjf_lds + S + S / 2;      // [NB]   bucket counters -> starts
  uint32_t* const d_words = r_cnt + NB;             // [NB-24] ranked words of the row whose output is pending
  uint32_t* const r_key = t_w;                      // [NB]   matched codes of shared buckets (the flag words are dead after the decide phase)
  // LDS byte offsets for the hand-written loops.  The dynamic block starts where the static variables end (a multiple of 16
  // here): taken from the low half of the flat address instead, every use paid the address-space cast's null check
  // (s_cmp_lg_u32 x, -1 + s_cselect_b32: seven pairs per wave and row).
  static_assert((sizeof(uint32_t) * (12 + NT / 64)) % 16 == 0, "the dynamic block follows the static variables without padding");
  const uint32_t keys_lds = __builtin_amdgcn_groupstaticsize();
  const uint32_t rkey_lds = keys_lds + 4u * (uint32_t)S;

  // this workgroup's shard of pairs, its size and the multiply-high that divides by it
  int f_shard, f_ps, f_sh;
  uint32_t f_magic, f_end, f_g;
  {
    RjfK* a = rjf_args();
    f_shard = (int)(blockIdx.x % (unsigned)a->nshards);
    const int hi = f_shard < a->n_hi ? 0 : 1;
    f_ps = a->ps[hi];
    f_magic = a->ps_magic[hi];
    f_sh = a->ps_sh[hi];
    f_end = (uint32_t)f_ps * (uint32_t)(a->H - 2 * GPC_R);  // tickets that are rows
    // The counter's address goes through an opaque per-lane zero: for an address it can prove uniform the compiler
    // makes ONE atomic per wave and broadcasts the result with v_readfirstlane -- which waits for it on the spot,
    // where the draw for the next row is meant to stay in flight for most of this one.
    if (threadIdx.x == 0) {
      uint32_t opaque0;
      asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
      s_ticket = __hip_atomic_fetch_add(a->tickets + f_shard * RJ_TICKET_STRIDE + opaque0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    f_g = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket);
  }
  // the row whose records are still in D: its pair, ticket row (-1: none), count
  int d_pair = 0, d_t = -1;
  uint32_t d_cnt = 0u;
  unsigned long long d_g0 = 0ull;  // first wave: the granules of the pending row's predecessors
  uint32_t f_nxt = 0u;             // second wave, lane 0: the ticket drawn for the next row

  // The codes of a row: one scalar base per side + a 32-bit lane offset + an immediate per slot.  No branch around a load
  // (the compiler sinks the key arithmetic into it and waits pair by pair), nor a clamp: pixel slots beyond W read into
  // the next row -- the code image is allocated with that slack -- and are masked later (with arithmetic, not a select
  // the compiler could turn into a branch around the load).  They are asked for at the very end of the row before
  // (asking earlier, right behind that row's decide barrier when its ticket is known, measured the same 578-586 us: a
  // wave's vector-memory counter is in order, so the wait for these loads then also waits for the acknowledges of the
  // pending row's record stores issued behind them; tools/exp/k_rowjoin_fused_round4_experiments.h.txt).
  uint32_t ncl[SPT], ncr[SPT];
  // the left row of ticket g (the right one lies H * W codes behind it)
  auto row_of = [&](uint32_t g) -> const uint32_t* {
    const uint32_t q = (f_ps == 1) ? g : (__umulhi(g, f_magic) >> f_sh);
    const int W = ka.W, H = ka.H;
    const int pair = f_shard + ka.nshards * (int)(g - q * (uint32_t)f_ps);
    const long ro = ((long)(pair * 2) * H + (GPC_R + (int)q)) * W;
    return ka.codes + ro;
  };
  auto issue_loads = [&](const uint32_t* rl, uint32_t (&cl)[SPT], uint32_t (&cr)[SPT]) {
    const uint32_t* rr_ = rl + (long)ka.H * ka.W;
    uint32_t lo = threadIdx.x * 4u;
    RJF_OPAQUE_V(lo);
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      cl[j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(rl) + lo + (uint32_t)(j * NT * 4));
      cr[j] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(rr_) + lo + (uint32_t)(j * NT * 4));
    }
  };
  if (f_g < f_end) issue_loads(row_of(f_g), ncl, ncr);

#pragma unroll 1
  while (f_g < f_end) {
    int tid = threadIdx.x;
    RJF_OPAQUE_V(tid);  // (per row: what is derived from it is made again instead of being kept -- and spilled -- across rows)
    const int lane = tid & 63;
    // ---- ticket -> (pair, row); the row's codes are in flight already
    const uint32_t q = (f_ps == 1) ? f_g : (__umulhi(f_g, f_magic) >> f_sh);
    const int f_t = (int)q, y = GPC_R + f_t;
    int pair, W;
    uint32_t nspl = 0u, nspr = 0u;  // WIDE: bit j = pixel slot j is a candidate whose code is 0xFFFFFFFF
    bool tail_row;
    int csh;
    {
      W = ka.W;
      const int H = ka.H;
      pair = f_shard + ka.nshards * (int)(f_g - q * (uint32_t)f_ps);
      const long ro = ((long)(pair * 2) * H + y) * W;
      // (scalar loads: the statistics were written by the launches before this one)
      typedef const int32_t __attribute__((address_space(4))) kint;
      kint* st = (kint*)ka.img_stats;
      // Tail quirks of the reference's merge scan concern only the largest right code of the last
      // right row that has candidates: it matches iff it occurs exactly TWICE on the right (then with the first of the
      // two in mask order) and once on the left.
      tail_row = (y == st[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_LASTROW]);
      // The NB rank buckets divide [0, 2^bits) with bits from the OR of every code k_hash computed for the left image:
      // bits the forest leaves constant cost no resolution.
      const uint32_t span = (uint32_t)st[(pair * 2) * GPC_STAT_STRIDE + GPC_STAT_CODEOR];
      csh = (span ? 32 - __builtin_clz(span) : 0) - (rjf_log2(NB + 1) - 1);  // bits of the span beyond those of a bucket index (NB buckets, or the power of two below)
      if (csh < 0) csh = 0;
      if (WIDE) {
        const uint8_t* cand = ka.cand;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
          const int x = j * NT + tid;
          // the hash kernel's candidate rule (k_hash.h): candidate byte set, inside the margin (the row is)
          const bool inm = x >= GPC_R && x < W - GPC_R;
          if (inm && ncl[j] == RJ_EMPTY && cand[ro + x]) nspl |= 1u << j;
          if (inm && ncr[j] == RJ_EMPTY && cand[ro + (long)H * W + x]) nspr |= 1u << j;
        }
      }
    }
    RJ_STAMP_INIT();
    RJF_PRIO(0);
    // ---- 0. this row's keys; key table and rank counters cleared (16-byte stores)
    uint32_t kl[SPT], kr[SPT];  // stored key = code + 1 (0 = no record in this pixel slot)
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const uint32_t inm = (uint32_t)((j * NT + tid - W) >> 31);   // all ones for a pixel slot inside the row
      kl[j] = (ncl[j] + 1u) & inm;   // (a non-candidate pixel holds GPC_NOCAND = 0xFFFFFFFF: + 1 = 0 as well)
      kr[j] = (ncr[j] + 1u) & inm;
#ifdef RJ_DBG_EMPTY   // experiment: every pixel slot without a record -- what a row costs before it holds anything
      kl[j] = kr[j] = 0u * (ncl[j] + ncr[j]);
#endif
    }
    const uint32_t spl = WIDE ? nspl : 0u, spr = WIDE ? nspr : 0u;
    {
      uint32_t z0;  // (made here: as a plain constant the zeros are kept in four registers across the row)
      asm volatile("v_mov_b32 %0, 0" : "=v"(z0));
      const uint4 zero = make_uint4(z0, z0, z0, z0);
      uint4* zk = reinterpret_cast<uint4*>(t_key);
      constexpr int NZK = S / 4, NZC = NB / 4;  // 16-byte words of the key table / of the counters
#pragma unroll
      for (int i = 0; i < (NZK + NT - 1) / NT; ++i)
        if (NZK % NT == 0 || tid + i * NT < NZK) zk[tid + i * NT] = zero;
      uint4* zc = reinterpret_cast<uint4*>(r_cnt);
#pragma unroll
      for (int i = 0; i < (NZC + NT - 1) / NT; ++i)
        if (NZC % NT == 0 || tid + i * NT < NZC) zc[tid + i * NT] = zero;
      if (tid == 0) {
        s_max_key = z0;
        s_tail_cnt = (int)z0;
        s_tail_minx = ~z0;
        s_cnt = z0;
        if (WIDE) {
          s_sp_l = (int)z0;
          s_sp_r = (int)z0;
          s_sp_minx = ~z0;
        }
      }
    }
    __syncthreads();  // B0: table and counters clear; the previous row's walk is over everywhere (its codes lie in the flag words)
    RJ_STAMP(0);
    RJF_PRIO(1);

    // ---- 1. build the ordered table from the left codes; the flag words are cleared meanwhile (the inserts touch keys only)
    uint32_t h0l[SPT];
    bool ldup[SPT];              // the code of left pixel slot j occurs at least twice on the left (lane masks, not bits of a register:
                                 // packing them cost a select and an OR per slot, testing them an AND and a compare)
    uint32_t carried[SPT];       // a displaced key this lane saw meet its copy (0: none -- the rule)
#pragma unroll
    for (int j = 0; j < SPT; ++j) carried[j] = 0u;
    {
      uint32_t old[SPT];
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
        // a pixel slot without a record inserts key 0 (a no-op) -- into a slot of its own: the atomics of
        // lanes that share an address are served one after the other
        h0l[j] = rjf_home(kl[j], (uint32_t)(j * NT + tid), hshift, smask, WIDE);
        old[j] = atomicMax(&t_key[h0l[j]], kl[j]);
      }
      {
        uint32_t z0;
        asm volatile("v_mov_b32 %0, 0" : "=v"(z0));
        uint4* zf = reinterpret_cast<uint4*>(t_w);
        constexpr int NZF = S / 2 / 4;
#pragma unroll
        for (int i = 0; i < (NZF + NT - 1) / NT; ++i)
          if (NZF % NT == 0 || tid + i * NT < NZF) zf[tid + i * NT] = make_uint4(z0, z0, z0, z0);
      }
      // A code that occurs twice among the left records is noticed HERE, by the lane whose carried key meets a copy of
      // itself (the probe returns the key it carries): total records - distinct keys such events, at least one per repeated
      // code, each seen by one lane.  Nearly always that lane carries its OWN key (it never displaced anything): it sets
      // the left-duplicate flag on its own slot after the lookup, and no left record needs a returning mark to find out
      // (round 3: four returning ds_or per wave and row).  A lane that carried a DISPLACED key into its copy (two copies
      // of a code and a larger code racing over the same slots) remembers that key and marks its slot after the barrier.
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
        uint32_t cur = kl[j], o = old[j];
        RJF_ASSERT_EXEC_ALL();
        rjf_insert_chain(keys_lds, cur, o, h0l[j], smask);
        const bool met = kl[j] && o == cur;
        ldup[j] = met && cur == kl[j];
        if (met && cur != kl[j]) carried[j] = cur;
      }
    }
    if (tail_row) {  // the largest right key of this row (block-uniform branch)
      uint32_t max_k = 0;
#pragma unroll
      for (int j = 0; j < SPT; ++j) max_k = max(max_k, kr[j]);
      max_k = wave_max_u32(max_k);
      if (lane == 0 && max_k) atomicMax(&s_max_key, max_k);
    }
    if (WIDE) {  // the code without a key: count its records on either side
      if (__ballot(spl != 0u) | __ballot(spr != 0u)) {
        if (spl) atomicAdd(&s_sp_l, __popc(spl));
        if (spr) {
          atomicAdd(&s_sp_r, __popc(spr));
          atomicMin(&s_sp_minx, (unsigned)((__ffs((int)spr) - 1) * NT + tid));
        }
      }
    }
    if (tid == 64) {  // the draw for this workgroup's NEXT row: in flight over the lookup and decide phases
      uint32_t opaque0;
      asm volatile("v_mov_b32 %0, 0" : "=v"(opaque0));
      f_nxt = __hip_atomic_fetch_add(rjf_args()->tickets + f_shard * RJ_TICKET_STRIDE + opaque0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();  // B1
    RJ_STAMP(1);
    RJF_PRIO(2);
    // The first wave asks for the granules of the PENDING row's predecessors here and looks at the answer after the
    // lookup phase (the wave has no other vector-memory operation in flight then: its counter is in order, and a wait
    // for an older load would wait for this one too).  Who does what is spread over the waves for the same reason: the
    // first wave asks and publishes, the second draws the tickets.  When to ask was measured (256 pairs, first windows
    // that still held a row that had not published its count, each costing a blocking poll): during the pending row's
    // own walk 33 %; at the top of the next row 18 %; here, after the next row's insert phase, 4 %.
    if (d_t >= 0 && tid < 64) {
      RjfK* a = rjf_args();
      d_g0 = rjf_lookback_ask(a->status + (long)d_pair * (ka.H - 2 * GPC_R) + d_t, d_t, lane, a->epoch);
    }

    // ---- 2. every record finds its code's slot (read-only) and marks it.  The marks of a side go out together (one
    //      LDS round trip for SPT returning atomics): a record without a slot ORs 0 into wherever its walk stopped.
    //      Flag word of slot h: halfword (h & 1) of t_w[h >> 1].
    uint32_t hl[SPT];
    {
      uint32_t h0r[SPT], f0l[SPT], f0r[SPT];
#pragma unroll
      for (int j = 0; j < SPT; ++j) {  // first probes of all records together
        h0r[j] = rjf_home(kr[j], (uint32_t)(j * NT + tid), hshift, smask, WIDE);
        f0l[j] = t_key[h0l[j]];
        f0r[j] = t_key[h0r[j]];
      }
      uint32_t seen[SPT], mv[SPT];
      // What a pixel slot searches for: its key -- or, without a record, 0xFFFFFFFF (key | sign mask of key - 1, one OR):
      // larger than every stored key (codes below 2^31: every arithmetic but WIDE), so its walk never starts, and equal to
      // none, so it never finds.  (Searching for 0 with a first probe forced to 0 cost a compare, a select and a wait
      // state per chain.)
      uint32_t ql[SPT], qr[SPT];
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
        ql[j] = WIDE ? kl[j] : (kl[j] | (uint32_t)((int32_t)(kl[j] - 1u) >> 31));
        qr[j] = WIDE ? kr[j] : (kr[j] | (uint32_t)((int32_t)(kr[j] - 1u) >> 31));
      }
      {
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
          uint32_t kk = WIDE ? (kl[j] ? f0l[j] : 0u) : f0l[j];
          RJF_ASSERT_EXEC_ALL();
          hl[j] = rjf_find_chain(keys_lds, ql[j], kk, h0l[j], smask);  // a left code is always found
        }
      }
#pragma unroll
      for (int j = 0; j < SPT; ++j)  // a repeated left code: the flag goes onto the code's slot (every copy of it reads that slot)
        if (ldup[j]) atomicOr(&t_w[hl[j] >> 1], F_LDUP << ((hl[j] & 1u) << 4));
      {  // the rare displaced key that met its copy: find its slot like any lookup and flag it (wave-uniform skip otherwise)
        uint32_t anyc = 0u;
#pragma unroll
        for (int j = 0; j < SPT; ++j) anyc |= carried[j];
        if (__ballot(anyc != 0u)) {
#pragma unroll
          for (int j = 0; j < SPT; ++j) {
            const uint32_t hc0 = rj_hash(carried[j], hshift);
            uint32_t kk = carried[j] ? t_key[hc0] : 0u;
            RJF_ASSERT_EXEC_ALL();
            const uint32_t hc = rjf_find_chain(keys_lds, carried[j], kk, hc0, smask);
            if (carried[j]) atomicOr(&t_w[hc >> 1], F_LDUP << ((hc & 1u) << 4));
#ifdef RJ_DBG_COUNT
            if (carried[j]) atomicAdd(rjf_args()->err + 3, 1);  // [3] displaced keys that met their copy
#endif
          }
        }
      }
      uint32_t hr[SPT];
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
        uint32_t kk = WIDE ? (kr[j] ? f0r[j] : 0u) : f0r[j];
        RJF_ASSERT_EXEC_ALL();
        hr[j] = rjf_find_chain(keys_lds, qr[j], kk, h0r[j], smask);
        // x goes into the zeroed low bits with the same atomic: several writers only when the code is not unique on
        // the right, and then x is not used
        const bool hit = WIDE ? (kr[j] && kk == kr[j]) : (kk == qr[j]);
        mv[j] = (hit ? (F_RSEEN | (uint32_t)(j * NT + tid)) : 0u) << ((hr[j] & 1u) << 4);
      }
#pragma unroll
      for (int j = 0; j < SPT; ++j) seen[j] = atomicOr(&t_w[hr[j] >> 1], mv[j]);
#pragma unroll
      for (int j = 0; j < SPT; ++j) {  // a second right record of this code (only the SEEN bit of the record's own halfword counts)
        const uint32_t again = seen[j] & mv[j] & (F_RSEEN | (F_RSEEN << 16));
        if (again) atomicOr(&t_w[hr[j] >> 1], again << 1);  // F_RDUP = F_RSEEN << 1
      }
    }
    // the key the tail rule applies to; none when the row's largest right code is the key-less 0xFFFFFFFF
    uint32_t tail_key = 0u;
    bool tail_sp = false;
    if (tail_row) {  // block-uniform
      tail_sp = WIDE && s_sp_r > 0;
      tail_key = tail_sp ? 0u : s_max_key;
#pragma unroll
      for (int j = 0; j < SPT; ++j)
        if (kr[j] && kr[j] == tail_key) {
          atomicAdd(&s_tail_cnt, 1);
          atomicMin(&s_tail_minx, (unsigned)(j * NT + tid));
        }
    }
    __syncthreads();  // B2
    RJ_STAMP(2);
    RJF_PRIO(3);
    if (d_t >= 0 && tid < 64) {  // the pending row's place: supports of the pair's rows before it
      RjfK* a = rjf_args();
      unsigned long long* d_st = a->status + (long)d_pair * (ka.H - 2 * GPC_R) + d_t;
      const uint32_t ep = a->epoch;
      const uint32_t base = rjf_lookback(d_st, d_t, lane, ep, d_g0, a->err);
      if (lane == 0) {
        if (d_t > 0) __hip_atomic_store(d_st, rjf_granule(ep, RJ_ST_PREFIX, base + d_cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_base = base;
      }
    }

    // ---- 3. decide every left candidate, and count the matches per rank bucket right away (the counters have their
    //      own LDS and were cleared with the table: no barrier between deciding and counting)
    bool ok[SPT];  // pixel slot j is a match (lane masks: as bits of a register every use was an AND and a compare)
    uint32_t xr[SPT];
    uint32_t wc = 0u;   // matches of this wave
    const int disp_high = ka.disp_high, apply_filter = ka.apply_filter;
#ifndef RJF_OLD_DECIDE
    if (!WIDE && !tail_row) {
      // Every row but the one the tail rule applies to (one per pair), straight-line: the four flag halfwords are asked for
      // together -- a slot without a record has an address of its own and reads it for nothing -- and a match is
      // (flags & (LDUP | RSEEN | RDUP)) == RSEEN, inside the disparity range, of a slot that holds a record: compares and
      // mask arithmetic, ~10 instructions per slot.  Written with a branch per condition it was ~36, each slot's read
      // waited for by itself inside its own EXEC region, and the scalar unit spent more instructions merging lane masks
      // than the vector unit comparing.
      uint32_t w[SPT];
#pragma unroll
      for (int j = 0; j < SPT; ++j) w[j] = reinterpret_cast<const uint16_t*>(t_w)[hl[j]];
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
        xr[j] = w[j] & F_XMASK;
        uint32_t has = kl[j];
        asm volatile("" : "+v"(has));  // (compared afresh: the lane masks of "holds a record" kept from the lookup phase cost spilled scalar pairs)
        const bool hit = (has != 0u) & ((w[j] & (F_LDUP | F_RSEEN | F_RDUP)) == F_RSEEN);
        const bool near = (apply_filter == 0) | ((int)__usad((uint32_t)(j * NT + tid), xr[j], 0u) <= disp_high);
        const bool good = hit & near;
        wc += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(good));
        ok[j] = good;
      }
    } else
#endif
    {
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
        xr[j] = 0u;
        bool good = false;
        if (kl[j]) {
          // (the slot's own halfword with a 16-bit read: one address operation on the way to the flags where word + shift
          // took five: 548-550 -> 545-547 us)
          const uint32_t w = reinterpret_cast<const uint16_t*>(t_w)[hl[j]];
          const bool tail = tail_row && kl[j] == tail_key;
          good = !(w & F_LDUP) && (tail ? (s_tail_cnt == 2) : ((w & (F_RSEEN | F_RDUP)) == F_RSEEN));
          xr[j] = tail ? s_tail_minx : (w & F_XMASK);
        } else if (WIDE && ((spl >> j) & 1u)) {
          good = (s_sp_l == 1) && (s_sp_r == (tail_sp ? 2 : 1));
          xr[j] = s_sp_minx;
        }
        if (good && apply_filter) good = abs((int)(j * NT + tid) - (int)xr[j]) <= disp_high;
        ok[j] = good;
        wc += (uint32_t)__popcll(__ballot(good));
      }
    }
    uint32_t rb[SPT], rs[SPT];  // rank bucket / arrival order in it (later: the bucket's first rank / order | matches in the bucket << 16)
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      rb[j] = (kl[j] - 1u) >> csh;
      rs[j] = 0u;
      if (ok[j]) rs[j] = atomicAdd(&r_cnt[rb[j]], 1u);
    }
    if (lane == 0 && wc) atomicAdd(&s_cnt, wc);  // the row's support count, one LDS add per wave that has a match
    if (tid == 64) s_ticket = f_nxt;  // (waits for the draw made at the top of the insert phase)
    __syncthreads();  // B3: every match is counted; the flag words are dead from here on (their LDS takes the codes of shared buckets)
    RJ_STAMP(3);
    RJF_PRIO(4);
    // this row's count goes out at once (later rows of the pair may be waiting for it)
    const uint32_t f_cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_cnt);
    if (tid == 0) {
      RjfK* a = rjf_args();
      __hip_atomic_store(a->status + (long)pair * (ka.H - 2 * GPC_R) + f_t, rjf_granule(a->epoch, f_t == 0 ? RJ_ST_PREFIX : RJ_ST_AGG, f_cnt),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t f_gn = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket);  // the next row of this workgroup

    // ---- 4. output position = rank of the code among the row's matches (counting rank on NB buckets).
    // How many matches share the bucket is read BEFORE the scan turns the counters into starts (every such read is done
    // before block_exscan's first barrier, the scan's stores come after it): a match alone in its bucket needs neither a
    // place in r_key nor the walk -- its rank is its bucket's start.
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (ok[j]) rs[j] |= r_cnt[rb[j]] << 16;
    if (d_t >= 0) {  // the pending row's records leave (D is rewritten by this row's walk, three barriers on)
      const uint32_t base = s_base;
      rjf_emit_row<NT>(d_words, d_cnt, base, d_pair, GPC_R + d_t, tid);
      if (tid == 0 && d_t == ka.H - 2 * GPC_R - 1) rjf_finish_pair(d_pair, base + d_cnt);  // the pair's last row
    }
    RJ_STAMP(4);
    RJF_PRIO(5);
    block_exscan<SPT, NT, false>(r_cnt, s_w, tid);  // r_cnt[b] = first rank of bucket b     (B4, B5)
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (ok[j]) {
        rb[j] = r_cnt[rb[j]];  // the bucket's first rank (the bucket index is not needed again)
        if ((rs[j] >> 16) > 1u) r_key[rb[j] + (rs[j] & 0xFFFFu)] = kl[j] - 1u;  // the code (WIDE: the key-less 0xFFFFFFFF ranks last)
      }
    __syncthreads();  // B6
    RJ_STAMP(5);
    RJF_PRIO(6);
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const bool m = ok[j];
      RJF_ASSERT_EXEC_ALL();
      // (rs is 0 for a slot that is no match: its population field says "nothing to walk" by itself)
      const uint32_t rank = rjf_walk(rkey_lds, rb[j], rs[j] >> 16, kl[j] - 1u);
      if (m) d_words[rank] = (uint32_t)(j * NT + tid) | (xr[j] << 16);  // the ranked words wait in D for the row's place in the output
    }
    // this row is the pending one now; the next row's clear touches neither D nor the flag words
    d_pair = pair;
    d_t = f_t;
    d_cnt = f_cnt;
    f_g = f_gn;
    // (the row's address worked out right behind the decide barrier, where the ticket is known, instead of here: 531 vs 525 us --
    // it delays the count's publication and the pending row's records, which the whole pair's look-backs wait for)
    if (f_g < f_end) issue_loads(row_of(f_g), ncl, ncr);
#ifdef GPC_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    RJ_STAMP(6);
    RJ_STAMP_FLUSH();
#ifdef GPC_WGLIFE
    ++wl_rows;
#endif
  }  // rows of this workgroup

  {
    RjfK* a = rjf_args();
    const int nrows = a->H - 2 * GPC_R;
    // every workgroup of the shard has drawn its last ticket: the counter starts over for the next launch
    const uint32_t f_last = f_end + (gridDim.x - (unsigned)f_shard + (unsigned)a->nshards - 1u) / (unsigned)a->nshards - 1u;
    if (threadIdx.x == 0 && f_g == f_last)
      __hip_atomic_store(a->tickets + f_shard * RJ_TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d_t >= 0) {  // the last row of this workgroup is still pending
      if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned long long* d_st = a->status + (long)d_pair * nrows + d_t;
        const uint32_t ep = a->epoch;
        const uint32_t base = rjf_lookback(d_st, d_t, lane, ep, rjf_lookback_ask(d_st, d_t, lane, ep), a->err);
        if (lane == 0) {
          if (d_t > 0) __hip_atomic_store(d_st, rjf_granule(ep, RJ_ST_PREFIX, base + d_cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_base = base;
        }
      }
      __syncthreads();  // (also: the last row's walk has filled D)
      const uint32_t base = s_base;
      rjf_emit_row<NT>(d_words, d_cnt, base, d_pair, GPC_R + d_t, threadIdx.x);
      if (threadIdx.x == 0 && d_t == nrows - 1) rjf_finish_pair(d_pair, base + d_cnt);
    }
  }
#ifdef GPC_WGLIFE
  if (threadIdx.x == 0 && blockIdx.x < 4096u) {
    unsigned long long t1_;
    unsigned hw_, xcc_;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_)::"memory");
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
    g_rjf_wg[3 * blockIdx.x] = wl_t0;
    g_rjf_wg[3 * blockIdx.x + 1] = t1_;
    g_rjf_wg[3 * blockIdx.x + 2] = ((unsigned long long)wl_rows << 40) | ((unsigned long long)(xcc_ & 0xFu) << 32) | hw_;
  }
#endif
}

}  // namespace gpc
