// k_hashtable.h -- the reference's optional hash-table matcher (settings.useHashtable_ == true).
//
// Replaces the Hashmatch branch of Forest::depthPriorFast (inference.hpp:204-225) and
// ndb::Hashmatch / OrderedLinkedList (hashmatch.hpp:91-131, 162-197, 252-263): 214673 buckets
// `state % 214673`, per bucket an ordered list that keeps only the FIRST 10 inserted elements
// (all source descriptors are inserted before all targets, each in mask order), and a walk of
// every list that emits (source, target) pairs of equal states under the reference's pair /
// triplet rules.  Output order = bucket index, then list order.
//
// On the GPU the records of both images (built in insertion order by k_g_rowcount /
// k_g_build, k_global.h) get their bucket id, a stable LSD radix sort (3 x 8 bits, the passes
// of k_global.h) groups them by bucket without disturbing insertion order, and k_ht_pairs replays
// the capped ordered insert and the list walk per bucket: a workgroup stages a tile of sorted
// records (+ a 9-record halo) in LDS with every thread gathering its own record, each record
// computes its position in its bucket's ordered list by counting (stable rank among the first 10),
// and the thread of the bucket's first record walks the <= 10 ordered entries.
// This mode returns a slightly different (smaller) match set than the sort matcher, exactly
// as in the reference.
#pragma once
#include "gpc_device.h"
#include "k_global.h"

namespace gpc {

#define HM_CAP 10

// skey[i] = state % 214673, sval[i] = i  for the N records (code, side<<31|k) in insertion order;
// rec[i] = (code, kv) interleaved, so that the bucket replay fetches a record with ONE 8-byte gather
__global__ __launch_bounds__(256) void k_ht_bucket_ids(const uint32_t* __restrict__ codes0,
                                                       const uint32_t* __restrict__ kv0,
                                                       const int32_t* __restrict__ gmisc, GpcDivW wd, int epipolar,
                                                       uint32_t* __restrict__ skey, uint32_t* __restrict__ sval,
                                                       uint2* __restrict__ rec, GpcBatchStrides bs) {
  codes0 += blockIdx.y * bs.recs;
  kv0 += blockIdx.y * bs.recs;
  skey += blockIdx.y * bs.recs;
  sval += blockIdx.y * bs.recs;
  rec += blockIdx.y * bs.recs;
  gmisc += blockIdx.y * GM_STRIDE;
  const int N = gmisc[GM_N];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  unsigned long long state = codes0[i];
  if (epipolar) state |= (unsigned long long)(uint32_t)divw(kv0[i] & 0x7FFFFFFFu, wd) << 32;
  skey[i] = (uint32_t)(state % HM_BUCKETS);
  sval[i] = (uint32_t)i;
  rec[i] = make_uint2(codes0[i], kv0[i]);
}

#ifndef HP_RPT
#define HP_RPT 2                        // sorted records per thread (measured: 1 -> 692, 2 -> 536, 4 -> 600, 8 -> 1220 us per 32 pairs)
#endif
#define HP_THREADS 256
#define HP_TILE (HP_THREADS * HP_RPT)   // sorted records per workgroup
#define HP_SPAN (HP_TILE + HM_CAP - 1)  // a bucket that starts in the tile keeps at most 9 records beyond it

// OrderedLinkedList::getDuplicates (hashmatch.hpp:162-197) on an ordered list of n <= 10 entries,
// given as bit masks so that the walk itself touches no memory:
//   eq bit u = entries u and u+1 hold the same state,  df bit u = they come from different images.
// Returns the mask of u for which the pair (entry u = source, entry u+1 = target) is reported.
// The reference's walk, position by position:
//     i = 0; while (i < n) { p = i++; if (i < n && eq[p]) { if (df[p]) { if (i + 1 >= n || !eq[i]) report p;
//                                                                       if (i + 1 < n && i + 2 >= n) break; }   // "last triplet"
//                                                          else if (i + 1 < n && df[i]) ++i; } }               // skip a false pair
// without the loop: a position p is jumped over iff p - 1 was visited, eq[p-1], !df[p-1], df[p] and p + 1 < n (a
// recurrence on one bit per position); a visited p with eq[p] & df[p] reports unless eq[p+1] with p + 2 < n; such a p
// at n - 3 ends the walk before n - 2.  Checked against the loop for every (n, eq, df) on the host.
__device__ __forceinline__ uint32_t ht_walk_bits(uint32_t eq, uint32_t df, int n) {
  if (n < 2) return 0u;
  const uint32_t valid1 = (1u << (n - 1)) - 1u;  // p + 1 < n
  const uint32_t valid2 = (1u << (n - 2)) - 1u;  // p + 2 < n
  eq &= valid1;
  df &= valid1;
  const uint32_t jump = (eq << 1) & ~(df << 1) & df;
  uint32_t v = 1u;
#pragma unroll
  for (int p = 1; p < HM_CAP; ++p) v |= (~((v >> (p - 1)) & (jump >> p)) & 1u) << p;
  const uint32_t hit = v & eq & df;
  uint32_t emit = hit & ~((eq >> 1) & valid2);
  if (n >= 3 && ((hit >> (n - 3)) & 1u)) emit &= ~(1u << (n - 2));
  return emit;
}

// pass A (!WRITE): gathers the records (sval = index into rec), leaves them in sorted order in
//   scode / skv for pass B, and counts the pairs per workgroup -> blkcnt (scanned by k_g_scan);
// pass B (WRITE): streams scode / skv and writes the pairs themselves, in bucket order
template <bool WRITE>
__global__ __launch_bounds__(HP_THREADS) void k_ht_pairs(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sval,
                                                      const uint2* __restrict__ rec, uint32_t* __restrict__ scode,
                                                      uint32_t* __restrict__ skv,
                                                      const int32_t* __restrict__ gmisc, GpcDivW wd, int epipolar,
                                                      int disp_high, int vtol, int apply_filter,
                                                      int32_t* __restrict__ blkcnt, int mode, void* __restrict__ out, int cap,
                                                      int32_t* __restrict__ count_out, const int32_t* __restrict__ stats,
                                                      int32_t* __restrict__ ncand_out, GpcBatchStrides bs) {
  __shared__ uint32_t s_key[HP_SPAN];             // bucket id, 0xFFFFFFFF beyond the last record
  __shared__ unsigned long long s_st[HP_SPAN];    // state, insertion order
  __shared__ uint32_t s_kv[HP_SPAN];
  __shared__ unsigned long long s_sst[HP_SPAN];   // state, ordered within the bucket (first 10 insertions only)
  __shared__ uint32_t s_skv[HP_SPAN];
  __shared__ uint32_t s_w[HP_THREADS / 64];
  __shared__ uint32_t s_prev;
  skey += blockIdx.y * bs.recs;
  sval += blockIdx.y * bs.recs;
  rec += blockIdx.y * bs.recs;
  scode += blockIdx.y * bs.recs;
  skv += blockIdx.y * bs.recs;
  blkcnt += blockIdx.y * bs.blk;
  gmisc += blockIdx.y * GM_STRIDE;
  stats += blockIdx.y * 2 * GPC_STAT_STRIDE;
  if (WRITE) {
    out = reinterpret_cast<char*>(out) + blockIdx.y * bs.out;
    count_out += blockIdx.y;
    if (ncand_out) ncand_out += 2 * blockIdx.y;
  }
  const int N = gmisc[GM_N];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * HP_TILE;

  // ---- stage the tile.  Pass A: every thread gathers its own record(s) through the sorted index;
  //      pass B: streams what pass A left behind (already ordered within each bucket)
  for (int i = tid; i < HP_SPAN; i += HP_THREADS) {
    const int j = j0 + i;
    uint32_t key = 0xFFFFFFFFu, kv = 0u;
    unsigned long long st = 0ull;
    if (j < N) {
      key = skey[j];
      if (WRITE) {
        st = scode[j];
        kv = skv[j];
      } else {
        const uint2 rc = rec[sval[j]];
        st = rc.x;
        kv = rc.y;
      }
      if (epipolar) st |= (unsigned long long)(uint32_t)divw(kv & 0x7FFFFFFFu, wd) << 32;
    }
    s_key[i] = key;
    if (WRITE) {
      s_sst[i] = st;
      s_skv[i] = kv;
    } else {
      s_st[i] = st;
      s_kv[i] = kv;
    }
  }
  if (tid == 0) s_prev = (j0 > 0 && j0 - 1 < N) ? skey[j0 - 1] : 0xFFFFFFFEu;
  __syncthreads();

  // ---- OrderedLinkedList::insert for every bucket that starts in this tile: a full list (10) drops
  //      the value, otherwise it goes behind every element <= it  ==  stable order by state of the
  //      first 10 insertions; each record counts its own place (pass A only; the ordered records
  //      also go to scode / skv for pass B)
  if (!WRITE) {
    for (int i = tid; i < HP_SPAN; i += HP_THREADS) {
      const uint32_t key = s_key[i];
      if (key == 0xFFFFFFFFu) continue;
      uint32_t back = 0u;  // same-bucket records right before this one (statically indexed, pipelined reads)
#pragma unroll
      for (int u = 1; u <= HM_CAP; ++u) {
        const bool in = i - u >= 0;
        back |= (in && s_key[in ? i - u : 0] == key) ? (1u << (u - 1)) : 0u;
      }
      const int t = __builtin_ctz(~back);          // insertion index within the bucket
      if (t >= HM_CAP) continue;                   // dropped: the list was full
      const int h = i - t;                         // the bucket's first record
      if (h == 0 && s_prev == key) continue;       // bucket of the previous tile (handled there, in its halo)
      if (h >= HP_TILE) continue;                  // bucket of the next tile
      const unsigned long long st = s_st[i];
      int r = 0;
#pragma unroll
      for (int u = 0; u < HM_CAP; ++u) {  // sorted by bucket: the bucket's records are contiguous from h on
        const int q = h + u;
        const bool in = q < HP_SPAN;
        const unsigned long long o = s_st[in ? q : 0];
        const bool same = in && s_key[in ? q : 0] == key;
        r += (same && (o < st || (o == st && u < t))) ? 1 : 0;
      }
      const uint32_t kv = s_kv[i];
      s_sst[h + r] = st;
      s_skv[h + r] = kv;
      scode[j0 + h + r] = (uint32_t)st;
      skv[j0 + h + r] = kv;
    }
    __syncthreads();
  }

  // ---- the thread that owns a bucket's first record walks its ordered list: the <= 10 entries are
  //      fetched with statically indexed (pipelined) LDS reads, the walk runs on bit masks.
  //      A thread owns HP_RPT consecutive positions, so thread order == bucket order.
  uint32_t em[HP_RPT];
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < HP_RPT; ++k) {
    const int p = tid * HP_RPT + k;
    uint32_t emit = 0u;
    const uint32_t key = s_key[p];
    const bool head = key != 0xFFFFFFFFu && (p == 0 ? s_prev != key : s_key[p - 1] != key);
    if (head) {
      uint32_t same = 0u;
      unsigned long long e[HM_CAP];
      uint32_t q[HM_CAP];
#pragma unroll
      for (int u = 0; u < HM_CAP; ++u) {
        const bool in = p + u < HP_SPAN;
        same |= (in && s_key[in ? p + u : 0] == key) ? (1u << u) : 0u;
        e[u] = s_sst[in ? p + u : 0];
        q[u] = s_skv[in ? p + u : 0];
      }
      const int n = __builtin_ctz(~same);  // leading entries of this bucket, at most 10 (bit 10 of ~same is set)
      uint32_t eq = 0u, df = 0u;
#pragma unroll
      for (int u = 0; u + 1 < HM_CAP; ++u) {
        eq |= (e[u] == e[u + 1]) ? (1u << u) : 0u;
        df |= ((q[u] ^ q[u + 1]) >> 31) << u;
      }
      emit = ht_walk_bits(eq, df, n);
      if (apply_filter) {
        uint32_t todo = emit;
        while (todo) {  // the reported pairs only (mostly one)
          const int u = __builtin_ctz(todo);
          todo &= todo - 1u;
          const uint32_t a = s_skv[p + u] & 0x7FFFFFFFu, b = s_skv[p + u + 1] & 0x7FFFFFFFu;
          const int ya = divw(a, wd), yb = divw(b, wd);
          const int xa = (int)a - ya * wd.W, xb = (int)b - yb * wd.W;
          if (!(abs(ya - yb) <= vtol && abs(xa - xb) <= disp_high)) emit &= ~(1u << u);
        }
      }
    }
    em[k] = emit;
    cnt += __popc(emit);
  }
  const uint32_t incl = wave_incl_scan((uint32_t)cnt);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = incl - (uint32_t)cnt;
  uint32_t total = 0;
  for (int w = 0; w < HP_THREADS / 64; ++w) {
    if (w < wave) base += s_w[w];
    total += s_w[w];
  }
  if (!WRITE) {
    if (tid == 0) blkcnt[blockIdx.x] = (int32_t)total;
    return;
  }
  const int off = blkcnt[blockIdx.x];  // exclusive prefix over the pair's workgroups (k_g_scan)
  int pos = off + (int)base;
#pragma unroll
  for (int k = 0; k < HP_RPT; ++k) {
    uint32_t emit = em[k];
    while (emit) {
      const int u = __builtin_ctz(emit);
      emit &= emit - 1u;
      if (pos < cap) {
        const int p = tid * HP_RPT + k + u;
        const uint32_t a = s_skv[p] & 0x7FFFFFFFu, b = s_skv[p + 1] & 0x7FFFFFFFu;
        const int ya = divw(a, wd), yb = divw(b, wd);
        const int4 m = make_int4((int)a - ya * wd.W, ya, (int)b - yb * wd.W, yb);
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
  }
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    *count_out = off + (int)total;
    if (ncand_out) {
      ncand_out[0] = stats[GPC_STAT_NCAND];
      ncand_out[1] = stats[GPC_STAT_STRIDE + GPC_STAT_NCAND];
    }
  }
}

}  // namespace gpc
