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

#define HM_BUCKETS 214673u
#define HM_CAP 10

// skey[i] = state % 214673, sval[i] = i  for the N records (code, side<<31|k) in insertion order;
// rec[i] = (code, kv) interleaved, so that the bucket replay fetches a record with ONE 8-byte gather
__global__ __launch_bounds__(256) void k_ht_bucket_ids(const uint32_t* __restrict__ codes0,
                                                       const uint32_t* __restrict__ kv0,
                                                       const int32_t* __restrict__ gmisc, int W, int epipolar,
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
  if (epipolar) state |= (unsigned long long)((kv0[i] & 0x7FFFFFFFu) / (uint32_t)W) << 32;
  skey[i] = (uint32_t)(state % HM_BUCKETS);
  sval[i] = (uint32_t)i;
  rec[i] = make_uint2(codes0[i], kv0[i]);
}

#define HP_TILE 256
#define HP_SPAN (HP_TILE + HM_CAP - 1)  // a bucket that starts in the tile keeps at most 9 records beyond it

// OrderedLinkedList::getDuplicates (hashmatch.hpp:162-197) on the ordered entries st[0..n) / kv[0..n)
// (source records have bit 31 of kv clear); emit(kv_source, kv_target) for every pair it reports.
template <class F>
__device__ __forceinline__ void ht_walk(const unsigned long long* st, const uint32_t* kv, int n, F&& emit) {
  int i = 0;
  while (i < n) {
    const int p = i;
    ++i;
    if (i < n && st[p] == st[i]) {
      if ((kv[p] ^ kv[i]) >> 31) {
        const bool e = (i + 1 < n) ? (st[i + 1] != st[i]) : true;
        if (e) emit(kv[p], kv[i]);
        if (i + 1 < n && i + 2 >= n) return;  // "last triplet": the reference leaves the bucket
      } else if (i + 1 < n && ((kv[i] ^ kv[i + 1]) >> 31)) {
        ++i;  // skip over a false pair
      }
    }
  }
}

// pass A (!WRITE): gathers the records (sval = index into rec), leaves them in sorted order in
//   scode / skv for pass B, and counts the pairs per workgroup -> blkcnt (scanned by k_g_scan);
// pass B (WRITE): streams scode / skv and writes the pairs themselves, in bucket order
template <bool WRITE>
__global__ __launch_bounds__(HP_TILE) void k_ht_pairs(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sval,
                                                      const uint2* __restrict__ rec, uint32_t* __restrict__ scode,
                                                      uint32_t* __restrict__ skv,
                                                      const int32_t* __restrict__ gmisc, int W, int epipolar,
                                                      int disp_high, int vtol, int apply_filter,
                                                      int32_t* __restrict__ blkcnt, int mode, void* __restrict__ out, int cap,
                                                      int32_t* __restrict__ count_out, const int32_t* __restrict__ stats,
                                                      int32_t* __restrict__ ncand_out, GpcBatchStrides bs) {
  __shared__ uint32_t s_key[HP_SPAN];             // bucket id, 0xFFFFFFFF beyond the last record
  __shared__ unsigned long long s_st[HP_SPAN];    // state, insertion order
  __shared__ uint32_t s_kv[HP_SPAN];
  __shared__ unsigned long long s_sst[HP_SPAN];   // state, ordered within the bucket (first 10 insertions only)
  __shared__ uint32_t s_skv[HP_SPAN];
  __shared__ uint32_t s_w[HP_TILE / 64];
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
  for (int i = tid; i < HP_SPAN; i += HP_TILE) {
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
      if (epipolar) st |= (unsigned long long)((kv & 0x7FFFFFFFu) / (uint32_t)W) << 32;
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
    for (int i = tid; i < HP_SPAN; i += HP_TILE) {
      const uint32_t key = s_key[i];
      if (key == 0xFFFFFFFFu) continue;
      int t = 0;  // insertion index within the bucket
      while (t < HM_CAP && i - t - 1 >= 0 && s_key[i - t - 1] == key) ++t;
      if (t >= HM_CAP) continue;                   // dropped: the list was full
      const int h = i - t;                         // the bucket's first record
      if (h == 0 && s_prev == key) continue;       // bucket of the previous tile (handled there, in its halo)
      if (h >= HP_TILE) continue;                  // bucket of the next tile
      const unsigned long long st = s_st[i];
      int r = 0;
      for (int u = 0; u < HM_CAP; ++u) {
        const int q = h + u;
        if (q >= HP_SPAN || s_key[q] != key) break;
        const unsigned long long o = s_st[q];
        r += (o < st || (o == st && u < t)) ? 1 : 0;
      }
      const uint32_t kv = s_kv[i];
      s_sst[h + r] = st;
      s_skv[h + r] = kv;
      scode[j0 + h + r] = (uint32_t)st;
      skv[j0 + h + r] = kv;
    }
    __syncthreads();
  }

  // ---- the thread of a bucket's first record walks its ordered list
  int n = 0;
  {
    const uint32_t key = s_key[tid];
    const bool head = key != 0xFFFFFFFFu && (tid == 0 ? s_prev != key : s_key[tid - 1] != key);
    if (head)
      while (n < HM_CAP && tid + n < HP_SPAN && s_key[tid + n] == key) ++n;
  }
  auto passes = [&](uint32_t ks, uint32_t kt, int4& m) {
    const int a = (int)(ks & 0x7FFFFFFFu), b = (int)(kt & 0x7FFFFFFFu);
    m = make_int4(a % W, a / W, b % W, b / W);
    return !apply_filter || (abs(m.y - m.w) <= vtol && abs(m.x - m.z) <= disp_high);
  };
  int cnt = 0;
  ht_walk(s_sst + tid, s_skv + tid, n, [&](uint32_t ks, uint32_t kt) {
    int4 m;
    cnt += passes(ks, kt, m) ? 1 : 0;
  });
  const uint32_t incl = wave_incl_scan((uint32_t)cnt);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = incl - (uint32_t)cnt;
  uint32_t total = 0;
  for (int w = 0; w < HP_TILE / 64; ++w) {
    if (w < wave) base += s_w[w];
    total += s_w[w];
  }
  if (!WRITE) {
    if (tid == 0) blkcnt[blockIdx.x] = (int32_t)total;
    return;
  }
  const int off = blkcnt[blockIdx.x];  // exclusive prefix over the pair's workgroups (k_g_scan)
  int pos = off + (int)base;
  ht_walk(s_sst + tid, s_skv + tid, n, [&](uint32_t ks, uint32_t kt) {
    int4 m;
    if (!passes(ks, kt, m)) return;
    if (pos < cap) {
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
  });
  if (blockIdx.x == gridDim.x - 1 && tid == 0) {
    *count_out = off + (int)total;
    if (ncand_out) {
      ncand_out[0] = stats[GPC_STAT_NCAND];
      ncand_out[1] = stats[GPC_STAT_STRIDE + GPC_STAT_NCAND];
    }
  }
}

}  // namespace gpc
