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
// of k_global.h) groups them by bucket without disturbing insertion order, and one thread per
// bucket replays the capped ordered insert and the list walk on at most 10 records.
// This mode returns a slightly different (smaller) match set than the sort matcher, exactly
// as in the reference; it is not on the benchmarked path and is not tuned.
#pragma once
#include "gpc_device.h"
#include "k_global.h"

namespace gpc {

#define HM_BUCKETS 214673u
#define HM_CAP 10

// skey[i] = state % 214673, sval[i] = i  for the N records (code, side<<31|k) in insertion order
__global__ __launch_bounds__(256) void k_ht_bucket_ids(const uint32_t* __restrict__ codes0,
                                                       const uint32_t* __restrict__ kv0,
                                                       const int32_t* __restrict__ gmisc, int W, int epipolar,
                                                       uint32_t* __restrict__ skey, uint32_t* __restrict__ sval,
                                                       GpcBatchStrides bs) {
  codes0 += blockIdx.y * bs.recs;
  kv0 += blockIdx.y * bs.recs;
  skey += blockIdx.y * bs.recs;
  sval += blockIdx.y * bs.recs;
  gmisc += blockIdx.y * GM_STRIDE;
  const int N = gmisc[GM_N];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  unsigned long long state = codes0[i];
  if (epipolar) state |= (unsigned long long)((kv0[i] & 0x7FFFFFFFu) / (uint32_t)W) << 32;
  skey[i] = (uint32_t)(state % HM_BUCKETS);
  sval[i] = (uint32_t)i;
}

struct HtPairs {
  int n;
  int4 p[HM_CAP / 2];
};

// Replays hashmatch.hpp for the bucket whose first sorted record is j.
__device__ __forceinline__ void ht_bucket_pairs(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sval,
                                                const uint32_t* __restrict__ codes0, const uint32_t* __restrict__ kv0,
                                                int j, int N, int W, int epipolar, int disp_high, int vtol,
                                                int apply_filter, HtPairs& out) {
  out.n = 0;
  unsigned long long st[HM_CAP];
  uint32_t kv[HM_CAP];
  int n = 0;
  const uint32_t b = skey[j];
  // OrderedLinkedList::insert: a full list drops the value; otherwise it goes behind every element <= it
  for (int t = 0; t < HM_CAP && j + t < N && skey[j + t] == b; ++t) {
    const uint32_t idx = sval[j + t];
    const uint32_t v = kv0[idx];
    unsigned long long s = codes0[idx];
    if (epipolar) s |= (unsigned long long)((v & 0x7FFFFFFFu) / (uint32_t)W) << 32;
    int pos = n;
    while (pos > 0 && st[pos - 1] > s) {
      st[pos] = st[pos - 1];
      kv[pos] = kv[pos - 1];
      --pos;
    }
    st[pos] = s;
    kv[pos] = v;
    ++n;
  }
  // OrderedLinkedList::getDuplicates (hashmatch.hpp:162-197); source records have bit 31 clear
  int i = 0;
  while (i < n) {
    const int p = i;
    ++i;
    if (i < n && st[p] == st[i]) {
      if ((kv[p] ^ kv[i]) >> 31) {
        const bool emit = (i + 1 < n) ? (st[i + 1] != st[i]) : true;
        if (emit) {
          const int ks = (int)(kv[p] & 0x7FFFFFFFu), kt = (int)(kv[i] & 0x7FFFFFFFu);
          const int4 m = make_int4(ks % W, ks / W, kt % W, kt / W);
          if (!apply_filter || (abs(m.y - m.w) <= vtol && abs(m.x - m.z) <= disp_high)) out.p[out.n++] = m;
        }
        if (i + 1 < n && i + 2 >= n) return;  // "last triplet": the reference leaves the bucket
      } else if (i + 1 < n && ((kv[i] ^ kv[i + 1]) >> 31)) {
        ++i;  // skip over a false pair
      }
    }
  }
}

// pass A: pairs per workgroup; pass B (WRITE): the pairs themselves, in bucket order
template <bool WRITE>
__global__ __launch_bounds__(256) void k_ht_pairs(const uint32_t* __restrict__ skey, const uint32_t* __restrict__ sval,
                                                  const uint32_t* __restrict__ codes0, const uint32_t* __restrict__ kv0,
                                                  const int32_t* __restrict__ gmisc, int W, int epipolar,
                                                  int disp_high, int vtol, int apply_filter,
                                                  int32_t* __restrict__ blkcnt, int mode, void* __restrict__ out, int cap,
                                                  int32_t* __restrict__ count_out, const int32_t* __restrict__ stats,
                                                  int32_t* __restrict__ ncand_out, GpcBatchStrides bs) {
  __shared__ uint32_t s_w[4];
  skey += blockIdx.y * bs.recs;
  sval += blockIdx.y * bs.recs;
  codes0 += blockIdx.y * bs.recs;
  kv0 += blockIdx.y * bs.recs;
  blkcnt += blockIdx.y * bs.blk;
  gmisc += blockIdx.y * GM_STRIDE;
  stats += blockIdx.y * 2 * GPC_STAT_STRIDE;
  if (WRITE) {
    out = reinterpret_cast<char*>(out) + blockIdx.y * bs.out;
    count_out += blockIdx.y;
    if (ncand_out) ncand_out += 2 * blockIdx.y;
  }
  const int N = gmisc[GM_N];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  HtPairs pr;
  pr.n = 0;
  if (j < N && (j == 0 || skey[j] != skey[j - 1]))
    ht_bucket_pairs(skey, sval, codes0, kv0, j, N, W, epipolar, disp_high, vtol, apply_filter, pr);
  const uint32_t incl = wave_incl_scan((uint32_t)pr.n);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = incl - (uint32_t)pr.n;
  for (int w = 0; w < wave; ++w) base += s_w[w];
  const uint32_t total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  if (!WRITE) {
    if (threadIdx.x == 0) blkcnt[blockIdx.x] = (int32_t)total;
    return;
  }
  const int off = block_prefix_rows(blkcnt, 0, blockIdx.x);
  for (int q = 0; q < pr.n; ++q) {
    const int pos = off + (int)base + q;
    if (pos >= cap) break;
    const int4 m = pr.p[q];
    if (mode == 0) {
      uint32_t* o = reinterpret_cast<uint32_t*>(out) + (long)pos * 3;
      o[0] = m.x;
      o[1] = m.y;
      o[2] = __float_as_uint((float)(m.x - m.z));
    } else {
      reinterpret_cast<int4*>(out)[pos] = m;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    *count_out = off + (int)total;
    if (ncand_out) {
      ncand_out[0] = stats[GPC_STAT_NCAND];
      ncand_out[1] = stats[GPC_STAT_STRIDE + GPC_STAT_NCAND];
    }
  }
}

}  // namespace gpc
