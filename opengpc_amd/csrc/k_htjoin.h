// k_htjoin.h -- the hash-table matcher (settings.useHashtable_ == true) without a device-wide sort.
//
// Replaces the Hashmatch branch of Forest::depthPriorFast (inference.hpp:204-225) and ndb::Hashmatch /
// OrderedLinkedList (hashmatch.hpp:91-131, 162-197, 252-263) -- like k_hashtable.h, which stays as the fallback
// (three 8-bit radix passes over the bucket ids + a replay of the sorted records).
//
// The reference's table has 214673 buckets `state % 214673`; a bucket keeps the FIRST 10 elements inserted (all
// source descriptors before all targets, each in mask order) ordered by state, and the lists are walked in bucket
// order.  Insertion order is the order of (side, pixel index), which every record carries -- so nothing has to be
// kept stable on the way:
//   1. k_gp_hist<true> / k_g_scan / k_gp_scatter<true> (k_partition.h) partition the records of either image by the top
//      bits of their bucket: 210 bins of 1024 buckets, ~1360 records per side and bin for a 1024x436 pair (larger
//      images: 420 / 839 bins of 512 / 256 buckets, HtjArgs::lbits);
//   2. k_ht_join, one workgroup per bin: counting sort of the bin's records by bucket in LDS (arrival order), every
//      record counts the records of its bucket inserted before it (kept iff < 10), then its rank among the kept ones
//      and its successor; thread b replays the walk of bucket b's list (ht_walk_bits, k_hashtable.h) on link bits, a
//      block scan of the emitted pairs gives their places in the bin's output;
//   3. k_ht_gather: bins in ascending order = bucket order -> gpc_support / gpc_correspondence.
// A bin that holds more than HTJ_CAP records (images beyond ~2.5 M candidates, or heavily repeated states) raises
// the overflow word (k_ht_check, right after the histogram) and the host takes the radix path.
#pragma once
#include "gpc_device.h"
#include "k_hashtable.h"
#include "k_partition.h"

namespace gpc {

#define HTJ_THREADS 1024
#define HTJ_BUCKETS (1 << HTJ_LBITS)  // == HTJ_THREADS: thread b owns bucket b of the bin
// k_ht_join<RPT>: RPT records per thread; a bin may hold HTJ_THREADS * RPT records of both images, 8 bytes of LDS each.
// RPT = 4 (64 VGPRs, two workgroups per CU) is the rule; RPT = 8 (one workgroup per CU) takes the bins of images
// whose 839 bins of 256 buckets would still overflow 4096 records (1920x1080).
template <int RPT>
struct HtjOcc { static constexpr int kWaves = RPT <= 4 ? 8 : 4; };
// k_ht_join<RPT, NT>: NT = 1024 threads is the rule; NT = 512 takes bins of at most 512 buckets and 2048 records -- 1920x1080:
// 128 buckets of 13 records, 1664 per bin -- where 1024 threads left 60 % of their record slots empty and every wave issued the
// whole instruction stream for them (four workgroups per CU instead of two).

struct HtjArgs {
  const uint2* kv;         // [npairs][recs]: records (code, pixel index), left image's then (at recs / 2) the right image's, by bin
  const int32_t* tabs;     // scanned (bin, chunk) tables of k_gp_hist: entry (b, 0) = start of bin b
  const int32_t* stats;
  uint2* staged;           // [npairs][recs / 2]: (y << 14 | x left, right) of a bin's pairs from the bin's left start on
  int32_t* bincnt;         // [npairs][nbins]: pairs per bin
  long recs;
  int nbins, nchunk, epi, disp_high, vtol, apply_filter;
  int lbits;               // log2(buckets per bin): HTJ_LBITS, or less for large images (then only the first threads own a bucket)
  const int32_t* biglist;  // [npairs][1 + HTJ_BIGCAP]: count, then the bins with more than 2048 records (k_ht_check)
  int use_list;            // 1: workgroup b takes bin biglist[pair][1 + b] (the launch's grid is the longest list of the batch)
  int min_recs;            // this launch takes the bins with more than min_recs (and at most NT * RPT) records: the 512-thread
                           // instantiation takes the bins up to 2048, a 1024-thread launch beside it the few larger ones
  int mid;                 // buckets with 11 .. mid records keep one thread per record, fuller ones are taken by a wave each
                           // (HM_CAP: every bucket with more than ten records goes to a wave -- the host's choice, see there)
  GpcDivW dw;
};

// Diagnostic build only (-DGPC_STAMPS, tools/stamp_profile.py ht): s_memtime at phase boundaries of thread 0.
#ifdef GPC_STAMPS
__device__ unsigned long long g_hj_stamps[16];
#define HJ_STAMP(i)                                                                              \
  do {                                                                                           \
    unsigned long long t_;                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    hj_acc[i] += t_ - hj_t0;                                                                     \
    hj_t0 = t_;                                                                                  \
  } while (0)
#define HJ_STAMP_FLUSH()                                                                         \
  if (threadIdx.x == 0 && (blockIdx.x & 15) == 5)                                                \
    for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_hj_stamps[i_], hj_acc[i_])
#define HJ_STAMP_INIT()                                                                          \
  unsigned long long hj_t0;                                                                      \
  unsigned long long hj_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                          \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hj_t0)::"memory")
#else
#define HJ_STAMP(i)
#define HJ_STAMP_INIT()
#define HJ_STAMP_FLUSH()
#endif

// A record's position travels as side << 31 | y << 14 | x (x < 16384: check_dims; y < 2^17: the host takes the radix
// path for taller images): same order as (side, pixel index), and row and column come out with a shift and a mask.
#define HTJ_POS 0x7FFFFFFFu
#define HTJ_XBITS 14
#define HTJ_XMASK 0x3FFFu
#define HTJ_MAXH (1 << 17)
__device__ __forceinline__ uint32_t htj_row(uint32_t kv) { return (kv >> HTJ_XBITS) & 0x1FFFFu; }
#define HTJ_SIDE 0x80000000u

// Raises the overflow word if some bin holds more records than one workgroup takes (known once the chunk tables are
// scanned: before anything is scattered).  grid: (npairs); 256 threads
#define HTJ_BIGCAP 1024  // entries of a pair's list of bins beyond the 512-thread join (more: that launch covers the whole grid)
__global__ void k_ht_check(const int32_t* __restrict__ tabs, const int32_t* __restrict__ stats, int nbins, int nchunk,
                           int cap, int32_t* __restrict__ overflow, int split_at, int32_t* __restrict__ biglist) {
  const int pair = blockIdx.x;
  const int32_t* tl = tabs + (long)(pair * 2) * nbins * nchunk;
  const int32_t* tr = tl + (long)nbins * nchunk;
  const int NL = stats[(pair * 2) * GPC_STAT_STRIDE + GPC_STAT_NCAND], NR = stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
  int mx = 0;
  for (int bin = threadIdx.x; bin < nbins; bin += blockDim.x) {
    const int nl = (bin + 1 < nbins ? tl[(long)(bin + 1) * nchunk] : NL) - tl[(long)bin * nchunk];
    const int nr = (bin + 1 < nbins ? tr[(long)(bin + 1) * nchunk] : NR) - tr[(long)bin * nchunk];
    mx = max(mx, nl + nr);
    if (nl + nr > split_at) {  // the work list of the 1024-thread launch beside the 512-thread one: [count | bins] per pair
      const int at = atomicAdd(&biglist[(long)pair * (HTJ_BIGCAP + 1)], 1);
      if (at < HTJ_BIGCAP) biglist[(long)pair * (HTJ_BIGCAP + 1) + 1 + at] = bin;
    }
  }
  // (one atomic per wave: one per bin on a single word took 19 us per batch of 8 at 1678 bins)
  mx = (int)wave_max_u32((uint32_t)mx);
  if ((threadIdx.x & 63) == 0) {
    if (mx > cap) atomicOr(overflow, 1);
    atomicMax(overflow + 1, mx);  // the batch's largest bin: the host sizes its next attempt by it
  }
  if (threadIdx.x == 0) atomicMax(overflow + 2, NL + NR);  // the batch's largest pair: records per bucket (HtjArgs::mid)
  __syncthreads();  // the list's length is final
  if (threadIdx.x == 0) atomicMax(overflow + 3, biglist[(long)pair * (HTJ_BIGCAP + 1)]);
}

// rank += (oy : oc : okv) < (y : c : kv) as 96-bit numbers, i.e. "state, then insertion order": a borrow chain.
__device__ __forceinline__ int htj_rank_add(int rank, uint32_t oy, uint32_t oc, uint32_t okv, uint32_t y, uint32_t c,
                                            uint32_t kv, bool epi) {
  int out;
  uint32_t tmp;
  if (epi)
    asm("v_sub_co_u32 %1, vcc, %2, %3\n\tv_subb_co_u32 %1, vcc, %4, %5, vcc\n\tv_subb_co_u32 %1, vcc, %6, %7, vcc\n\t"
        "v_addc_co_u32 %0, vcc, 0, %8, vcc"
        : "=v"(out), "=&v"(tmp)
        : "v"(okv), "v"(kv), "v"(oc), "v"(c), "v"(oy), "v"(y), "v"(rank)
        : "vcc");
  else
    asm("v_sub_co_u32 %1, vcc, %2, %3\n\tv_subb_co_u32 %1, vcc, %4, %5, vcc\n\tv_addc_co_u32 %0, vcc, 0, %6, vcc"
        : "=v"(out), "=&v"(tmp)
        : "v"(okv), "v"(kv), "v"(oc), "v"(c), "v"(rank)
        : "vcc");
  return out;
}

// One workgroup per bin.  A record is handled by ONE thread from arrival to output: it counts its rank r in its
// bucket's ordered list (entries before it), moves there, publishes "equal state" / "other image" / "passes the
// disparity filter" of the link (r, r + 1) as bits r, 10 + r, 20 + r of the bucket's word; the bucket's thread
// replays OrderedLinkedList::getDuplicates on those bits and hands back the emitted links + their place.
// Buckets with more than 10 records keep the first ten in insertion order (= order of kv).  Where they are the exception
// (1024x436: 2.7 records per bucket; repeated states -- the zero code of a flat stretch of a row puts hundreds of records
// into one bucket) one WAVE takes each: ten times the smallest kv above the last one (a strided pass + a DPP minimum),
// the ten winners ranked among themselves.  Where they are the rule (1920x1080: 13 records per bucket) the buckets up
// to HtjArgs::mid records stay with their records' threads (see "insert where a list fills up" below).
// The kernel is bound by the number of LDS operations (random addresses, 32 waves per CU): a record is one 8-byte
// LDS element, a bucket's start and count one word.
// grid: (nbins, npairs); dynamic LDS: 8 * HTJ_THREADS * RPT bytes
template <int RPT, int NT = HTJ_THREADS>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(HtjOcc<RPT>::kWaves, 8))) void k_ht_join(HtjArgs a) {
  constexpr int HTJ_RPT = RPT, HTJ_CAP = NT * RPT;
  extern __shared__ __attribute__((aligned(16))) uint2 htj_rec[];  // (code, side << 31 | y << 14 | x), by bucket
  __shared__ uint32_t s_cs[NT];    // records per bucket (thread b owns bucket b: 1 << a.lbits <= NT), then start | count << 16
  __shared__ uint32_t s_bits[NT];  // link bits, then emitted links | first output place << 10
  __shared__ uint32_t s_list[HTJ_CAP / (HM_CAP + 1) + 1];  // the buckets with more than 10 records
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_nbig, s_nmid;
  const int pair = blockIdx.y;
  int bin = blockIdx.x;
  if (a.use_list) {  // block-uniform
    const int32_t* bl = a.biglist + (long)pair * (HTJ_BIGCAP + 1);
    if (bin >= bl[0]) return;
    bin = bl[1 + bin];
  }
  const int tid = threadIdx.x;
  HJ_STAMP_INIT();
  const int32_t* tl = a.tabs + (long)(pair * 2) * a.nbins * a.nchunk;
  const int32_t* tr = tl + (long)a.nbins * a.nchunk;
  const int NL = a.stats[(pair * 2) * GPC_STAT_STRIDE + GPC_STAT_NCAND], NR = a.stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
  const int ol = tl[(long)bin * a.nchunk], orr = tr[(long)bin * a.nchunk];
  const int nl = (bin + 1 < a.nbins ? tl[(long)(bin + 1) * a.nchunk] : NL) - ol;
  const int nr = (bin + 1 < a.nbins ? tr[(long)(bin + 1) * a.nchunk] : NR) - orr;
  const int n = nl + nr;
  int32_t* bincnt = a.bincnt + (long)pair * a.nbins;
  if (n > HTJ_CAP || n <= a.min_recs) return;  // block-uniform: another launch's bin (a bin beyond every launch: k_ht_check has
                                                // sent the batch to the radix path already)
  HJ_STAMP(0);  // bin bounds (scalar loads)
  // one base per array and a 32-bit index (left records at ol + i, right ones at recs / 2 + orr + i - nl): per-lane
  // 64-bit pointers would cost two registers per record and array
  const uint2* rb_ = a.kv + pair * a.recs;
  const uint32_t roff = (uint32_t)(a.recs / 2) + (uint32_t)orr - (uint32_t)nl;

  // ---- the bin's records -> registers (loads first, the counters are cleared under them).
  // (Two or three bins per workgroup with the next bin's records prefetched under the last phases were tried: 291 vs
  // 241 us per 32 pairs -- the eight prefetched values per thread do not fit beside the rank phase at 64 VGPRs.)
  uint32_t code[HTJ_RPT], kv[HTJ_RPT];
#pragma unroll
  for (int j = 0; j < HTJ_RPT; ++j) {
    const int i = j * NT + tid;
    code[j] = 0u;
    kv[j] = 0xFFFFFFFFu;  // no record
    if (i < n) {
      const bool right = i >= nl;
      const uint32_t idx = (uint32_t)i + (right ? roff : (uint32_t)ol);
      const uint2 rec = rb_[idx];
      code[j] = rec.x;
      const uint32_t pix = rec.y;
      const uint32_t y = (uint32_t)divw(pix, a.dw);
      kv[j] = (right ? HTJ_SIDE : 0u) | (y << HTJ_XBITS) | (pix - y * (uint32_t)a.dw.W);
    }
  }
  s_cs[tid] = 0u;
  s_bits[tid] = 0u;
  if (tid == 0) s_nbig = s_nmid = 0u;
  HJ_STAMP(1);  // records arrive
  __syncthreads();
  int lb[HTJ_RPT], place[HTJ_RPT];
  uint32_t yy[HTJ_RPT];
#pragma unroll
  for (int j = 0; j < HTJ_RPT; ++j) {
    lb[j] = -1;
    yy[j] = 0u;
    place[j] = 0;
    if (kv[j] != 0xFFFFFFFFu) {
      yy[j] = a.epi ? htj_row(kv[j]) : 0u;
      lb[j] = (int)(hm_bucket(code[j], yy[j]) & ((1u << a.lbits) - 1u));
      place[j] = (int)atomicAdd(&s_cs[lb[j]], 1u);  // arrival rank within the bucket
    }
  }
  __syncthreads();
  HJ_STAMP(2);  // buckets + counts
  const uint32_t own_cnt = s_cs[tid];
  uint32_t own_s;
  {
    uint32_t total;
    own_s = gp_block_exscan<NT / 64>(own_cnt, s_w, &total);
    s_cs[tid] = own_s | (own_cnt << 16);
    if (own_cnt > (uint32_t)a.mid) s_list[atomicAdd(&s_nbig, 1u)] = (uint32_t)tid;
    else if (own_cnt > HM_CAP) s_nmid = 1u;  // (whoever writes, writes 1)
  }
  __syncthreads();
  HJ_STAMP(3);  // scan
  int bs[HTJ_RPT], len[HTJ_RPT];  // the bucket's stretch
#pragma unroll
  for (int j = 0; j < HTJ_RPT; ++j) {
    bs[j] = len[j] = 0;
    if (lb[j] >= 0) {
      const uint32_t w = s_cs[lb[j]];
      bs[j] = (int)(w & 0xFFFFu);
      htj_rec[bs[j] + place[j]] = make_uint2(code[j], kv[j]);
      if ((w >> 16) > (uint32_t)a.mid) lb[j] = -1;  // a wave takes that bucket
      else len[j] = (int)(w >> 16);
    }
  }
  __syncthreads();
  HJ_STAMP(4);  // placed

  // ---- OrderedLinkedList::insert where a list fills up: a full list drops the value, so a bucket keeps the first ten
  //      records in insertion order (= order of kv).  Buckets with 11 .. a.mid records (1920x1080: 13 records per
  //      bucket on average, nearly every bucket) stay with their records' threads: a record counts the records of its
  //      bucket inserted before it (len reads of 4 bytes, all of a step in flight together); the first ten then move
  //      to the front of the bucket's stretch, the others drop out, and the bucket goes on as one of ten records.
  //      (One wave per such bucket, as for the fuller ones below, was 43 % of the kernel at 1920x1080.)
  const bool has_mid = s_nmid != 0u;  // block-uniform: bins without such a bucket skip the pass and its barriers
  if (has_mid) {
#pragma unroll
    for (int j = 0; j < HTJ_RPT; ++j) place[j] = 0;  // (the arrival rank has been used)
    // (no branch around a read: with one the compiler sinks the compare into the branch and waits for every read on the
    // spot -- eight LDS round trips per step instead of one; a lane without an entry re-reads its bucket's first)
    int maxlen = 0;
#pragma unroll
    for (int j = 0; j < HTJ_RPT; ++j) maxlen = max(maxlen, len[j] > HM_CAP ? len[j] : 0);
    maxlen = (int)wave_max_u32((uint32_t)maxlen);  // wave-uniform trip count
#pragma unroll 1
    for (int u = 0; u < maxlen; u += 2) {
      uint32_t o0[HTJ_RPT], o1[HTJ_RPT];
#pragma unroll
      for (int j = 0; j < HTJ_RPT; ++j) {
        const bool mid = len[j] > HM_CAP;
        o0[j] = htj_rec[bs[j] + ((mid && u < len[j]) ? u : 0)].y;
        o1[j] = htj_rec[bs[j] + ((mid && u + 1 < len[j]) ? u + 1 : 0)].y;
      }
#pragma unroll
      for (int j = 0; j < HTJ_RPT; ++j) {
        const bool mid = len[j] > HM_CAP;
        place[j] += ((mid && u < len[j] && o0[j] < kv[j]) ? 1 : 0) + ((mid && u + 1 < len[j] && o1[j] < kv[j]) ? 1 : 0);
      }
    }
    __syncthreads();  // every count is complete: the stretches may be rewritten
#pragma unroll
    for (int j = 0; j < HTJ_RPT; ++j)
      if (len[j] > HM_CAP) {
        if (place[j] < HM_CAP) {
          htj_rec[bs[j] + place[j]] = make_uint2(code[j], kv[j]);
          len[j] = HM_CAP;
        } else {  // the list was full when this record came
          lb[j] = -1;
          len[j] = 0;
        }
      }
  }
  // ---- the buckets with more than a.mid records (repeated states; every bucket beyond ten where a.mid is ten), one wave each
  {
    const int lane = tid & 63, wave = tid >> 6;
    const int nbig = (int)s_nbig;  // block-uniform
    for (int k = wave; k < nbig; k += NT / 64) {
      const int b = (int)s_list[k];
      const uint32_t w = s_cs[b];
      const int s0 = (int)(w & 0xFFFFu), nb = (int)(w >> 16);
      // ins = insertion index (order of kv) of this lane's record if it is one of the first ten, else -1
      int ins = -1;
      uint2 me = make_uint2(0u, 0u);
      if (nb <= 64) {  // (nearly all of them: 11 .. 16 records) one record per lane, counted against the others
        if (lane < nb) me = htj_rec[s0 + lane];
        int t = 0;
        for (int o = 0; o < nb; ++o) t += ((uint32_t)__builtin_amdgcn_readlane((int)me.y, o) < me.y) ? 1 : 0;
        if (lane < nb && t < HM_CAP) ins = t;
      } else {  // ten times the smallest kv above the last one; lane r keeps the r-th
        uint32_t prev = 0u;
        int myidx = s0;
        for (int r = 0; r < HM_CAP; ++r) {
          uint32_t m = 0xFFFFFFFFu;
          int midx = s0;
          for (int i = s0 + lane; i < s0 + nb; i += 64) {
            const uint32_t v = htj_rec[i].y;
            if ((r == 0 || v > prev) && v < m) {
              m = v;
              midx = i;
            }
          }
          const uint32_t wm = ~wave_max_u32(~m);
          const unsigned long long bal = __ballot(m == wm);  // kv are distinct: one lane
          const int widx = __builtin_amdgcn_readlane(midx, (int)__builtin_ctzll(bal));
          if (lane == r) myidx = widx;
          prev = wm;
        }
        if (lane < HM_CAP) {
          me = htj_rec[myidx];
          ins = lane;
        }
      }
      const uint32_t my = a.epi ? htj_row(me.y) : 0u;
      int r = 0;  // "behind every element <= it": stable order by state among the ten
#pragma unroll
      for (int o = 0; o < HM_CAP; ++o) {
        const int src = (int)__builtin_ctzll(__ballot(ins == o));  // exactly one lane
        const uint32_t oc = (uint32_t)__builtin_amdgcn_readlane((int)me.x, src);
        const uint32_t okv = (uint32_t)__builtin_amdgcn_readlane((int)me.y, src);
        const uint32_t oy = (uint32_t)__builtin_amdgcn_readlane((int)my, src);
        const bool before = oy < my || (oy == my && (oc < me.x || (oc == me.x && okv < me.y)));
        r += before ? 1 : 0;
      }
      __builtin_amdgcn_wave_barrier();
      if (ins >= 0) htj_rec[s0 + r] = me;  // (every read above has completed: same wave, in order)
      __builtin_amdgcn_wave_barrier();
      if (lane + 1 < HM_CAP) {  // the link (lane, lane + 1)
        const uint2 e0_ = htj_rec[s0 + lane], e1_ = htj_rec[s0 + lane + 1];
        const int ya = (int)htj_row(e0_.y), yb = (int)htj_row(e1_.y);
        uint32_t lw = (e0_.x == e1_.x && (!a.epi || ya == yb)) ? (1u << lane) : 0u;
        lw |= (((e0_.y ^ e1_.y) >> 31) & 1u) << (10 + lane);
        bool pass = true;
        if (a.apply_filter) {
          const int xa = (int)(e0_.y & HTJ_XMASK), xb = (int)(e1_.y & HTJ_XMASK);
          pass = abs(ya - yb) <= a.vtol && abs(xa - xb) <= a.disp_high;
        }
        lw |= (pass ? 1u : 0u) << (20 + lane);
        atomicOr(&s_bits[b], lw);
      }
    }
  }
  if (has_mid) __syncthreads();  // the kept records of the 11 .. a.mid buckets are in place
  HJ_STAMP(5);  // 10-cap
  // ---- the other buckets: rank r = entries before this one, all of the thread's records side by side; a (wave, j)
  //      without entries left issues nothing
  int rank[HTJ_RPT];
#pragma unroll
  for (int j = 0; j < HTJ_RPT; ++j) rank[j] = lb[j] >= 0 ? 0 : -1;
#pragma unroll
  for (int u = 0; u < HM_CAP; u += 2) {  // two entries per step; all loads of a step are in flight together
    uint2 o0[HTJ_RPT], o1[HTJ_RPT];
    bool any_more = false;
#pragma unroll
    for (int j = 0; j < HTJ_RPT; ++j) {
      o0[j] = o1[j] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);  // no entry: behind everything
      if (__any(u < len[j])) {
        any_more = true;
        if (u < len[j]) o0[j] = htj_rec[bs[j] + u];
        if (u + 1 < len[j]) o1[j] = htj_rec[bs[j] + u + 1];
      }
    }
    if (!any_more) break;
#pragma unroll
    for (int j = 0; j < HTJ_RPT; ++j) {
      uint32_t y0 = 0u, y1 = 0u;
      if (a.epi) {  // the row is the upper half of the state
        y0 = htj_row(o0[j].y);
        y1 = htj_row(o1[j].y);
      }
      rank[j] = htj_rank_add(rank[j], y0, o0[j].x, o0[j].y, yy[j], code[j], kv[j], a.epi);
      rank[j] = htj_rank_add(rank[j], y1, o1[j].x, o1[j].y, yy[j], code[j], kv[j], a.epi);
    }
  }
#pragma unroll
  for (int j = 0; j < HTJ_RPT; ++j)
    if (lb[j] < 0) rank[j] = -1;
  __syncthreads();
  HJ_STAMP(6);  // ranks
#pragma unroll
  for (int j = 0; j < HTJ_RPT; ++j)
    if (rank[j] >= 0) htj_rec[bs[j] + rank[j]] = make_uint2(code[j], kv[j]);
  __syncthreads();
  uint32_t succ[HTJ_RPT];
#pragma unroll
  for (int j = 0; j < HTJ_RPT; ++j) {
    succ[j] = 0u;
    if (rank[j] >= 0 && rank[j] + 1 < len[j]) {  // the link (r, r + 1)
      const int r = rank[j];
      const uint2 nx = htj_rec[bs[j] + r + 1];
      succ[j] = nx.y;
      const int ya = (int)htj_row(kv[j]), yb = (int)htj_row(nx.y);
      uint32_t w = (nx.x == code[j] && (!a.epi || ya == yb)) ? (1u << r) : 0u;
      w |= (((kv[j] ^ nx.y) >> 31) & 1u) << (10 + r);
      bool pass = true;
      if (a.apply_filter) {
        const int xa = (int)(kv[j] & HTJ_XMASK), xb = (int)(nx.y & HTJ_XMASK);
        pass = abs(ya - yb) <= a.vtol && abs(xa - xb) <= a.disp_high;
      }
      w |= (pass ? 1u : 0u) << (20 + r);
      atomicOr(&s_bits[lb[j]], w);
    }
  }
  __syncthreads();
  HJ_STAMP(7);  // list order + links

  // ---- OrderedLinkedList::getDuplicates: thread b replays the walk of bucket b on the link bits
  uint32_t emit = 0u;
  {
    const int m = min((int)own_cnt, HM_CAP);
    const uint32_t w = s_bits[tid];
    if (m > 1 && (w & (w >> 10) & (w >> 20) & 0x3FFu))  // a reported link is "equal", "other image" and "passes"
      emit = ht_walk_bits(w & 0x3FFu, (w >> 10) & 0x3FFu, m) & (w >> 20);
  }
  uint32_t total;
  const uint32_t base = gp_block_exscan<NT / 64>((uint32_t)__popc(emit), s_w, &total);
  s_bits[tid] = emit | (base << 10);
  if (tid == 0) bincnt[bin] = (int32_t)total;
  uint2* st = a.staged + pair * (a.recs / 2) + ol;  // every pair has its own left record: at most nl of them
  if (own_cnt > (uint32_t)a.mid) {  // a wave's bucket: its thread writes the (few) pairs
    uint32_t pos = base;
    for (uint32_t todo = emit; todo; todo &= todo - 1u) {
      const int u = __builtin_ctz(todo);
      st[pos++] = make_uint2(htj_rec[own_s + u].y & HTJ_POS, htj_rec[own_s + u + 1].y & HTJ_POS);
    }
  }
  __syncthreads();
  HJ_STAMP(8);  // walk + scan
#pragma unroll
  for (int j = 0; j < HTJ_RPT; ++j)
    if (rank[j] >= 0) {
      const uint32_t w = s_bits[lb[j]];
      if ((w >> rank[j]) & 1u) {
        const int pos = (int)(w >> 10) + __popc(w & ((1u << rank[j]) - 1u) & 0x3FFu);
        st[pos] = make_uint2(kv[j] & HTJ_POS, succ[j] & HTJ_POS);
      }
    }
  HJ_STAMP(9);  // output
  HJ_STAMP_FLUSH();
}

// grid: (ceil(nbins / HTG_BINS), npairs)
#define HTG_BINS 2
__global__ __launch_bounds__(RM_THREADS) void k_ht_gather(HtjArgs a, int mode, void* __restrict__ out, long out_stride_bytes,
                                                          int cap, int32_t* __restrict__ counts, int32_t* __restrict__ ncand) {
  const int pair = blockIdx.y;
  const int32_t* bc = a.bincnt + (long)pair * a.nbins;
  const int32_t* tl = a.tabs + (long)(pair * 2) * a.nbins * a.nchunk;
  const int b0 = blockIdx.x * HTG_BINS, bend = min(b0 + HTG_BINS, a.nbins);
  int off = block_prefix_rows(bc, 0, b0);
  char* o = reinterpret_cast<char*>(out) + pair * out_stride_bytes;
  for (int b = b0; b < bend; ++b) {
    const int cnt = bc[b];
    const uint2* st = a.staged + pair * (a.recs / 2) + tl[(long)b * a.nchunk];
    for (int i = threadIdx.x; i < cnt; i += RM_THREADS) {
      const int pos = off + i;
      if (pos >= cap) break;
      const uint2 v = st[i];
      const int yl = (int)(v.x >> HTJ_XBITS), yr = (int)(v.y >> HTJ_XBITS);
      const int xl = (int)(v.x & HTJ_XMASK), xr = (int)(v.y & HTJ_XMASK);
      if (mode == 0) {
        uint32_t* q = reinterpret_cast<uint32_t*>(o) + (long)pos * 3;
        q[0] = xl;
        q[1] = yl;
        q[2] = __float_as_uint((float)(xl - xr));
      } else {
        reinterpret_cast<int4*>(o)[pos] = make_int4(xl, yl, xr, yr);
      }
    }
    off += cnt;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    counts[pair] = off;
    if (ncand) {
      ncand[2 * pair + 0] = a.stats[(pair * 2 + 0) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
      ncand[2 * pair + 1] = a.stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
    }
  }
}

}  // namespace gpc
