// k_partition.h -- non-epipolar ("global") sort-matcher without a device-wide sort.
//
// Replaces, for settings.epipolarMode_ == false and useHashtable_ == false, the two std::sort calls and the
// merge scan of Forest::findCorrespondences (inference.hpp:227-254) and the filter of Forest::rectifiedMatch
// (inference.hpp:384-391) -- like k_global.h, which stays as the fallback.
//
// A code must be unique over the whole left image and over the whole right image, and the output is ordered by
// code.  Instead of sorting all records (four 8-bit radix passes over ~570 k records per pair), the records are
// PARTITIONED into contiguous code ranges of a few hundred to ~2000 records per side, and every partition goes
// through the same LDS hash join + counting rank as an image row of the epipolar matcher (k_rowjoin.h, VIRT):
// partitions in ascending order and ranks inside a partition give the reference's output order.
//   1. k_gp_hist     histogram of the top code bits (256 bins; 512 / 1024 for images beyond ~1 M pixels) per chunk of
//                    rows, in LDS; k_g_scan over the (bin, chunk) table gives every chunk the start of its records in
//                    every bin;
//   2. k_gp_plan     per pair: cuts where the running count max(nL, nR) passes a multiple of the target
//                    size (so a partition is a run of consecutive bins, adapted to the image's code distribution) and
//                    around every bin that is large by itself, start positions of every bin and partition, and an
//                    OVERFLOW flag when a partition exceeds what one workgroup can join (a single bin of more than 4096
//                    records: heavily duplicated codes, striped images) -- the host then takes the radix-sort path of
//                    k_global.h;
//   3. k_gp_scatter  records (code, pixel index) to their bin's stretch (positions from an LDS copy of the chunk's starts);
//   4. k_row_join<4, 1024, WIDE, true>  one workgroup per partition;
//   5. k_gp_gather   matches -> gpc_support / gpc_correspondence in partition order.
// Order inside a bin's stretch is arbitrary (LDS atomics); nothing depends on it: the join carries
// positions, and the one place the reference's order among EQUAL codes matters (tail quirk Q2: first of two equal
// targets in mask order) takes the smaller pixel index explicitly.
#pragma once
#include <type_traits>
#include "gpc_device.h"
#include "k_global.h"
#include "k_rows.h"

namespace gpc {

#define GP_MAXBINS 2048   // histogram / scatter kernels come for 256, 1024 and 2048 bins (template MB).  256 = 8 top code bits is the
                          // rule: more destinations make the scatter's runs too short to coalesce (measured per 32 pairs:
                          // 4096 bins 445 us, 1024: 337, 512: 262, 256: 181 -- before the tiles were staged through LDS);
                          // images beyond ~1 M pixels take 512 or 1024 so that a bin still fits one workgroup's LDS
#define GP_NB 4096          // records per side a partition may hold: k_row_join<4, 1024> (GpLayout::cap; 8192 with k_row_join<8, 1024>
                            // where a single bin is larger: skewed top code bits on large images)
#define GP_THREADS 1024
#ifndef GPS_THREADS
#define GPS_THREADS 1024    // threads of a histogram / scatter workgroup (scatter per 32 pairs: 1024 -> 96 us, 512 -> 128, 256 -> 183: shorter runs)
#endif
// tabs: [npairs * 2][nbins * nchunk] int32 -- records per (bin, chunk of rows) of one image, bin-major; after the
//       exclusive scan (k_g_scan) entry (b, c) is where chunk c's records of bin b start in the image's record array.
// plan: per-pair block of int32 (stride ps): [off L : pmax + 1][off R : pmax + 1][rowcnt : pmax][misc : 8]
// misc: 0 number of partitions, 1 overflow flag, 2 last partition with right records, 3 number of over-large partitions,
//       8 .. 8 + GP_BIGCAP - 1 their numbers
// batch words (after the plan blocks): [0] some pair overflowed, [1] largest number of partitions of a pair,
//                                       [2] largest bin (records of one side), [3] largest list of over-large partitions
struct GpLayout {
  int nbins, bshift, nchunk, rows_per_chunk, pmax, target;
  int cap;                       // records per side a partition aims to stay below: bins larger than cap - target stand alone
  int cap_hard;                  // records per side the largest join kernel takes (a single bin beyond it: overflow)
  int epi;                       // HT only: the state carries the row (epipolar mode)
  long ps;                       // ints per pair in the plan blocks
  int o_off, o_rowcnt, o_misc;
};
// Which bin a record goes to.  HT (hash-table matcher, k_htjoin.h): the top bits of its Hashmatch bucket.
#define HTJ_LBITS 10  // buckets per bin = 1024: one per thread of the joining workgroup
template <bool HT>
__device__ __forceinline__ uint32_t gp_bin(uint32_t code, int y, const GpLayout& g) {
  if (HT) return hm_bucket(code, g.epi ? (uint32_t)y : 0u) >> g.bshift;  // bshift = log2(buckets per bin), 8 .. HTJ_LBITS
  return code >> g.bshift;
}
#define GP_NPARTS 0
#define GP_OVERFLOW 1
#define GP_LASTR 2
#define GP_NBIG 3         // partitions with more than GP_NB records on a side (single bins): listed at misc + 8
#define GP_BIGCAP 248     // entries of that list (more of them: the second launch covers the whole grid again)

// No global atomics anywhere (scattered ones run at ~20 G/s on MI355X: 18 M records took 0.76 ms to count and 1.5 ms to
// place that way): a workgroup counts the records of its chunk of rows per bin in LDS, the table is scanned, and the
// scatter hands out positions from an LDS copy of its chunk's starts.
// grid: (nchunk, 2, npairs)
template <bool HT, int MB>
__global__ __launch_bounds__(GPS_THREADS) void k_gp_hist(const uint32_t* __restrict__ codes,
                                                        const uint8_t* __restrict__ cand, int W, int H, long codes_stride,
                                                        int32_t* __restrict__ tabs, GpLayout g, GpcDivW wd) {
  __shared__ int s_cnt[MB];
  const int chunk = blockIdx.x, side = blockIdx.y, pair = blockIdx.z;
  for (int i = threadIdx.x; i < g.nbins; i += GPS_THREADS) s_cnt[i] = 0;
  __syncthreads();
  const int y0 = GPC_R + chunk * g.rows_per_chunk, y1 = min(y0 + g.rows_per_chunk, H - GPC_R);
  const long img = pair * codes_stride + (long)side * H * W;
  const uint32_t* im = codes + img;
  const uint8_t* cm = cand ? cand + img : nullptr;
  // the chunk's rows are one contiguous stretch of the code image (margin pixels hold the sentinel): four pixels of
  // one row per thread and step (W is a multiple of 16)
  for (int q = y0 * W + 4 * (int)threadIdx.x; q < y1 * W; q += 4 * GPS_THREADS) {
    const uint4 c4 = *reinterpret_cast<const uint4*>(im + q);
    const uint32_t c[4] = {c4.x, c4.y, c4.z, c4.w};
    const int yy = (HT || cm) ? divw((uint32_t)q, wd) : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (g_is_record(c[k], cm ? cm + (long)yy * W : nullptr, q + k - yy * W, W)) atomicAdd(&s_cnt[gp_bin<HT>(c[k], yy, g)], 1);
  }
  __syncthreads();
  int32_t* tab = tabs + (long)(pair * 2 + side) * g.nbins * g.nchunk;
  for (int i = threadIdx.x; i < g.nbins; i += GPS_THREADS) tab[i * g.nchunk + chunk] = s_cnt[i];
}

// exclusive block scan of one value per thread (NW waves); returns the exclusive prefix, *total gets the sum
template <int NW = 16>  // waves of the workgroup
__device__ __forceinline__ uint32_t gp_block_exscan(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan(v);
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t base = incl - v, tot = 0;
  for (int w = 0; w < NW; ++w) {
    if (w < wave) base += s_w[w];
    tot += s_w[w];
  }
  __syncthreads();
  *total = tot;
  return base;
}

// grid: (npairs); 1024 threads, nbins / 1024 consecutive bins each.  tabs are scanned: entry (b, 0) = start of bin b.
__global__ __launch_bounds__(GP_THREADS) void k_gp_plan(const int32_t* __restrict__ tabs, const int32_t* __restrict__ stats,
                                                        int32_t* __restrict__ plan, GpLayout g,
                                                        int32_t* __restrict__ batch_overflow) {
  __shared__ uint32_t s_w[16];
  __shared__ uint32_t s_praw[GP_MAXBINS];  // running count max(nL, nR) before the bin, divided by the target
  __shared__ int32_t s_sl[GP_MAXBINS + 1], s_sr[GP_MAXBINS + 1];  // start of every bin per side (+ the totals)
  extern __shared__ int32_t s_off[];       // [2][pmax + 1]
  const int pair = blockIdx.x;
  int32_t* blk = plan + pair * g.ps;
  int32_t* misc = blk + g.o_misc;
  const int32_t* tl_ = tabs + (long)(pair * 2) * g.nbins * g.nchunk;
  const int32_t* tr_ = tl_ + (long)g.nbins * g.nchunk;
  const int NL = stats[(pair * 2) * GPC_STAT_STRIDE + GPC_STAT_NCAND], NR = stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
  const int tid = threadIdx.x;
  for (int b = tid; b < g.nbins; b += GP_THREADS) {
    s_sl[b] = tl_[(long)b * g.nchunk];
    s_sr[b] = tr_[(long)b * g.nchunk];
  }
  if (tid == 0) {
    s_sl[g.nbins] = NL;
    s_sr[g.nbins] = NR;
  }
  __syncthreads();
  constexpr int BPT = GP_MAXBINS >= GP_THREADS ? GP_MAXBINS / GP_THREADS : 1;
  const int bpt = (g.nbins + GP_THREADS - 1) / GP_THREADS;  // <= BPT
  uint32_t m[BPT], sm = 0;
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int b = tid * bpt + i;
    const bool in = i < bpt && b < g.nbins;
    m[i] = in ? (uint32_t)max(s_sl[b + 1] - s_sl[b], s_sr[b + 1] - s_sr[b]) : 0u;
    sm += m[i];
  }
  {  // the largest bin of the batch: what decides between the 4096- and the 8192-record join and the radix path
    uint32_t mx = 0u;
#pragma unroll
    for (int i = 0; i < BPT; ++i) mx = max(mx, m[i]);
    mx = wave_max_u32(mx);
    if ((tid & 63) == 0 && mx) atomicMax(batch_overflow + 2, (int32_t)mx);
  }
  uint32_t tm;
  uint32_t am = gp_block_exscan(sm, s_w, &tm);
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int b = tid * bpt + i;
    if (i < bpt && b < g.nbins) {
      s_praw[b] = am / (uint32_t)g.target;
      am += m[i];
    }
  }
  __syncthreads();
  // A bin starts a partition when its raw number differs from its predecessor's (the predecessor crossed a multiple of
  // the target), or when it or its predecessor is BIG (more than cap - target records on a side): a big bin stands
  // alone, and a run of other bins holds at most target - 1 + (cap - target) records -- so a partition exceeds the
  // capacity only if a single bin does.  Partition id = cuts before it.
  const uint32_t bigsz = (uint32_t)(g.cap - g.target);
  auto starts = [&](int b) {
    if (b == 0 || s_praw[b] != s_praw[b - 1]) return true;
    const uint32_t mb = (uint32_t)max(s_sl[b + 1] - s_sl[b], s_sr[b + 1] - s_sr[b]);
    const uint32_t mp = (uint32_t)max(s_sl[b] - s_sl[b - 1], s_sr[b] - s_sr[b - 1]);
    return mb > bigsz || mp > bigsz;
  };
  uint32_t ncut = 0;
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int b = tid * bpt + i;
    if (i < bpt && b < g.nbins) ncut += starts(b) ? 1u : 0u;
  }
  uint32_t nparts;
  uint32_t id = gp_block_exscan(ncut, s_w, &nparts);
#pragma unroll
  for (int i = 0; i < BPT; ++i) {
    const int b = tid * bpt + i;
    if (i < bpt && b < g.nbins && starts(b)) {
      if ((int)id <= g.pmax) {
        s_off[id] = s_sl[b];
        s_off[g.pmax + 1 + id] = s_sr[b];
      }
      ++id;
    }
  }
  if (tid == 0 && (int)nparts <= g.pmax) {
    s_off[nparts] = NL;
    s_off[g.pmax + 1 + nparts] = NR;
  }
  __syncthreads();
  const bool too_many = (int)nparts > g.pmax;  // cannot happen while pmax >= records / target + 2; checked anyway
  int over = too_many ? 1 : 0, last_r = -1;
  if (!too_many) {
    for (int p = tid; p <= (int)nparts; p += GP_THREADS) {
      blk[g.o_off + p] = s_off[p];
      blk[g.o_off + g.pmax + 1 + p] = s_off[g.pmax + 1 + p];
      if (p < (int)nparts) {
        const int nl = s_off[p + 1] - s_off[p], nr = s_off[g.pmax + 1 + p + 1] - s_off[g.pmax + 1 + p];
        if (nl > g.cap_hard || nr > g.cap_hard) over = 1;
        if (nr > 0) last_r = p;
        if (nl > GP_NB || nr > GP_NB) {  // the 8192-record join's work list
          const int at = atomicAdd(&misc[GP_NBIG], 1);
          if (at < GP_BIGCAP) misc[8 + at] = p;
        }
      }
    }
  }
  if (__ballot(over) && (tid & 63) == 0) {
    atomicOr(&misc[GP_OVERFLOW], 1);
    atomicOr(batch_overflow, 1);  // one word for the whole batch: what the host reads
  }
  for (int o = 32; o > 0; o >>= 1) last_r = max(last_r, __shfl_xor(last_r, o));
  if ((tid & 63) == 0 && last_r >= 0) atomicMax(&misc[GP_LASTR], last_r);
  __syncthreads();  // the list's length is final
  if (tid == 0) {
    misc[GP_NPARTS] = too_many ? 0 : (int32_t)nparts;
    if (!too_many) atomicMax(batch_overflow + 1, (int32_t)nparts);  // the batch's largest partition count: the join's grid
    atomicMax(batch_overflow + 3, misc[GP_NBIG]);                    // ... and the largest work list of over-large partitions
  }
}

// grid: (nchunk, 2, npairs); records (code, pixel index) of side s of a pair live at kv + pair * recs + s * (recs / 2).
// The chunk's pixels go through in tiles of GP_TILE: the tile's records are first put in bin order in LDS, then
// written out by consecutive threads, so that every bin's run of a tile leaves as one contiguous piece.
#ifndef GP_GROUPS
#define GP_GROUPS 2   // per 32 pairs, scatter of the non-epipolar / hash-table mode: 1 -> 96 / 97 us, 2 -> 81 / 93, 3 -> 86 / 97 (LDS: one workgroup per CU)
#endif
#define GP_TILE (4 * GP_GROUPS * GPS_THREADS)
template <bool HT, int MB>
__global__ __launch_bounds__(GPS_THREADS) void k_gp_scatter(const uint32_t* __restrict__ codes,
                                                           const uint8_t* __restrict__ cand, int W, int H, long codes_stride,
                                                           const int32_t* __restrict__ tabs, GpLayout g, GpcDivW wd,
                                                           uint2* __restrict__ kv, long recs) {
  typedef typename std::conditional<(MB > 256), uint16_t, uint8_t>::type bin_t;
  __shared__ int s_cur[MB];      // where the chunk's next record of a bin goes (global position)
  __shared__ __attribute__((aligned(16))) int s_tcnt[MB];  // records of the tile per bin, then their first place in the tile
  __shared__ int s_gofs[MB];     // global position of the tile's first record of a bin minus its place in the tile
  __shared__ uint32_t s_key[GP_TILE], s_val[GP_TILE];
  __shared__ bin_t s_bin[HT ? GP_TILE : 4];  // HT: the bin is not a shift of the key
  __shared__ uint32_t s_wsum[1];
  const int chunk = blockIdx.x, side = blockIdx.y, pair = blockIdx.z;
  const int tid = threadIdx.x;
  const int32_t* tab = tabs + (long)(pair * 2 + side) * g.nbins * g.nchunk;
  for (int i = tid; i < g.nbins; i += GPS_THREADS) s_cur[i] = tab[i * g.nchunk + chunk];
  uint2* k = kv + pair * recs + side * (recs / 2);
  const int y0 = GPC_R + chunk * g.rows_per_chunk, y1 = min(y0 + g.rows_per_chunk, H - GPC_R);
  const long img = pair * codes_stride + (long)side * H * W;
  const uint32_t* im = codes + img;
  const uint8_t* cm = cand ? cand + img : nullptr;
  constexpr int PPT = GP_TILE / GPS_THREADS;
  constexpr int NG = GP_GROUPS;  // groups of four consecutive pixels (one 16-byte load each) a thread takes per tile
  const int qend = y1 * W;
  uint4 nxt[NG];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int qb = y0 * W + 4 * (gi * GPS_THREADS + tid);
    nxt[gi] = make_uint4(0u, 0u, 0u, 0u);
    if (qb < qend) nxt[gi] = *reinterpret_cast<const uint4*>(im + qb);
  }
  for (int q0 = y0 * W; q0 < qend; q0 += GP_TILE) {
    for (int i = tid; i < MB; i += GPS_THREADS) s_tcnt[i] = 0;
    __syncthreads();
    uint32_t c[PPT], pix[PPT], bin[PPT];
    int lr[PPT];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      const int qb = q0 + 4 * (gi * GPS_THREADS + tid);  // four pixels of one row: W is a multiple of 16
      const uint4 c4 = nxt[gi];
      if (qb + GP_TILE < qend) nxt[gi] = *reinterpret_cast<const uint4*>(im + qb + GP_TILE);  // the next tile's, under this one's work
      c[4 * gi + 0] = c4.x;
      c[4 * gi + 1] = c4.y;
      c[4 * gi + 2] = c4.z;
      c[4 * gi + 3] = c4.w;
      const int yy = (qb < qend && (HT || cm)) ? divw((uint32_t)qb, wd) : 0;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const int i = 4 * gi + k4, q = qb + k4;
        lr[i] = -1;
        pix[i] = (uint32_t)q;
        bin[i] = 0u;
        if (qb < qend && g_is_record(c[i], cm ? cm + (long)yy * W : nullptr, q - yy * W, W)) {
          bin[i] = gp_bin<HT>(c[i], yy, g);
          lr[i] = atomicAdd(&s_tcnt[bin[i]], 1);
        }
      }
    }
    __syncthreads();
    if (tid < 64) {  // one wave scans: MB / 64 consecutive bins per lane
      constexpr int Q = MB / 256;  // groups of four bins per lane
      int4 cnt[Q];
      uint32_t sum = 0u;
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        cnt[q] = *reinterpret_cast<int4*>(&s_tcnt[4 * (Q * tid + q)]);
        sum += (uint32_t)(cnt[q].x + cnt[q].y + cnt[q].z + cnt[q].w);
      }
      const uint32_t incl = wave_incl_scan(sum);
      int run = (int)(incl - sum);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        int4 ex;
        ex.x = run;
        ex.y = ex.x + cnt[q].x;
        ex.z = ex.y + cnt[q].y;
        ex.w = ex.z + cnt[q].z;
        run = ex.w + cnt[q].w;
        *reinterpret_cast<int4*>(&s_tcnt[4 * (Q * tid + q)]) = ex;  // first place of the bin in the tile
      }
      if (tid == 63) s_wsum[0] = incl;
    }
    __syncthreads();
    const int ntile = (int)s_wsum[0];
#pragma unroll
    for (int i = 0; i < PPT; ++i)
      if (lr[i] >= 0) {
        const int place = s_tcnt[bin[i]] + lr[i];
        s_key[place] = c[i];
        s_val[place] = pix[i];
        if (HT) s_bin[place] = (bin_t)bin[i];
      }
    for (int i = tid; i < g.nbins; i += GPS_THREADS) s_gofs[i] = s_cur[i] - s_tcnt[i];
    __syncthreads();
    for (int i = tid; i < g.nbins; i += GPS_THREADS) {  // advance the chunk's cursors by what this tile holds of the bin
      const int nx = (i + 1 < MB) ? s_tcnt[i + 1] : ntile;
      s_cur[i] += nx - s_tcnt[i];
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int j = i * GPS_THREADS + tid;
      if (j < ntile) {
        const uint32_t kk = s_key[j];
        const int pos = s_gofs[HT ? (uint32_t)s_bin[j] : kk >> g.bshift] + j;
        k[pos] = make_uint2(kk, s_val[j]);  // (one 8-byte store: half as many partial cache lines as two 4-byte arrays)
      }
    }
    __syncthreads();
  }
}

// Matches of the partitions (staged by the join as the two pixel indices) -> the caller's
// array in partition order.  grid: (ceil(pmax / GPG_PARTS), npairs)
#define GPG_PARTS 2
__global__ __launch_bounds__(RM_THREADS) void k_gp_gather(const uint32_t* __restrict__ staged, const int32_t* __restrict__ part,
                                                          GpLayout g, long recs, GpcDivW wd,
                                                          int mode, void* __restrict__ out, long out_stride_bytes, int cap,
                                                          int32_t* __restrict__ counts, const int32_t* __restrict__ stats,
                                                          int32_t* __restrict__ ncand) {
  const int pair = blockIdx.y;
  const int32_t* blk = part + pair * g.ps;
  const int nparts = blk[g.o_misc + GP_NPARTS];
  const int p0 = blockIdx.x * GPG_PARTS;
  const bool last_wg = blockIdx.x == gridDim.x - 1;
  if (p0 >= nparts && !last_wg) return;
  const int32_t* rc = blk + g.o_rowcnt;
  int off = block_prefix_rows(rc, 0, min(p0, nparts));
  const int pend = min(p0 + GPG_PARTS, nparts);
  const uint2* st = reinterpret_cast<const uint2*>(staged) + pair * (recs / 2);
  char* o = reinterpret_cast<char*>(out) + pair * out_stride_bytes;
  for (int p = p0; p < pend; ++p) {
    const int cnt = rc[p], ol = blk[g.o_off + p];
    for (int i = threadIdx.x; i < cnt; i += RM_THREADS) {
      const int pos = off + i;
      if (pos >= cap) break;
      const uint2 v = st[ol + i];
      const uint32_t kl = v.x, kr = v.y;
      const int yl = divw(kl, wd), yr = divw(kr, wd);
      const int xl = (int)kl - yl * wd.W, xr = (int)kr - yr * wd.W;
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
  if (last_wg && threadIdx.x == 0) {
    int total = off;
    for (int p = pend; p < nparts; ++p) total += rc[p];  // (pend == nparts for the last workgroup: nothing to add)
    counts[pair] = total;
    if (ncand) {
      ncand[2 * pair + 0] = stats[(pair * 2 + 0) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
      ncand[2 * pair + 1] = stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_NCAND];
    }
  }
}

}  // namespace gpc
