// k_train.h -- the fern TRAINING scoring loop.
//
// Replaces Fern::evalSplit (Fern.hpp:209-262), Fern::markSplitSamples (Fern.hpp:271-291) and the
// candidate loop of Fern::train (Fern.hpp:336-351) over a device-resident set of patch triplets.
//
// Layout ("train set"): the reference keeps a triplet as three 27x27 byte patches (729 bytes each,
// file order of storeAllTriplets, Feature.hpp:247-256) and a test reads two bytes of each patch.
// On the device the set is stored TRANSPOSED -- planes[patch][pixel][triplet], one byte each -- so
// that a test (i, j) reads six contiguous byte rows, 4 triplets per 32-bit load: 6 bytes of HBM
// traffic per triplet and candidate instead of six 64-byte lines.
//   flags[t]: bit 0 pos.split, bit 1 neg.split (GPCDescriptor::split, Feature.hpp:65),
//             bit 2 "ref == pos on all committed levels", bit 3 "ref != neg on some committed level",
//             bit 7 padding entry (no sample; marks forced to 3 so that nothing ever counts it).
// While level L of a fern is searched, the levels 0..L-1 are fixed, so the comparison of the three
// code words reduces to the two cached bits and the candidate's own three decisions.  Code words are
// 64 bits in the reference: the host limits a fern to 64 levels, for which the bits are exact.
#pragma once
#include "gpc_device.h"

namespace gpc {

#define TS_PATCH 729
#define TS_THREADS 256
#define TS_ITER 1024      // triplets per workgroup and loop iteration of the evaluation kernel (4 per lane)
#define TS_MAXTAU 64      // intercept values searched per candidate

#define TSF_POS 1u
#define TSF_NEG 2u
#define TSF_EQ 4u
#define TSF_NE 8u
#define TSF_PAD 128u

struct GpcSplit {
  int32_t i, j, tau;
};

// ---- host layout [n][3][729] -> planes [3][729][np]   (np = n rounded up to a multiple of 256)
// grid: (np / 64, 3, 3): 64 triplets x one patch x 243 of its 729 bytes
__global__ __launch_bounds__(TS_THREADS) void k_ts_transpose(const uint8_t* __restrict__ aos, int n, long np,
                                                             uint8_t* __restrict__ planes) {
  __shared__ uint8_t s[64][244];
  const int t0 = blockIdx.x * 64, patch = blockIdx.y, c0 = blockIdx.z * 243;
  for (int e = threadIdx.x; e < 64 * 243; e += TS_THREADS) {
    const int tt = e / 243, c = e - tt * 243;
    const int t = t0 + tt;
    s[tt][c] = (t < n) ? aos[((long)t * 3 + patch) * TS_PATCH + c0 + c] : (uint8_t)0;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 243; e += TS_THREADS) {
    const int c = e >> 6, tt = e & 63;
    planes[((long)patch * TS_PATCH + c0 + c) * np + t0 + tt] = s[tt][c];
  }
}

// start of a fern: no level committed (ref == pos == neg == 0); optionally resetMarkOnSamples (Fern.hpp:296-301)
__global__ __launch_bounds__(TS_THREADS) void k_ts_begin(uint8_t* __restrict__ flags, int n, long np, int reset_marks) {
  const long t = (long)blockIdx.x * TS_THREADS + threadIdx.x;
  if (t >= np) return;
  uint32_t f = flags[t];
  if (t >= n) f = TSF_PAD | TSF_POS | TSF_NEG;
  else f = (reset_marks ? 0u : (f & (TSF_POS | TSF_NEG)));
  flags[t] = (uint8_t)(f | TSF_EQ);
}

__device__ __forceinline__ bool ts_dec(uint32_t a, uint32_t b, int tau) { return ((int)a - (int)b) < tau; }

// ---- candidates of ONE level: tp / fp per (candidate, tau); blockIdx.y = candidate, blockIdx.x = chunk of
// iters * 1024 triplets (the host picks iters: every workgroup ends with 2 * ntau global atomics, so
// chunks are as long as the grid stays large enough to fill the device)
// tp[c * ntau + k], fp[...] must be zero on entry.  The samples that count (tot) do not depend on
// the candidate: k_ts_tot.  fn = tot - tp - fp.
__global__ __launch_bounds__(TS_THREADS) void k_ts_eval_level(const uint8_t* __restrict__ planes,
                                                              const uint8_t* __restrict__ flags, long np,
                                                              const GpcSplit* __restrict__ cand, int taulo, int ntau, int iters,
                                                              int32_t* __restrict__ tp, int32_t* __restrict__ fp) {
  __shared__ int s_tp[TS_MAXTAU], s_fp[TS_MAXTAU];
  const int tid = threadIdx.x, lane = tid & 63;
  if (tid < TS_MAXTAU) { s_tp[tid] = 0; s_fp[tid] = 0; }
  __syncthreads();
  const GpcSplit cd = cand[blockIdx.y];
  const uint8_t* ri = planes + (long)cd.i * np;
  const uint8_t* rj = planes + (long)cd.j * np;
  const uint8_t* pi = ri + (long)TS_PATCH * np;
  const uint8_t* pj = rj + (long)TS_PATCH * np;
  const uint8_t* ni = pi + (long)TS_PATCH * np;
  const uint8_t* nj = pj + (long)TS_PATCH * np;
  const long c0 = (long)blockIdx.x * iters * TS_ITER;  // a workgroup owns `iters` x 1024 consecutive triplets
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    const long t = c0 + ((long)it * TS_THREADS + tid) * 4;  // 4 triplets per lane (np is a multiple of 256... and of 4)
    uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0, d0 = 0, d1 = 0, fl = 0x03030303u;
    if (t < np) {
      a0 = *reinterpret_cast<const uint32_t*>(ri + t);
      a1 = *reinterpret_cast<const uint32_t*>(rj + t);
      b0 = *reinterpret_cast<const uint32_t*>(pi + t);
      b1 = *reinterpret_cast<const uint32_t*>(pj + t);
      d0 = *reinterpret_cast<const uint32_t*>(ni + t);
      d1 = *reinterpret_cast<const uint32_t*>(nj + t);
      fl = *reinterpret_cast<const uint32_t*>(flags + t);
    }
    int dr[4], dp[4], dn[4];
    bool counted[4], eq[4], ne[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int sh = 8 * b;
      dr[b] = (int)((a0 >> sh) & 0xFFu) - (int)((a1 >> sh) & 0xFFu);
      dp[b] = (int)((b0 >> sh) & 0xFFu) - (int)((b1 >> sh) & 0xFFu);
      dn[b] = (int)((d0 >> sh) & 0xFFu) - (int)((d1 >> sh) & 0xFFu);
      const uint32_t f = (fl >> sh) & 0xFFu;
      counted[b] = !((f & TSF_POS) && (f & TSF_NEG));  // Fern.hpp:239
      eq[b] = (f & TSF_EQ) != 0;
      ne[b] = (f & TSF_NE) != 0;
    }
    if (ntau == 1) {  // zero optimizer: one intercept, the kernel is a pure byte stream
      int wtp = 0, wfp = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const bool r = dr[b] < taulo, p = dp[b] < taulo, q = dn[b] < taulo;  // Feature::getDecisions, Feature.hpp:101-109
        const bool e = eq[b] && (r == p), d = ne[b] || (r != q);
        wtp += __popcll(__ballot(counted[b] && e && d));                     // ref == pos, ref != neg
        wfp += __popcll(__ballot(counted[b] && !e && !d));                   // ref != pos, ref == neg
      }
      if (lane == 0) {
        if (wtp) atomicAdd(&s_tp[0], wtp);
        if (wfp) atomicAdd(&s_fp[0], wfp);
      }
    } else {
      // All intercepts at once: a decision x(i) - x(j) < tau switches on at one intercept, so over the
      // index k (tau = taulo + k) it is the bit mask ~0 << (diff - taulo + 1), and the class of a triplet
      // is a few AND / XORs of three such masks.  A class mask changes at <= 4 places: instead of
      // counting every k, each change is added to a difference array (a prefix sum at the end gives the
      // count per k); the changes at k = 0 -- most of them -- are counted with a ballot, not an atomic.
      const unsigned long long valid = (ntau >= 64) ? ~0ull : ((1ull << ntau) - 1ull);
      int base_tp = 0, base_fp = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        auto step = [&](int diff) -> unsigned long long {
          const int k = diff - taulo + 1;  // first index whose intercept exceeds diff
          return k <= 0 ? ~0ull : (k >= 64 ? 0ull : (~0ull << k));
        };
        const unsigned long long rm = step(dr[b]), pm = step(dp[b]), qm = step(dn[b]);
        const unsigned long long am = rm ^ pm, bm = rm ^ qm;  // ref != pos ; ref != neg on this level
        unsigned long long tpm = 0ull, fpm = 0ull;
        if (counted[b]) {
          const unsigned long long e = eq[b] ? ~am : 0ull, d = ne[b] ? ~0ull : bm;
          tpm = (e & d) & valid;
          fpm = (~e & ~d) & valid;
        }
        base_tp += __popcll(__ballot((tpm & 1ull) != 0ull));
        base_fp += __popcll(__ballot((fpm & 1ull) != 0ull));
        // rising / falling edges at k >= 1
        unsigned long long up = tpm & ~(tpm << 1) & ~1ull, dn_ = ~tpm & (tpm << 1) & valid;
        while (up) { atomicAdd(&s_tp[__builtin_ctzll(up)], 1); up &= up - 1ull; }
        while (dn_) { atomicAdd(&s_tp[__builtin_ctzll(dn_)], -1); dn_ &= dn_ - 1ull; }
        up = fpm & ~(fpm << 1) & ~1ull;
        dn_ = ~fpm & (fpm << 1) & valid;
        while (up) { atomicAdd(&s_fp[__builtin_ctzll(up)], 1); up &= up - 1ull; }
        while (dn_) { atomicAdd(&s_fp[__builtin_ctzll(dn_)], -1); dn_ &= dn_ - 1ull; }
      }
      if (lane == 0) {
        if (base_tp) atomicAdd(&s_tp[0], base_tp);
        if (base_fp) atomicAdd(&s_fp[0], base_fp);
      }
    }
  }
  __syncthreads();
  if (ntau > 1 && tid < 2) {  // difference arrays -> counts per intercept
    int* a = tid ? s_fp : s_tp;
    int acc = 0;
    for (int k = 0; k < ntau; ++k) {
      acc += a[k];
      a[k] = acc;
    }
  }
  __syncthreads();
  if (tid < ntau) {
    if (s_tp[tid]) atomicAdd(&tp[blockIdx.y * ntau + tid], s_tp[tid]);
    if (s_fp[tid]) atomicAdd(&fp[blockIdx.y * ntau + tid], s_fp[tid]);
  }
}

// samples evalSplit counts at all: !(pos.split && neg.split)   (*tot zero on entry)
__global__ __launch_bounds__(TS_THREADS) void k_ts_tot(const uint8_t* __restrict__ flags, long np, int32_t* __restrict__ tot) {
  const long t = ((long)blockIdx.x * TS_THREADS + threadIdx.x) * 4;
  int c = 0;
  if (t < np) {
    const uint32_t fl = *reinterpret_cast<const uint32_t*>(flags + t);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t f = (fl >> (8 * b)) & 0xFFu;
      c += !((f & TSF_POS) && (f & TSF_NEG));
    }
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(tot, c);
}

// the level's winner joins the fixed levels; before that, markSplitSamples over the levels fixed so far
// (Fern.hpp:355-356 passes `level`, i.e. the levels BEFORE the one just chosen)
__global__ __launch_bounds__(TS_THREADS) void k_ts_commit(const uint8_t* __restrict__ planes, uint8_t* __restrict__ flags,
                                                          int n, long np, GpcSplit best, int mark_split) {
  const long t = (long)blockIdx.x * TS_THREADS + threadIdx.x;
  if (t >= n) return;
  uint32_t f = flags[t];
  if (mark_split) {
    if (f & TSF_EQ) f |= TSF_POS;
    if (f & TSF_NE) f |= TSF_NEG;
  }
  const long oi = (long)best.i * np + t, oj = (long)best.j * np + t;
  const long pp = (long)TS_PATCH * np;
  const bool r = ts_dec(planes[oi], planes[oj], best.tau);
  const bool p = ts_dec(planes[oi + pp], planes[oj + pp], best.tau);
  const bool q = ts_dec(planes[oi + 2 * pp], planes[oj + 2 * pp], best.tau);
  if (r != p) f &= ~TSF_EQ;
  if (r != q) f |= TSF_NE;
  flags[t] = (uint8_t)f;
}

// ---- the general forms, for any parameter list (nothing cached): Fern::evalSplit / markSplitSamples
// counts: tp, fp, fn, tot (zero on entry).  nparams <= 64.
template <bool MARK>
__global__ __launch_bounds__(TS_THREADS) void k_ts_eval_split(const uint8_t* __restrict__ planes, uint8_t* __restrict__ flags,
                                                              int n, long np, const GpcSplit* __restrict__ params,
                                                              int nparams, int32_t* __restrict__ counts) {
  __shared__ GpcSplit s_p[64];
  if ((int)threadIdx.x < nparams) s_p[threadIdx.x] = params[threadIdx.x];
  __syncthreads();
  const long t = (long)blockIdx.x * TS_THREADS + threadIdx.x;
  const bool in = t < n;
  bool eq = true, ne = false;
  const long pp = (long)TS_PATCH * np;
  if (in) {
    for (int l = 0; l < nparams; ++l) {
      const GpcSplit sp = s_p[l];
      const long oi = (long)sp.i * np + t, oj = (long)sp.j * np + t;
      const bool r = ts_dec(planes[oi], planes[oj], sp.tau);
      const bool p = ts_dec(planes[oi + pp], planes[oj + pp], sp.tau);
      const bool q = ts_dec(planes[oi + 2 * pp], planes[oj + 2 * pp], sp.tau);
      eq = eq && (r == p);
      ne = ne || (r != q);
    }
  }
  if (MARK) {
    if (in) {
      uint32_t f = flags[t];
      if (eq) f |= TSF_POS;
      if (ne) f |= TSF_NEG;
      flags[t] = (uint8_t)f;
    }
    return;
  }
  const uint32_t f = in ? flags[t] : (TSF_POS | TSF_NEG);
  const bool counted = !((f & TSF_POS) && (f & TSF_NEG));
  const int ctp = __popcll(__ballot(counted && eq && ne));
  const int cfp = __popcll(__ballot(counted && !eq && !ne));
  const int ctot = __popcll(__ballot(counted));
  if ((threadIdx.x & 63) == 0) {
    if (ctp) atomicAdd(&counts[0], ctp);
    if (cfp) atomicAdd(&counts[1], cfp);
    if (ctot - ctp - cfp) atomicAdd(&counts[2], ctot - ctp - cfp);
    if (ctot) atomicAdd(&counts[3], ctot);
  }
}

// marks in / out for the host (bit 0 pos.split, bit 1 neg.split)
__global__ __launch_bounds__(TS_THREADS) void k_ts_set_marks(uint8_t* __restrict__ flags, const uint8_t* __restrict__ marks, int n) {
  const long t = (long)blockIdx.x * TS_THREADS + threadIdx.x;
  if (t < n) flags[t] = (uint8_t)((flags[t] & ~(TSF_POS | TSF_NEG)) | (marks[t] & (TSF_POS | TSF_NEG)));
}
__global__ __launch_bounds__(TS_THREADS) void k_ts_get_marks(const uint8_t* __restrict__ flags, uint8_t* __restrict__ marks, int n) {
  const long t = (long)blockIdx.x * TS_THREADS + threadIdx.x;
  if (t < n) marks[t] = (uint8_t)(flags[t] & (TSF_POS | TSF_NEG));
}

}  // namespace gpc
