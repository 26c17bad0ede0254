// ref_harness.cpp -- C-ABI wrappers around the REAL reference kernels.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit includes the reference's
// lib/gpc/filter.hpp from where it lies under /root/reference (nothing is copied
// into this repository) and exports its raw-pointer kernels so that the C
// restatement in gpc_oracle.c can be validated against them, and so that bench.py
// can time the reference SSE kernels as part of the CPU baseline.
//
// filter.hpp includes "gpc/buffer.hpp" only for its transitive std headers; that
// header needs Eigen, which this image does not have, so the build passes
// -D__NDB_BUFFER (buffer.hpp's own include guard, buffer.hpp:31) which makes the
// include expand to nothing.  No stand-in header is written.  inference.hpp /
// buffer.hpp themselves (Forest, Buffer<T>) cannot be built here (Eigen) -- the
// API-level behaviour is pinned by SURVEY.md Appendix C instead.
//
// Build: see oracle/Makefile (target _ref/libgpc_ref.so).
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <vector>

#include "gpc/filter.hpp"
#include "gpc/hashmatch.hpp"  // templates only; its buffer.hpp include is guarded out like filter.hpp's

extern "C" {

// filter.hpp:293
void gpc_ref_box(const uint8_t* in, uint8_t* out, int w, int h) {
  ndb::box(const_cast<uint8_t*>(in), out, w, h, 1);
}

// filter.hpp:404
void gpc_ref_sobel(const uint8_t* in, uint8_t* grad, int w, int h, int thr) {
  ndb::sobel(const_cast<uint8_t*>(in), grad, w, h, (uint8_t)thr, 1);
}

// filter.hpp:60 -- `a` must be 32-byte aligned and readable up to the next
// multiple of 32 bytes; `ind` needs n entries.
int gpc_ref_arr2ind(const uint8_t* a, int n, int32_t* ind) {
  int m = 0;
  ndb::arr2ind(a, n, ind, &m);
  return m;
}

// filter.hpp:547 / :619 -- codes must be zero-filled by the caller, as
// inference.hpp:274 does.  ntests <= 32.
void gpc_ref_hash(const uint8_t* smooth, const uint8_t* grad, uint32_t* codes,
                  const int32_t* offs, const int32_t* tau, int ntests, int type, int w,
                  int h, int nthreads) {
  std::vector<int32_t> mask(offs, offs + 2 * ntests);
  std::vector<int> idx;  // only used by the non-SSE fallback
  if (type == 0) {
    ndb::gpcFilter(const_cast<uint8_t*>(smooth), grad, codes, mask, idx, w, h, nthreads);
  } else {
    std::vector<int> t(tau, tau + ntests);
    ndb::gpcFilterTau(const_cast<uint8_t*>(smooth), grad, codes, mask, t, idx, w, h, nthreads);
  }
}

// ndb::Hashmatch<T> (hashmatch.hpp:217-272) is a template over the element type; it is
// instantiated here with a harness-owned element that offers the operators the template uses
// (<=, ==, !=, %, diffImgs) and driven exactly like depthPriorFast does (inference.hpp:204-225):
// 214673 buckets, all source elements first, then all targets, then getDuplicates.
struct HmElem {
  uint64_t state;
  int32_t k;
  bool src;
  bool operator==(const HmElem& d) const { return state == d.state; }
  bool operator!=(const HmElem& d) const { return state != d.state; }
  bool operator<=(const HmElem& d) const { return state <= d.state; }
  bool diffImgs(const HmElem& d) { return src != d.src; }
  int operator%(const int& d) const { return state % d; }
};

int gpc_ref_hashmatch(const uint64_t* ss, const int32_t* sk, int ns, const uint64_t* ts, const int32_t* tk,
                      int nt, int32_t* out_pairs /* 2 ints per pair */) {
  ndb::Hashmatch<HmElem> hm(214673, ns + nt);
  for (int i = 0; i < ns; ++i) { HmElem e = {ss[i], sk[i], true}; hm.insert(e); }
  for (int i = 0; i < nt; ++i) { HmElem e = {ts[i], tk[i], false}; hm.insert(e); }
  std::vector<std::pair<HmElem, HmElem>> corr;
  hm.getDuplicates(corr);
  for (size_t i = 0; i < corr.size(); ++i) {
    out_pairs[2 * i] = corr[i].first.k;
    out_pairs[2 * i + 1] = corr[i].second.k;
  }
  return (int)corr.size();
}

// same, with the candidate index list (only the -DSSE=OFF build reads it)
void gpc_ref_hash_idx(const uint8_t* smooth, const uint8_t* grad, uint32_t* codes, const int32_t* offs,
                      const int32_t* tau, int ntests, int type, int w, int h, const int32_t* idx, int nidx) {
  std::vector<int32_t> mask(offs, offs + 2 * ntests);
  std::vector<int> ix(idx, idx + nidx);
  if (type == 0) {
    ndb::gpcFilter(const_cast<uint8_t*>(smooth), grad, codes, mask, ix, w, h, 1);
  } else {
    std::vector<int> t(tau, tau + ntests);
    ndb::gpcFilterTau(const_cast<uint8_t*>(smooth), grad, codes, mask, t, ix, w, h, 1);
  }
}

int gpc_ref_is_sse(void) {
#ifdef _INTRINSICS_SSE
  return 1;
#else
  return 0;
#endif
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// CPU baseline for bench.py: the timed region of samples/sparsematch.cpp:45-52 with the
// reference's REAL SSE kernels (above) and a C++ port of the inference.hpp glue around
// them that keeps the reference's data structures and algorithms, so that its run time is
// representative of the reference binary: 32-byte aligned zero-filled images, a zero-filled
// uint32 code image per evaluation (inference.hpp:274), 24-byte descriptors
// {Point, uint64 state, bool} (buffer.hpp:58-62), std::sort on them (inference.hpp:231-233),
// the merge scan (:236-252) and the disparity filter (:384-391).  Single thread, like the
// reference's default (inference.hpp:89).  Its output is tested against the oracle.
// ---------------------------------------------------------------------------------------
namespace {

struct Pt { int x, y; };
struct Desc {
  Pt point;
  uint64_t state;
  bool srcDescr;
  bool operator<(const Desc& d) const { return state < d.state; }
};
static_assert(sizeof(Desc) == 24, "descriptor layout of buffer.hpp:58-62");

struct Img {
  uint8_t* base;
  uint8_t* p;
  explicit Img(size_t n) {
    base = (uint8_t*)aligned_alloc(32, ((n + 128 + 31) / 32) * 32);
    memset(base, 0, ((n + 128 + 31) / 32) * 32);
    p = base + 64;  // slack in front: the SSE kernels read in[-1]
  }
  ~Img() { free(base); }
  Img(const Img&) = delete;
};

struct Pre {
  Img smooth, grad;
  std::vector<int> mask;
  explicit Pre(size_t n) : smooth(n), grad(n) {}
};

void preprocess(const uint8_t* raw, int W, int H, int thr, Pre& out) {
  const size_t n = (size_t)W * H;
  Img in(n);
  memcpy(in.p, raw, n);
  ndb::box(in.p, out.smooth.p, W, H, 1);
  uint8_t* s = out.smooth.p;  // clearBoundary, buffer.hpp:630-654
  for (int y = 0; y < H; ++y) { s[(size_t)y * W] = 0; s[(size_t)y * W + 1] = 0; s[(size_t)y * W + W - 1] = 0; }
  memset(s, 0, W);
  memset(s + (size_t)(H - 2) * W, 0, (size_t)2 * W);
  ndb::sobel(in.p, out.grad.p, W, H, (uint8_t)thr, 1);
  std::vector<int> idx(n + 64);
  int m = 0;
  ndb::arr2ind(out.grad.p, (int)n, idx.data(), &m);
  out.mask.clear();
  for (int i = 0; i < m; ++i) {  // inference.hpp:318-325
    int x = idx[i] % W, y = idx[i] / W;
    if (y >= 13 && y < H - 13 && x >= 13 && x < W - 13) out.mask.push_back(idx[i]);
  }
}

std::vector<Desc> evaluate(Pre& im, const std::vector<int32_t>& mask, const std::vector<int>& tau, int type,
                           int W, int H, bool epipolar) {
  std::vector<uint32_t> codes((size_t)W * H, 0);
  std::vector<int> idx;
  if (type == 0) ndb::gpcFilter(im.smooth.p, im.grad.p, codes.data(), mask, idx, W, H, 1);
  else ndb::gpcFilterTau(im.smooth.p, im.grad.p, codes.data(), mask, tau, idx, W, H, 1);
  std::vector<Desc> out(im.mask.size());
  size_t j = 0;
  for (int k : im.mask) {
    Desc d;
    d.point.x = k % W;
    d.point.y = k / W;
    d.state = codes[k];
    d.srcDescr = false;
    if (epipolar) d.state |= (uint64_t)d.point.y << 32;
    out[j++] = d;
  }
  return out;
}

}  // namespace

extern "C" int gpc_ref_cpu_baseline_pair(const uint8_t* rawL, const uint8_t* rawR, int W, int H,
                                         const int32_t* offs, const int32_t* tau, int ntests, int type,
                                         int thr, int disp_high, int vtol, int epipolar,
                                         int32_t* out_xyd /* 3 ints per support */, int cap,
                                         double* ms_pre, double* ms_match) {
  typedef std::chrono::high_resolution_clock clk;
  std::vector<int32_t> mask(offs, offs + 2 * ntests);
  std::vector<int> taus(tau, tau + ntests);
  const size_t n = (size_t)W * H;
  auto t0 = clk::now();
  Pre L(n), R(n);
  preprocess(rawL, W, H, thr, L);
  preprocess(rawR, W, H, thr, R);
  auto t1 = clk::now();
  std::vector<Desc> S = evaluate(L, mask, taus, type, W, H, epipolar != 0);
  std::vector<Desc> T = evaluate(R, mask, taus, type, W, H, epipolar != 0);
  std::sort(S.begin(), S.end());
  std::sort(T.begin(), T.end());
  int count = 0;
  if (!T.empty()) {
    size_t j = 0;
    const size_t last = T.size() - 1;
    for (size_t i = 0; i < S.size(); ++i) {
      bool unique = true;
      while (i + 1 < S.size() && S[i].state == S[i + 1].state) { ++i; unique = false; }
      if (!unique) continue;
      while (j < last && T[j].state < S[i].state) ++j;
      if (j != last && T[j].state == S[i].state && (j + 1 == last || T[j + 1].state != T[j].state)) {
        const int dy = S[i].point.y - T[j].point.y, dx = S[i].point.x - T[j].point.x;
        if (std::abs(dy) <= vtol && std::abs(dx) <= disp_high) {
          if (count < cap) {
            out_xyd[3 * count] = S[i].point.x;
            out_xyd[3 * count + 1] = S[i].point.y;
            out_xyd[3 * count + 2] = dx;
          }
          ++count;
        }
      }
    }
  }
  auto t2 = clk::now();
  if (ms_pre) *ms_pre = std::chrono::duration<double, std::milli>(t1 - t0).count();
  if (ms_match) *ms_match = std::chrono::duration<double, std::milli>(t2 - t1).count();
  return count;
}
