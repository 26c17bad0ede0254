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
#include <cstdint>
#include <cstring>
#include <functional>
#include <iostream>
#include <vector>

#include "gpc/filter.hpp"

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

int gpc_ref_is_sse(void) {
#ifdef _INTRINSICS_SSE
  return 1;
#else
  return 0;
#endif
}

}  // extern "C"
