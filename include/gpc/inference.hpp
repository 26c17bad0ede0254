// inference.hpp -- gpc::inference::{InferenceSettings, Forest} on MI355X.
//
// Host-side mirror of the reference's lib/gpc/inference.hpp: same class, method and field
// names, same by-value conventions, same messages on stdout, so that a caller such as
// samples/sparsematch.cpp compiles unchanged.  Every hot method forwards to the C ABI of
// libgpc_hip.so (include/gpc_hip.h) exactly where the reference calls its SSE kernels:
//
//   readForest       (inference.hpp:404-446)  -> gpc_hip_read_forest (host parser) + gpc_hip_warmup: the reference's sample
//                                                calls it BEFORE its clock starts (samples/sparsematch.cpp:42-45) and it is
//                                                the one call that knows the image size, so the device context, the code
//                                                objects and every workspace of that size are made here, not in the first
//                                                timed call (GPC_HIP_NO_WARMUP=1: a host parser only, as in the reference)
//   preprocessImage  (inference.hpp:302-333)  -> gpc_hip_preprocess_begin / _fetch (the image also stays on the device:
//                                                a PreprocessedImage handed to rectifiedMatch / stereoMatch unchanged is
//                                                matched from there, include/gpc_hip.h "Resident images")
//   stereoMatch      (inference.hpp:344-361)  -> gpc_hip_stereo_match
//   rectifiedMatch   (inference.hpp:375-393)  -> gpc_hip_rectified_match
//   matchPair        (extension: the whole t0..t2 region of sparsematch.cpp:45-52 in one
//                     call, raw images in, supports out -> gpc_hip_match_pair)
//
// There is no CPU fallback and, like the reference, no error channel on these methods: a call that cannot run (no gfx950
// device, a HIP error) prints the reason to std::cout and returns an EMPTY result -- the reference's own convention for
// what it can report (readPNG -> 1, readForest -> empty mask).  GPC_HIP_ABORT_ON_ERROR=1 aborts the process instead.
// One device context per host thread (thread_local), device chosen by GPC_HIP_DEVICE.
#ifndef GPC_AMD_INFERENCE_HPP
#define GPC_AMD_INFERENCE_HPP

#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "gpc/buffer.hpp"
#include "gpc_hip.h"

using namespace std;  // the reference's headers do this; callers rely on it (sparsematch.cpp)

namespace gpc {
namespace inference {

typedef std::chrono::high_resolution_clock::time_point time_point;
inline time_point sysTick() { return std::chrono::high_resolution_clock::now(); }
inline float tickToMs(time_point t0, time_point t1) {
  return (float)std::abs(1000. * std::chrono::duration_cast<std::chrono::duration<double>>(t1 - t0).count());
}

struct InferenceSettings {
  uint8_t gradientThreshold_ = 10;
  int dispHigh_ = 128;
  int verticalTolerance_ = 1;
  bool epipolarMode_ = false;
  bool useHashtable_ = false;
  int numThreads_ = 1;

  InferenceSettings(uint8_t gradientThreshold, int dispHigh, int verticalTolerance, bool epipolarMode,
                    bool useHashtable, int numThreads)
      : gradientThreshold_(gradientThreshold), dispHigh_(dispHigh), verticalTolerance_(verticalTolerance),
        epipolarMode_(epipolarMode), useHashtable_(useHashtable), numThreads_(numThreads) {}
  InferenceSettings() {}
  InferenceSettings& builder(void) { return *this; }
  InferenceSettings& gradientThreshold(uint8_t v) { gradientThreshold_ = v; return *this; }
  InferenceSettings& dispHigh(int v) { dispHigh_ = v; return *this; }
  InferenceSettings& verticalTolerance(int v) { verticalTolerance_ = v; return *this; }
  InferenceSettings& epipolarMode(bool v) { epipolarMode_ = v; return *this; }
  InferenceSettings& useHashtable(bool v) { useHashtable_ = v; return *this; }
  InferenceSettings& numThreads(int v) {  // clamped like inference.hpp:122-128; unused on the GPU
    const int hw = (int)std::thread::hardware_concurrency();
    numThreads_ = (v > hw) ? hw : v;
    return *this;
  }
  gpc_settings toC() const {
    gpc_settings s;
    s.gradient_threshold = gradientThreshold_;
    s.disp_high = dispHigh_;
    s.vertical_tolerance = verticalTolerance_;
    s.epipolar_mode = epipolarMode_ ? 1 : 0;
    s.use_hashtable = useHashtable_ ? 1 : 0;
    s.num_threads = numThreads_;
    return s;
  }
};

namespace detail {
struct ContextHolder {
  gpc_hip_ctx* ctx = nullptr;
  // FilterMask last uploaded (the reference passes it by value into every call)
  gpc_filter_mask uploaded;
  bool have = false;
  size_t support_hint = 0;  // supports of this thread's last matchPair call (+ slack): how large the next result array starts
  gpc_filter_mask warmed;   // the forest (with its image size) readForest last warmed this thread's context for
  bool have_warmed = false;
  // matchPair's page-locked staging: the two images go in and the supports come out through memory the device reads and
  // writes directly (ndb::Buffer and std::vector are pageable: the runtime would stage every copy itself, synchronously)
  void* pin_in = nullptr;
  void* pin_out = nullptr;
  size_t pin_in_cap = 0, pin_out_cap = 0;
  bool pinned(void** p, size_t* cap, size_t bytes) {
    if (bytes <= *cap) return true;
    if (*p) gpc_hip_host_free(ctx, *p);
    *p = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 4;
    if (gpc_hip_host_alloc(ctx, want, p) != GPC_OK) return false;
    *cap = want;
    return true;
  }
  ~ContextHolder() {
    if (ctx && pin_in) gpc_hip_host_free(ctx, pin_in);
    if (ctx && pin_out) gpc_hip_host_free(ctx, pin_out);
    if (ctx) gpc_hip_destroy(ctx);
  }
};
// Status of the last gpc:: call on this thread that reached the library (GPC_OK = 0): the reference's API returns results
// by value and has no error channel, so a failed call returns an EMPTY result -- which also is what "no matches" looks
// like.  A pipeline that must tell the two apart asks gpc::inference::lastStatus() / lastError() after the call.
inline int& last_status() {
  static thread_local int st = GPC_OK;
  return st;
}
inline std::string& last_error() {
  static thread_local std::string msg;
  return msg;
}
inline void fail(int st, gpc_hip_ctx* ctx, const char* what) {
  last_status() = st;
  last_error() = std::string(what) + " failed: " + gpc_hip_status_string(st) +
                 ((st == GPC_E_HIP && ctx) ? std::string(" (") + gpc_hip_last_error(ctx) + ")" : std::string());
  std::cout << "gpc_hip: " << what << " failed: " << gpc_hip_status_string(st);
  if (st == GPC_E_HIP && ctx) std::cout << " (" << gpc_hip_last_error(ctx) << ")";
  std::cout << std::endl;
  if (std::getenv("GPC_HIP_ABORT_ON_ERROR")) std::abort();
}
// quiet: a context that cannot be made is not reported (readForest's warm-up: the call that needs the device will say so)
inline ContextHolder& holder(bool quiet = false) {
  static thread_local ContextHolder h;
  if (!h.ctx) {
    const char* dev = std::getenv("GPC_HIP_DEVICE");
    const int st = gpc_hip_create(dev ? std::atoi(dev) : 0, &h.ctx);
    if (st != GPC_OK) {
      if (!quiet) fail(st, nullptr, "gpc_hip_create");
      h.ctx = nullptr;
      return h;  // the caller returns an empty result
    }
    // Like the reference, the arithmetic variant is a build-time choice of the CALLER's
    // translation unit: -D_INTRINSICS_SSE (the reference's default, samples/CMakeLists.txt:13-17)
    // selects the SSE-exact kernels, its absence the *Naive ones (filter.hpp:157-282).
#ifdef _INTRINSICS_SSE
    gpc_hip_set_arithmetic(h.ctx, GPC_ARITH_SSE);
#else
    gpc_hip_set_arithmetic(h.ctx, GPC_ARITH_NAIVE);
#endif
  }
  return h;
}
}  // namespace detail

// Extension (the reference has no error channel): status / message of this thread's last failed gpc:: call, GPC_OK (0)
// and "" when none failed since clearStatus().  An empty result with lastStatus() == 0 means "no matches".
inline int lastStatus() { return detail::last_status(); }
inline const std::string& lastError() { return detail::last_error(); }
inline void clearStatus() {
  detail::last_status() = GPC_OK;
  detail::last_error().clear();
}

class Forest {
 public:
  struct FilterMask {
    std::vector<int32_t> mask;
    std::vector<int> tau;
    int width;
    int height;
    int type;
    FilterMask(std::vector<int32_t> mask, int width, int height, int type)
        : mask(mask), width(width), height(height), type(type) {}
    FilterMask(std::vector<int32_t> mask, std::vector<int> tau, int width, int height, int type)
        : mask(mask), tau(tau), width(width), height(height), type(type) {}
  };
  struct PreprocessedImage {
    ndb::Buffer<uint8_t> smooth;
    ndb::Buffer<uint8_t> grad;
    std::vector<int> mask;
    PreprocessedImage(ndb::Buffer<uint8_t>& smooth, ndb::Buffer<uint8_t>& grad, std::vector<int>& mask)
        : smooth(smooth), grad(grad), mask(mask) {}
    PreprocessedImage() {}  // what a call that could not run returns: no candidates
  };
  enum CorrMethod { sorting = 's', hashtable = 'h' };

  // inference.hpp:404-446
  FilterMask readForest(std::string path, int width, int height) {
    gpc_filter_mask fm;
    const int st = gpc_hip_read_forest(path.c_str(), width, height, &fm);
    if (st == GPC_E_IO && fm.num_tests == 0 && fm.discarded == 0) {
      cout << "Error opening forest file" << endl;
      return FilterMask(std::vector<int32_t>(), width, height, 0);
    }
    // the reference prints the fern count and one note per discarded test (:417, :431)
    cout << "number of ferns:" << countFerns(path) << endl;
    for (int i = 0; i < fm.discarded; ++i)
      cout << "Note: A maximum of 32 fern features are allowed, discarding remainder of forest." << endl;
    std::vector<int32_t> mask(fm.mask, fm.mask + 2 * fm.num_tests);
    std::vector<int> tau;
    if (fm.type != 0) tau.assign(fm.tau, fm.tau + fm.num_tests);
    FilterMask result = fm.type == 0 ? FilterMask(mask, width, height, 0) : FilterMask(mask, tau, width, height, 1);
    if (st == GPC_OK && fm.num_tests > 0 && !std::getenv("GPC_HIP_NO_WARMUP")) warmUp(result);
    return result;
  }

  // What the reference's caller does next -- preprocessImage x2, rectifiedMatch / matchPair on images of this size --
  // done once here on a synthetic image and thrown away: context, code objects, workspaces and page-locked staging
  // (gpc_hip_warmup), and this thread's own staging and the allocator's state for result arrays of this size (the dry run
  // below).  Says nothing and leaves no status when it cannot run (no device: the first real call reports that).
  void warmUp(FilterMask& forestmask) {
    detail::ContextHolder& h = detail::holder(true);
    if (!h.ctx) return;
    gpc_filter_mask key;
    if (!toC(forestmask, key)) return;
    if (h.have_warmed && memcmp(&h.warmed, &key, sizeof key) == 0) return;
    if (gpc_hip_set_forest(h.ctx, &key) != GPC_OK) return;
    h.uploaded = key;
    h.have = true;
    if (gpc_hip_warmup(h.ctx, forestmask.width, forestmask.height, nullptr) != GPC_OK) return;
    ndb::Buffer<uint8_t> img(forestmask.height, forestmask.width);
    for (int y = 0; y < img.rows(); ++y)
      for (int x = 0; x < img.cols(); ++x) {
        uint32_t v = ((uint32_t)(x >> 2) * 73856093u) ^ ((uint32_t)(y >> 2) * 19349663u);
        v ^= v >> 16; v *= 0x85ebca6bu; v ^= v >> 13;
        img(y, x) = (uint8_t)(v >> 9);
      }
    const int st0 = detail::last_status();
    const std::string err0 = detail::last_error();
    InferenceSettings sparse(5, 128, 0, true, false, 1);
    for (int it = 0; it < 2; ++it) {
      PreprocessedImage a = preprocessImage(img, sparse), b = preprocessImage(img, sparse);
      std::vector<ndb::Support> r = rectifiedMatch(a, b, forestmask, sparse);
      std::vector<ndb::Support> f = matchPair(img, img, forestmask, sparse);
      h.support_hint = 0;  // (a pair matched against itself says nothing about the caller's)
    }
    detail::last_status() = st0;
    detail::last_error() = err0;
    h.warmed = key;
    h.have_warmed = true;
  }

  // inference.hpp:302-333
  PreprocessedImage preprocessImage(ndb::Buffer<uint8_t>& img, InferenceSettings settings) {
    assert((settings.gradientThreshold_ >= 0 && settings.gradientThreshold_ <= 255) &&
           "gradientThreshold needs to be within 0...255");
    detail::ContextHolder& h = detail::holder();
    if (!h.ctx) return PreprocessedImage();
    int n = 0;
    int st = gpc_hip_preprocess_begin(h.ctx, img.data(), img.cols(), img.rows(), settings.gradientThreshold_, &n);
    if (st != GPC_OK) {
      detail::fail(st, h.ctx, "gpc_hip_preprocess");
      return PreprocessedImage();
    }
    // the arrays the caller keeps are written once, by the library (no zero fill first, no copy of a copy)
    PreprocessedImage r;
    r.smooth = ndb::Buffer<uint8_t>::uninitialized(img.rows(), img.cols());
    r.smooth.width = img.width;
    r.grad = ndb::Buffer<uint8_t>::uninitialized(img.rows(), img.cols());
    r.grad.width = img.width;
    r.mask.resize((size_t)n);
    st = gpc_hip_preprocess_fetch(h.ctx, r.smooth.data(), r.grad.data(), r.mask.data(), n);
    if (st != GPC_OK) {
      detail::fail(st, h.ctx, "gpc_hip_preprocess");
      return PreprocessedImage();
    }
    return r;
  }

  // inference.hpp:344-361
  std::vector<ndb::Correspondence> stereoMatch(PreprocessedImage& simg, PreprocessedImage& timg,
                                               FilterMask& forestmask, InferenceSettings settings) {
    assert((forestmask.width == simg.smooth.cols() && forestmask.height == simg.smooth.rows()) &&
           "Source Image: dimension does not fit dimension of supplied forest mask");
    assert((forestmask.width == timg.smooth.cols() && forestmask.height == simg.smooth.rows()) &&
           "Targe Image: dimension does not fit dimension of supplied forest mask");
    detail::ContextHolder& h = detail::holder();
    if (!h.ctx || !upload(h, forestmask)) return std::vector<ndb::Correspondence>();
    const gpc_settings s = settings.toC();
    // the results land in page-locked memory of this thread's context (the join writes them there over the link) and
    // become the vector in one pass
    const size_t cap = std::min(simg.mask.size(), timg.mask.size()) + 1;
    if (!h.pinned(&h.pin_out, &h.pin_out_cap, cap * sizeof(gpc_correspondence))) {
      detail::fail(GPC_E_HIP, h.ctx, "gpc_hip_host_alloc");
      return std::vector<ndb::Correspondence>();
    }
    int n = 0;
    const int st = gpc_hip_stereo_match(
        h.ctx, simg.smooth.data(), simg.grad.data(), simg.mask.data(), (int)simg.mask.size(), timg.smooth.data(),
        timg.grad.data(), timg.mask.data(), (int)timg.mask.size(), simg.smooth.cols(), simg.smooth.rows(), &s,
        static_cast<gpc_correspondence*>(h.pin_out), (int)cap, &n);
    if (st != GPC_OK) {
      detail::fail(st, h.ctx, "gpc_hip_stereo_match");
      return std::vector<ndb::Correspondence>();
    }
    const ndb::Correspondence* res = static_cast<const ndb::Correspondence*>(h.pin_out);
    return std::vector<ndb::Correspondence>(res, res + n);
  }

  // inference.hpp:375-393
  std::vector<ndb::Support> rectifiedMatch(PreprocessedImage& simg, PreprocessedImage& timg,
                                           FilterMask& forestmask, InferenceSettings settings) {
    assert((forestmask.width == simg.smooth.cols() && forestmask.height == simg.smooth.rows()) &&
           "Source Image: dimension does not fit dimension of supplied forest mask");
    detail::ContextHolder& h = detail::holder();
    if (!h.ctx || !upload(h, forestmask)) return std::vector<ndb::Support>();
    const gpc_settings s = settings.toC();
    const size_t cap = std::min(simg.mask.size(), timg.mask.size()) + 1;  // (page-locked staging: see stereoMatch)
    if (!h.pinned(&h.pin_out, &h.pin_out_cap, cap * sizeof(gpc_support))) {
      detail::fail(GPC_E_HIP, h.ctx, "gpc_hip_host_alloc");
      return std::vector<ndb::Support>();
    }
    int n = 0;
    const int st = gpc_hip_rectified_match(
        h.ctx, simg.smooth.data(), simg.grad.data(), simg.mask.data(), (int)simg.mask.size(), timg.smooth.data(),
        timg.grad.data(), timg.mask.data(), (int)timg.mask.size(), simg.smooth.cols(), simg.smooth.rows(), &s,
        static_cast<gpc_support*>(h.pin_out), (int)cap, &n);
    if (st != GPC_OK) {
      detail::fail(st, h.ctx, "gpc_hip_rectified_match");
      return std::vector<ndb::Support>();
    }
    const ndb::Support* res = static_cast<const ndb::Support*>(h.pin_out);
    return std::vector<ndb::Support>(res, res + n);
  }

  // Extension: preprocessImage x2 + rectifiedMatch without bringing the intermediates back
  // to the host (the timed region of samples/sparsematch.cpp:45-52 as one device pipeline).
  std::vector<ndb::Support> matchPair(ndb::Buffer<uint8_t>& simg, ndb::Buffer<uint8_t>& timg, FilterMask& forestmask,
                                      InferenceSettings settings, int* candidatesL = nullptr,
                                      int* candidatesR = nullptr) {
    if (candidatesL) *candidatesL = 0;  // (defined on every path out, the failing ones included)
    if (candidatesR) *candidatesR = 0;
    detail::ContextHolder& h = detail::holder();
    if (!h.ctx || !upload(h, forestmask)) return std::vector<ndb::Support>();
    const gpc_settings s = settings.toC();
    // Images and supports pass through page-locked staging memory of this thread's context (one memcpy each way on the
    // host; the kernels read the images and write the 12-byte supports over the link themselves: the single-pair path of
    // gpc_hip_match_batch).  The staging array is sized by the last call's count rather than by the worst case (one
    // support per pixel), and never so small that a textured pair needs the call twice (the second attempt below).
    const size_t npix = (size_t)simg.rows() * simg.cols();
    size_t cap0 = h.support_hint ? h.support_hint : npix * 3 / 4 + 1;
    if (cap0 > npix + 1) cap0 = npix + 1;
    const size_t in_bytes = (npix + 15) / 16 * 16;  // the second image starts 16-byte aligned
    if (!h.pinned(&h.pin_in, &h.pin_in_cap, 2 * in_bytes) || !h.pinned(&h.pin_out, &h.pin_out_cap, cap0 * sizeof(gpc_support))) {
      detail::fail(GPC_E_HIP, h.ctx, "gpc_hip_host_alloc");
      return std::vector<ndb::Support>();
    }
    uint8_t* pl = static_cast<uint8_t*>(h.pin_in);
    uint8_t* pr = pl + in_bytes;
    memcpy(pl, simg.data(), npix);
    memcpy(pr, timg.data(), npix);
    int n = 0;
    int st = gpc_hip_match_pair(h.ctx, pl, pr, simg.cols(), simg.rows(), &s, static_cast<gpc_support*>(h.pin_out), (int)cap0, &n,
                                candidatesL, candidatesR);
    if (st == GPC_E_CAPACITY) {
      cap0 = (size_t)n + 1;
      if (!h.pinned(&h.pin_out, &h.pin_out_cap, cap0 * sizeof(gpc_support))) {
        detail::fail(GPC_E_HIP, h.ctx, "gpc_hip_host_alloc");
        return std::vector<ndb::Support>();
      }
      st = gpc_hip_match_pair(h.ctx, pl, pr, simg.cols(), simg.rows(), &s, static_cast<gpc_support*>(h.pin_out), (int)cap0, &n,
                              candidatesL, candidatesR);
    }
    if (st != GPC_OK) {
      detail::fail(st, h.ctx, "gpc_hip_match_pair");
      if (candidatesL) *candidatesL = 0;
      if (candidatesR) *candidatesR = 0;
      return std::vector<ndb::Support>();
    }
    const ndb::Support* res = static_cast<const ndb::Support*>(h.pin_out);
    std::vector<ndb::Support> supp(res, res + n);  // (one copy; no element-by-element value-initialisation first)
    h.support_hint = (size_t)n + (size_t)n / 8 + 1024;
    return supp;
  }

 private:
  static int countFerns(const std::string& path) {
    FILE* fp = fopen(path.c_str(), "rb");
    int n = 0;
    if (fp) {
      if (fscanf(fp, "%d", &n) != 1) n = 0;
      fclose(fp);
    }
    return n;
  }
  static bool toC(const FilterMask& f, gpc_filter_mask& fm) {
    memset(&fm, 0, sizeof fm);
    fm.num_tests = (int)(f.mask.size() / 2);
    if (fm.num_tests > GPC_MAX_TESTS) fm.num_tests = GPC_MAX_TESTS;  // filter.hpp:574 `i < 64`
    for (int i = 0; i < 2 * fm.num_tests; ++i) fm.mask[i] = f.mask[i];
    for (int i = 0; i < fm.num_tests && i < (int)f.tau.size(); ++i) fm.tau[i] = f.tau[i];
    fm.type = f.type;
    fm.width = f.width;
    fm.height = f.height;
    return true;
  }
  static bool upload(detail::ContextHolder& h, const FilterMask& f) {
    gpc_filter_mask fm;
    toC(f, fm);
    if (h.have && memcmp(&h.uploaded, &fm, sizeof fm) == 0) return true;
    const int st = gpc_hip_set_forest(h.ctx, &fm);
    if (st != GPC_OK) {
      detail::fail(st, h.ctx, "gpc_hip_set_forest");
      return false;
    }
    h.uploaded = fm;
    h.have = true;
    return true;
  }
};

}  // namespace inference
}  // namespace gpc
#endif
