// gpc/Feature.hpp -- MI355X-native mirror of the reference's training feature (lib/gpc/Feature.hpp):
// patch triplets, the parameters of one fern level, the decision rule, hyperplane sampling and the
// triplet file format.  Host-side data handling only; the scoring itself runs on the GPU through
// gpc::training::Fern (gpc/Fern.hpp -> gpc_hip_train_*, include/gpc_hip.h).
#ifndef _GPC_feature
#define _GPC_feature

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <iterator>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "gpc/buffer.hpp"
#include "gpc/inference.hpp"  // ndb box filter entry point + the per-thread device context

using namespace std;  // the reference's headers do this; samples/train.cpp relies on it

namespace gpc {
namespace training {

class Feature {
 private:
  std::mt19937 rng;
  std::uniform_int_distribution<int> randIJ7, randIJ17, randIJ27, randTAU;

 public:
  // Feature.hpp:61-67
  struct GPCDescriptor {
    ndb::Buffer<uint8_t> feature;
    int x = 0, y = 0;
    bool split = false;  // this sample has been split from the reference in training
    bool le = false;     // low-energy patch
  };
  // Feature.hpp:72-76
  struct GPCPatchTriplet {
    GPCDescriptor ref;
    GPCDescriptor pos;
    GPCDescriptor neg;
  };
  // Feature.hpp:82-89
  struct params {
    int i = 0, j = 0;
    int ix = 0, iy = 0;
    int jx = 0, jy = 0;
    int tau = 0;  // threshold for sign(i-j-tau)
  };

  // Feature.hpp:101-109
  inline void getDecisions(bool& ref, bool& pos, bool& neg, params& p, const GPCPatchTriplet& trip) {
    ref = ((int)trip.ref.feature(p.i) - (int)trip.ref.feature(p.j) < p.tau);
    pos = ((int)trip.pos.feature(p.i) - (int)trip.pos.feature(p.j) < p.tau);
    neg = ((int)trip.neg.feature(p.i) - (int)trip.neg.feature(p.j) < p.tau);
  }

  // Feature.hpp:111-119 (seeded from std::random_device, like the reference)
  Feature() {
    std::random_device rd2;
    init(rd2());
  }
  // extension: reproducible sampling (tests)
  void seed(unsigned s) { rng = std::mt19937(s); }

  // Feature.hpp:131-176.  A random test (i, j) whose two pixels lie in the centred side x side window of
  // the 27x27 patch: scale 2 -> 7x7, 1 -> 17x17, 0 -> 27x27.  Draw order as in the reference (window cell
  // of i, then of j, repeated until they differ; then the intercept), so a seeded generator yields the
  // same sequence.  Patch index of an offset (dx, dy): (dx + 13) + 27 * (dy + 13), at every scale.
  void inline sampleHyperplane(int scale, params& p) {
    const int side = scale == 2 ? 7 : (scale == 1 ? 17 : 27);
    if (scale >= 0 && scale <= 2) {
      std::uniform_int_distribution<int>& cell = scale == 2 ? randIJ7 : (scale == 1 ? randIJ17 : randIJ27);
      const int half = side / 2;
      do {
        const int ci = cell(rng);
        const int cj = cell(rng);
        p.ix = ci % side - half;
        p.iy = ci / side - half;
        p.jx = cj % side - half;
        p.jy = cj / side - half;
        p.i = patchIndex(p.ix, p.iy);
        p.j = patchIndex(p.jx, p.jy);
      } while (p.i == p.j);
    }
    p.tau = randTAU(rng);
  }
  static int patchIndex(int dx, int dy) { return (dx + 13) + 27 * (dy + 13); }

  // Feature.hpp:190-245: for every keypoint triple that lies more than 20 px inside the image, the 27x27
  // patches of the smoothed left (ref) and right (pos, neg) image.  Smoothing = 3x3 box + clearBoundary,
  // run on the GPU (the `smooth` output of preprocessImage).
  void extractAllTriplets(ndb::Buffer<uint8_t>& bwL, ndb::Buffer<uint8_t>& bwR, std::vector<ndb::Point>& ref,
                          std::vector<ndb::Point>& pos, std::vector<ndb::Point>& neg,
                          std::vector<GPCPatchTriplet>& triplets) {
    ndb::Buffer<uint8_t> smoothL = smoothed(bwL), smoothR = smoothed(bwR);
    const int cols = bwL.cols(), rows = bwL.rows();
    auto inside = [cols, rows](const ndb::Point& kp) {
      return kp.x > 20 && kp.y > 20 && kp.x < cols - 20 && kp.y < rows - 20;
    };
    auto cut = [](const ndb::Buffer<uint8_t>& img, const ndb::Point& at, GPCDescriptor& d) {
      d.x = at.x;
      d.y = at.y;
      img.getPatch(d.feature, at.x, at.y, 27);
    };
    for (size_t k = 0; k < ref.size(); ++k) {
      if (!(inside(ref[k]) && inside(pos[k]) && inside(neg[k]))) continue;
      triplets.emplace_back();
      cut(smoothL, ref[k], triplets.back().ref);
      cut(smoothR, pos[k], triplets.back().pos);
      cut(smoothR, neg[k], triplets.back().neg);
    }
  }

  // Feature.hpp:254-263: 3 x 729 bytes per triplet (ref, pos, neg), nothing else in the file
  void storeAllTriplets(std::vector<GPCPatchTriplet>& data, std::string path) {
    std::ofstream out(path, std::ios::binary);
    for (const GPCPatchTriplet& t : data)
      for (const GPCDescriptor* d : {&t.ref, &t.pos, &t.neg})
        out.write(reinterpret_cast<const char*>(d->feature.data()), kPatchBytes);
  }
  // Feature.hpp:272-297: a file whose size is not a multiple of 3 x 729 is refused with the reference's message
  std::vector<GPCPatchTriplet> loadAllTriplets(std::string path) {
    std::vector<GPCPatchTriplet> data;
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    const uint32_t filesize = (uint32_t)in.tellg();  // 32 bits, like the reference
    if (filesize % (3 * kPatchBytes)) {
      cout << "ERR: File is not a training set of this feature type" << endl;
      cout << "FS: " << filesize << endl;
      return data;
    }
    data.resize(filesize / (3 * kPatchBytes));
    in.seekg(0);
    for (GPCPatchTriplet& t : data)
      for (GPCDescriptor* d : {&t.ref, &t.pos, &t.neg}) {
        d->feature.resize(27, 27);
        in.read(reinterpret_cast<char*>(d->feature.data()), kPatchBytes);
      }
    return data;
  }
  static constexpr int kPatchBytes = 27 * 27;

 private:
  void init(unsigned s) {
    rng = std::mt19937(s);
    randIJ7 = std::uniform_int_distribution<int>(0, 48);
    randIJ17 = std::uniform_int_distribution<int>(0, 17 * 17 - 1);
    randIJ27 = std::uniform_int_distribution<int>(0, 27 * 27 - 1);
    randTAU = std::uniform_int_distribution<int>(-15, 15);
  }
  // ndb::box + clearBoundary of the whole image (Feature.hpp:197-205) = the `smooth` output of preprocessImage
  static ndb::Buffer<uint8_t> smoothed(ndb::Buffer<uint8_t>& bw) {
    gpc::inference::Forest forest;
    gpc::inference::InferenceSettings s;
    return forest.preprocessImage(bw, s).smooth;
  }
};  // Feature
}  // namespace training
}  // namespace gpc
#endif
