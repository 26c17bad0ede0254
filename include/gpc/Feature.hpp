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

  // Feature.hpp:131-176: a random test inside the 7x7 (scale 2), 17x17 (1) or 27x27 (0) centre of the patch
  void inline sampleHyperplane(int scale, params& p) {
    if (scale == 2) {
      p.i = p.j;
      while (p.i == p.j) {
        int i = randIJ7(rng);
        int j = randIJ7(rng);
        p.ix = i % 7 - 3;
        p.iy = i / 7 - 3;
        p.jx = j % 7 - 3;
        p.jy = j / 7 - 3;
        p.i = 280 + (p.ix + 3) + 27 * (p.iy + 3);
        p.j = 280 + (p.jx + 3) + 27 * (p.jy + 3);
      }
    } else if (scale == 1) {
      p.i = p.j;
      while (p.i == p.j) {
        int i = randIJ17(rng);
        int j = randIJ17(rng);
        p.ix = i % 17 - 8;
        p.iy = i / 17 - 8;
        p.jx = j % 17 - 8;
        p.jy = j / 17 - 8;
        p.i = 140 + (p.ix + 8) + 27 * (p.iy + 8);
        p.j = 140 + (p.jx + 8) + 27 * (p.jy + 8);
      }
    } else if (scale == 0) {
      p.i = p.j;
      while (p.i == p.j) {
        p.i = randIJ27(rng);
        p.j = randIJ27(rng);
        p.ix = p.i % 27 - 13;
        p.iy = p.i / 27 - 13;
        p.jx = p.j % 27 - 13;
        p.jy = p.j / 27 - 13;
        p.i = (p.ix + 13) + 27 * (p.iy + 13);
        p.j = (p.jx + 13) + 27 * (p.jy + 13);
      }
    }
    p.tau = randTAU(rng);
  }

  // Feature.hpp:190-245: patches of the box-filtered images around the three keypoint lists.
  // The 3x3 box + clearBoundary run on the GPU (the same preprocess kernel as inference).
  void extractAllTriplets(ndb::Buffer<uint8_t>& bwL, ndb::Buffer<uint8_t>& bwR, std::vector<ndb::Point>& ref,
                          std::vector<ndb::Point>& pos, std::vector<ndb::Point>& neg,
                          std::vector<GPCPatchTriplet>& triplets) {
    ndb::Buffer<uint8_t> LL = smoothed(bwL), RR = smoothed(bwR);
    auto f = [=](ndb::Point& kp) {
      if (kp.x > 20 && kp.y > 20 && kp.x < bwL.cols() - 20 && kp.y < bwL.rows() - 20) return false;
      else return true;
    };
    for (std::vector<ndb::Point>::size_type i = 0; i != ref.size(); i++) {
      if (!f(ref[i]) && !f(pos[i]) && !f(neg[i])) {
        GPCPatchTriplet newPatch;
        newPatch.ref.x = ref[i].x;
        newPatch.ref.y = ref[i].y;
        LL.getPatch(newPatch.ref.feature, ref[i].x, ref[i].y, 27);
        newPatch.pos.x = pos[i].x;
        newPatch.pos.y = pos[i].y;
        RR.getPatch(newPatch.pos.feature, pos[i].x, pos[i].y, 27);
        newPatch.neg.x = neg[i].x;
        newPatch.neg.y = neg[i].y;
        RR.getPatch(newPatch.neg.feature, neg[i].x, neg[i].y, 27);
        triplets.push_back(std::move(newPatch));
      }
    }
  }

  // Feature.hpp:254-263
  void storeAllTriplets(std::vector<GPCPatchTriplet>& data, std::string path) {
    ofstream fout;
    fout.open(path, ios::binary | ios::out);
    for (auto& triplet : data) {
      fout.write((char*)triplet.ref.feature.data(), 27 * 27);
      fout.write((char*)triplet.pos.feature.data(), 27 * 27);
      fout.write((char*)triplet.neg.feature.data(), 27 * 27);
    }
    fout.close();
  }
  // Feature.hpp:272-297
  std::vector<GPCPatchTriplet> loadAllTriplets(std::string path) {
    std::vector<GPCPatchTriplet> data;
    std::ifstream in(path, std::ifstream::ate | std::ifstream::binary);
    uint32_t filesize = in.tellg();
    if (filesize % ((27 * 27) * 3)) {
      cout << "ERR: File is not a training set of this feature type" << endl;
      cout << "FS: " << filesize << endl;
      return data;
    }
    int numSamples = filesize / ((27 * 27) * 3);
    data.resize(numSamples);
    ifstream fin;
    fin.open(path, ios::binary | ios::in);
    for (auto& datum : data) {
      datum.ref.feature.resize(27, 27);
      datum.pos.feature.resize(27, 27);
      datum.neg.feature.resize(27, 27);
      fin.read((char*)datum.ref.feature.data(), 27 * 27);
      fin.read((char*)datum.pos.feature.data(), 27 * 27);
      fin.read((char*)datum.neg.feature.data(), 27 * 27);
    }
    fin.close();
    return data;
  }

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
