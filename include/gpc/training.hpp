// gpc/training.hpp -- MI355X-native mirror of the reference's forest trainer (lib/gpc/training.hpp):
// ForestSettings and Forest::trainAndExport (bootstrap of the training set per fern, Fern::train on
// the GPU, export in the text format Forest::readForest reads).
//
// The Sintel dataset readers of the reference (SintelOpticalFlow.hpp, SintelStereo.hpp: PNG / flow
// file walking and keypoint selection) are outside this build; gpc::datasource::SintelOpticalFlow
// offers only loadTrainingData(), which is all samples/train.cpp needs.
#ifndef _GPC_training
#define _GPC_training
#include <sys/stat.h>

#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "gpc/Feature.hpp"
#include "gpc/Fern.hpp"
#include "gpc/buffer.hpp"

namespace gpc {
namespace datasource {
// SintelOpticalFlow.hpp:181-190
class SintelOpticalFlow {
  typedef gpc::training::Feature::GPCPatchTriplet GPCTriplet_t;
  gpc::training::Feature Feature;

 public:
  std::vector<GPCTriplet_t> loadTrainingData(std::string path) {
    struct stat buffer;
    if (stat(path.c_str(), &buffer) != 0) {
      std::vector<GPCTriplet_t> emptyset;
      cout << "ERR: No extracted training set found at given path" << endl;
      return emptyset;
    } else {
      return Feature.loadAllTriplets(path);
    }
  }
};
}  // namespace datasource

namespace training {
// training.hpp:59-74
struct ForestSettings {
  enum FernType { Zero, Tau };
  FernType fernType;
  std::string getFernTypeName() { return fernType == FernType::Zero ? "zero" : "tau"; }
  double sampleFraction;
  std::vector<gpc::training::Fern> ferns;
  ForestSettings(std::vector<gpc::training::Fern> ferns, double sampleFraction)
      : sampleFraction(sampleFraction), ferns(ferns) {}
};

// training.hpp:91-159
class Forest {
 private:
  typedef gpc::training::Feature F;
  typedef F::GPCPatchTriplet GPCTriplet_t;
  std::mt19937 rng;
  std::uniform_int_distribution<int> randSample;

 public:
  Forest() {}
  void trainAndExport(std::vector<GPCTriplet_t>& trainingSamples, gpc::training::ForestSettings forestSettings,
                      gpc::training::OptimizerSettings optSettings, std::string filename) {
    std::chrono::high_resolution_clock::time_point t0, t1;
    std::random_device rd2;
    if (trainingSamples.size() == 0) {
      cout << "ERR: Training set is empty. Aborting." << endl;
      return;
    }
    rng = std::mt19937(rd2());
    randSample =
        std::uniform_int_distribution<int>(0, int(forestSettings.sampleFraction * trainingSamples.size()) - 1);
    int fernIndex = 1;
    for (auto& fern : forestSettings.ferns) {
      std::vector<GPCTriplet_t> subSample;  // with replacement
      for (int i = 0; i < int(forestSettings.sampleFraction * trainingSamples.size()); i++)
        subSample.push_back(trainingSamples[randSample(rng)]);
      cout << "Fern(" << fernIndex++ << "/" << forestSettings.ferns.size() << ") num samples:" << subSample.size();
      cout << endl << std::string(90, '*') << endl;
      t0 = std::chrono::high_resolution_clock::now();
      fern.train(subSample, optSettings);
      t1 = std::chrono::high_resolution_clock::now();
      cout << "done in " << std::chrono::duration_cast<std::chrono::duration<double>>(t1 - t0).count() << " s"
           << endl
           << endl;
    }
    cout << "Exporting forest" << endl;
    std::fstream file(filename, std::ofstream::out | std::ofstream::trunc);
    file << forestSettings.ferns.size() << endl;
    int f = 0;
    for (auto& fern : forestSettings.ferns) {
      std::vector<gpc::training::Feature::params> fparams = fern.getParameters();
      int scale = fern.getScale();  // 2: small, 1: medium, 0: large
      file << f << " " << ((scale == 2) ? "s" : ((scale == 1) ? "m" : "l")) << " " << fparams.size() << endl;
      int i = 0;
      for (auto& p : fparams) {
        file << int(i) << " " << int(p.ix) << " " << int(p.iy) << " " << int(p.jx) << " " << int(p.jy) << " "
             << int(p.tau) << endl;
        i++;
      }
      f++;
    }
    file.close();
  }
};  // Forest
}  // namespace training
}  // namespace gpc
#endif
