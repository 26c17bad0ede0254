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
  typedef gpc::training::Feature::GPCPatchTriplet GPCTriplet_t;
  std::mt19937 rng;
  std::uniform_int_distribution<int> randSample;

  // one line per fern ("<index> <s|m|l> <levels>") followed by one line per level
  // ("<level> <ix> <iy> <jx> <jy> <tau>"): the format Forest::readForest parses (inference.hpp:404-446)
  static void writeForest(std::vector<gpc::training::Fern>& ferns, const std::string& filename) {
    std::ofstream file(filename, std::ios::out | std::ios::trunc);
    file << ferns.size() << endl;
    for (size_t f = 0; f < ferns.size(); ++f) {
      const std::vector<gpc::training::Feature::params> levels = ferns[f].getParameters();
      const int scale = ferns[f].getScale();  // 2: small, 1: medium, 0: large
      file << f << " " << (scale == 2 ? "s" : (scale == 1 ? "m" : "l")) << " " << levels.size() << endl;
      for (size_t l = 0; l < levels.size(); ++l)
        file << l << " " << levels[l].ix << " " << levels[l].iy << " " << levels[l].jx << " " << levels[l].jy << " "
             << levels[l].tau << endl;
    }
  }

 public:
  Forest() {}
  // Every fern trains on its own bootstrap sample (with replacement) of sampleFraction * N triplets, drawn
  // from the first sampleFraction * N triplets like the reference does (training.hpp:113-121).
  void trainAndExport(std::vector<GPCTriplet_t>& trainingSamples, gpc::training::ForestSettings forestSettings,
                      gpc::training::OptimizerSettings optSettings, std::string filename) {
    if (trainingSamples.empty()) {
      cout << "ERR: Training set is empty. Aborting." << endl;
      return;
    }
    std::random_device seedSource;
    rng = std::mt19937(seedSource());
    const int perFern = int(forestSettings.sampleFraction * trainingSamples.size());
    randSample = std::uniform_int_distribution<int>(0, perFern - 1);
    const size_t total = forestSettings.ferns.size();
    for (size_t f = 0; f < total; ++f) {
      std::vector<GPCTriplet_t> bootstrap;
      bootstrap.reserve(perFern > 0 ? perFern : 0);
      for (int k = 0; k < perFern; ++k) bootstrap.push_back(trainingSamples[randSample(rng)]);
      cout << "Fern(" << (f + 1) << "/" << total << ") num samples:" << bootstrap.size() << endl
           << std::string(90, '*') << endl;
      const auto started = std::chrono::high_resolution_clock::now();
      forestSettings.ferns[f].train(bootstrap, optSettings);
      const std::chrono::duration<double> took = std::chrono::high_resolution_clock::now() - started;
      cout << "done in " << took.count() << " s" << endl << endl;
    }
    cout << "Exporting forest" << endl;
    writeForest(forestSettings.ferns, filename);
  }
};  // Forest
}  // namespace training
}  // namespace gpc
#endif
