// train -- trains a forest of ferns on extracted patch triplets and writes it in the text format
// Forest::readForest reads.  Same role and defaults as the reference's samples/train.cpp; the scoring
// loops run on the MI355X (gpc/Fern.hpp -> libgpc_hip.so).
//
// usage: train <extracted triplets .bin> <forest out .txt> [zero|tau] [ferns per scale] [depth] [resamples]
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "gpc/training.hpp"

int main(int argc, char** argv) {
  std::string dataset = "../../data/SintelOpticalFlow-extracted.bin";
  std::string forest = "../../forests/defaultZeroForest.txt";
  if (argc >= 3) {
    dataset = argv[1];
    forest = argv[2];
  } else {
    std::cout << "Usage: " << argv[0]
              << " <extracted dataset path> <forest path> [zero|tau] [ferns per scale] [depth] [resamples]"
              << std::endl;
    std::cout << "Trying defaults: " << dataset << " -> " << forest << std::endl;
  }
  const bool tau = argc >= 4 && !std::strcmp(argv[3], "tau");
  const int per_scale = argc >= 5 ? std::atoi(argv[4]) : 2;
  const int depth = argc >= 6 ? std::atoi(argv[5]) : 5;
  const int resamples = argc >= 7 ? std::atoi(argv[6]) : 10;

  // reference defaults: 10 resamples, all samples scored on every level, precision and recall weighted equally;
  // the tau optimizer additionally searches the intercept on [-10, 10)
  gpc::training::OptimizerSettings optimizer =
      tau ? gpc::training::OptimizerSettings(gpc::training::TauOptimizerSettings()
                                                 .builder()
                                                 .taulo(-10)
                                                 .tauhi(10)
                                                 .numResamples(resamples)
                                                 .onlyScoreNonSplitSamples(false)
                                                 .w1(0.5))
          : gpc::training::OptimizerSettings(gpc::training::ZeroOptimizerSettings()
                                                 .builder()
                                                 .numResamples(resamples)
                                                 .onlyScoreNonSplitSamples(false)
                                                 .w1(0.5));
  // small / medium / large ferns, 70 % bootstrap per fern
  gpc::training::ForestSettings settings(gpc::training::FernFactory(per_scale, per_scale, per_scale, depth), 0.7);

  gpc::datasource::SintelOpticalFlow source;
  std::cout << "Loading dataset" << std::endl;
  std::vector<gpc::training::Feature::GPCPatchTriplet> data = source.loadTrainingData(dataset);
  gpc::training::Forest trainer;
  trainer.trainAndExport(data, settings, optimizer, forest);
  return data.empty() ? 1 : 0;
}
