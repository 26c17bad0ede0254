// train_api_check -- drives gpc::training::{Feature, Fern, Forest} (include/gpc/*.hpp) for tests/test_training_api.py.
//   sample <scale> <count> <seed>                       -> the hyperplanes Feature::sampleHyperplane draws
//   api <triplets.bin> <seed> <depth> <resamples> <taulo> <tauhi> <only> <w1> <scale>
//        -> evalSplit / markSplitSamples / train on the loaded triplets (text on stdout)
//   forest <triplets.bin> <out.txt>                     -> Forest::trainAndExport with small settings
//   patch <out.bin>                                     -> Buffer::getPatch / store / load round trip
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "gpc/training.hpp"

using gpc::training::Feature;

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string cmd = argv[1];
  if (cmd == "sample") {
    Feature f;
    f.seed((unsigned)std::atoi(argv[4]));
    Feature::params p;
    for (int k = 0; k < std::atoi(argv[3]); ++k) {
      f.sampleHyperplane(std::atoi(argv[2]), p);
      std::printf("HP %d %d %d %d %d %d %d\n", p.i, p.j, p.ix, p.iy, p.jx, p.jy, p.tau);
    }
    return 0;
  }
  if (cmd == "api") {
    Feature f;
    std::vector<Feature::GPCPatchTriplet> data = f.loadAllTriplets(argv[2]);
    const int depth = std::atoi(argv[4]), nres = std::atoi(argv[5]);
    gpc::training::OptimizerSettings opt(std::atoi(argv[6]), std::atoi(argv[7]), nres, std::atoi(argv[8]) != 0,
                                         std::atof(argv[9]));
    gpc::training::Fern fern(gpc::training::FernSettings(depth, std::atoi(argv[10])));
    fern.seed((unsigned)std::atoi(argv[3]));
    // a fixed parameter list for evalSplit / markSplitSamples
    std::vector<Feature::params> params(3);
    const int pi[3] = {5, 364, 700}, pj[3] = {33, 365, 2}, pt[3] = {0, 2, -3};
    for (int l = 0; l < 3; ++l) {
      params[l].i = pi[l];
      params[l].j = pj[l];
      params[l].tau = pt[l];
    }
    for (size_t k = 0; k < data.size(); ++k) {  // some marks to start from
      data[k].pos.split = (k % 3) == 0;
      data[k].neg.split = (k % 5) == 0;
    }
    gpc::training::splitStats s;
    fern.evalSplit(data, params, gpc::training::FernSettings(depth, 0), opt, 2, s);
    std::printf("EVAL %d %d %d %d %.17g %.17g %.17g %.17g\n", s.tp, s.fp, s.fn, s.tot, s.prec, s.rec, s.hmean, s.convcomb);
    fern.markSplitSamples(data, params, 2);
    std::printf("MARKS");
    for (auto& t : data) std::printf(" %d", (t.pos.split ? 1 : 0) | (t.neg.split ? 2 : 0));
    std::printf("\n");
    fern.train(data, opt);
    std::printf("PARAMS");
    for (auto& p : fern.getParameters()) std::printf(" %d %d %d %d %d %d %d", p.i, p.j, p.tau, p.ix, p.iy, p.jx, p.jy);
    std::printf("\nMARKS2");
    for (auto& t : data) std::printf(" %d", (t.pos.split ? 1 : 0) | (t.neg.split ? 2 : 0));
    std::printf("\n");
    return 0;
  }
  if (cmd == "forest") {
    gpc::datasource::SintelOpticalFlow src;
    auto data = src.loadTrainingData(argv[2]);
    gpc::training::OptimizerSettings opt = gpc::training::ZeroOptimizer(4, true, 0.5);
    gpc::training::ForestSettings fs(gpc::training::FernFactory(1, 1, 1, 3), 0.7);
    gpc::training::Forest forest;
    forest.trainAndExport(data, fs, opt, argv[3]);
    auto none = src.loadTrainingData("/nonexistent/triplets.bin");
    std::printf("MISSING %zu\n", none.size());
    return 0;
  }
  if (cmd == "patch") {
    ndb::Buffer<uint8_t> img(40, 48);
    for (int y = 0; y < 40; ++y)
      for (int x = 0; x < 48; ++x) img.setPixel(x, y, (uint8_t)(7 * x + 13 * y));
    Feature f;
    std::vector<Feature::GPCPatchTriplet> v(1);
    img.getPatch(v[0].ref.feature, 20, 18, 27);
    img.getPatch(v[0].pos.feature, 21, 18, 27);
    img.getPatch(v[0].neg.feature, 20, 19, 27);
    // patch(row = ix, col = iy) = pixel(x + ix - 13, y + iy - 13): the row index follows the image x offset
    std::printf("PATCH %d %d %d\n", (int)v[0].ref.feature(0), (int)v[0].ref.feature(1), (int)v[0].ref.feature(27));
    f.storeAllTriplets(v, argv[2]);
    auto back = f.loadAllTriplets(argv[2]);
    std::printf("ROUNDTRIP %zu %d\n", back.size(),
                (int)(std::memcmp(back[0].neg.feature.data(), v[0].neg.feature.data(), 729) == 0));
    return 0;
  }
  return 2;
}
