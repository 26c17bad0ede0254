// sparsematch -- sparse stereo matching of one rectified pair on an MI355X.
//
// Command line, defaults and the result line follow the reference's sample
// (samples/sparsematch.cpp) so the binary can stand in for it:
//     sparsematch <forest path> <left image path> <right image path> [--fused] [--repeat N]
// writes disparity.png (left image with the supports painted in the KITTI colour ramp).
// --fused runs the timed region as one device pipeline (Forest::matchPair) instead of the
// two API calls preprocessImage + rectifiedMatch; the supports are identical.
#include <cstring>
#include <iostream>

#include "gpc/inference.hpp"

int main(int argc, char** argv) {
  std::string forestPath = "../forests/defaultZeroForest.txt";
  std::string leftPath = "../data/left.png";
  std::string rightPath = "../data/right.png";
  bool fused = false;
  int repeat = 1;
  std::vector<std::string> pos;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--fused")) fused = true;
    else if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
    else pos.push_back(argv[i]);
  }
  if (pos.size() == 3) {
    forestPath = pos[0];
    leftPath = pos[1];
    rightPath = pos[2];
  } else {
    std::cout << "Usage: " << argv[0] << " <forest path> <left image path> <right image path>" << std::endl;
    std::cout << "Trying defaults:" << std::endl;
    std::cout << "Forest path: " << forestPath << std::endl;
    std::cout << "Left image : " << leftPath << std::endl;
    std::cout << "Right image: " << rightPath << std::endl;
  }
#ifdef _INTRINSICS_SSE
  std::cout << "Using SSE intrinsics" << std::endl;  // arithmetic of the reference's default build
#endif
  std::cout << "Using HIP kernels (gfx950)" << std::endl;

  typedef gpc::inference::Forest Forest;
  Forest forest;
  gpc::inference::InferenceSettings settings = gpc::inference::InferenceSettings()
                                                   .builder()
                                                   .gradientThreshold(5)
                                                   .verticalTolerance(0)
                                                   .dispHigh(128)
                                                   .epipolarMode(true)
                                                   .useHashtable(false);

  ndb::Buffer<uint8_t> left, right;
  if (left.readPNG(leftPath) || right.readPNG(rightPath)) {
    std::cout << "No image data \n";
    return -1;
  }
  Forest::FilterMask fm = forest.readForest(forestPath, left.cols(), left.rows());

  std::vector<ndb::Support> supp;
  for (int it = 0; it < repeat; ++it) {
    size_t nl = 0, nr = 0;
    gpc::inference::time_point t0 = gpc::inference::sysTick(), t1, t2;
    if (fused) {
      int cl = 0, cr = 0;
      supp = forest.matchPair(left, right, fm, settings, &cl, &cr);
      t1 = t2 = gpc::inference::sysTick();
      nl = cl;
      nr = cr;
    } else {
      Forest::PreprocessedImage lp = forest.preprocessImage(left, settings);
      Forest::PreprocessedImage rp = forest.preprocessImage(right, settings);
      t1 = gpc::inference::sysTick();
      supp = forest.rectifiedMatch(lp, rp, fm, settings);
      t2 = gpc::inference::sysTick();
      nl = lp.mask.size();
      nr = rp.mask.size();
    }
    std::cout << "tPreprocess: " << gpc::inference::tickToMs(t1, t0) << " ms"
              << ", #candidatesL:" << nl << ", #candidatesR:" << nr
              << ", tMatch: " << gpc::inference::tickToMs(t2, t1) << " ms"
              << ", num matches:" << supp.size() << std::endl;
  }
  ndb::Buffer<ndb::RGBColor> render = ndb::getDisparityVisualization(left, supp);
  render.writePNGRGB("disparity.png");
  return 0;
}
