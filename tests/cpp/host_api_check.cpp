// host_api_check -- exercises the host-only parts of include/gpc/*.hpp for pytest.
//   host_api_check read <png> <out.raw>        -> prints "rc cols rows width height", dumps Buffer bytes
//   host_api_check write_gray <w> <h> <out.png>  (pixel = (x*3 + y*7) & 0xFF)
//   host_api_check write_rgb <w> <h> <out.png>   (r = x, g = y, b = x ^ y, all & 0xFF)
//   host_api_check vis <w> <h> <out.png>         supports on a diagonal, d = x/2
//   host_api_check ramp <out.raw>                disparityColor for d = -4.0, -3.5 .. 260.0 as RGB bytes (529 x 3)
//   host_api_check settings                      prints InferenceSettings defaults and builder result
//   host_api_check clear <w> <h> <out.raw>       clearBoundary on an all-255 buffer of visible width w
#include <cstdio>
#include <cstring>
#include <iostream>

#include "gpc/inference.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::string cmd = argv[1];
  if (cmd == "read" && argc == 4) {
    ndb::Buffer<uint8_t> b;
    int rc = b.readPNG(argv[2]);
    printf("RESULT %d %d %d %d %d\n", rc, b.cols(), b.rows(), b.width, b.height);
    FILE* f = fopen(argv[3], "wb");
    if (b.size()) fwrite(b.data(), 1, b.size(), f);
    fclose(f);
    return 0;
  }
  if ((cmd == "write_gray" || cmd == "write_rgb" || cmd == "vis" || cmd == "clear") && argc == 5) {
    int w = atoi(argv[2]), h = atoi(argv[3]);
    if (cmd == "write_gray") {
      ndb::Buffer<uint8_t> b(h, w);
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) b.setPixel(x, y, (uint8_t)((x * 3 + y * 7) & 0xFF));
      b.writePNG(argv[4]);
    } else if (cmd == "write_rgb") {
      ndb::Buffer<ndb::RGBColor> b(h, w);
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) b.setPixel(x, y, ndb::RGBColor(x & 0xFF, y & 0xFF, (x ^ y) & 0xFF));
      b.writePNGRGB(argv[4]);
    } else if (cmd == "vis") {
      ndb::Buffer<uint8_t> b(h, w);
      for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) b.setPixel(x, y, (uint8_t)((x + y) & 0xFF));
      std::vector<ndb::Support> s;
      for (int i = 0; i < std::min(w, h); ++i) s.push_back(ndb::Support(i, i, (float)(i / 2)));
      ndb::Buffer<ndb::RGBColor> v = ndb::getDisparityVisualization(b, s);
      v.writePNGRGB(argv[4]);
    } else {
      ndb::Buffer<uint8_t> b(h, w, 255);
      b.clearBoundary();
      FILE* f = fopen(argv[4], "wb");
      fwrite(b.data(), 1, b.size(), f);
      fclose(f);
      printf("RESULT %d %d\n", b.cols(), b.rows());
    }
    return 0;
  }
  if (cmd == "ramp" && argc == 3) {
    FILE* f = fopen(argv[2], "wb");
    for (int i = -8; i <= 520; ++i) {
      const ndb::RGBColor c = ndb::disparityColor(0.5f * (float)i);
      const uint8_t px[3] = {c.r, c.g, c.b};
      fwrite(px, 1, 3, f);
    }
    fclose(f);
    return 0;
  }
  if (cmd == "settings") {
    gpc::inference::InferenceSettings d;
    printf("DEFAULT %d %d %d %d %d %d\n", d.gradientThreshold_, d.dispHigh_, d.verticalTolerance_, d.epipolarMode_,
           d.useHashtable_, d.numThreads_);
    gpc::inference::InferenceSettings s = gpc::inference::InferenceSettings().builder().gradientThreshold(5)
        .verticalTolerance(0).dispHigh(64).epipolarMode(true).useHashtable(false).numThreads(100000);
    printf("BUILT %d %d %d %d %d %d\n", s.gradientThreshold_, s.dispHigh_, s.verticalTolerance_, s.epipolarMode_,
           s.useHashtable_, s.numThreads_ <= (int)std::thread::hardware_concurrency());
    printf("SIZES %zu %zu %zu\n", sizeof(ndb::Descriptor), sizeof(ndb::Support), sizeof(ndb::Correspondence));
    gpc::inference::Forest f;
    gpc::inference::Forest::FilterMask m = f.readForest("/nonexistent/forest.txt", 96, 64);
    printf("MISSING %zu %d %d %d\n", m.mask.size(), m.type, m.width, m.height);
    return 0;
  }
  // host_api_check api <forest> <left.png> <right.png> <epipolar 0/1> <hashtable 0/1> <out.bin>
  //   runs Forest::preprocessImage x2, stereoMatch, rectifiedMatch and matchPair on the GPU and dumps
  //   [n_corr][corr...][n_supp][supp...][n_fused][fused...][maskL size][maskR size] as int32/float32 words
  if (cmd == "api" && argc == 8) {
    ndb::Buffer<uint8_t> L, R;
    if (L.readPNG(argv[3]) || R.readPNG(argv[4])) return 3;
    gpc::inference::Forest forest;
    gpc::inference::InferenceSettings st = gpc::inference::InferenceSettings().builder().gradientThreshold(5)
        .verticalTolerance(1).dispHigh(64).epipolarMode(atoi(argv[5]) != 0).useHashtable(atoi(argv[6]) != 0);
    gpc::inference::Forest::FilterMask fm = forest.readForest(argv[2], L.cols(), L.rows());
    gpc::inference::Forest::PreprocessedImage lp = forest.preprocessImage(L, st);
    gpc::inference::Forest::PreprocessedImage rp = forest.preprocessImage(R, st);
    std::vector<ndb::Correspondence> corr = forest.stereoMatch(lp, rp, fm, st);
    std::vector<ndb::Support> supp = forest.rectifiedMatch(lp, rp, fm, st);
    std::vector<ndb::Support> fused = forest.matchPair(L, R, fm, st);
    FILE* f = fopen(argv[7], "wb");
    int n = (int)corr.size();
    fwrite(&n, 4, 1, f);
    if (n) fwrite(corr.data(), sizeof(ndb::Correspondence), n, f);
    n = (int)supp.size();
    fwrite(&n, 4, 1, f);
    if (n) fwrite(supp.data(), sizeof(ndb::Support), n, f);
    n = (int)fused.size();
    fwrite(&n, 4, 1, f);
    if (n) fwrite(fused.data(), sizeof(ndb::Support), n, f);
    n = (int)lp.mask.size();
    fwrite(&n, 4, 1, f);
    n = (int)rp.mask.size();
    fwrite(&n, 4, 1, f);
    fclose(f);
    printf("API %zu %zu %zu\n", corr.size(), supp.size(), fused.size());
    return 0;
  }
  // host_api_check nodevice <forest>: matchPair with GPC_HIP_DEVICE pointing at a device that does not exist -- the call must
  // return an empty result WITH a status (the reference's API has no error channel) and defined candidate counts
  if (cmd == "nodevice" && argc == 3) {
    ndb::Buffer<uint8_t> L(64, 96, 7), R(64, 96, 9);
    gpc::inference::Forest forest;
    gpc::inference::InferenceSettings st;
    gpc::inference::Forest::FilterMask fm = forest.readForest(argv[2], L.cols(), L.rows());
    int cl = -12345, cr = -12345;
    printf("BEFORE %d [%s]\n", gpc::inference::lastStatus(), gpc::inference::lastError().c_str());
    std::vector<ndb::Support> r = forest.matchPair(L, R, fm, st, &cl, &cr);
    printf("AFTER %zu %d %d %d [%s]\n", r.size(), cl, cr, gpc::inference::lastStatus(), gpc::inference::lastError().c_str());
    gpc::inference::clearStatus();
    printf("CLEARED %d [%s]\n", gpc::inference::lastStatus(), gpc::inference::lastError().c_str());
    return 0;
  }
  if (cmd == "forest" && argc == 5) {
    gpc::inference::Forest f;
    gpc::inference::Forest::FilterMask m = f.readForest(argv[2], atoi(argv[3]), atoi(argv[4]));
    printf("FOREST %zu %zu %d", m.mask.size(), m.tau.size(), m.type);
    for (size_t i = 0; i < m.mask.size(); ++i) printf(" %d", m.mask[i]);
    printf("\n");
    return 0;
  }
  return 2;
}
