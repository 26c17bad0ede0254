// three_call_timeline -- where the warm three-call path (Forest::preprocessImage x2 + Forest::rectifiedMatch, as
// include/gpc/inference.hpp runs them) spends its time: the same C-ABI calls and the same allocations, a timestamp after
// every stage, medians over the warm iterations.  usage: three_call_timeline <forest> <W> <H> [iterations]
#include <malloc.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "gpc/buffer.hpp"
#include "gpc_hip.h"

static double now_us() {
  using namespace std::chrono;
  return duration_cast<duration<double, std::micro>>(steady_clock::now().time_since_epoch()).count();
}
static uint32_t mix(uint32_t a, uint32_t b) {
  uint32_t h = a * 73856093u ^ b * 19349663u;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
static std::map<std::string, std::vector<double>> g_t;
static std::vector<std::string> g_order;
struct Lap {
  double t = now_us();
  void operator()(const char* name) {
    const double n = now_us();
    if (!g_t.count(name)) g_order.push_back(name);
    g_t[name].push_back(n - t);
    t = now_us();
  }
};
struct Pre {
  ndb::Buffer<uint8_t> smooth, grad;
  std::vector<int> mask;
};

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int W = atoi(argv[2]), H = atoi(argv[3]), iters = argc > 4 ? atoi(argv[4]) : 30;
  if (getenv("TL_MALLOPT")) {
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
  }
  ndb::Buffer<uint8_t> L(H, W), R(H, W);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      L(y, x) = (uint8_t)((((mix((x + 24) >> 2, y >> 2) & 0xFF) * 3 + (mix(x + 24, y) & 0x3F)) >> 2));
      R(y, x) = (uint8_t)((((mix((x + 48) >> 2, y >> 2) & 0xFF) * 3 + (mix(x + 48, y) & 0x3F)) >> 2));
    }
  gpc_hip_ctx* c = nullptr;
  if (gpc_hip_create(0, &c)) return 1;
  gpc_filter_mask fm;
  gpc_hip_read_forest(argv[1], W, H, &fm);
  gpc_hip_set_forest(c, &fm);
  gpc_settings s = {5, 128, 0, 1, 0, 1};
  gpc_hip_warmup(c, W, H, &s);
  const size_t maxcand = (size_t)(W - 26) * (H - 26);
  std::vector<gpc_support> supp;
  for (int it = 0; it < iters; ++it) {
    if (it == 5) { g_t.clear(); g_order.clear(); }
    const bool show = it < 4 && getenv("TL_FIRST");   // the first calls of the process, stage by stage
    Pre p[2];
    Lap lap;
    double t_pre = 0;
    const double t0 = now_us();
    for (int side = 0; side < 2; ++side) {
      ndb::Buffer<uint8_t>& img = side ? R : L;
      gpc_hip_preprocess_begin(c, img.data(), W, H, 5);
      lap("pre: begin (queue)");
      p[side].smooth = ndb::Buffer<uint8_t>::uninitialized(H, W);
      p[side].grad = ndb::Buffer<uint8_t>::uninitialized(H, W);
      lap("pre: 2 image buffers");
      p[side].mask.resize(maxcand);
      lap("pre: mask.resize(max)");
      int n = 0;
      gpc_hip_preprocess_fetch(c, p[side].smooth.data(), p[side].grad.data(), p[side].mask.data(), (int)maxcand, &n);
      lap("pre: fetch (wait + copies)");
      p[side].mask.resize((size_t)n);
      lap("pre: mask.resize(n)");
    }
    t_pre = now_us() - t0;
    lap.t = now_us();
    gpc_hip_rectified_match_begin(c, p[0].smooth.data(), p[0].grad.data(), p[0].mask.data(), (int)p[0].mask.size(),
                                  p[1].smooth.data(), p[1].grad.data(), p[1].mask.data(), (int)p[1].mask.size(), W, H, &s);
    lap("match: begin (queue)");
    std::vector<gpc_support> v;
    v.resize(std::min(p[0].mask.size(), p[1].mask.size()) + 1);
    lap("match: result.resize(cap)");
    int n = 0;
    gpc_hip_match_fetch(c, v.data(), (int)v.size(), &n, nullptr, nullptr);
    lap("match: fetch (wait + copy)");
    v.resize((size_t)n);
    supp = std::move(v);   // the caller's `supp = forest.rectifiedMatch(...)`
    lap("match: resize(n) + move");
    const double t_all = now_us() - t0;
    if (!g_t.count("tPreprocess")) { g_order.push_back("tPreprocess"); g_order.push_back("tPreprocess + tMatch"); }
    g_order.erase(std::unique(g_order.begin(), g_order.end()), g_order.end());
    g_t["tPreprocess"].push_back(t_pre);
    g_t["tPreprocess + tMatch"].push_back(t_all);
    if (show) {
      printf("call %d:", it + 1);
      for (auto& k : g_order) printf("  %s %.0f", k.c_str(), g_t[k].back());
      printf("\n");
    }
  }
  printf("%dx%d, %zu supports, %d warm iterations%s; medians in us (per call; the preprocess stages are per image)\n", W, H, supp.size(),
         iters - 5, getenv("TL_MALLOPT") ? ", mallopt(no trim, no mmap)" : "");
  for (auto& k : g_order) {
    std::vector<double>& v = g_t[k];
    std::sort(v.begin(), v.end());
    printf("  %-30s %9.1f   (min %.1f, max %.1f)\n", k.c_str(), v[v.size() / 2], v.front(), v.back());
  }
  gpc_hip_destroy(c);
  return 0;
}
