// cold_start_probe -- where a run-once process (the reference's sparsematch CLI) spends its first milliseconds on the
// C ABI: library load, device discovery, context, forest upload, warm-up, and then the reference's timed region
// (samples/sparsematch.cpp:45-52: preprocessImage x2 + rectifiedMatch) call by call.  One line per stage, wall clock.
// usage: cold_start_probe <forest> <W> <H> [--no-warmup] [--repeat N]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gpc_hip.h"

static double now_ms() {
  using namespace std::chrono;
  return duration_cast<duration<double, std::milli>>(steady_clock::now().time_since_epoch()).count();
}
static uint32_t mix(uint32_t a, uint32_t b) {  // the synthetic texture of SURVEY 8d
  uint32_t h = a * 73856093u ^ b * 19349663u;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return h;
}
static void synth(std::vector<uint8_t>& img, int W, int H, int shift) {
  img.resize((size_t)W * H);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const int xs = x + shift;
      img[(size_t)y * W + x] = (uint8_t)((((mix(xs >> 2, y >> 2) & 0xFF) * 3 + (mix(xs, y) & 0x3F)) >> 2));
    }
}

int main(int argc, char** argv) {
  if (argc < 4) { printf("usage: %s <forest> <W> <H> [--no-warmup] [--repeat N]\n", argv[0]); return 2; }
  const int W = atoi(argv[2]), H = atoi(argv[3]);
  bool warm = true;
  int repeat = 3;
  for (int i = 4; i < argc; ++i) {
    if (!strcmp(argv[i], "--no-warmup")) warm = false;
    if (!strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = atoi(argv[++i]);
  }
  const double t_start = now_ms();
  std::vector<uint8_t> L, R;
  synth(L, W, H, 24);
  synth(R, W, H, 48);
  double t = now_ms();
  printf("%-34s %9.3f ms\n", "synthetic pair (host)", t - t_start);
  double t0 = now_ms();
  int ndev = 0;
  gpc_hip_device_count(&ndev);
  t = now_ms(); printf("%-34s %9.3f ms  (%d devices)\n", "gpc_hip_device_count (hipInit)", t - t0, ndev); t0 = t;
  gpc_hip_ctx* c = nullptr;
  int st = gpc_hip_create(0, &c);
  t = now_ms(); printf("%-34s %9.3f ms  (status %d)\n", "gpc_hip_create", t - t0, st); t0 = t;
  if (st) return 1;
  gpc_filter_mask fm;
  st = gpc_hip_read_forest(argv[1], W, H, &fm);
  t = now_ms(); printf("%-34s %9.3f ms  (status %d, %d tests)\n", "gpc_hip_read_forest (host parse)", t - t0, st, fm.num_tests); t0 = t;
  st = gpc_hip_set_forest(c, &fm);
  t = now_ms(); printf("%-34s %9.3f ms  (status %d)\n", "gpc_hip_set_forest (first hipMalloc)", t - t0, st); t0 = t;
  gpc_settings s = {5, 128, 0, 1, 0, 1};
  if (warm) {
    st = gpc_hip_warmup(c, W, H, &s);
    t = now_ms(); printf("%-34s %9.3f ms  (status %d)\n", "gpc_hip_warmup", t - t0, st); t0 = t;
  }
  const size_t n = (size_t)W * H;
  for (int it = 0; it < repeat; ++it) {
    std::vector<uint8_t> smL(n), grL(n), smR(n), grR(n);
    std::vector<int32_t> mL(n), mR(n);
    int nl = 0, nr = 0, ns = 0;
    t0 = now_ms();
    st = gpc_hip_preprocess(c, L.data(), W, H, 5, smL.data(), grL.data(), mL.data(), (int)n, &nl);
    const double t1 = now_ms();
    st |= gpc_hip_preprocess(c, R.data(), W, H, 5, smR.data(), grR.data(), mR.data(), (int)n, &nr);
    const double t2 = now_ms();
    std::vector<gpc_support> out((size_t)(nl < nr ? nl : nr) + 1);
    st |= gpc_hip_rectified_match(c, smL.data(), grL.data(), mL.data(), nl, smR.data(), grR.data(), mR.data(), nr, W, H, &s,
                                  out.data(), (int)out.size(), &ns);
    const double t3 = now_ms();
    printf("call %d: preprocess L %8.3f ms, R %8.3f ms, rectified_match %8.3f ms, sum %8.3f ms  (status %d, candidates %d / %d, supports %d)\n",
           it + 1, t1 - t0, t2 - t1, t3 - t2, t3 - t0, st, nl, nr, ns);
  }
  for (int it = 0; it < repeat; ++it) {
    std::vector<gpc_support> out(n);
    int ns = 0, nl = 0, nr = 0;
    t0 = now_ms();
    st = gpc_hip_match_pair(c, L.data(), R.data(), W, H, &s, out.data(), (int)out.size(), &ns, &nl, &nr);
    t = now_ms();
    printf("match_pair %d: %8.3f ms  (status %d, supports %d)\n", it + 1, t - t0, st, ns);
  }
  t0 = now_ms();
  gpc_hip_destroy(c);
  t = now_ms(); printf("%-34s %9.3f ms\n", "gpc_hip_destroy", t - t0);
  printf("%-34s %9.3f ms\n", "whole process (main)", t - t_start);
  return 0;
}
