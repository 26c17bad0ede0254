// sanitize_host -- the host-only code paths of the product under AddressSanitizer / UBSan / ThreadSanitizer (CPU build
// container only; sanitizers never run on the GPU box).  Built twice by tests/test_sanitizers.py:
//   -fsanitize=address,undefined  against the library compiled the same way:  forest | expand | pool | png
//   -fsanitize=thread             against the library compiled the same way:  pool
// Every sub-command prints "OK <what>" lines; a sanitizer report aborts the process with a non-zero status.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gpc_hip.h"
#ifndef NO_PNG
#include "gpc/png_io.hpp"
#endif

extern "C" int gpc_hip_debug_expand_pool(const uint32_t* packed, const int32_t* rows, int hpad, int H, int npairs,
                                         const int32_t* counts, int cap, int threads, int parts, gpc_support* out,
                                         const uint8_t* copy_src, uint8_t* copy_dst, size_t copy_bytes);

extern "C" int gpc_hip_debug_fingerprint(const uint8_t* smooth, const uint8_t* grad, size_t n, const int32_t* mask, int n_mask,
                                         int full, uint64_t* fp, const void* a, size_t na, const void* b, size_t nb, int* overlap);

static uint32_t rng_state = 12345u;
static uint32_t rnd() {
  rng_state = rng_state * 1664525u + 1013904223u;
  return rng_state >> 8;
}

// forest texts the parser must survive: truncated at every byte, non-numeric tokens, > 32 tests, absurd counts
static int cmd_forest(const char* path) {
  FILE* fp = fopen(path, "rb");
  if (!fp) return 2;
  std::string text;
  char buf[4096];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, got);
  fclose(fp);
  gpc_filter_mask fm;
  int st = gpc_hip_parse_forest(text.c_str(), 1024, 436, &fm);
  printf("OK forest whole status %d tests %d type %d\n", st, fm.num_tests, fm.type);
  int ok = 0, bad = 0;
  for (size_t cut = 0; cut < text.size(); cut += 1 + text.size() / 400) {   // every prefix (stepped): truncated files
    std::string t = text.substr(0, cut);
    st = gpc_hip_parse_forest(t.c_str(), 1024, 436, &fm);
    (st == GPC_OK ? ok : bad)++;
    if (fm.num_tests < 0 || fm.num_tests > GPC_MAX_TESTS) return 3;
  }
  printf("OK forest prefixes accepted %d refused %d\n", ok, bad);
  const char* nasty[] = {"", " ", "x", "1", "1 0", "1 0 s", "1 0 s 2 0 1 2", "-1", "999999999", "1 0 s 999999999 0 1 1 1 1 0",
                         "2 0 s 1 0 a b c d e", "1 0 s 1 0 99999999999999999999 1 1 1 0", "1 0 s 1 0 1 1 1 1 99999999999",
                         "1 0 s 1 0 -2147483648 -2147483648 2147483647 2147483647 -2147483648"};
  for (const char* t : nasty) {
    st = gpc_hip_parse_forest(t, 1024, 436, &fm);
    if (fm.num_tests < 0 || fm.num_tests > GPC_MAX_TESTS) return 3;
  }
  printf("OK forest nasty texts %zu\n", sizeof nasty / sizeof nasty[0]);
  // 16 ferns x 20 tests: the first 32 are kept, 288 discarded (inference.hpp:425-432)
  std::string big = "16\n";
  for (int f = 0; f < 16; ++f) {
    big += std::to_string(f) + " s 20\n";
    for (int t = 0; t < 20; ++t) big += std::to_string(t) + " 1 -2 3 -4 " + std::to_string((t % 5) - 2) + "\n";
  }
  st = gpc_hip_parse_forest(big.c_str(), 1024, 436, &fm);
  printf("OK forest 16x20 status %d tests %d discarded %d\n", st, fm.num_tests, fm.discarded);
  return (fm.num_tests == 32 && fm.discarded == 288) ? 0 : 4;
}

struct Packed {
  int H, npairs, hpad;
  std::vector<int32_t> rows, counts;
  std::vector<uint32_t> words;
};
static Packed make_packed(int H, int npairs, int maxrow, int cap) {
  Packed p;
  p.H = H;
  p.npairs = npairs;
  p.hpad = (H + 3) & ~3;
  p.rows.assign((size_t)npairs * p.hpad, 0);
  p.counts.assign(npairs, 0);
  for (int i = 0; i < npairs; ++i) {
    long tot = 0;
    for (int y = 13; y < H - 13; ++y) {
      const int c = (rnd() % 7 == 0) ? 0 : (int)(rnd() % (unsigned)(maxrow + 1));   // ragged rows, empty ones among them
      p.rows[(size_t)i * p.hpad + y] = c;
      tot += c;
    }
    p.counts[i] = (int32_t)tot;
    const long lim = tot < cap ? tot : cap;
    for (long k = 0; k < lim; ++k) p.words.push_back((rnd() & 0x3FF) | ((rnd() & 0x3FF) << 16));
  }
  p.words.reserve(p.words.size() + 1);   // (data() of an empty vector may be null: the entry point refuses null arrays)
  return p;
}
static uint64_t fnv(const void* data, size_t n) {
  const uint8_t* b = (const uint8_t*)data;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}

// gpc_hip_expand_packed: capacities of 0, 1, exactly the count, and beyond it; aligned and unaligned output arrays
static int cmd_expand() {
  for (int H : {27, 64, 436}) {
    for (int trial = 0; trial < 6; ++trial) {
      Packed p = make_packed(H, 1, 40, 1 << 30);
      const int n = p.counts[0];
      for (int take : {0, 1, n / 2, n}) {
        if (take > n) continue;
        for (int mis = 0; mis < 2; ++mis) {   // 16-byte aligned (SSE2 streaming stores) and not
          std::vector<uint8_t> raw((size_t)take * sizeof(gpc_support) + 64);
          uintptr_t a = ((uintptr_t)raw.data() + 15) & ~(uintptr_t)15;
          gpc_support* out = (gpc_support*)(a + (mis ? 4 : 0));
          if (gpc_hip_expand_packed(p.words.data(), p.rows.data(), H, take, out) != GPC_OK) return 3;
          long pos = 0;   // check against the definition
          for (int y = 13; y < H - 13 && pos < take; ++y)
            for (int k = 0; k < p.rows[y] && pos < take; ++k, ++pos) {
              const uint32_t w = p.words[pos];
              const int xl = w & 0xFFFF, xr = w >> 16;
              if (out[pos].x != xl || out[pos].y != y || out[pos].d != (float)(xl - xr)) return 4;
            }
        }
      }
    }
  }
  if (gpc_hip_expand_packed(nullptr, nullptr, 436, 0, nullptr) != GPC_E_INVALID) return 5;
  printf("OK expand\n");
  return 0;
}

// the expansion pool with 1 .. 8 workers, 1 .. 4 row ranges per pair, capacities below and above the pairs' counts
static int cmd_pool() {
  uint64_t want = 0;
  for (int threads : {1, 2, 3, 8}) {
    for (int parts : {1, 2, 4}) {
      rng_state = 777u;
      const int H = 96, npairs = 23, cap = 1500;
      Packed p = make_packed(H, npairs, 60, cap);   // some pairs have more supports than `cap`
      std::vector<gpc_support> out((size_t)npairs * cap);
      memset(out.data(), 0, out.size() * sizeof(gpc_support));
      std::vector<uint8_t> src(300000), dst(300000, 0);
      for (size_t i = 0; i < src.size(); ++i) src[i] = (uint8_t)rnd();
      if (gpc_hip_debug_expand_pool(p.words.data(), p.rows.data(), p.hpad, H, npairs, p.counts.data(), cap, threads, parts,
                                    out.data(), src.data(), dst.data(), src.size()) != GPC_OK)
        return 3;
      if (memcmp(src.data(), dst.data(), src.size()) != 0) return 4;
      const uint64_t h = fnv(out.data(), out.size() * sizeof(gpc_support));
      if (!want) want = h;
      if (h != want) return 5;   // the same records whatever the split
      // and equal to the single-threaded definition
      const uint32_t* recs = p.words.data();
      for (int i = 0; i < npairs; ++i) {
        const int lim = p.counts[i] < cap ? p.counts[i] : cap;
        std::vector<gpc_support> one(lim + 1);
        gpc_hip_expand_packed(recs, p.rows.data() + (size_t)i * p.hpad, H, lim, one.data());
        if (memcmp(one.data(), out.data() + (size_t)i * cap, (size_t)lim * sizeof(gpc_support)) != 0) return 6;
        recs += lim;
      }
    }
  }
  printf("OK pool\n");
  return 0;
}

#ifndef NO_PNG
// PNG files the decoder must refuse without reading or allocating out of bounds: every truncation of a valid file,
// headers that promise more than the data can hold, filter bytes that do not exist, IDAT that inflates to the wrong size
static int cmd_png(const char* dir) {
  const std::string d(dir);
  ndb::pngio::Image img;
  int st = ndb::pngio::decode_file(d + "/good.png", img);
  if (st != 0 || img.width != 40 || img.height != 30) return 3;
  FILE* fp = fopen((d + "/good.png").c_str(), "rb");
  std::vector<uint8_t> good;
  uint8_t buf[4096];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, fp)) > 0) good.insert(good.end(), buf, buf + got);
  fclose(fp);
  auto write = [&](const std::vector<uint8_t>& v) {
    FILE* f = fopen((d + "/t.png").c_str(), "wb");
    if (!v.empty()) fwrite(v.data(), 1, v.size(), f);
    fclose(f);
  };
  int refused = 0, accepted = 0;
  for (size_t cut = 0; cut < good.size(); ++cut) {   // truncated at every byte
    write(std::vector<uint8_t>(good.begin(), good.begin() + cut));
    ndb::pngio::Image t;
    (ndb::pngio::decode_file(d + "/t.png", t) == 0 ? accepted : refused)++;
  }
  printf("OK png truncations refused %d accepted %d\n", refused, accepted);
  // forged headers: width / height / depth / colour type / interlace of the IHDR chunk (bytes 16 .. 28); chunk CRCs are not checked
  const uint32_t dims[] = {0u, 1u, 41u, 0x7FFFFFFFu, 0xFFFFFFFFu, 1u << 24, 65536u};
  for (uint32_t w : dims)
    for (uint32_t h : dims) {
      std::vector<uint8_t> v = good;
      v[16] = w >> 24; v[17] = w >> 16; v[18] = w >> 8; v[19] = w;
      v[20] = h >> 24; v[21] = h >> 16; v[22] = h >> 8; v[23] = h;
      write(v);
      ndb::pngio::Image t;
      if (ndb::pngio::decode_file(d + "/t.png", t) == 0 && !(w == 40 && h == 30)) return 4;   // nothing but the true size decodes
    }
  for (int depth : {0, 1, 4, 8, 16, 32})
    for (int ct = 0; ct < 8; ++ct)
      for (int il = 0; il < 2; ++il) {
        std::vector<uint8_t> v = good;
        v[24] = (uint8_t)depth; v[25] = (uint8_t)ct; v[28] = (uint8_t)il;
        write(v);
        ndb::pngio::Image t;
        (void)ndb::pngio::decode_file(d + "/t.png", t);
      }
  printf("OK png forged headers\n");
  // every filter type incl. ones that do not exist, random pixels: files made here with zlib
  for (int ft = 0; ft < 7; ++ft) {
    const int W = 33, H = 9;
    std::vector<uint8_t> raw((size_t)(W + 1) * H);
    for (auto& b : raw) b = (uint8_t)rnd();
    for (int y = 0; y < H; ++y) raw[(size_t)(W + 1) * y] = (uint8_t)ft;
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(clen);
    compress(z.data(), &clen, raw.data(), (uLong)raw.size());
    FILE* f = fopen((d + "/t.png").c_str(), "wb");
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    uint8_t ihdr[13] = {0, 0, 0, (uint8_t)W, 0, 0, 0, (uint8_t)H, 8, 0, 0, 0, 0};
    ndb::pngio::put_chunk(f, "IHDR", ihdr, 13);
    ndb::pngio::put_chunk(f, "IDAT", z.data(), clen);
    ndb::pngio::put_chunk(f, "IEND", nullptr, 0);
    fclose(f);
    ndb::pngio::Image t;
    st = ndb::pngio::decode_file(d + "/t.png", t);
    if ((ft <= 4) != (st == 0)) return 5;
  }
  printf("OK png filters\n");
  return 0;
}
#endif

// The fingerprint of a resident image's host copies: arrays of EXACTLY the sizes given (heap blocks: a read past either end
// is a sanitizer report), sizes from 0 bytes up and not multiples of 8, both modes; one changed byte anywhere changes the
// full fingerprint, a changed first / last word changes the sampled one; the overlap test on touching and nested ranges.
static int cmd_fingerprint() {
  int checked = 0;
  for (size_t n : {size_t(1), size_t(7), size_t(8), size_t(9), size_t(63), size_t(64), size_t(521), size_t(4099), size_t(70001)}) {
    for (int n_mask : {0, 1, 2, 3, 17, 1000, 20011}) {
      std::vector<uint8_t> sm(n), gr(n);
      std::vector<int32_t> mk((size_t)n_mask);
      for (auto& v : sm) v = (uint8_t)rnd();
      for (auto& v : gr) v = (uint8_t)(rnd() & 1 ? 255 : 0);
      for (auto& v : mk) v = (int32_t)rnd();
      for (int full = 0; full < 2; ++full) {
        uint64_t f0 = 0, f1 = 0;
        if (gpc_hip_debug_fingerprint(sm.data(), gr.data(), n, mk.data(), n_mask, full, &f0, nullptr, 0, nullptr, 0, nullptr)) return 3;
        if (gpc_hip_debug_fingerprint(sm.data(), gr.data(), n, mk.data(), n_mask, full, &f1, nullptr, 0, nullptr, 0, nullptr) || f0 != f1) return 4;
        const size_t at = full ? (size_t)(rnd() % n) : (rnd() & 1 ? 0 : n - 1);   // sampled: first and last word are always in
        sm[at] ^= 0x40;
        if (gpc_hip_debug_fingerprint(sm.data(), gr.data(), n, mk.data(), n_mask, full, &f1, nullptr, 0, nullptr, 0, nullptr) || f0 == f1) return 5;
        sm[at] ^= 0x40;
        if (n_mask) {
          const size_t am = full ? (size_t)(rnd() % (unsigned)n_mask) : (rnd() & 1 ? 0 : (size_t)n_mask - 1);
          mk[am] ^= 1;
          if (gpc_hip_debug_fingerprint(sm.data(), gr.data(), n, mk.data(), n_mask, full, &f1, nullptr, 0, nullptr, 0, nullptr) || f0 == f1) return 6;
          mk[am] ^= 1;
        }
        ++checked;
      }
    }
  }
  char buf[64];
  struct { size_t a0, na, b0, nb; int want; } cases[] = {{0, 16, 16, 16, 0}, {0, 17, 16, 16, 1}, {8, 8, 0, 64, 1}, {0, 64, 8, 8, 1},
                                                         {0, 0, 0, 64, 0}, {32, 8, 0, 32, 0}, {0, 64, 63, 1, 1}};
  uint64_t f = 0;
  uint8_t one = 0;
  for (auto& c : cases) {
    int ov = -1;
    if (gpc_hip_debug_fingerprint(&one, &one, 1, nullptr, 0, 0, &f, buf + c.a0, c.na, buf + c.b0, c.nb, &ov) || ov != c.want) return 7;
  }
  printf("OK fingerprint %d array sets, sampled and full, overlap cases %zu\n", checked, sizeof cases / sizeof cases[0]);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string cmd = argv[1];
  if (cmd == "fingerprint") return cmd_fingerprint();
  if (cmd == "forest" && argc == 3) return cmd_forest(argv[2]);
  if (cmd == "expand") return cmd_expand();
  if (cmd == "pool") return cmd_pool();
#ifndef NO_PNG
  if (cmd == "png" && argc == 3) return cmd_png(argv[2]);
#endif
  return 2;
}
