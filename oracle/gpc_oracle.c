/*
 * gpc_oracle.c -- scalar C restatement of the openGPC SSE inference path.
 *
 * TEST INFRASTRUCTURE ONLY (see gpc_oracle.h).  Every function cites the
 * reference lines (relative to /root/reference) whose *behaviour* it restates;
 * the code is written from the arithmetic specification in SURVEY.md 8(a), not
 * transliterated from the intrinsics.
 *
 * Conventions shared with the HIP path:
 *   - images are u8 [H][W] row-major with W % 16 == 0;
 *   - bytes the reference never writes (smooth/grad rows 0, H-3.. ) are 0;
 *   - the one out-of-buffer read the reference performs (in[-1], reached by
 *     box/sobel at y=1,x=0 through linear addressing) reads as 0.
 */
#include "gpc_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* synthetic inputs + checksums           */
/* ------------------------------------------------------------------ */

uint32_t gpc_oracle_mix(uint32_t a, uint32_t b) {
  uint32_t h = a * 73856093u ^ b * 19349663u;
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

static uint8_t synth_px(int s, int x, int y) {
  uint32_t coarse = gpc_oracle_mix((uint32_t)((x >> 2) + s * 4099), (uint32_t)(y >> 2)) & 0xFFu;
  uint32_t fine = gpc_oracle_mix((uint32_t)(x + s * 4099), (uint32_t)y) & 0x3Fu;
  return (uint8_t)((coarse * 3u + fine) >> 2);
}

void gpc_oracle_synth_pair(uint8_t* left, uint8_t* right, int W, int H, int s, int D) {
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      left[(size_t)y * W + x] = synth_px(s, x + D, y);
      right[(size_t)y * W + x] = synth_px(s, x + 2 * D, y);
    }
}

uint64_t gpc_oracle_fnv1a64(const void* data, uint64_t nbytes) {
  const uint8_t* p = (const uint8_t*)data;
  uint64_t h = 1469598103934665603ull;
  for (uint64_t i = 0; i < nbytes; ++i) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  return h;
}

/* ------------------------------------------------------------------ */
/* box  -- lib/gpc/filter.hpp:293-392                                  */
/* ------------------------------------------------------------------ */

/* linear-addressed pixel: the SSE loads at row-1 / row+1 simply run into the
 * neighbouring row at the image's left/right edge (filter.hpp:325-327). */
static inline int lin_px(const uint8_t* in, long k) { return k < 0 ? 0 : in[k]; }

static inline int third(int s) { return (s * 21846) >> 16; } /* mulhi_epi16(s, 21846) */
static inline int ninth(int s) { return (s * 7282) >> 16; }  /* mulhi_epi16(s, 7282)  */

static inline int hsum3(const uint8_t* in, long k) {
  return lin_px(in, k - 1) + lin_px(in, k) + lin_px(in, k + 1);
}

void gpc_oracle_box(const uint8_t* in, uint8_t* out, int W, int H) {
  /* rows are produced two at a time from y=1 while y < H-3 (filter.hpp:307,388):
   * last written row is H-4 for even H and H-3 for odd H. */
  int last = (H % 2 == 0) ? H - 4 : H - 3;
  for (int y = 1; y <= last; ++y)
    for (int x = 0; x < W; ++x) {
      long k = (long)y * W + x;
      int v = third(third(hsum3(in, k - W)) + third(hsum3(in, k)) + third(hsum3(in, k + W)));
      out[k] = (uint8_t)(v > 255 ? 255 : v);
    }
}

/* clearBoundary -- lib/gpc/buffer.hpp:630-654 (width == aligned width here) */
void gpc_oracle_clear_boundary(uint8_t* buf, int W, int H) {
  for (int y = 0; y < H; ++y) {
    buf[(size_t)y * W + 0] = 0;
    buf[(size_t)y * W + 1] = 0;
    buf[(size_t)y * W + W - 1] = 0;
  }
  memset(buf, 0, (size_t)W);
  if (H >= 2) memset(buf + (size_t)(H - 2) * W, 0, (size_t)2 * W);
}

/* ------------------------------------------------------------------ */
/* sobel -- lib/gpc/filter.hpp:404-519                                 */
/* ------------------------------------------------------------------ */

static int sobel_decision(const uint8_t* in, int W, int x, int y, int16_t thr_sq) {
  long k = (long)y * W + x;
  int l0 = lin_px(in, k - W - 1), c0 = lin_px(in, k - W), r0 = lin_px(in, k - W + 1);
  int l1 = lin_px(in, k - 1), r1 = lin_px(in, k + 1);
  int l2 = lin_px(in, k + W - 1), c2 = lin_px(in, k + W), r2 = lin_px(in, k + W + 1);
  int gx = ninth(l0 + l2 + 2 * l1) - ninth(r0 + r2 + 2 * r1);   /* filter.hpp:466-475 */
  int gy = ninth(l0 + r0 + 2 * c0) - ninth(l2 + r2 + 2 * c2);   /* filter.hpp:482-491 */
  int16_t mag = (int16_t)(gx * gx + gy * gy);                   /* <= 25538, no wrap  */
  return mag > thr_sq;                                          /* cmpgt_epi16 :505   */
}

void gpc_oracle_sobel(const uint8_t* in, uint8_t* grad, int W, int H, int thr) {
  /* threshold^2 goes through _mm_set1_epi16 (filter.hpp:418): it wraps for thr >= 182 */
  int16_t thr_sq = (int16_t)(uint16_t)((thr & 0xFF) * (thr & 0xFF));
  for (int y = 1; y <= H - 4; ++y)
    for (int x = 0; x < W; ++x) {
      /* unpacklo_epi8 of the 16-bit compare mask (filter.hpp:504-507): every 8-pixel
       * group shows its first four decisions, each twice. */
      int xs = (x & ~7) + ((x & 7) >> 1);
      grad[(size_t)y * W + x] = sobel_decision(in, W, xs, y, thr_sq) ? 255 : 0;
    }
}

/* arr2ind -- lib/gpc/filter.hpp:60-75 */
int gpc_oracle_arr2ind(const uint8_t* a, int n, int32_t* ind) {
  int m = 0;
  for (int i = 0; i < n; ++i)
    if (a[i]) ind[m++] = i;
  return m;
}

/* margin lambda -- lib/gpc/inference.hpp:318-325 */
int gpc_oracle_margin(const int32_t* idx, int m, int W, int H, int32_t* out) {
  int n = 0;
  for (int i = 0; i < m; ++i) {
    int x = idx[i] % W, y = idx[i] / W;
    if (y >= 13 && y < H - 13 && x >= 13 && x < W - 13) out[n++] = idx[i];
  }
  return n;
}

/* ------------------------------------------------------------------ */
/* gpcFilter / gpcFilterTau -- lib/gpc/filter.hpp:547-606, 619-683     */
/* ------------------------------------------------------------------ */

static inline int sat_s8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }

/* Pixels left of the 13-pixel margin in row 13 reach in front of the buffer (the
 * reference reads heap there; such pixels are never candidates).  Out-of-range
 * taps read as 0 here so that the whole code image is defined. */
static inline unsigned tap(const uint8_t* img, long k, long n) {
  return (k < 0 || k >= n) ? 0u : img[k];
}

static uint32_t fern_code(const uint8_t* smooth, long k, long n, int x, const gpc_oracle_forest* f) {
  uint32_t code = 0;
  for (int t = 0; t < f->num_tests; ++t) {
    unsigned a = tap(smooth, k + f->offs[2 * t], n);
    unsigned b = tap(smooth, k + f->offs[2 * t + 1], n);
    if (f->type != 0) /* _mm_subs_epi8(b, set1_epi8(tau)) then unsigned compare :647-652 */
      b = (unsigned)(uint8_t)sat_s8((int)(int8_t)b - (int)(int8_t)f->tau[t]);
    if (!(a > b)) continue;
    /* bit placement of the 4 byte planes + the 64-bit-lane carry of bitMask+=bitMask
     * at test 8 (filter.hpp:574-595) */
    if (t < 8) code |= 1u << t;
    else if (t == 8) code |= (x & 7) ? 1u : 0u;
    else code |= 1u << (t - 1);
  }
  return code;
}

void gpc_oracle_hash(const uint8_t* smooth, const uint8_t* grad, uint32_t* codes,
                     const gpc_oracle_forest* f, int W, int H) {
  /* rows 13 .. H-16 (filter.hpp:602); 16-pixel groups with no gradient byte are
   * skipped (:566) and keep whatever the caller put there (the API zero-fills). */
  for (int y = 13; y < H - 15; ++y)
    for (int x0 = 0; x0 < W; x0 += 16) {
      const uint8_t* g = grad + (size_t)y * W + x0;
      int any = 0;
      for (int i = 0; i < 16; ++i) any |= g[i];
      if (!any) continue;
      for (int i = 0; i < 16; ++i) {
        long k = (long)y * W + x0 + i;
        codes[k] = fern_code(smooth, k, (long)W * H, x0 + i, f);
      }
    }
}

/* ------------------------------------------------------------------ */
/* The -DSSE=OFF build: boxNaive / sobelNaive / gpcFilter(Tau)Naive     */
/* lib/gpc/filter.hpp:157-282                                          */
/* ------------------------------------------------------------------ */

/* Both naive filters walk nine pointers linearly over the image: output position o takes the
 * 3x3 window centred on o in LINEAR addressing, for o = W+1 .. (H-1)*W (filter.hpp:175-186,
 * 213-221).  The last windows reach two bytes past the buffer (read as 0 here). */
static inline int lin_px2(const uint8_t* in, long k, long n) { return (k < 0 || k >= n) ? 0 : in[k]; }

void gpc_oracle_box_naive(const uint8_t* in, uint8_t* out, int W, int H) {
  const long n = (long)W * H;
  for (long o = W + 1; o <= (long)(H - 1) * W; ++o) {
    int s = 0;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) s += lin_px2(in, o + (long)dy * W + dx, n);
    out[o] = (uint8_t)(s / 9); /* :218 */
  }
}

void gpc_oracle_sobel_naive(const uint8_t* in, uint8_t* grad, int W, int H, int thr) {
  const long n = (long)W * H;
  const int thr_sq = thr * thr; /* plain int (:159), no 16-bit wrap */
  for (long o = W + 1; o <= (long)(H - 1) * W; ++o) {
    const int p11 = lin_px2(in, o - W - 1, n), p12 = lin_px2(in, o - W, n), p13 = lin_px2(in, o - W + 1, n);
    const int p21 = lin_px2(in, o - 1, n), p23 = lin_px2(in, o + 1, n);
    const int p31 = lin_px2(in, o + W - 1, n), p32 = lin_px2(in, o + W, n), p33 = lin_px2(in, o + W + 1, n);
    const int sx = (p11 + p31 + 2 * p21 - p13 - 2 * p23 - p33) / 9; /* C division truncates (:179) */
    const int sy = (p11 + p13 + 2 * p12 - p31 - 2 * p32 - p33) / 9;
    grad[o] = (sx * sx + sy * sy > thr_sq) ? 255 : 0;
  }
}

void gpc_oracle_hash_naive(const uint8_t* smooth, const int32_t* mask, int n, uint32_t* codes,
                           const gpc_oracle_forest* f, int W, int H) {
  const long npx = (long)W * H;
  for (int i = 0; i < n; ++i) {
    const long k = mask[i];
    uint32_t code = 0;
    for (int t = 0; t < f->num_tests; ++t) { /* MSB first (:245-249) */
      const int a = (int)tap(smooth, k + f->offs[2 * t], npx);
      const int b = (int)tap(smooth, k + f->offs[2 * t + 1], npx);
      code <<= 1;
      if (f->type != 0 ? (a > b - f->tau[t]) : (a > b)) code |= 1u; /* plain int compare (:276) */
    }
    codes[k] = code;
  }
}

int gpc_oracle_preprocess_naive(const uint8_t* raw, int W, int H, int thr,
                                uint8_t* smooth, uint8_t* grad, int32_t* mask) {
  size_t n = (size_t)W * H;
  memset(smooth, 0, n);
  memset(grad, 0, n);
  gpc_oracle_box_naive(raw, smooth, W, H);
  gpc_oracle_clear_boundary(smooth, W, H);
  gpc_oracle_sobel_naive(raw, grad, W, H, thr);
  int32_t* idx = (int32_t*)malloc(n * sizeof(int32_t));
  int m = gpc_oracle_arr2ind(grad, (int)n, idx);
  int k = gpc_oracle_margin(idx, m, W, H, mask);
  free(idx);
  return k;
}

/* ------------------------------------------------------------------ */
/* readForest -- lib/gpc/inference.hpp:404-446                         */
/* ------------------------------------------------------------------ */

static const char* next_token(const char* p, char* tok, int cap) {
  while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v') ++p;
  if (!*p) return NULL;
  int n = 0;
  while (*p && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f' || *p == '\v')) {
    if (n < cap - 1) tok[n++] = *p;
    ++p;
  }
  tok[n] = 0;
  return p;
}

static int next_int(const char** p, int* v) {
  char tok[64];
  char* end;
  const char* q = next_token(*p, tok, sizeof tok);
  if (!q) return -1;
  long l = strtol(tok, &end, 10);
  if (end == tok) return -1;
  *v = (int)l;
  *p = q;
  return 0;
}

int gpc_oracle_parse_forest_text(const char* text, int W, int H, gpc_oracle_forest* f) {
  memset(f, 0, sizeof *f);
  f->width = W;
  f->height = H;
  const char* p = text;
  int num_ferns, nonzero_tau = 0;
  if (next_int(&p, &num_ferns)) return -2;
  for (int i = 0; i < num_ferns; ++i) {
    int fern_id, num_tests;
    char scale[64];
    if (next_int(&p, &fern_id)) return -2;
    p = next_token(p, scale, sizeof scale);
    if (!p) return -2;
    if (next_int(&p, &num_tests)) return -2;
    for (int j = 0; j < num_tests; ++j) {
      int v[6]; /* levelID ix iy jx jy tau; levelID is parsed and ignored (:421-424) */
      for (int q = 0; q < 6; ++q)
        if (next_int(&p, &v[q])) return -2;
      if (f->num_tests < GPC_ORACLE_MAX_TESTS) { /* :426 */
        int t = f->num_tests++;
        f->offs[2 * t] = v[1] + v[2] * W;
        f->offs[2 * t + 1] = v[3] + v[4] * W;
        f->dxy[4 * t + 0] = v[1];
        f->dxy[4 * t + 1] = v[2];
        f->dxy[4 * t + 2] = v[3];
        f->dxy[4 * t + 3] = v[4];
        f->tau[t] = v[5];
      } else {
        f->discarded++; /* "Note: A maximum of 32 fern features..." :431 */
      }
      if (v[5] != 0) nonzero_tau++; /* counted for discarded tests too (:433) */
    }
  }
  f->type = nonzero_tau ? 1 : 0;
  return 0;
}

int gpc_oracle_read_forest(const char* path, int W, int H, gpc_oracle_forest* f) {
  FILE* fp = fopen(path, "rb");
  if (!fp) { /* "Error opening forest file" -> empty mask, type 0 (:409-412) */
    memset(f, 0, sizeof *f);
    f->width = W;
    f->height = H;
    return -1;
  }
  fseek(fp, 0, SEEK_END);
  long n = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  char* text = (char*)malloc((size_t)n + 1);
  size_t got = fread(text, 1, (size_t)n, fp);
  text[got] = 0;
  fclose(fp);
  int rc = gpc_oracle_parse_forest_text(text, W, H, f);
  free(text);
  return rc;
}

/* ------------------------------------------------------------------ */
/* preprocessImage -- lib/gpc/inference.hpp:302-333                    */
/* ------------------------------------------------------------------ */

int gpc_oracle_preprocess(const uint8_t* raw, int W, int H, int thr,
                          uint8_t* smooth, uint8_t* grad, int32_t* mask) {
  size_t n = (size_t)W * H;
  memset(smooth, 0, n);
  memset(grad, 0, n);
  gpc_oracle_box(raw, smooth, W, H);
  gpc_oracle_clear_boundary(smooth, W, H);
  gpc_oracle_sobel(raw, grad, W, H, thr); /* on the RAW image (:313) */
  int32_t* idx = (int32_t*)malloc(n * sizeof(int32_t));
  int m = gpc_oracle_arr2ind(grad, (int)n, idx);
  int k = gpc_oracle_margin(idx, m, W, H, mask);
  free(idx);
  return k;
}

/* evalFastMaskOnSubsetSSE gather + epipolar key -- inference.hpp:282-290, 192-197 */
void gpc_oracle_descriptors(const uint32_t* codes, const int32_t* mask, int n, int W,
                            int epipolar, uint64_t* state) {
  for (int i = 0; i < n; ++i) {
    uint64_t s = codes[mask[i]];
    if (epipolar) s |= (uint64_t)(uint32_t)(mask[i] / W) << 32;
    state[i] = s;
  }
}

/* ------------------------------------------------------------------ */
/* findCorrespondences -- lib/gpc/inference.hpp:227-254                */
/* ------------------------------------------------------------------ */

typedef struct {
  uint64_t state;
  int32_t k;
} desc_t;

/* The reference uses std::sort (unstable).  Order among equal states only ever
 * matters for tail quirk Q2; this oracle and the HIP path both
 * define it as "mask order" (a stable sort). */
static void merge_sort(desc_t* a, desc_t* tmp, int n) {
  for (int w = 1; w < n; w *= 2) {
    for (int lo = 0; lo < n; lo += 2 * w) {
      int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int i = lo, j = mid, o = lo;
      while (i < mid && j < hi) tmp[o++] = (a[j].state < a[i].state) ? a[j++] : a[i++];
      while (i < mid) tmp[o++] = a[i++];
      while (j < hi) tmp[o++] = a[j++];
    }
    memcpy(a, tmp, (size_t)n * sizeof(desc_t));
  }
}

int gpc_oracle_find_correspondences(const uint64_t* ss, const int32_t* sk, int ns,
                                    const uint64_t* ts, const int32_t* tk, int nt,
                                    int W, gpc_oracle_corr* out) {
  if (ns <= 0 || nt <= 0) return 0; /* nt == 0 is UB in the reference (size()-1) */
  int cap = ns > nt ? ns : nt;
  desc_t* S = (desc_t*)malloc((size_t)ns * sizeof(desc_t));
  desc_t* T = (desc_t*)malloc((size_t)nt * sizeof(desc_t));
  desc_t* tmp = (desc_t*)malloc((size_t)cap * sizeof(desc_t));
  for (int i = 0; i < ns; ++i) { S[i].state = ss[i]; S[i].k = sk[i]; }
  for (int i = 0; i < nt; ++i) { T[i].state = ts[i]; T[i].k = tk[i]; }
  merge_sort(S, tmp, ns);
  merge_sort(T, tmp, nt);
  int n = 0;
  int j = 0;
  const int last = nt - 1;
  for (int i = 0; i < ns;) {
    int run = 1;
    while (i + run < ns && S[i + run].state == S[i].state) ++run;
    if (run == 1) {
      uint64_t s = S[i].state;
      /* lower bound restricted to [0, nt-1): the last sorted target is never
       * inspected by the search (:243-246) ... */
      while (j < last && T[j].state < s) ++j;
      /* ... so it can never match (Q1), and a hit at nt-2 skips the target
       * uniqueness test (Q2) (:248-249). */
      if (j != last && T[j].state == s && (j + 1 == last || T[j + 1].state != s)) {
        out[n].sx = S[i].k % W;
        out[n].sy = S[i].k / W;
        out[n].tx = T[j].k % W;
        out[n].ty = T[j].k / W;
        ++n;
      }
    }
    i += run;
  }
  free(S);
  free(T);
  free(tmp);
  return n;
}

/* ------------------------------------------------------------------ */
/* Hashmatch -- lib/gpc/hashmatch.hpp, driven by inference.hpp:204-225  */
/* ------------------------------------------------------------------ */

#define HM_BUCKETS 214673 /* inference.hpp:210 */
#define HM_CAP 10         /* terminateAfter, hashmatch.hpp:93 */

typedef struct {
  uint64_t state;
  int32_t k;
  int32_t src; /* srcDescr */
} hm_elem;

typedef struct {
  hm_elem e[HM_CAP];
  int n;
} hm_bucket;

/* OrderedLinkedList::insert (hashmatch.hpp:91-131): full buckets drop the value; otherwise it
 * goes behind every element <= it, so equal states keep their insertion order. */
static void hm_insert(hm_bucket* b, hm_elem v) {
  if (b->n >= HM_CAP) return;
  int pos = 0;
  while (pos < b->n && b->e[pos].state <= v.state) ++pos;
  for (int i = b->n; i > pos; --i) b->e[i] = b->e[i - 1];
  b->e[pos] = v;
  b->n++;
}

/* OrderedLinkedList::getDuplicates (hashmatch.hpp:162-197) on the ordered array. */
static int hm_pairs(const hm_bucket* b, int W, gpc_oracle_corr* out) {
  int n = 0, i = 0;
  const hm_elem* a = b->e;
  while (i < b->n) {
    int p = i;
    ++i;
    if (i < b->n && a[p].state == a[i].state) {
      if (a[p].src != a[i].src) {
        int emit = 0;
        if (i + 1 < b->n) {
          emit = a[i + 1].state != a[i].state;          /* third element differs (:175) */
        } else {
          emit = 1;                                     /* no third element (:180) */
        }
        if (emit) {
          out[n].sx = a[p].k % W;
          out[n].sy = a[p].k / W;
          out[n].tx = a[i].k % W;
          out[n].ty = a[i].k / W;
          ++n;
        }
        if (i + 1 < b->n && i + 2 >= b->n) return n;    /* "last triplet": leave the bucket (:178) */
      } else if (i + 1 < b->n && a[i].src != a[i + 1].src) {
        ++i;                                            /* skip over a false pair (:188-192) */
      }
    }
  }
  return n;
}

int gpc_oracle_hash_correspondences(const uint64_t* ss, const int32_t* sk, int ns,
                                    const uint64_t* ts, const int32_t* tk, int nt,
                                    int W, gpc_oracle_corr* out) {
  hm_bucket* tab = (hm_bucket*)calloc(HM_BUCKETS, sizeof(hm_bucket));
  for (int i = 0; i < ns; ++i) { /* all source descriptors first (inference.hpp:213-216) */
    hm_elem v = {ss[i], sk[i], 1};
    hm_insert(&tab[ss[i] % HM_BUCKETS], v);
  }
  for (int i = 0; i < nt; ++i) {
    hm_elem v = {ts[i], tk[i], 0};
    hm_insert(&tab[ts[i] % HM_BUCKETS], v);
  }
  int n = 0;
  for (int b = 0; b < HM_BUCKETS; ++b)
    if (tab[b].n) n += hm_pairs(&tab[b], W, out + n);
  free(tab);
  return n;
}

/* rectifiedMatch filter -- lib/gpc/inference.hpp:384-391 */
int gpc_oracle_rectified_filter(const gpc_oracle_corr* c, int n,
                                const gpc_oracle_settings* s, gpc_oracle_support* out) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    int dy = c[i].sy - c[i].ty, dx = c[i].sx - c[i].tx;
    if (abs(dy) <= s->vertical_tolerance && abs(dx) <= s->disp_high) {
      out[m].x = c[i].sx;
      out[m].y = c[i].sy;
      out[m].d = (float)dx;
      ++m;
    }
  }
  return m;
}

/* ------------------------------------------------------------------ */
/* the timed region of samples/sparsematch.cpp:45-52                   */
/* ------------------------------------------------------------------ */

int gpc_oracle_match_pair(const uint8_t* rawL, const uint8_t* rawR, int W, int H,
                          const gpc_oracle_forest* f, const gpc_oracle_settings* s,
                          gpc_oracle_support* out, int32_t* n_cand_l, int32_t* n_cand_r) {
  size_t n = (size_t)W * H;
  const uint8_t* raw[2] = {rawL, rawR};
  uint8_t* smooth = (uint8_t*)malloc(n);
  uint8_t* grad = (uint8_t*)malloc(n);
  uint32_t* codes = (uint32_t*)malloc(n * sizeof(uint32_t));
  int32_t* mask[2];
  uint64_t* state[2];
  int cnt[2];
  for (int im = 0; im < 2; ++im) {
    mask[im] = (int32_t*)malloc(n * sizeof(int32_t));
    memset(codes, 0, n * sizeof(uint32_t)); /* Buffer<uint32_t>(rows, cols, 0) :274 */
    if (s->naive) {
      cnt[im] = gpc_oracle_preprocess_naive(raw[im], W, H, s->gradient_threshold, smooth, grad, mask[im]);
      gpc_oracle_hash_naive(smooth, mask[im], cnt[im], codes, f, W, H);
    } else {
      cnt[im] = gpc_oracle_preprocess(raw[im], W, H, s->gradient_threshold, smooth, grad, mask[im]);
      gpc_oracle_hash(smooth, grad, codes, f, W, H);
    }
    state[im] = (uint64_t*)malloc((size_t)(cnt[im] + 1) * sizeof(uint64_t));
    gpc_oracle_descriptors(codes, mask[im], cnt[im], W, s->epipolar_mode, state[im]);
  }
  int capc = cnt[0] > 0 ? cnt[0] : 1;
  gpc_oracle_corr* corr = (gpc_oracle_corr*)malloc((size_t)capc * sizeof(gpc_oracle_corr));
  int nc = s->use_hashtable
               ? gpc_oracle_hash_correspondences(state[0], mask[0], cnt[0], state[1], mask[1], cnt[1], W, corr)
               : gpc_oracle_find_correspondences(state[0], mask[0], cnt[0], state[1], mask[1], cnt[1], W, corr);
  int m = gpc_oracle_rectified_filter(corr, nc, s, out);
  if (n_cand_l) *n_cand_l = cnt[0];
  if (n_cand_r) *n_cand_r = cnt[1];
  free(corr);
  for (int im = 0; im < 2; ++im) { free(mask[im]); free(state[im]); }
  free(codes);
  free(grad);
  free(smooth);
  return m;
}
