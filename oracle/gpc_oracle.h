/*
 * gpc_oracle.h -- CPU oracle for the openGPC sparse-stereo hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a scalar C restatement of the reference's
 * SSE code path (lib/gpc/filter.hpp, lib/gpc/inference.hpp, lib/gpc/buffer.hpp)
 * used as the checker in tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py.  Nothing in the product path (opengpc_amd/, include/) may
 * link, import or call it.
 *
 * Parity pin: SURVEY.md Appendix C known-answer vectors (FNV-1a-64 hashes of
 * raw/smooth/grad/mask/codes/supports produced by the compiled reference) are
 * committed as tests/golden/appendix_c.json and checked by
 * tests/test_oracle_golden.py; the raw-pointer kernels are additionally checked
 * against the real reference kernels (oracle/_ref, built from
 * /root/reference/lib/gpc/filter.hpp where it lies) by tests/test_oracle_vs_ref.py.
 */
#ifndef GPC_ORACLE_H
#define GPC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPC_ORACLE_MAX_TESTS 32

typedef struct {
  int32_t x, y;
  float d;
} gpc_oracle_support;

typedef struct {
  int32_t sx, sy, tx, ty;
} gpc_oracle_corr;

typedef struct {
  int32_t offs[2 * GPC_ORACLE_MAX_TESTS]; /* mask[2t]=ix+iy*W, mask[2t+1]=jx+jy*W */
  int32_t dxy[4 * GPC_ORACLE_MAX_TESTS];  /* ix,iy,jx,jy per kept test            */
  int32_t tau[GPC_ORACLE_MAX_TESTS];
  int32_t num_tests;  /* kept tests (<=32)                         */
  int32_t type;       /* 0 = all tau zero, 1 = some tau non-zero   */
  int32_t discarded;  /* tests beyond the 32nd                     */
  int32_t width, height;
} gpc_oracle_forest;

typedef struct {
  int32_t gradient_threshold; /* 0..255 */
  int32_t disp_high;
  int32_t vertical_tolerance;
  int32_t epipolar_mode;
  int32_t use_hashtable; /* 0: sort-match (findCorrespondences), 1: ndb::Hashmatch */
  int32_t naive;         /* 0: the reference built with -D_INTRINSICS_SSE (default, parity target);
                            1: the reference built with SSE=OFF (*Naive kernels, filter.hpp:157-282) */
} gpc_oracle_settings;

/* synthetic inputs, SURVEY.md 8(d) */
uint32_t gpc_oracle_mix(uint32_t a, uint32_t b);
void gpc_oracle_synth_pair(uint8_t* left, uint8_t* right, int W, int H, int s, int D);
uint64_t gpc_oracle_fnv1a64(const void* data, uint64_t nbytes);

/* filter.hpp kernels (SSE semantics) */
void gpc_oracle_box(const uint8_t* in, uint8_t* out, int W, int H);
void gpc_oracle_clear_boundary(uint8_t* buf, int W, int H);
void gpc_oracle_sobel(const uint8_t* in, uint8_t* grad, int W, int H, int thr);
int gpc_oracle_arr2ind(const uint8_t* a, int n, int32_t* ind);
int gpc_oracle_margin(const int32_t* idx, int m, int W, int H, int32_t* out);
void gpc_oracle_hash(const uint8_t* smooth, const uint8_t* grad, uint32_t* codes,
                     const gpc_oracle_forest* f, int W, int H);

/* filter.hpp *Naive kernels (the reference's -DSSE=OFF build) */
void gpc_oracle_box_naive(const uint8_t* in, uint8_t* out, int W, int H);
void gpc_oracle_sobel_naive(const uint8_t* in, uint8_t* grad, int W, int H, int thr);
/* codes only at mask positions (gpcFilterNaive / gpcFilterTauNaive walk idx) */
void gpc_oracle_hash_naive(const uint8_t* smooth, const int32_t* mask, int n, uint32_t* codes,
                           const gpc_oracle_forest* f, int W, int H);
int gpc_oracle_preprocess_naive(const uint8_t* raw, int W, int H, int thr,
                                uint8_t* smooth, uint8_t* grad, int32_t* mask);

/* inference.hpp */
int gpc_oracle_parse_forest_text(const char* text, int W, int H, gpc_oracle_forest* f);
int gpc_oracle_read_forest(const char* path, int W, int H, gpc_oracle_forest* f);
/* smooth, grad: W*H bytes (never-written rows come out 0); mask: capacity W*H */
int gpc_oracle_preprocess(const uint8_t* raw, int W, int H, int thr,
                          uint8_t* smooth, uint8_t* grad, int32_t* mask);
/* states in mask order (x = k%W, y = k/W, state = code | y<<32 if epipolar) */
void gpc_oracle_descriptors(const uint32_t* codes, const int32_t* mask, int n, int W,
                            int epipolar, uint64_t* state);
/* findCorrespondences on (state, linear index) sets; returns count */
int gpc_oracle_find_correspondences(const uint64_t* ss, const int32_t* sk, int ns,
                                    const uint64_t* ts, const int32_t* tk, int nt,
                                    int W, gpc_oracle_corr* out);
/* Hashmatch path of depthPriorFast (inference.hpp:204-225, hashmatch.hpp): 214673 buckets of
 * ordered lists capped at 10 entries, pair extraction per bucket.  Same in/out as above. */
int gpc_oracle_hash_correspondences(const uint64_t* ss, const int32_t* sk, int ns,
                                    const uint64_t* ts, const int32_t* tk, int nt,
                                    int W, gpc_oracle_corr* out);
int gpc_oracle_rectified_filter(const gpc_oracle_corr* c, int n,
                                const gpc_oracle_settings* s, gpc_oracle_support* out);
/* whole timed region t0..t2 of samples/sparsematch.cpp: raw pair -> supports.
 * out capacity must be >= W*H.  Optional outputs may be NULL. */
int gpc_oracle_match_pair(const uint8_t* rawL, const uint8_t* rawR, int W, int H,
                          const gpc_oracle_forest* f, const gpc_oracle_settings* s,
                          gpc_oracle_support* out, int32_t* n_cand_l, int32_t* n_cand_r);

#ifdef __cplusplus
}
#endif
#endif
