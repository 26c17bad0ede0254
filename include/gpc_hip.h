/*
 * gpc_hip.h -- C ABI of libgpc_hip.so, the MI355X (gfx950) implementation of the
 * openGPC sparse-stereo hot path: preprocess (3x3 box, binary Sobel, candidate
 * mask) -> fern hash codes -> unique-code collision matching -> disparity filter.
 *
 * This is the drop-in boundary.  The reference has no FFI layer (header-only C++),
 * so each entry point names the reference function it stands in for; the C++ API
 * in include/gpc/inference.hpp forwards to these exactly where the reference's
 * Forest methods call the raw-pointer kernels of lib/gpc/filter.hpp.
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns a gpc_status (0 = ok);
 *   - images are 8-bit, row-major, `width` a multiple of 16 (reference asserts this,
 *     filter.hpp:294,405,549), tightly packed (stride == width); width <= 16384 and
 *     width * height <= 2^30, larger images are refused with GPC_E_UNSUPPORTED (the reference
 *     has no such limit);
 *   - outputs are caller-allocated with an explicit capacity; the true count is
 *     always returned, GPC_E_CAPACITY if it did not fit (the first `cap` entries are
 *     valid);
 *   - a context belongs to one device and one host thread at a time (the reference's
 *     Forest is stateless and re-entrant; use one context per thread);
 *   - there is NO CPU fallback: without a usable gfx950 device gpc_hip_create fails.
 */
#ifndef GPC_HIP_H
#define GPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPC_HIP_ABI_VERSION 1
#define GPC_MAX_TESTS 32   /* readForest keeps the first 32 tests, inference.hpp:426 */
#define GPC_PATCH_RADIUS 13 /* 27x27 patch; also the candidate margin, inference.hpp:322 */

typedef enum {
  GPC_OK = 0,
  GPC_E_INVALID = 1,      /* bad argument (null pointer, width % 16, size mismatch)      */
  GPC_E_NO_DEVICE = 2,    /* no HIP device / not gfx950 / HIP runtime error at create    */
  GPC_E_HIP = 3,          /* a HIP call failed; see gpc_hip_last_error                   */
  GPC_E_CAPACITY = 4,     /* output did not fit; count returned is the true count        */
  GPC_E_NO_FOREST = 5,    /* match/hash called before gpc_hip_set_forest                 */
  GPC_E_FOREST_RANGE = 6, /* a test offset leaves the 27x27 patch                        */
  GPC_E_IO = 7,           /* forest file could not be opened / parsed                    */
  GPC_E_UNSUPPORTED = 8   /* a size / setting this build cannot honour (width > 16384, ...) */
} gpc_status;

/* == ndb::Support, lib/gpc/buffer.hpp:91-97 (12 bytes) */
typedef struct {
  int32_t x, y;
  float d;
} gpc_support;

/* == ndb::Correspondence, lib/gpc/buffer.hpp:99-102 (two ndb::Point) */
typedef struct {
  int32_t src_x, src_y, tar_x, tar_y;
} gpc_correspondence;

/* == gpc::inference::InferenceSettings, lib/gpc/inference.hpp:71-131 */
typedef struct {
  int32_t gradient_threshold; /* gradientThreshold_ (uint8_t), default 10 */
  int32_t disp_high;          /* dispHigh_, default 128                    */
  int32_t vertical_tolerance; /* verticalTolerance_, default 1             */
  int32_t epipolar_mode;      /* epipolarMode_, default 0                  */
  int32_t use_hashtable;      /* useHashtable_: 1 = ndb::Hashmatch matcher */
  int32_t num_threads;        /* numThreads_: accepted, ignored            */
} gpc_settings;

/* == gpc::inference::Forest::FilterMask, lib/gpc/inference.hpp:137-156 */
typedef struct {
  int32_t mask[2 * GPC_MAX_TESTS]; /* mask[2t]=ix+iy*width, mask[2t+1]=jx+jy*width */
  int32_t tau[GPC_MAX_TESTS];
  int32_t num_tests;
  int32_t type;      /* 0: all tau == 0 (gpcFilter), 1: gpcFilterTau */
  int32_t width, height;
  int32_t discarded; /* tests dropped beyond the 32nd                */
} gpc_filter_mask;

typedef struct gpc_hip_ctx gpc_hip_ctx;

/* ---- library / context ------------------------------------------------------ */
int gpc_hip_abi_version(void);
const char* gpc_hip_status_string(int status);
int gpc_hip_device_count(int* count);
/* Creates a context on `device` with its own HIP stream. */
int gpc_hip_create(int device, gpc_hip_ctx** ctx);
int gpc_hip_destroy(gpc_hip_ctx* ctx);
/* Last HIP error text of this context (static storage inside the context). */
const char* gpc_hip_last_error(const gpc_hip_ctx* ctx);
/* Borrow an externally owned hipStream_t (e.g. torch's current stream); NULL restores
 * the context's own stream. */
int gpc_hip_set_stream(gpc_hip_ctx* ctx, void* hip_stream);
int gpc_hip_synchronize(gpc_hip_ctx* ctx);
/* Pre-size device workspaces for `max_pairs` pairs of width x height (optional; the
 * entry points below grow them on demand, which costs a hipMalloc + sync). */
int gpc_hip_reserve(gpc_hip_ctx* ctx, int width, int height, int max_pairs);

/* Pays, outside the caller's timed region, for everything a first call would otherwise pay inside it: the library's code
 * objects, the workspaces of one pair of width x height, page-locked staging, streams and worker threads.  The reference's
 * sample starts its clock after Forest::readForest(path, width, height) (samples/sparsematch.cpp:42-45) -- the one call
 * that knows the image size -- so the C++ API calls this from there; a C caller does it after gpc_hip_set_forest (the
 * forest must be set: GPC_E_NO_FOREST otherwise).  Runs the host entry points once on a synthetic pair and discards the
 * results.  settings == NULL: all four matcher modes (epipolar x hashtable). */
int gpc_hip_warmup(gpc_hip_ctx* ctx, int width, int height, const gpc_settings* settings);

/* The reference has two arithmetic variants chosen at BUILD time (samples/CMakeLists.txt:13-20):
 * SSE=ON (-D_INTRINSICS_SSE, the default and the parity target of this library) and SSE=OFF
 * (boxNaive / sobelNaive / gpcFilter(Tau)Naive, filter.hpp:157-282), which produce different
 * smooth images, masks and codes.  A context starts in GPC_ARITH_SSE. */
#define GPC_ARITH_SSE 0
#define GPC_ARITH_NAIVE 1
int gpc_hip_set_arithmetic(gpc_hip_ctx* ctx, int mode);

/* Page-locked host memory (hipHostMalloc) for the host-buffer entry points: with pageable
 * buffers the PCIe copies run at a fraction of the link rate.  Free with gpc_hip_host_free. */
int gpc_hip_host_alloc(gpc_hip_ctx* ctx, uint64_t bytes, void** ptr);
int gpc_hip_host_free(gpc_hip_ctx* ctx, void* ptr);

/* ---- forest ------------------------------------------------------------------ */
/* Forest::readForest (inference.hpp:404-446): parses the text forest for an image of
 * `width` x `height`.  Host only.  A missing file yields GPC_E_IO and an empty mask of
 * type 0, as the reference does. */
int gpc_hip_read_forest(const char* path, int width, int height, gpc_filter_mask* out);
int gpc_hip_parse_forest(const char* text, int width, int height, gpc_filter_mask* out);
/* Uploads the tests (what gpcFilter/gpcFilterTau receive as `fastmask`/`tau`,
 * filter.hpp:547,619).  Offsets are decoded back to (dx,dy) with |d| <= 13. */
int gpc_hip_set_forest(gpc_hip_ctx* ctx, const gpc_filter_mask* fm);

/* ---- host-buffer entry points (drop-in for the Forest methods) ---------------- */
/* Forest::preprocessImage (inference.hpp:302-333): box + clearBoundary, sobel on the
 * raw image, ascending candidate indices with the 13-pixel margin.
 * smooth, grad: width*height bytes; mask: capacity mask_cap ints. */
int gpc_hip_preprocess(gpc_hip_ctx* ctx, const uint8_t* raw, int width, int height,
                       int gradient_threshold, uint8_t* smooth, uint8_t* grad,
                       int32_t* mask, int mask_cap, int* n_mask);
/* The same in two steps, so that the caller can allocate while the device works (the by-value PreprocessedImage of
 * Forest::preprocessImage, inference.hpp:161-165, 302-333): _begin QUEUES the kernels -- the device writes smooth, grad and
 * the candidate list into page-locked staging memory of the context over the link -- and returns; _fetch waits, copies the
 * three results into the caller's arrays with the library's worker threads (any may be NULL; the candidate count is
 * returned; GPC_E_CAPACITY if mask_cap is short, the first mask_cap indices are delivered).  An image has at most
 * (width - 26) * (height - 26) candidates.  Exactly one _fetch per _begin, no other call on the context in between.
 *
 * Resident images.  The image also STAYS on the device (the last two per context), and the arrays handed to _fetch (or to
 * gpc_hip_preprocess) are remembered as its host copies.  gpc_hip_rectified_match / gpc_hip_stereo_match recognise them --
 * same addresses, same sizes, same arithmetic mode, and a fingerprint of their contents (66 words spread over each array)
 * unchanged -- and then hash and match from the device copies instead of uploading smooth, grad and mask again: the
 * reference's by-value PreprocessedImage without its round trip over the link.  Anything else (copies of the arrays,
 * edited arrays, arrays from another context) takes the upload path; the results are the same either way.  A caller that
 * EDITS a delivered array in place in a way 66 samples can miss must set GPC_HIP_RESIDENT=2 (every byte is hashed) or
 * GPC_HIP_RESIDENT=0 (never resident).  Every library call that writes over a remembered array forgets it. */
int gpc_hip_preprocess_begin(gpc_hip_ctx* ctx, const uint8_t* raw, int width, int height, int gradient_threshold);
int gpc_hip_preprocess_fetch(gpc_hip_ctx* ctx, uint8_t* smooth, uint8_t* grad, int32_t* mask, int mask_cap, int* n_mask);
/* Match calls of this context served from resident images so far (tests, diagnostics). */
int gpc_hip_resident_hits(const gpc_hip_ctx* ctx);

/* ndb::gpcFilter / gpcFilterTau as called by evalFastMaskOnSubsetSSE
 * (inference.hpp:266-292): dense code image, width*height uint32, zero where the
 * reference leaves its zero-filled buffer untouched. */
int gpc_hip_hash_codes(gpc_hip_ctx* ctx, const uint8_t* smooth, const uint8_t* grad,
                       int width, int height, uint32_t* codes);
/* Forest::rectifiedMatch (inference.hpp:375-393) on already preprocessed images.
 * maskL/maskR are the candidate index lists of the PreprocessedImage. */
int gpc_hip_rectified_match(gpc_hip_ctx* ctx, const uint8_t* smoothL, const uint8_t* gradL,
                            const int32_t* maskL, int n_maskL, const uint8_t* smoothR,
                            const uint8_t* gradR, const int32_t* maskR, int n_maskR,
                            int width, int height, const gpc_settings* settings,
                            gpc_support* out, int cap, int* n_out);
/* Forest::stereoMatch (inference.hpp:344-361): correspondences, no disparity filter. */
int gpc_hip_stereo_match(gpc_hip_ctx* ctx, const uint8_t* smoothL, const uint8_t* gradL,
                         const int32_t* maskL, int n_maskL, const uint8_t* smoothR,
                         const uint8_t* gradR, const int32_t* maskR, int n_maskR,
                         int width, int height, const gpc_settings* settings,
                         gpc_correspondence* out, int cap, int* n_out);
/* The whole timed region of samples/sparsematch.cpp:45-52 for one pair:
 * preprocessImage x2 + rectifiedMatch, raw host images in, supports out. */
int gpc_hip_match_pair(gpc_hip_ctx* ctx, const uint8_t* rawL, const uint8_t* rawR,
                       int width, int height, const gpc_settings* settings,
                       gpc_support* out, int cap, int* n_out, int* n_cand_l, int* n_cand_r);

/* The three calls above in two steps each: *_begin queues the work and returns (the results go to page-locked memory of
 * the context), gpc_hip_match_fetch waits and copies min(count, cap) records into `out` (gpc_support for the rectified and
 * pair forms, gpc_correspondence for the stereo form) with the library's worker threads; *n_out is the true count
 * (GPC_E_CAPACITY when it exceeds cap: fetch again with a larger array -- the results stay until the next call on the
 * context).  n_cand_l / n_cand_r are filled after gpc_hip_match_pair_begin only.  Between the two steps the caller can
 * allocate (and let the allocator zero) the array the results go to: that is what a std::vector<ndb::Support> of the
 * reference's API costs, and here it runs beside the kernels instead of after them. */
int gpc_hip_rectified_match_begin(gpc_hip_ctx* ctx, const uint8_t* smoothL, const uint8_t* gradL,
                                  const int32_t* maskL, int n_maskL, const uint8_t* smoothR,
                                  const uint8_t* gradR, const int32_t* maskR, int n_maskR,
                                  int width, int height, const gpc_settings* settings);
int gpc_hip_stereo_match_begin(gpc_hip_ctx* ctx, const uint8_t* smoothL, const uint8_t* gradL,
                               const int32_t* maskL, int n_maskL, const uint8_t* smoothR,
                               const uint8_t* gradR, const int32_t* maskR, int n_maskR,
                               int width, int height, const gpc_settings* settings);
int gpc_hip_match_pair_begin(gpc_hip_ctx* ctx, const uint8_t* rawL, const uint8_t* rawR,
                             int width, int height, const gpc_settings* settings);
int gpc_hip_match_fetch(gpc_hip_ctx* ctx, void* out, int cap, int* n_out, int* n_cand_l, int* n_cand_r);

/* ---- device-resident batch entry points ------------------------------------- */
/* `npairs` raw pairs already in HBM ([npairs][height][width] each side) -> supports in
 * HBM: d_out[npairs][cap_per_pair], d_counts[npairs] (true counts), and, if non-NULL,
 * d_ncand[npairs][2] candidate counts.
 * With the reference's sparsematch settings (epipolar_mode = 1, use_hashtable = 0) the call is
 * ASYNCHRONOUS on the context's stream: three launches are queued and it returns.
 * With epipolar_mode = 0 or use_hashtable = 1 (the device-wide matchers) the HOST WAITS once inside the
 * call (the hash-table matcher once per planning attempt): those matchers partition the records by code /
 * bucket range and read a few words back to learn whether every partition fits a workgroup (if not -- heavily
 * repeated codes -- the whole batch takes the radix-sort path instead).  The wait is for an event in the middle
 * of what the call queues, not for the stream; part of their work runs on a second stream of the context that
 * is joined back into the context's stream before the call returns.  In either case the outputs may be read
 * only after gpc_hip_synchronize (or another wait on the stream).
 * The join that writes the supports (three launches: the last one) places a row behind the rows before it with a
 * bounded wait on other workgroups; a wait that ran out (not observed so far) makes the kernel store the number of
 * that launch into a host-visible word, and the outputs of that launch must not be used.  The word is examined by
 * gpc_hip_synchronize and by every entry point that synchronises itself (they return GPC_E_HIP, gpc_hip_last_error
 * names the launch), and -- for callers that gave the context their own stream and wait on it themselves -- at the
 * top of the NEXT call that queues such a join and in gpc_hip_destroy (which then returns GPC_E_HIP after freeing
 * everything): such callers should call gpc_hip_synchronize once before trusting a batch they did not wait for
 * through this library. */
int gpc_hip_match_batch_device(gpc_hip_ctx* ctx, const uint8_t* d_rawL, const uint8_t* d_rawR,
                               int width, int height, int npairs, const gpc_settings* settings,
                               gpc_support* d_out, int cap_per_pair, int32_t* d_counts,
                               int32_t* d_ncand);
/* A STREAM of batches (the same context called again and again): with two lanes, consecutive gpc_hip_match_batch_device
 * calls (epipolar sort-matcher) alternate between two sets of workspaces on two streams of the context's own, and batch
 * k+1's preprocess and hash kernels run beside batch k's join (measured: 0.905 instead of 0.945 ms per 256 pairs).  In
 * this mode
 *   - a call's inputs are what the context's stream has produced when the call is made;
 *   - a call's outputs are complete after gpc_hip_synchronize, or -- for a caller that waits on a stream of its own
 *     (gpc_hip_set_stream) -- after that stream has passed gpc_hip_pipeline_join, which makes it wait for every call
 *     queued so far;
 *   - two calls are in flight at a time: consecutive calls need distinct output arrays.
 * lanes = 1 (the default) is the strict form above: everything on the context's stream, call by call. */
int gpc_hip_set_pipeline(gpc_hip_ctx* ctx, int lanes);
int gpc_hip_pipeline_join(gpc_hip_ctx* ctx);
/* Same from/to host memory (pinned or pageable), synchronous.  With the reference's sparsematch settings
 * (epipolar mode, sort matcher) the results cross PCIe packed (4 bytes per support, see below) and are expanded
 * into `out` by worker threads of the library while later chunks are still on the link: settings->num_threads > 1
 * asks for that many workers, otherwise the CPUs the process may use minus two.  The records delivered are
 * bit-identical to the device path's. */
int gpc_hip_match_batch(gpc_hip_ctx* ctx, const uint8_t* rawL, const uint8_t* rawR,
                        int width, int height, int npairs, const gpc_settings* settings,
                        gpc_support* out, int cap_per_pair, int32_t* counts, int32_t* ncand);

/* Host threads gpc_hip_match_batch last used to expand packed results (0 before the first such call).  Default:
 * settings->num_threads if > 1, else the CPUs this process may use -- divided by LOCAL_WORLD_SIZE when a launcher
 * starts one process per GPU -- less the feeding thread, between 2 and 8. */
int gpc_hip_host_threads(const gpc_hip_ctx* ctx);
/* Where the last gpc_hip_match_batch / gpc_hip_match_batch_packed call of the context spent its time: the host's clock,
 * in ms since the call's entry, when it saw [0] the last chunk's upload complete, [1] the last chunk's kernels done (its
 * counts arrived), [2] the last chunk of packed records landed in host memory, [3] the expansion / delivery done (the
 * call returned).  All 0 for calls that took another path (one or two pairs written directly, the device-wide matchers). */
int gpc_hip_batch_stages(const gpc_hip_ctx* ctx, float* ms4);
/* The CPU each worker thread last ran a job on (returns the number of workers; fills at most cap entries). */
int gpc_hip_host_worker_cpus(gpc_hip_ctx* ctx, int* cpus, int cap);
/* Batch calls of this context that ran their chunk pipeline on a thread bound to the GPU's NUMA node because the caller's
 * thread was on another socket (GPC_HIP_NO_FEEDER=1: never). */
int gpc_hip_fed_calls(const gpc_hip_ctx* ctx);
/* The NUMA node of the host this context's GPU hangs off (the expansion workers are bound to its CPUs), -1 if unknown. */
int gpc_hip_host_numa_node(const gpc_hip_ctx* ctx);

/* ---- packed results ------------------------------------------------------------ */
/* Forest::rectifiedMatch (inference.hpp:375-393) in epipolar mode emits supports row by row, so a support
 * {x, y, float(x - xR)} (ndb::Support, buffer.hpp:91-97) is fully described by one 32-bit word x | xR << 16 plus
 * the number of supports per row: 4 bytes instead of 12.  d_packed[npairs][cap_per_pair] receives the words in
 * the reference's output order, d_rows[npairs][height] the per-row counts (rows outside 13 .. height-14 are not
 * written), d_counts[npairs] the true totals.  Epipolar sort-matcher only (GPC_E_UNSUPPORTED otherwise).
 * Asynchronous on the context's stream. */
int gpc_hip_match_batch_device_packed(gpc_hip_ctx* ctx, const uint8_t* d_rawL, const uint8_t* d_rawR,
                                      int width, int height, int npairs, const gpc_settings* settings,
                                      uint32_t* d_packed, int cap_per_pair, int32_t* d_rows,
                                      int32_t* d_counts, int32_t* d_ncand);
/* The same from / to HOST memory, synchronous (the chunk pipeline of gpc_hip_match_batch without its last stage): the
 * records stay as they crossed the link -- packed[npairs][cap_per_pair] words, rows[npairs][height] per-row counts (rows
 * outside 13 .. height-14 hold 0), counts[npairs] the true totals (GPC_E_CAPACITY when one exceeds cap_per_pair; the pair's
 * first cap_per_pair records are delivered).  For callers that consume supports row by row, or that expand them later /
 * elsewhere with gpc_hip_expand_packed: a batch of 256 pairs of 1024x436 leaves 209 MB in host memory where the
 * ndb::Support arrays of gpc_hip_match_batch are 625 MB -- with one process per GPU on an 8-GPU node those 12-byte records
 * are what the host's memory bandwidth runs out on.  Epipolar sort-matcher only (GPC_E_UNSUPPORTED otherwise). */
int gpc_hip_match_batch_packed(gpc_hip_ctx* ctx, const uint8_t* rawL, const uint8_t* rawR, int width, int height,
                               int npairs, const gpc_settings* settings, uint32_t* packed, int cap_per_pair,
                               int32_t* rows, int32_t* counts, int32_t* ncand);
/* Host only: the first n supports of one pair from its packed words and row counts. */
int gpc_hip_expand_packed(const uint32_t* packed, const int32_t* rows, int height, int n, gpc_support* out);

/* ---- fern training: the scoring loop -------------------------- */
/* Replaces Fern::evalSplit (Fern.hpp:209-262), Fern::markSplitSamples (Fern.hpp:271-291) and the
 * level / resample / tau loops of Fern::train (Fern.hpp:312-372) over a device-resident training
 * set.  A triplet is three 27x27 byte patches (ref, pos, neg), 729 bytes each in the byte order of
 * Feature::storeAllTriplets (Feature.hpp:247-256); a test is (i, j, tau) with i, j linear indices
 * into a patch (Feature::params, Feature.hpp:84-89; Feature::getDecisions, Feature.hpp:101-109).
 * Marks: one byte per triplet, bit 0 = pos.split, bit 1 = neg.split (GPCDescriptor::split).
 * Hyperplane SAMPLING stays with the caller (Feature::sampleHyperplane draws from std::mt19937);
 * these entry points score what was drawn. */
typedef struct gpc_hip_train_set gpc_hip_train_set;
typedef struct gpc_split {
  int32_t i, j, tau;
} gpc_split;
typedef struct gpc_split_stats { /* splitStats, Fern.hpp:52-68 */
  double prec, rec, hmean, convcomb;
  int32_t tp, fp, fn, tot;
} gpc_split_stats;

/* Uploads `n` triplets (n * 3 * 729 bytes of host memory) and lays them out for the device.
 * All marks start cleared.  The set belongs to `ctx` and is destroyed with it at the latest. */
int gpc_hip_train_set_create(gpc_hip_ctx* ctx, const uint8_t* triplets, int n, gpc_hip_train_set** out);
int gpc_hip_train_set_destroy(gpc_hip_ctx* ctx, gpc_hip_train_set* set);
int gpc_hip_train_set_size(const gpc_hip_train_set* set);
/* marks_in != NULL: replace the marks; marks_out != NULL: read them back (after the replacement) */
int gpc_hip_train_set_marks(gpc_hip_ctx* ctx, gpc_hip_train_set* set, const uint8_t* marks_in, uint8_t* marks_out);
/* Fern::evalSplit over params[0 .. score_until_level] (at most 64 levels: the reference's code
 * words have 64 bits).  w1 = OptimizerSettings::w1_. */
int gpc_hip_train_eval_split(gpc_hip_ctx* ctx, gpc_hip_train_set* set, const gpc_split* params,
                             int score_until_level, double w1, gpc_split_stats* stats);
/* Fern::markSplitSamples over params[0 .. num_params) */
int gpc_hip_train_mark_split_samples(gpc_hip_ctx* ctx, gpc_hip_train_set* set, const gpc_split* params,
                                     int num_params);
/* Fern::train for one fern with the hyperplane samples supplied by the caller:
 * cand[level * num_resamples + k] is the k-th draw of sampleHyperplane at `level` (its tau is
 * ignored: the loop over [taulo, tauhi) overwrites it, Fern.hpp:341-342).  Keeps the reference's
 * selection rule (first candidate whose hmean exceeds the float maximum so far; a level on which
 * nothing scores above 0 inherits the previous level's parameters) and returns per level the
 * parameters chosen and the statistics train() prints (those of the LAST candidate evaluated).
 * max_depth <= 64, tauhi - taulo <= 64. */
int gpc_hip_train_fern(gpc_hip_ctx* ctx, gpc_hip_train_set* set, int max_depth, const gpc_split* cand,
                       int num_resamples, int taulo, int tauhi, int only_score_non_split, double w1,
                       gpc_split* fernparams, gpc_split_stats* level_stats);
/* One level of the above in pieces, for callers that sample adaptively: begin a fern, score
 * `ncand` candidates of the current level for every tau in [taulo, tauhi) (tp/fp: [ncand][ntau],
 * fn = *tot - tp - fp), then fix the level's winner. */
int gpc_hip_train_begin_fern(gpc_hip_ctx* ctx, gpc_hip_train_set* set, int reset_marks);
int gpc_hip_train_eval_level(gpc_hip_ctx* ctx, gpc_hip_train_set* set, const gpc_split* cand, int ncand,
                             int taulo, int tauhi, int32_t* tp, int32_t* fp, int32_t* tot);
int gpc_hip_train_commit_level(gpc_hip_ctx* ctx, gpc_hip_train_set* set, const gpc_split* best, int mark_split);

/* ---- measurement -------------------------------------------------------------- */
/* Per-kernel HIP-event timing on the context's stream.  When enabled every launch of
 * the named kernels is bracketed by hipEvents; gpc_hip_kernel_time returns the summed
 * milliseconds and launch count since the last reset (synchronises the stream). */
int gpc_hip_enable_kernel_timing(gpc_hip_ctx* ctx, int enable);
/* Restrict the bracketing to the kernels whose index bit is set (default: all).  Every pair of
 * event records costs a little stream time, so a benchmark times only the kernel it reports. */
int gpc_hip_set_kernel_timing_mask(gpc_hip_ctx* ctx, unsigned mask);
int gpc_hip_reset_kernel_timing(gpc_hip_ctx* ctx);
int gpc_hip_kernel_count(void);
const char* gpc_hip_kernel_name(int index);
/* The profiler's (rocprofv3) name of the template instantiation this context last launched under timing slot
 * `index`, e.g. "gpc::k_row_join<4, 256, false>"; "" before the first launch. */
const char* gpc_hip_kernel_launch_name(const gpc_hip_ctx* ctx, int index);
int gpc_hip_kernel_time(gpc_hip_ctx* ctx, int index, float* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* GPC_HIP_H */
