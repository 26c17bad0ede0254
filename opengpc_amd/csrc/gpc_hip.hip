// gpc_hip.hip -- host side of libgpc_hip.so: context, workspaces, launches, C ABI.
//
// gfx950 only.  No CPU fallback: every entry point needs a live HIP device.
#include "../../include/gpc_hip.h"

#include <hip/hip_runtime.h>

#if defined(__SSE2__)
#include <emmintrin.h>   // the host expansion of packed results has an SSE2 path (x86 hosts); scalar otherwise
#endif
#include <sched.h>

#include <cctype>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <deque>
#include <map>
#include <utility>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gpc_device.h"
#include "k_global.h"
#include "k_hash.h"
#include "k_hashtable.h"
#include "k_htjoin.h"
#include "k_partition.h"
#include "k_preprocess.h"
#include "k_rowjoin.h"
#include "k_rowjoin_fused.h"
#include "k_rows.h"
#include "k_train.h"

namespace {

enum KernelId {
  KID_PREPROCESS = 0,
  KID_HASH,
  KID_ROW_JOIN,
  KID_GATHER_ROWS,
  KID_MASK,
  KID_GLOBAL_KEYS,
  KID_GLOBAL_SORT,
  KID_GLOBAL_MATCH,
  KID_TRAIN_EVAL,
  KID_COUNT
};
const char* const kKernelNames[KID_COUNT] = {
    "k_preprocess", "k_hash", "k_row_join", "k_gather_rows",
    "k_mask", "k_global_keys", "k_global_sort", "k_global_match", "k_train_eval"};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct TimedSpan {
  hipEvent_t a, b;
  int kid;
};

// where run_match puts PACKED results (k_gather_rows mode 2): words between the pairs' arrays
struct PackedOut {
  int32_t* rows = nullptr;
  long packed_stride = 0, rows_stride = 0;  // words between the pairs' arrays
  int32_t* totals = nullptr;                // non-null: scratch [npairs]; the pairs' records are then written back to back
};

// Host threads that expand packed results (gpc_hip_expand_packed) into the caller's gpc_support arrays while the
// next chunk is on the link.  Jobs carry the index of the staging slot they read; wait_slot() blocks until every
// job of a slot is done, so the slot can be overwritten.
struct ExpandJob {
  const uint32_t* packed;
  const int32_t* rows;
  int H, y0, y1, first, limit;  // rows [y0, y1) of a pair whose supports start at index `first`; stop at `limit`
  gpc_support* out;
  int slot;
  // a plain copy instead (pageable input images into the page-locked bounce buffer of their chunk): copy_bytes > 0
  const void* copy_src = nullptr;
  void* copy_dst = nullptr;
  size_t copy_bytes = 0;
};

class ExpandPool {
 public:
  ~ExpandPool() { stop(); }
  // bind: CPUs the workers may run on (the NUMA node of the GPU: the page-locked buffers they read and write live there,
  // and workers that land on the other socket made the same call take 7.3 instead of 5.4 ms); null = wherever
  // bind: the CPUs of the GPU's NUMA node.  groups: the same CPUs by last-level cache (one set per CCD): worker i is held on
  // group i mod n.  Left to the scheduler, workers woken by one thread are placed on idle CPUs of the WAKER's cache domain --
  // all eight on one CCD, whose link to the memory controllers then carries every record they write: the same 256-pair call
  // took 8.5 ms instead of 5.3 (profiles/r05_g_bench.json against r05_e: worker CPUs 64-71 + their SMT siblings).
  void start(int nthreads, const cpu_set_t* bind, const std::vector<cpu_set_t>* groups = nullptr) {
    if ((int)threads_.size() == nthreads) return;
    stop();
    quit_ = false;
    have_bind_ = bind != nullptr;
    if (bind) bind_ = *bind;
    groups_.clear();
    if (bind && groups) groups_ = *groups;
    cpus_.assign((size_t)nthreads, -1);
    for (int i = 0; i < nthreads; ++i)
      threads_.emplace_back([this, i] {
        if (!groups_.empty()) (void)sched_setaffinity(0, sizeof(cpu_set_t), &groups_[(size_t)i % groups_.size()]);
        else if (have_bind_) (void)sched_setaffinity(0, sizeof bind_, &bind_);
        run(i);
      });
  }
  void stop() {
    {
      std::lock_guard<std::mutex> g(m_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
    threads_.clear();
  }
  void push(const ExpandJob& j) {
    {
      std::lock_guard<std::mutex> g(m_);
      q_.push_back(j);
      ++pending_[j.slot & 7];
    }
    cv_.notify_one();
  }
  void wait_slot(int slot) {
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [&] { return pending_[slot & 7] == 0; });
  }
  void wait_all() {
    for (int s = 0; s < 8; ++s) wait_slot(s);
  }
  int size() const { return (int)threads_.size(); }
  // the CPU every worker last ran a job on (diagnostics: gpc_hip_host_worker_cpus)
  int worker_cpus(int* out, int cap) {
    std::lock_guard<std::mutex> g(m_);
    int n = 0;
    for (size_t i = 0; i < cpus_.size() && n < cap; ++i) out[n++] = cpus_[i];
    return (int)cpus_.size();
  }

 private:
  void run(int index);
  std::vector<std::thread> threads_;
  std::vector<int> cpus_;
  std::deque<ExpandJob> q_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  int pending_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool quit_ = false;
  bool have_bind_ = false;
  cpu_set_t bind_;
  std::vector<cpu_set_t> groups_;
};

}  // namespace

struct gpc_hip_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // gpc_hip_match_batch: upload / download streams and the events that chain a chunk's stages
  hipStream_t s_in = nullptr, s_out = nullptr, s_cnt = nullptr;
  hipStream_t s_aux = nullptr;            // the non-epipolar matcher's launch for over-large partitions runs beside the main one
  hipEvent_t e_fork = nullptr, e_join = nullptr;
  hipEvent_t e_flag = nullptr;            // the device-wide matchers' plan words have arrived on the host
  hipEvent_t e_in[4] = {}, e_comp[4] = {}, e_cnt[4] = {}, e_out[4] = {};
  DevBuf packed;                  // packed results of the chunks in flight (3 slots)
  void* h_stage = nullptr;        // page-locked landing area of packed results (4 slots)
  size_t h_stage_cap = 0;
  void* h_in = nullptr;           // page-locked bounce buffer for PAGEABLE input images (2 slots x 2 sides x a chunk)
  size_t h_in_cap = 0;
  int32_t* h_cnt = nullptr;       // page-locked landing area of counts [npairs] + candidate counts [npairs][2]: a copy to the
  size_t h_cnt_cap = 0;           // caller's (pageable) arrays would block the host until the chunk's kernels are done
  ExpandPool pool;
  // host clock at the stages of the last gpc_hip_match_batch / _packed call, ms since its entry (gpc_hip_batch_stages):
  // [0] last upload seen complete, [1] last chunk's kernels done (its counts have arrived), [2] last chunk of packed
  // records has landed in host memory, [3] expansion / delivery done (the call returns)
  float stage_ms[4] = {0.f, 0.f, 0.f, 0.f};
  bool grad_is_bits = false;      // c->grad holds k_preprocess's bit image (set by run_preprocess, read by run_hash)
  bool no_grad_bits = false;      // GPC_HIP_NO_GRAD_BITS: the batched pipelines keep the byte image (A/B checks)
  int upload_mode = 1;            // GPC_HIP_UPLOAD: single-pair host path -- 0: hipMemcpyAsync per side, 1: one k_upload2 launch, 2: k_preprocess reads the host's pages
  int direct_max = 2;             // GPC_HIP_DIRECT_MAX: batches up to this size with a page-locked `out` are written by the
                                  // kernels straight into the caller's array (no packed records, no host expansion); 0 = never
  bool have_node_cpus = false;    // CPUs of the NUMA node this GPU hangs off (from sysfs), within the process's affinity mask
  cpu_set_t node_cpus;
  std::vector<cpu_set_t> node_l3;  // node_cpus by last-level cache (CCD): the expansion workers are dealt over these
  int numa_node = -1;
  // Two lanes (gpc_hip_set_pipeline): consecutive device-resident batch calls alternate between two sets of workspaces on
  // two streams of the context's own, wired with events so that batch k+1's k_preprocess and k_hash run beside batch k's
  // join (which then takes one workgroup per CU less: a wave slot per SIMD stays free for them).  A lane is SWAPPED INTO
  // the members below for the duration of a call (LaneScope), so every run_* function works on it unchanged.
  struct Lane {
    hipStream_t s = nullptr;
    hipEvent_t e_in = nullptr, e_hash = nullptr, e_join = nullptr;
    DevBuf smooth, grad, codes, stats, jstate, staged, rowcnt;
    size_t jstate_granules = 0;
    uint32_t join_epoch = 0;
    bool grad_is_bits = false, used = false;
  };
  Lane lanes[2];
  int pipeline = 1, next_lane = 0;
  // experiment hooks (gpc_hip_debug_pipeline_events, tools/overlap_experiment.py; all null in the product): events the
  // device-resident batch call waits for before k_preprocess / between k_preprocess and k_hash, and records after k_hash /
  // after the join -- enough to let one batch's k_preprocess run beside the previous batch's join and nothing else
  hipEvent_t dbg_wait_pre = nullptr, dbg_wait_hash = nullptr, dbg_rec_hash = nullptr, dbg_rec_join = nullptr;
  bool no_pair_packed = false;    // GPC_HIP_NO_PAIR_PACKED: the two-step forms deliver 12-byte records over the link (A/B checks)
  bool no_feeder = false;         // GPC_HIP_NO_FEEDER: the chunk pipeline always runs on the calling thread (A/B checks)
  int fed_calls = 0;              // batch calls that ran on a feeder thread (gpc_hip_fed_calls)
  int chunk_pairs = 0;            // GPC_HIP_CHUNK: pairs per chunk of gpc_hip_match_batch (tuning)
  int expand_threads = 0;         // GPC_HIP_EXPAND_THREADS (tuning)
  char err[256] = {0};

  bool naive = false;  // gpc_hip_set_arithmetic: the reference's SSE=OFF (*Naive) arithmetic
  bool have_forest = false;
  GpcForestDev forest;        // tests in file order (SSE bit placement)
  GpcForestDev forest_naive;  // tests reversed: slot u = test T-1-u lands on bit u (MSB-first codes), raw int tau
  int forest_w = 0, forest_h = 0;
  gpc_filter_mask forest_src;  // what gpc_hip_set_forest was last given: the same forest again costs nothing

  // workspaces
  DevBuf raw, smooth, grad, candmap, codes, staged, rowcnt, stats, out, counts, ncand, mask;
  DevBuf gkv;  // records (code, pixel index) of the partitioned device-wide matchers, by bin
  DevBuf gkeys[2], gvals[2], ghist, gmisc, hkeys[2], hvals[2], hrec;
  DevBuf gpart;       // partition plan of the non-epipolar matcher (k_partition.h) + one overflow word for the batch
  int32_t* h_flag = nullptr;  // page-locked landing word of that overflow flag
  int no_partition = 0;       // GPC_HIP_NO_PARTITION: always take the radix-sort path (A/B checks)
  int rows_per_chunk = 16;    // GPC_HIP_ROWS_PER_CHUNK: rows one partition workgroup scatters (A/B checks)
  int gp_log2bins = 0;        // GPC_HIP_GP_LOG2BINS = 8 .. 11: force the number of code-range bins (A/B checks)
  int ht_hint_w = 0, ht_hint_h = 0, ht_hint_lbits = 0;  // what the hash-table planner ended on for the last image size: where it starts next time
  bool ht_no_half = false;    // GPC_HIP_HT_NO_HALF: never the 512-thread k_ht_join (A/B checks)
  int ht_mid = 0;             // GPC_HIP_HT_MID = 10 .. 64: force HtjArgs::mid (A/B checks; 10 = one wave per bucket beyond ten records)
  int ht_lbits = 0;           // GPC_HIP_HT_LBITS = 7 .. 10: force the buckets per bin of the hash-table matcher (A/B checks)
  int gp_target = 2000;       // GPC_HIP_GP_TARGET: records per side a partition of the non-epipolar matcher aims at. Per 32 pairs of
                              // 1024x436, join + gather: 1400 -> 208 us, 1800 -> 189, 2000 / 2200 -> 182, 2600 -> 192, 3000 -> 208
  int flat_chunks = 0;        // GPC_HIP_FLAT_CHUNKS: gpc_hip_match_batch with equal chunks only (A/B checks)
  DevBuf forest_dev;  // [0] = forest, [1] = forest_naive, [2] = forest with the tall tile's offsets: the hash kernel reads its tests from here (scalar loads)
  int hash_tall = -1;  // GPC_HIP_HASH_TALL = 0 | 1: never / always the 40-row tile where it exists (tests, A/B checks); default: by rounds

  // fused join + output (k_rowjoin.h, FUSE): ticket counters + look-back granules, launch epoch, error word
  DevBuf jstate;
  size_t jstate_granules = 0;
  uint32_t join_epoch = 0;
  int32_t* h_err = nullptr;   // page-locked, device-visible: a look-back of the fused join timed out
  int32_t* d_err = nullptr;   // the device's address of that word
  int no_fuse = 0;            // GPC_HIP_NO_FUSE: join + k_gather_rows as two launches (A/B checks)
  int fuse_always = 0;        // GPC_HIP_FUSE_ALWAYS: the fused join wherever it is possible, however small the launch (tests, A/B checks)
  int fuse_wgs = 0;           // GPC_HIP_FUSE_WGS: workgroups of the persistent join (tuning; default = what the device holds)
  int fuse_min_pairs = 1;     // GPC_HIP_FUSE_MIN_PAIRS: smaller batches take the two-launch path
  int pre_rows = 0;           // GPC_HIP_PRE_ROWS = 14 | 6 | 2: that strip height of k_preprocess whatever the launch's size (tests)
  int fuse_shards = 0;        // GPC_HIP_FUSE_SHARDS: ticket counters the pairs are dealt over (0: join_shards() chooses)
  int num_cus = 0;
  std::map<std::pair<const void*, size_t>, int> wgs_per_cu;  // occupancy of the persistent instantiations launched so far, per LDS size
  std::map<const void*, int> dyn_lds;     // largest dynamic-LDS size a kernel has been allowed so far (hipFuncSetAttribute once, not per call)

  int hash_tpw = 0;    // GPC_HIP_HASH_TPW: tiles per workgroup of the hash kernel (tuning)
  int join_nt = 0;     // GPC_HIP_JOIN_NT = 256 | 512 | 1024: force the join kernel's threads per row (tuning)

  // rocprof name of the instantiation last launched under each timing slot (bench.py reports it)
  char launch_name[KID_COUNT][96] = {{0}};

  // timing
  bool timing = false;
  unsigned timing_mask = 0xFFFFFFFFu;  // which KernelIds are bracketed by events when timing is on
  std::vector<TimedSpan> spans;
  std::vector<TimedSpan> free_spans;

  std::vector<gpc_hip_train_set*> train_sets;  // training sets created on this context

  // Forest::preprocessImage -> Forest::rectifiedMatch without the round trip (the reference's PreprocessedImage travels by
  // value through host memory, inference.hpp:161-165): the last two preprocessed images stay on the device beside the
  // host copies the caller received; a match call whose host arrays are recognised as those copies (resident_slot) skips
  // the upload.  Guarded by g_res_mu: a preprocess call of ANY context that writes over a recorded host array drops the record.
  struct Resident {
    bool valid = false;
    const uint8_t *smooth = nullptr, *grad = nullptr;
    const int32_t* mask = nullptr;
    int n_mask = 0, W = 0, H = 0;
    bool naive = false;
    uint64_t fp = 0;  // fingerprint of the three host arrays as delivered
  };
  Resident res[2];
  int res_next = 0;
  DevBuf res_smooth, res_grad;  // [2][H][W] bytes each
  size_t res_n = 0;             // bytes per slot
  int resident_mode = 1;        // GPC_HIP_RESIDENT = 0: never; 1: pointers + sizes + sampled fingerprint; 2: + every byte hashed
  int resident_hits = 0;        // match calls served from the device-resident images (gpc_hip_resident_hits)
  // Page-locked transfer arena of the host-buffer entry points.  The caller's arrays (std::vector, ndb::Buffer, numpy:
  // pageable) never reach hipMemcpy: the runtime pins such memory for the copy and KEEPS the pinning cached, and when the
  // caller later frees the array the driver must suspend and restore the process's queues to drop it -- the next
  // submission then waits ~27 ms (measured: profiles/r05_a_cold_start.txt).  So pageable memory is copied by CPU threads
  // into / out of this arena, and the device reads / writes the arena over the link.
  uint8_t* h_xfer = nullptr;
  size_t h_xfer_cap = 0;
  // page-locked staging of gpc_hip_preprocess_begin / _fetch: [raw | smooth | grad | mask | count]
  uint8_t* h_pre = nullptr;
  size_t h_pre_cap = 0;
  int pre_slot = -1, pre_W = 0, pre_H = 0;  // what _begin left for _fetch
  bool pre_have_mask = false;
  hipEvent_t e_pre = nullptr;   // smooth and grad of that image have landed in the staging block
  // what gpc_hip_*_match_begin left for gpc_hip_match_fetch: results in the transfer arena (or, `direct`, already in the
  // caller's page-locked array), the count and the candidate counts in h_cnt[0 .. 2]
  struct PendingMatch {
    bool active = false, direct = false, have_ncand = false;
    bool packed = false;   // the arena holds [H row counts | cap_dev words xL | xR << 16] (epipolar sort-matcher): 4 bytes
                           // per support over the link instead of 12, expanded into the caller's array by the workers
    size_t esz = 0;
    int cap_dev = 0, H = 0;
  } pend;
  bool debug = false;           // GPC_HIP_DEBUG: launch geometry on stderr
  bool debug_plan = false;      // GPC_HIP_DEBUG_PLAN: the hash-table planner's choice on stderr
};

struct gpc_hip_train_set {
  gpc_hip_ctx* owner = nullptr;
  int n = 0;
  long np = 0;                // n rounded up to a multiple of 256
  uint8_t* planes = nullptr;  // [3][729][np]
  uint8_t* flags = nullptr;   // [np]
  int32_t* counts = nullptr;  // scratch: tp/fp per (candidate, tau) + tot
  gpc::GpcSplit* d_cand = nullptr;
  size_t counts_cap = 0, cand_cap = 0;
  std::vector<int32_t> h_counts;  // host staging of the counters
};

namespace {

// every live context (gpc_hip_create .. gpc_hip_destroy) and the lock of their Resident records
std::mutex g_res_mu;
std::vector<gpc_hip_ctx*> g_ctxs;

#define HIPCHK(ctx, call)                                                              \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call,        \
               hipGetErrorString(e_), __FILE__, __LINE__);                             \
      return GPC_E_HIP;                                                                \
    }                                                                                  \
  } while (0)

#define CHK(expr)                  \
  do {                             \
    int s_ = (expr);               \
    if (s_ != GPC_OK) return s_;   \
  } while (0)

double host_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

int ensure(gpc_hip_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap) return GPC_OK;
  if (b.p) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
  }
  // slack: growth without reallocation, and the fused join reads whole pixel-slot rows (up to 16 KiB past a row's end)
  size_t want = bytes + bytes / 8 + 65536;
  HIPCHK(c, hipMalloc(&b.p, want));
  b.cap = want;
  return GPC_OK;
}

void release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

struct Timed {
  gpc_hip_ctx* c;
  TimedSpan s;
  bool on;
  Timed(gpc_hip_ctx* ctx, int kid) : c(ctx), on(ctx->timing && ((ctx->timing_mask >> kid) & 1u)) {
    if (!on) return;
    if (!c->free_spans.empty()) {
      s = c->free_spans.back();
      c->free_spans.pop_back();
    } else {
      if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) {
        on = false;
        return;
      }
    }
    s.kid = kid;
    (void)hipEventRecord(s.a, c->stream);
  }
  ~Timed() {
    if (!on) return;
    (void)hipEventRecord(s.b, c->stream);
    c->spans.push_back(s);
  }
};

// hipFuncAttributeMaxDynamicSharedMemorySize, raised only when a launch needs more than the kernel has been granted
int allow_dyn_lds(gpc_hip_ctx* c, const void* fn, size_t bytes) {
  int& have = c->dyn_lds[fn];
  if ((int)bytes <= have) return GPC_OK;
  HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  have = (int)bytes;
  return GPC_OK;
}

int check_dims(int W, int H) {
  if (W <= 0 || H <= 0 || (W % 16) != 0) return GPC_E_INVALID;
  if (W < 2 * GPC_R + 16 || H < 2 * GPC_R + 4) return GPC_E_INVALID;
  // the reference takes any multiple of 16; here a row must fit one workgroup's join table (k_rowjoin.h) and a
  // pixel index 30 bits
  if (W > 16384 || (long)W * H > (1l << 30)) return GPC_E_UNSUPPORTED;
  return GPC_OK;
}

// CPUs this process may really use: affinity mask and cgroup quota, at most 16
int usable_cpus() {
  int n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (FILE* fp = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32];
    long per = 0;
    if (fscanf(fp, "%31s %ld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
      const long lim = atol(q) / per;
      if (lim >= 1 && lim < n) n = (int)lim;
    }
    fclose(fp);
  }
  if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
    long quota = -1, per = 0;
    if (fscanf(fp, "%ld", &quota) == 1 && quota > 0) {
      if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (fscanf(fq, "%ld", &per) == 1 && per > 0 && quota / per >= 1 && quota / per < n) n = (int)(quota / per);
        fclose(fq);
      }
    }
    fclose(fp);
  }
  return n < 1 ? 1 : (n > 16 ? 16 : n);
}

// The CPUs of the GPU's NUMA node: /sys/bus/pci/devices/<bus id>/numa_node, then that node's cpulist, intersected with
// what the process may use.  GPC_HIP_NO_NUMA_BIND leaves the workers unbound.
void find_gpu_node_cpus(gpc_hip_ctx* c) {
  c->have_node_cpus = false;
  if (getenv("GPC_HIP_NO_NUMA_BIND")) return;
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, c->device) != hipSuccess) return;
  for (char* p = bdf; *p; ++p) *p = (char)tolower((unsigned char)*p);
  char path[256];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
  int node = -1;
  if (FILE* fp = fopen(path, "r")) {
    if (fscanf(fp, "%d", &node) != 1) node = -1;
    fclose(fp);
  }
  if (node < 0) return;
  snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
  FILE* fp = fopen(path, "r");
  if (!fp) return;
  char list[4096] = {0};
  const size_t got = fread(list, 1, sizeof list - 1, fp);
  fclose(fp);
  list[got] = 0;
  cpu_set_t node_set, mine;
  CPU_ZERO(&node_set);
  for (const char* p = list; *p;) {  // "0-15,128-143"
    while (*p && !isdigit((unsigned char)*p)) ++p;
    if (!*p) break;
    char* e = nullptr;
    long a = strtol(p, &e, 10), b = a;
    p = e;
    if (*p == '-') b = strtol(p + 1, &e, 10), p = e;
    for (long k = a; k <= b && k < CPU_SETSIZE; ++k) CPU_SET((int)k, &node_set);
  }
  if (sched_getaffinity(0, sizeof mine, &mine) != 0) return;
  CPU_AND(&c->node_cpus, &node_set, &mine);
  if (CPU_COUNT(&c->node_cpus) < 2) return;  // nothing sensible to bind to
  c->numa_node = node;
  c->have_node_cpus = true;
  // the node's CPUs by last-level cache: /sys/devices/system/cpu/cpuN/cache/index3/shared_cpu_list ("64-71,192-199")
  c->node_l3.clear();
  if (getenv("GPC_HIP_NO_CCD_SPREAD")) return;
  cpu_set_t seen;
  CPU_ZERO(&seen);
  for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) {
    if (!CPU_ISSET(cpu, &c->node_cpus) || CPU_ISSET(cpu, &seen)) continue;
    snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE* f3 = fopen(path, "r");
    if (!f3) { c->node_l3.clear(); return; }
    char l3[1024] = {0};
    const size_t g3 = fread(l3, 1, sizeof l3 - 1, f3);
    fclose(f3);
    l3[g3] = 0;
    cpu_set_t grp;
    CPU_ZERO(&grp);
    for (const char* p = l3; *p;) {
      while (*p && !isdigit((unsigned char)*p)) ++p;
      if (!*p) break;
      char* e = nullptr;
      long a = strtol(p, &e, 10), b = a;
      p = e;
      if (*p == '-') b = strtol(p + 1, &e, 10), p = e;
      for (long k = a; k <= b && k < CPU_SETSIZE; ++k)
        if (CPU_ISSET((int)k, &c->node_cpus)) CPU_SET((int)k, &grp);
    }
    if (CPU_COUNT(&grp) == 0) CPU_SET(cpu, &grp);
    CPU_OR(&seen, &seen, &grp);
    c->node_l3.push_back(grp);
  }
  if (c->node_l3.size() < 2) c->node_l3.clear();  // one cache domain: nothing to spread over
}

// Worker threads gpc_hip_match_batch may start for the host expansion when nobody said how many: the process's CPUs
// shared among the ranks of this node (one process per GPU sets LOCAL_WORLD_SIZE, torch.distributed.run does), less one
// for the thread that feeds the GPU; at least 2, at most 8 (3 .. 10 workers all keep up with the link).
int default_expand_threads() {
  int share = 0;
  int local_world = 1;
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 64) local_world = v;
  }
  cpu_set_t set;
  int visible = 1;
  if (sched_getaffinity(0, sizeof set, &set) == 0) visible = CPU_COUNT(&set);
  // a one-GPU box gives the process a quota of its own (usable_cpus sees it); on a node shared by the ranks of one
  // job every rank sees all CPUs and must share them
  share = local_world > 1 ? visible / local_world - 1 : usable_cpus() - 2;
  return share < 2 ? 2 : (share > 8 ? 8 : share);
}

// Packed supports (xL | xR << 16, rows in ascending order, rows[y] of them in row y) -> ndb::Support records
// {x, y, float(xL - xR)} (inference.hpp:384-391, buffer.hpp:91-97) for the rows [y0, y1) of one pair whose first
// support has index `first`; stops at index `limit`.  Four records are 48 bytes = three 16-byte streaming
// stores (no read-for-ownership of the 625 MB a 256-pair batch expands to).
void expand_rows(const uint32_t* packed, const int32_t* rows, int y0, int y1, long first, long limit, gpc_support* out) {
  long pos = first;
  const uint32_t* src = packed + first;
#if defined(__SSE2__)
  const bool aligned = (reinterpret_cast<uintptr_t>(out) & 15u) == 0;
  const __m128i m16 = _mm_set1_epi32(0xFFFF);
#endif
  for (int y = y0; y < y1 && pos < limit; ++y) {
    const long cnt = rows[y];
    long n = cnt;
    if (pos + n > limit) n = limit - pos;
    long i = 0;
    auto one = [&](uint32_t v) {
      const int xl = (int)(v & 0xFFFFu), xr = (int)(v >> 16);
      out[pos].x = xl;
      out[pos].y = y;
      out[pos].d = (float)(xl - xr);
      ++pos;
    };
#if defined(__SSE2__)
    if (aligned) {
      for (; i < n && (pos & 3); ++i) one(src[i]);
      const __m128 Y = _mm_castsi128_ps(_mm_set1_epi32(y));
      for (; i + 4 <= n; i += 4, pos += 4) {
        const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
        const __m128i xl = _mm_and_si128(v, m16);
        const __m128 X = _mm_castsi128_ps(xl);
        const __m128 D = _mm_cvtepi32_ps(_mm_sub_epi32(xl, _mm_srli_epi32(v, 16)));
        const __m128 t0 = _mm_castsi128_ps(_mm_unpacklo_epi32(xl, _mm_castps_si128(Y)));  // x0 y x1 y
        const __m128 t1 = _mm_castsi128_ps(_mm_unpackhi_epi32(xl, _mm_castps_si128(Y)));  // x2 y x3 y
        const __m128 a = _mm_shuffle_ps(t0, _mm_shuffle_ps(D, X, _MM_SHUFFLE(1, 1, 0, 0)), _MM_SHUFFLE(2, 0, 1, 0));  // x0 y d0 x1
        const __m128 b = _mm_shuffle_ps(_mm_shuffle_ps(Y, D, _MM_SHUFFLE(1, 1, 0, 0)), t1, _MM_SHUFFLE(1, 0, 2, 0));  // y d1 x2 y
        const __m128 c = _mm_shuffle_ps(_mm_shuffle_ps(D, X, _MM_SHUFFLE(3, 3, 2, 2)),
                                        _mm_shuffle_ps(Y, D, _MM_SHUFFLE(3, 3, 0, 0)), _MM_SHUFFLE(2, 0, 2, 0));      // d2 x3 y d3
        __m128i* o = reinterpret_cast<__m128i*>(out + pos);
        _mm_stream_si128(o, _mm_castps_si128(a));
        _mm_stream_si128(o + 1, _mm_castps_si128(b));
        _mm_stream_si128(o + 2, _mm_castps_si128(c));
      }
    }
#endif
    for (; i < n; ++i) one(src[i]);
    src += cnt;
  }
#if defined(__SSE2__)
  _mm_sfence();
#endif
}

void ExpandPool::run(int index) {
  for (;;) {
    ExpandJob j;
    {
      std::unique_lock<std::mutex> g(m_);
      cv_.wait(g, [&] { return quit_ || !q_.empty(); });
      if (q_.empty()) return;
      j = q_.front();
      q_.pop_front();
      cpus_[(size_t)index] = sched_getcpu();
    }
    if (j.copy_bytes) memcpy(j.copy_dst, j.copy_src, j.copy_bytes);
    else expand_rows(j.packed, j.rows, j.y0, j.y1, j.first, j.limit, j.out);
    {
      std::lock_guard<std::mutex> g(m_);
      if (--pending_[j.slot & 7] == 0) done_.notify_all();
    }
  }
}

int pow2_at_least(int v) {
  int p = 2;
  while (p < v) p <<= 1;
  return p;
}

// ---------------------------------------------------------------- forest parsing

bool next_tok(const char*& p, std::string& tok) {
  while (*p && isspace((unsigned char)*p)) ++p;
  if (!*p) return false;
  const char* s = p;
  while (*p && !isspace((unsigned char)*p)) ++p;
  tok.assign(s, p - s);
  return true;
}

bool next_int(const char*& p, int& v) {
  std::string tok;
  if (!next_tok(p, tok)) return false;
  char* end = nullptr;
  long l = strtol(tok.c_str(), &end, 10);
  if (end == tok.c_str()) return false;
  v = (int)l;
  return true;
}

// ---------------------------------------------------------------- pipeline stages

int run_global_match(gpc_hip_ctx* c, int W, int H, int npairs, const gpc_settings* s, int mode, const uint8_t* d_cand,
                     void* d_out, int cap, int32_t* d_counts, int32_t* d_ncand);
int run_hashtable_match(gpc_hip_ctx* c, int W, int H, int npairs, const gpc_settings* s, int mode, const uint8_t* d_cand,
                        void* d_out, int cap, int32_t* d_counts, int32_t* d_ncand);

// raw0/raw1 device pointers; fills smooth, grad for npairs*sides images
// gradbits: the caller's only reader of the gradient image is run_hash (a batched pipeline): it may leave as one bit per pixel
// d_smooth_out / d_grad_out: where the images go instead of the context's workspaces (byte gradient image only)
int run_preprocess(gpc_hip_ctx* c, const uint8_t* d_raw0, const uint8_t* d_raw1, int W, int H,
                   int npairs, int sides, int thr, bool gradbits = false, uint8_t* d_smooth_out = nullptr,
                   uint8_t* d_grad_out = nullptr) {
  const int nimg = npairs * sides;
  const size_t n = (size_t)W * H;
  if (!d_smooth_out) {
    CHK(ensure(c, c->smooth, n * nimg));
    CHK(ensure(c, c->grad, n * nimg));
    d_smooth_out = (uint8_t*)c->smooth.p;
    d_grad_out = (uint8_t*)c->grad.p;
  } else {
    gradbits = false;
  }
  CHK(ensure(c, c->stats, sizeof(int32_t) * GPC_STAT_STRIDE * nimg));
  // threshold^2 passes through _mm_set1_epi16 in the SSE build (filter.hpp:418); sobelNaive keeps the int (:159)
  const int thr_sq = c->naive ? (thr & 0xFF) * (thr & 0xFF) : (int)(int16_t)(uint16_t)((thr & 0xFF) * (thr & 0xFF));
  // Rows per thread (the strip a thread marches down): 14 read every raw row 1.14 times and are what a launch that fills the
  // device many times over wants; a smaller launch is a matter of ROUNDS -- the device holds 8 workgroups per CU, and a launch
  // of 1.2 rounds takes two -- so the strip height is the one whose rounds x (rows + 2 read + ~2 of set-up) comes out lowest:
  // 64 images of 1024x436 are 1024 workgroups of 14-row strips (half a round of long chains: 23 us), 2432 of 6-row ones
  // (1.2 rounds: 21 us) and exactly 2048 of 7-row ones (one round).
  const int gx = (W / PP_PX + PP_TX - 1) / PP_TX;
  auto blocks_with = [&](int r) { return (long)gx * ((H + PP_TY * r - 1) / (PP_TY * r)) * nimg; };
  int rows = c->pre_rows;
  if (!rows) {
    static const int heights[] = {PP_ROWS, 10, 7, PP_ROWS_MID, 4, PP_ROWS_SMALL};
    const long places = 8l * (c->num_cus > 0 ? c->num_cus : 256);
    long best = -1;
    for (int r : heights) {
      const long cost = (blocks_with(r) + places - 1) / places * (r + 4);
      if (best < 0 || cost < best) { best = cost; rows = r; }   // (ties go to the taller strip: fewer re-read rows)
    }
  }
  dim3 grid(gx, (H + PP_TY * rows - 1) / (PP_TY * rows), nimg);
  Timed t(c, KID_PREPROCESS);
  // (the SSE=OFF arithmetic keeps the byte image: its 32-test codes need the candidate BYTES in the matchers, wide_codes())
  c->grad_is_bits = gradbits && !c->naive && !c->no_grad_bits;
  snprintf(c->launch_name[KID_PREPROCESS], sizeof c->launch_name[0], "gpc::k_preprocess<%s, %d%s>", c->naive ? "true" : "false", rows,
           c->grad_is_bits ? ", true" : "");
#define LAUNCH_PRE(NAIVE, ROWS, BITS)                                                                        \
  hipLaunchKernelGGL((gpc::k_preprocess<NAIVE, ROWS, BITS>), grid, dim3(PP_TX * PP_TY), 0, c->stream, d_raw0, d_raw1, \
                     d_smooth_out, d_grad_out, W, H, sides, thr_sq, (int32_t*)c->stats.p)
#define LAUNCH_PRE_ROWS(NAIVE, BITS)                                  \
  do {                                                                \
    if (rows == PP_ROWS) LAUNCH_PRE(NAIVE, PP_ROWS, BITS);            \
    else if (rows == 10) LAUNCH_PRE(NAIVE, 10, BITS);                 \
    else if (rows == 7) LAUNCH_PRE(NAIVE, 7, BITS);                   \
    else if (rows == PP_ROWS_MID) LAUNCH_PRE(NAIVE, PP_ROWS_MID, BITS); \
    else if (rows == 4) LAUNCH_PRE(NAIVE, 4, BITS);                   \
    else LAUNCH_PRE(NAIVE, PP_ROWS_SMALL, BITS);                      \
  } while (0)
  if (c->naive) LAUNCH_PRE_ROWS(true, false);
  else if (c->grad_is_bits) LAUNCH_PRE_ROWS(false, true);
  else LAUNCH_PRE_ROWS(false, false);
#undef LAUNCH_PRE_ROWS
#undef LAUNCH_PRE
  HIPCHK(c, hipGetLastError());
  return GPC_OK;
}

// smooth/grad/(candmap) device pointers for nimg images -> code image
int run_hash(gpc_hip_ctx* c, const uint8_t* d_smooth, const uint8_t* d_grad, const uint8_t* d_cand,
             int W, int H, int nimg, bool dense, uint32_t* d_codes) {
  if (!c->have_forest) return GPC_E_NO_FOREST;
  // tiles per workgroup: each CU holds 2 workgroups (67 KiB of LDS each); walking several
  // vertically adjacent tiles hides the next window's load latency, but the grid must still fill
  // the 512 slots and split the tile rows evenly.  cost ~ rounds * tiles per workgroup.
  // tiles cover the candidate rows 13 .. H-14 only (k_hash.h)
  // the gradient image is k_preprocess's bit image: the batched SSE pipelines (run_preprocess(..., gradbits))
  const bool gbits = c->grad_is_bits && d_grad == (const uint8_t*)c->grad.p && !dense && !c->naive && d_cand == nullptr;
  const int slots = 2 * (c->num_cus > 0 ? c->num_cus : 256);  // workgroups of k_hash the device holds at once (2 per CU: 67-76 KiB of LDS each)
  const int gx = (W + HT_X - 1) / HT_X;
  int ty = HT_Y;
  {
    // Launches too small for several tiles per workgroup run in ROUNDS of `slots` one-tile workgroups: a single 1920x1080
    // pair is 8 x 33 x 2 = 528 tiles of 32 rows -- a second round for 16 of them, 39 us of a 74 us step -- and 432 tiles
    // of 40 rows.  The taller tile (batched SSE pipelines only) is taken where rounds x window rows comes out lower.
    const long n32 = (long)gx * ((H - 2 * GPC_R + HT_Y - 1) / HT_Y) * nimg, n40 = (long)gx * ((H - 2 * GPC_R + HT_Y_TALL - 1) / HT_Y_TALL) * nimg;
    const long c32 = (n32 + slots - 1) / slots * (HT_Y + 2 * GPC_R), c40 = (n40 + slots - 1) / slots * (HT_Y_TALL + 2 * GPC_R);
    if (gbits && n32 < 2l * slots && c40 < c32) ty = HT_Y_TALL;
    if (gbits && c->hash_tall == 1) ty = HT_Y_TALL;
    if (c->hash_tall == 0) ty = HT_Y;
  }
  const int tiles_y = (H - 2 * GPC_R + ty - 1) / ty;
  int tpw = 1;
  if ((long)gx * tiles_y * nimg >= 2l * slots) {  // small launches keep one tile per workgroup (parallelism first)
    // What a split costs is the longest chain of tiles one of the `slots` resident workgroup places works through.  A column
    // of tiles_y tiles becomes ceil(tiles_y / t) workgroups of t tiles and one of the remainder; the hardware starts the
    // next workgroup wherever one ends, so the places fill like bins: as many whole workgroups as fit the rounds, the
    // short ones last.  (Scored as "idle tail of the last round + 4 % for a ragged split" before: at 32 pairs of 1024x436
    // -- 13 tile rows, 256 columns -- that took 4 + 4 + 4 + 1 tiles in two rounds (8 tiles on the longest chain) over
    // 7 + 6 in one (7): k_hash 58 -> 52 us.)  A workgroup's start (tap tables, first window not overlapped) is ~0.4 tile.
    double best = 1e30;
    for (int t = 2; t <= 16 && t <= tiles_y; ++t) {
      const long per_col = (tiles_y + t - 1) / t, ncol = (long)gx * nimg;
      const long nwg = per_col * ncol;
      // (a split that leaves a few places empty still counts: one 3840x2160 pair is 15 x 67 x 2 tiles = 510 workgroups of 4 for
      //  512 places -- 57 us where 1020 workgroups of 2 take 60, and 68 with k_hash's wave priorities)
      if (nwg * 8 < (long)slots * 7) break;
      const int rem = tiles_y - (int)(per_col - 1) * t;             // tiles of a column's last workgroup (1 .. t)
      const long n_full = (per_col - 1) * ncol, n_rem = ncol;         // workgroups of t tiles / of rem tiles
      // full-size workgroups dealt over the places first, the short ones onto the least loaded places
      const long full_rounds = n_full / slots, full_left = n_full % slots;
      double chain = (double)full_rounds * (t + 0.4);
      long short_left = n_rem;
      if (full_left) {  // `full_left` places carry one more full workgroup; the others take short ones meanwhile
        const long free_places = slots - full_left;
        const long fit = free_places * (long)((t + 0.4) / (rem + 0.4));  // short workgroups that fit beside that round
        short_left = n_rem > fit ? n_rem - fit : 0;
        chain += t + 0.4;
      }
      chain += (double)((short_left + slots - 1) / slots) * (rem + 0.4);
      if (chain < best - 1e-9) { best = chain; tpw = t; }
    }
  }
  if (c->hash_tpw > 0) tpw = c->hash_tpw < tiles_y ? c->hash_tpw : tiles_y;  // GPC_HIP_HASH_TPW (tuning)
  dim3 grid(gx, (tiles_y + tpw - 1) / tpw, nimg);
  // the last `slots` workgroups in dispatch order: the launch's last round (k_hash.h: wave priority by the tiles left)
  const long nwg_all = (long)grid.x * grid.y * grid.z;
  const int last_from = nwg_all > slots ? (int)(nwg_all - slots) : 0;
  Timed t(c, KID_HASH);
  const bool tau = c->forest.type != 0;
  int32_t* st = (int32_t*)c->stats.p;
  snprintf(c->launch_name[KID_HASH], sizeof c->launch_name[0], "gpc::k_hash<%s, %s, %s, %s, %d>", tau ? "true" : "false",
           dense ? "true" : "false", c->naive ? "true" : "false", gbits ? "true" : "false", ty);
#define LAUNCH_HASH(TAU, DENSE, NAIVE)                                                                    \
  hipLaunchKernelGGL((gpc::k_hash<TAU, DENSE, NAIVE>), grid, dim3(HT_THREADS), 0, c->stream, d_smooth, d_grad, \
                     d_cand, d_codes, W, H, (const GpcForestDev*)c->forest_dev.p + (NAIVE ? 1 : 0), st, tpw, last_from)
  if (gbits) {
#define LAUNCH_HASH_BITS(TAU, TY, FD)                                                                                      \
  hipLaunchKernelGGL((gpc::k_hash<TAU, false, false, true, TY>), grid, dim3(HT_THREADS), 0, c->stream, d_smooth, d_grad, \
                     d_cand, d_codes, W, H, (const GpcForestDev*)c->forest_dev.p + FD, st, tpw, last_from)
    if (ty == HT_Y_TALL) {
      if (tau) LAUNCH_HASH_BITS(true, HT_Y_TALL, 2); else LAUNCH_HASH_BITS(false, HT_Y_TALL, 2);
    } else {
      if (tau) LAUNCH_HASH_BITS(true, HT_Y, 0); else LAUNCH_HASH_BITS(false, HT_Y, 0);
    }
#undef LAUNCH_HASH_BITS
  } else if (c->naive) {
    if (tau && dense) LAUNCH_HASH(true, true, true);
    else if (tau) LAUNCH_HASH(true, false, true);
    else if (dense) LAUNCH_HASH(false, true, true);
    else LAUNCH_HASH(false, false, true);
  } else {
    if (tau && dense) LAUNCH_HASH(true, true, false);
    else if (tau) LAUNCH_HASH(true, false, false);
    else if (dense) LAUNCH_HASH(false, true, false);
    else LAUNCH_HASH(false, false, false);
  }
#undef LAUNCH_HASH
  HIPCHK(c, hipGetLastError());
  return GPC_OK;
}

// Significant bits of the codes the current forest can produce (the partitioned matcher bins on the top ones of
// these, k_partition.h).  SSE placement (filter.hpp:574-595): test t -> bit t for
// t <= 7, test 8 is OR-ed into bit 0, test t -> bit t-1 for t >= 9; Naive (filter.hpp:245-249): T bits.
int code_bits(const gpc_hip_ctx* c) {
  const int T = c->forest.num_tests;
  if (c->naive) return T;
  return T <= 8 ? T : T - 1;
}

// SSE=OFF arithmetic with 32 tests: codes use bit 31 and 0xFFFFFFFF is a code (k_rowjoin.h, WIDE)
bool wide_codes(const gpc_hip_ctx* c) { return c->naive && c->forest.num_tests == 32; }

// How the join kernel covers a row of W pixels: NT threads x SPT pixel slots, table of 1 << log2s slots.
struct JoinPlan {
  int nt, spt, log2s;
  size_t lds;
};

JoinPlan plan_join(const gpc_hip_ctx* c, int W) {
  JoinPlan p;
  p.spt = 1;
  p.nt = 256;
  while (p.spt * p.nt < W && p.spt < 4) p.spt <<= 1;    // <= 1024 px: 256 threads, 8 workgroups per CU
  while (p.spt * p.nt < W && p.nt < 1024) p.nt <<= 1;   // <= 4096 px: more threads per row
  if (c->join_nt && 4 * c->join_nt >= W) {  // tuning override (GPC_HIP_JOIN_NT): threads per row, slots per thread follow
    p.nt = c->join_nt;
    p.spt = 1;
    while (p.spt * p.nt < W) p.spt <<= 1;
  }
  while (p.spt * p.nt < W && p.spt < 16) p.spt <<= 1;   // <= 16384 px: 1024 threads with 8 / 16 slots each
  // only left codes are inserted: S >= 2*(W-26) keeps the load factor <= 0.5 (up to the 16384 slots = 128 KiB
  // one workgroup can have: beyond 8218 px the table fills further, W-26 < 16384 keys always fit);
  // S >= NT*SPT because the rank phase reuses the key table as bucket counters
  p.log2s = 1;
  while ((1 << p.log2s) < p.nt * p.spt || ((1 << p.log2s) < 2 * (W - 2 * GPC_R) && p.log2s < 14)) ++p.log2s;
  p.lds = ((size_t)8 * ((1u << p.log2s) + 1) + 15) / 16 * 16;  // keys + flag/x words
  return p;
}

// A look-back of the fused join gave up (k_rowjoin_fused.h, RJ_SPIN_LIMIT): the results of that launch are not to be
// used.  The kernel stores the launch's epoch (never 0) into a page-locked word with a plain system-scope store; the word
// is looked at wherever an entry point has just synchronised the stream, at the top of the NEXT fused launch (callers that
// wait on their own stream -- set_stream + their own synchronisation -- learn of it there at the latest) and in
// gpc_hip_destroy, and says WHICH launch failed.
int check_join_err(gpc_hip_ctx* c) {
#ifdef RJ_DBG_COUNT
  if (c->h_err && c->h_err[1]) {
    fprintf(stderr, "[RJ_DBG_COUNT] look-backs %d, windows with a row that had not published %d, displaced keys that met their copy %d\n", c->h_err[1], c->h_err[2], c->h_err[3]);
    c->h_err[1] = c->h_err[2] = c->h_err[3] = 0;
  }
#endif
  if (c->h_err) {
    // (one exchange: a launch still in flight may store its epoch between a load and a clearing store, and would be lost)
    const int32_t ep = __atomic_exchange_n(c->h_err, 0, __ATOMIC_ACQ_REL);
    if (ep) {
      snprintf(c->err, sizeof(c->err), "k_row_join_fused: a row waited too long for the rows before it in fused launch %d of this "
               "context (%u launched so far): THAT launch's supports are not to be used; a call that reports this at its start "
               "has queued nothing itself", ep, c->join_epoch);
      return GPC_E_HIP;
    }
  }
  return GPC_OK;
}

// Ticket counters (shards of pairs) of a fused launch.  Workgroup b serves shard b % shards and lands on XCD b % 8, so
//   * 8 | shards pins every shard -- its pairs -- to one XCD, and the XCDs finish apart: 256 pairs with 8 / 16 / 32 / 64
//     shards 567-570 us, with 3 .. 13 shards 550-554 us (15 / 17 / 21 / 31: 561 / 562 / 565 / 573; 2 shards 653 and one
//     shard 1246: same-address atomics on the counter);
//   * a shard's workgroups take only its rows, so shards of unequal size finish apart as well: 8 pairs of 1920x1080 over
//     7 shards (one of them two pairs) 155 us against 123 over 8.
// So: 3 .. 13 shards, not a multiple of 8, the least time the fullest shard runs alone; ties go to the count nearest 7.
int join_shards(const gpc_hip_ctx* c, int npairs) {
  if (c->fuse_shards > 0) return npairs < c->fuse_shards ? npairs : c->fuse_shards;
  if (npairs <= 3) return npairs;
  int best = 3;
  long best_waste = -1;
  for (int n = 3; n <= 13 && n <= npairs; ++n) {
    if (n % 8 == 0) continue;
    long waste = ((long)((npairs + n - 1) / n) * n - npairs) * 1000 / npairs;   // per mille of the launch the fullest shard runs alone
    if (waste <= 15) waste = 0;                                                   // (below the boxes' run-to-run spread)
    const bool nearer = abs(n - 7) < abs(best - 7);
    if (best_waste < 0 || waste < best_waste || (waste == best_waste && nearer)) {
      best = n;
      best_waste = waste;
    }
  }
  return best;
}

// State of the fused join: ticket counters + one granule per (pair, row); zeroed when (re)allocated, then kept
// consistent by the kernel itself (the last draw resets a counter; granules carry the launch's epoch).
int ensure_join_state(gpc_hip_ctx* c, size_t granules) {
  const size_t tk_bytes = sizeof(uint32_t) * RJ_SHARDS * RJ_TICKET_STRIDE;
  if (!c->h_err) {
    HIPCHK(c, hipHostMalloc((void**)&c->h_err, 64, hipHostMallocMapped));
    memset(c->h_err, 0, 64);
    HIPCHK(c, hipHostGetDevicePointer((void**)&c->d_err, c->h_err, 0));
  }
  if (granules > c->jstate_granules || c->join_epoch >= (1u << 30) - 2u) {
    CHK(ensure(c, c->jstate, tk_bytes + sizeof(unsigned long long) * granules));
    HIPCHK(c, hipMemsetAsync(c->jstate.p, 0, c->jstate.cap, c->stream));
    c->jstate_granules = (c->jstate.cap - tk_bytes) / sizeof(unsigned long long);
    c->join_epoch = 0;
  }
  return GPC_OK;
}

int ensure_flag_event(gpc_hip_ctx* c) {
  if (!c->e_flag) HIPCHK(c, hipEventCreateWithFlags(&c->e_flag, hipEventDisableTiming));
  return GPC_OK;
}

// The second stream of the device-wide matchers: a launch for the few over-large partitions / bins runs beside the main one.
int ensure_aux_stream(gpc_hip_ctx* c) {
  if (c->s_aux) return GPC_OK;
  HIPCHK(c, hipStreamCreateWithFlags(&c->s_aux, hipStreamNonBlocking));
  HIPCHK(c, hipEventCreateWithFlags(&c->e_fork, hipEventDisableTiming));
  HIPCHK(c, hipEventCreateWithFlags(&c->e_join, hipEventDisableTiming));
  return GPC_OK;
}

// code images of npairs pairs -> supports / correspondences in d_out.
// d_cand: the candidate bytes the hash kernel used ([2*npairs][H][W]: grad, or the scattered mask list)
// mode 0: gpc_support, 1: gpc_correspondence, 2: packed supports (epipolar sort-match only; `po` says where)
int run_match(gpc_hip_ctx* c, int W, int H, int npairs, const gpc_settings* s, int mode, const uint8_t* d_cand,
              void* d_out, int cap, int32_t* d_counts, int32_t* d_ncand, const PackedOut* po = nullptr) {
  const int apply_filter = (mode != 1);
  if (mode == 2 && (s->use_hashtable || !s->epipolar_mode || !po)) return GPC_E_UNSUPPORTED;
  if (s->use_hashtable) return run_hashtable_match(c, W, H, npairs, s, mode, d_cand, d_out, cap, d_counts, d_ncand);
  if (s->epipolar_mode) {
    CHK(ensure(c, c->staged, sizeof(uint32_t) * (size_t)W * H * npairs));
    CHK(ensure(c, c->rowcnt, sizeof(int32_t) * (size_t)H * npairs * 2));
    // |dy| is 0 for every epipolar match; a negative tolerance rejects everything
    int disp_high = s->disp_high;
    if (apply_filter && s->vertical_tolerance < 0) disp_high = -1;
    const JoinPlan jp = plan_join(c, W);
    if (jp.nt * jp.spt < W) return GPC_E_UNSUPPORTED;  // unreachable below check_dims' 16384 px
    // One launch for join + output (k_rowjoin_fused.h) for rows up to 4096 px (12 bits of x beside the flags) when the
    // records' place follows from the rows before them alone (not the gap-free packing of `totals`)
    const size_t flds = RJF_LDS_BYTES(jp.nt * jp.spt);
    bool fuse = !c->no_fuse && npairs >= c->fuse_min_pairs && jp.spt <= 4 && jp.nt * jp.spt <= 4096 &&
                !(po && po->totals) && (long)npairs * (H - 2 * GPC_R) < (1l << 31) - 65536;
    if (fuse && !c->fuse_always) {
      // The persistent launch pays off once every workgroup takes several rows (its output lags one row behind, and the
      // last row of a workgroup is placed with a blocking look-back): measured against join + k_gather_rows (round 5, with
      // the wave priorities of k_rowjoin_fused.h), 1024x436 steps: 4 .. 16 pairs 4-14 % slower, 20 / 24 pairs 3 % faster,
      // 28 pairs 8 %, 32 .. 256 pairs 8-11 %; 1920x1080: one pair 6 % slower, 8 pairs 8 % faster; one 3840x2160 pair 9 %
      // faster.  Rows per resident workgroup >= 4, or >= 3 for rows of 2048 px and more, is where it wins.
      const long lds_wgs = (long)(160 * 1024 / (flds + 64)), wave_wgs = 32 / (jp.nt / 64);
      const long resident = (long)c->num_cus * (lds_wgs < wave_wgs ? lds_wgs : wave_wgs);
      const long rows_total = (long)npairs * (H - 2 * GPC_R);
      fuse = rows_total >= 4 * resident || (W >= 2048 && rows_total >= 3 * resident);
    }
    if (fuse) {
      // k_row_join_fused (k_rowjoin_fused.h): one by-value parameter the kernel reads from its argument segment
      const int nrows = H - 2 * GPC_R;
      CHK(check_join_err(c));  // a look-back of an EARLIER launch that timed out is reported before anything new is queued
      CHK(ensure_join_state(c, (size_t)npairs * nrows));
      const size_t lds = flds;
      gpc::RjfArgs a;
      memset(&a, 0, sizeof a);
      a.codes = (const uint32_t*)c->codes.p;
      a.cand = d_cand;
      a.img_stats = (const int32_t*)c->stats.p;
      a.tickets = (uint32_t*)c->jstate.p;
      a.status = (unsigned long long*)((uint32_t*)c->jstate.p + RJ_SHARDS * RJ_TICKET_STRIDE);
      a.err = c->d_err;
      a.out = d_out;
      a.counts = d_counts;
      a.ncand = d_ncand;
      a.rows_out = po ? po->rows : nullptr;
      a.packed_stride = po ? po->packed_stride : 0l;
      a.rows_stride = po ? po->rows_stride : 0l;
      a.W = W;
      a.H = H;
      a.disp_high = disp_high;
      a.apply_filter = apply_filter;
      a.epoch = ++c->join_epoch;
      a.npairs = npairs;
      a.mode = mode;
      a.cap = cap;
      Timed t(c, KID_ROW_JOIN);
      const bool wide = wide_codes(c);
      snprintf(c->launch_name[KID_ROW_JOIN], sizeof c->launch_name[0], "gpc::k_row_join_fused<%d, %d, %s>", jp.spt, jp.nt,
               wide ? "true" : "false");
      c->launch_name[KID_GATHER_ROWS][0] = 0;
#define LAUNCH_RJF(SPT, NT, WIDE)                                                                               \
  do {                                                                                                          \
    const void* fn_ = reinterpret_cast<const void*>(gpc::k_row_join_fused<SPT, NT, WIDE>);                      \
    int& per_cu = c->wgs_per_cu[std::make_pair(fn_, lds)];                                                      \
    if (per_cu == 0) {                                                                                          \
      if (lds > 48 * 1024) CHK(allow_dyn_lds(c, fn_, lds));                                                     \
      HIPCHK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn_, NT, lds));                           \
      if (per_cu < 1) per_cu = 1;                                                                               \
    }                                                                                                           \
    /* (two lanes: one workgroup per CU less, the other lane's k_preprocess / k_hash need a wave slot per SIMD) */ \
    long nwg = c->fuse_wgs > 0 ? c->fuse_wgs : (long)((c->pipeline > 1 && per_cu > 1) ? per_cu - 1 : per_cu) * c->num_cus; \
    if (nwg > (long)npairs * nrows) nwg = (long)npairs * nrows;                                                 \
    /* every shard needs a workgroup that draws its tickets (workgroup b serves shard b % nshards) */           \
    int nsh = join_shards(c, npairs);                                                                           \
    if (nsh > nwg) nsh = (int)nwg;                                                                              \
    a.nshards = nsh;                                                                                            \
    if (c->debug) fprintf(stderr, "[gpc_hip] k_row_join_fused<%d, %d>: %d workgroups per CU by the occupancy API, %ld workgroups, %d shards, %zu B of LDS\n", SPT, NT, per_cu, nwg, nsh, lds); \
    a.n_hi = npairs % nsh;                                                                                      \
    a.ps[0] = npairs / nsh + (a.n_hi ? 1 : 0);                                                                  \
    a.ps[1] = npairs / nsh;                                                                                     \
    for (int k_ = 0; k_ < 2; ++k_) {                                                                            \
      const GpcDivW dv_ = make_divw(a.ps[k_] > 1 ? a.ps[k_] : 2);                                     \
      a.ps_magic[k_] = dv_.magic;                                                                               \
      a.ps_sh[k_] = dv_.sh;                                                                                     \
    }                                                                                                           \
    hipLaunchKernelGGL((gpc::k_row_join_fused<SPT, NT, WIDE>), dim3((unsigned)nwg), dim3(NT), lds, c->stream, a); \
  } while (0)
#define LAUNCH_RJF_W(SPT, NT) do { if (wide) LAUNCH_RJF(SPT, NT, true); else LAUNCH_RJF(SPT, NT, false); } while (0)
#define LAUNCH_RJF_S(NT)                      \
  switch (jp.spt) {                           \
    case 1: LAUNCH_RJF_W(1, NT); break;       \
    case 2: LAUNCH_RJF_W(2, NT); break;       \
    default: LAUNCH_RJF_W(4, NT); break;      \
  }
      if (jp.nt == 1024) { LAUNCH_RJF_S(1024) }
      else if (jp.nt == 512) { LAUNCH_RJF_S(512) }
      else { LAUNCH_RJF_S(256) }
#undef LAUNCH_RJF_S
#undef LAUNCH_RJF_W
#undef LAUNCH_RJF
      HIPCHK(c, hipGetLastError());
      return GPC_OK;
    }
    {
      Timed t(c, KID_ROW_JOIN);
      const int rpw = 1;  // (the kernel takes one row per workgroup)
      const dim3 jgrid((H - 2 * GPC_R + rpw - 1) / rpw, npairs);
      const bool wide = wide_codes(c);
      snprintf(c->launch_name[KID_ROW_JOIN], sizeof c->launch_name[0], "gpc::k_row_join<%d, %d, %s>", jp.spt, jp.nt,
               wide ? "true" : "false");
#define LAUNCH_JOIN(SPT, NT, WIDE)                                                                            \
  do {                                                                                                        \
    const void* fn_ = reinterpret_cast<const void*>(gpc::k_row_join<SPT, NT, WIDE>);                          \
    if (jp.lds > 48 * 1024) CHK(allow_dyn_lds(c, fn_, jp.lds));                                               \
    hipLaunchKernelGGL((gpc::k_row_join<SPT, NT, WIDE>), jgrid, dim3(NT), jp.lds, c->stream,                  \
                       (const uint32_t*)c->codes.p, d_cand, W, H, disp_high, apply_filter,                    \
                       (const int32_t*)c->stats.p, (uint32_t*)c->staged.p, (int32_t*)c->rowcnt.p, jp.log2s,   \
                       rpw, gpc::RjVirt());                                                    \
  } while (0)
#define LAUNCH_JOIN_W(SPT, NT) do { if (wide) LAUNCH_JOIN(SPT, NT, true); else LAUNCH_JOIN(SPT, NT, false); } while (0)
#define LAUNCH_JOIN_S(NT)                      \
  switch (jp.spt) {                            \
    case 1: LAUNCH_JOIN_W(1, NT); break;       \
    case 2: LAUNCH_JOIN_W(2, NT); break;       \
    default: LAUNCH_JOIN_W(4, NT); break;      \
  }
      if (jp.nt == 1024) {
        if (jp.spt == 16) LAUNCH_JOIN_W(16, 1024);
        else if (jp.spt == 8) LAUNCH_JOIN_W(8, 1024);
        else LAUNCH_JOIN_S(1024)
      } else if (jp.nt == 512) { LAUNCH_JOIN_S(512) }
      else { LAUNCH_JOIN_S(256) }
#undef LAUNCH_JOIN_S
#undef LAUNCH_JOIN_W
#undef LAUNCH_JOIN
      HIPCHK(c, hipGetLastError());
    }
    {
      Timed t(c, KID_GATHER_ROWS);
      snprintf(c->launch_name[KID_GATHER_ROWS], sizeof c->launch_name[0], "gpc::k_gather_rows");
      if (po && po->totals)
        hipLaunchKernelGGL(gpc::k_pair_totals, dim3(npairs), dim3(RM_THREADS), 0, c->stream, (const int32_t*)c->rowcnt.p, H,
                           po->totals);
      const int gr = ((long)((H - 2 * GPC_R + GR_ROWS - 1) / GR_ROWS) * npairs >= 2048) ? GR_ROWS : 1;
      hipLaunchKernelGGL(gpc::k_gather_rows, dim3((H - 2 * GPC_R + gr - 1) / gr, npairs), dim3(RM_THREADS), 0, c->stream,
                         (const uint32_t*)c->staged.p, (const int32_t*)c->rowcnt.p, W, H, mode, d_out,
                         cap, d_counts, (const int32_t*)c->stats.p, d_ncand, gr, po ? po->rows : nullptr,
                         po ? po->packed_stride : 0l, po ? po->rows_stride : 0l,
                         po ? (const int32_t*)po->totals : (const int32_t*)nullptr);
      HIPCHK(c, hipGetLastError());
    }
    return GPC_OK;
  }
  return run_global_match(c, W, H, npairs, s, mode, d_cand, d_out, cap, d_counts, d_ncand);
}

// Shared set-up of the two device-wide-sort matchers (k_global.h, k_hashtable.h); every launch
// covers the whole batch (pair = blockIdx.y / .z).
struct GlobalPlan {
  int nmax, nblk, nmblk;
  gpc::GpcBatchStrides bs;
  int32_t *hist, *blkcnt, *gmisc, *rowcnt;
  size_t esz;
};

int plan_global(gpc_hip_ctx* c, int W, int H, int npairs, int mode, int cap, bool hashtable, GlobalPlan& g) {
  const size_t n = (size_t)W * H;
  g.nmax = 2 * (W - 2 * GPC_R) * (H - 2 * GPC_R);
  g.nblk = (g.nmax + GS_TILE - 1) / GS_TILE;
  g.nmblk = (g.nmax + RM_THREADS - 1) / RM_THREADS;
  g.esz = mode == 0 ? sizeof(gpc_support) : sizeof(gpc_correspondence);
  const size_t recs = (size_t)g.nmax * npairs;
  for (int i = 0; i < 2; ++i) {
    CHK(ensure(c, c->gkeys[i], sizeof(uint32_t) * recs));
    CHK(ensure(c, c->gvals[i], sizeof(uint32_t) * recs));
    if (hashtable) {
      CHK(ensure(c, c->hkeys[i], sizeof(uint32_t) * recs));
      CHK(ensure(c, c->hvals[i], sizeof(uint32_t) * recs));
      CHK(ensure(c, c->hrec, 2 * sizeof(uint32_t) * recs));
    }
  }
  CHK(ensure(c, c->ghist, sizeof(int32_t) * ((size_t)256 * g.nblk + g.nmblk) * npairs));
  CHK(ensure(c, c->gmisc, sizeof(int32_t) * GM_STRIDE * npairs));
  CHK(ensure(c, c->rowcnt, sizeof(int32_t) * (size_t)H * 2 * npairs));
  g.hist = (int32_t*)c->ghist.p;
  g.blkcnt = g.hist + (size_t)256 * g.nblk * npairs;
  g.gmisc = (int32_t*)c->gmisc.p;
  g.rowcnt = (int32_t*)c->rowcnt.p;
  g.bs.codes = (long)(2 * n);
  g.bs.recs = g.nmax;
  g.bs.hist = (long)256 * g.nblk;
  g.bs.blk = g.nmblk;
  g.bs.out = (long)cap * (long)g.esz;
  g.bs.rows = 2 * H;
  return GPC_OK;
}

// stable LSD radix sort of (keys, vals) by `passes` 8-bit digits; result in buffer index passes & 1
int radix_passes(gpc_hip_ctx* c, const GlobalPlan& g, int npairs, uint32_t* keys[2], uint32_t* vals[2], int passes) {
  for (int pass = 0; pass < passes; ++pass) {
    const int src = pass & 1, dst = src ^ 1;
    hipLaunchKernelGGL(gpc::k_g_hist, dim3(g.nblk, npairs), dim3(GS_THREADS), 0, c->stream,
                       (const uint32_t*)keys[src], (const int32_t*)g.gmisc, 8 * pass, g.hist, g.nblk, g.bs);
    hipLaunchKernelGGL(gpc::k_g_scan, dim3(1, npairs), dim3(1024), 0, c->stream, g.hist, 256 * g.nblk, g.bs.hist);
    hipLaunchKernelGGL(gpc::k_g_scatter, dim3(g.nblk, npairs), dim3(GS_THREADS), 0, c->stream,
                       (const uint32_t*)keys[src], (const uint32_t*)vals[src], keys[dst], vals[dst],
                       (const int32_t*)g.gmisc, 8 * pass, (const int32_t*)g.hist, g.nblk, g.bs);
  }
  HIPCHK(c, hipGetLastError());
  return GPC_OK;
}

// Non-epipolar mode: one device-wide stable radix sort per pair (k_global.h).
// Non-epipolar matcher by partition + LDS join (k_partition.h).  *done = false when some partition is too large for one
// workgroup (heavily duplicated codes): nothing has been written then and the caller takes the radix-sort path.
// The overflow word is read back, so this mode synchronises the stream once per call.
int run_partition_match(gpc_hip_ctx* c, const GlobalPlan& g, int W, int H, int npairs, const gpc_settings* s, int mode,
                        const uint8_t* d_cand, void* d_out, int cap, int32_t* d_counts, int32_t* d_ncand, bool* done) {
  *done = false;
  const bool wide = wide_codes(c);
  const int bits = wide ? 32 : code_bits(c);
  gpc::GpLayout L = {};
  // 256 bins (8 top code bits) up to ~1 M record slots per pair; 512 / 1024 for larger images, so that a bin of a textured
  // image still holds about a partition's worth of records (the scatter's runs get shorter: k_partition.h)
  // ... and 2048 beyond 1920x1080
  int lb = 8;
  while (lb < 11 && (double)g.nmax * 0.7 / (double)(1 << lb) > 2800.0) ++lb;
  if (c->gp_log2bins > 0) lb = c->gp_log2bins;
  if (lb > bits) lb = bits;
  L.nbins = 1 << lb;
  L.bshift = bits - lb;
  L.target = c->gp_target;  // records per side a partition aims at: half of what k_row_join<4, 1024> holds (skewed bins, zero-code rows)
  L.cap = GP_NB;
  L.cap_hard = 2 * GP_NB;  // a bin that is large by itself (skewed top code bits) goes to the 8192-record join
  // cuts happen where a running count <= nmax passes a multiple of the target, and around bins of more than
  // GP_NB - target records (k_gp_plan); the join's grid is what the plan really made (read back with the overflow word)
  L.pmax = g.nmax / L.target + 2 * (g.nmax / (GP_NB - L.target)) + 2;
  // ... and a partition is a run of whole bins, so there are never more partitions than bins: without this bound the
  // plan kernel's LDS (8 bytes per possible partition) outgrew a workgroup's 160 KiB from ~6.5 M pixels on and the
  // launch failed where the radix path would have served (3840x2160: 189 KB)
  if (L.pmax > L.nbins) L.pmax = L.nbins;
  const int rows = H - 2 * GPC_R;
  L.rows_per_chunk = rows >= 64 ? c->rows_per_chunk : (rows + 3) / 4;  // >= 4 chunks per image
  L.nchunk = (rows + L.rows_per_chunk - 1) / L.rows_per_chunk;
  L.o_off = 0;
  L.o_rowcnt = L.o_off + 2 * (L.pmax + 1);
  L.o_misc = L.o_rowcnt + L.pmax;
  L.ps = L.o_misc + 8 + GP_BIGCAP;
  const size_t tab_ints = (size_t)2 * npairs * L.nbins * L.nchunk;
  const size_t plan_bytes = sizeof(int32_t) * ((size_t)L.ps * npairs + 4);
  CHK(ensure(c, c->gpart, sizeof(int32_t) * tab_ints + plan_bytes));
  CHK(ensure(c, c->staged, sizeof(uint2) * (size_t)(g.nmax / 2) * npairs));  // (left, right) pixel index per match
  if (!c->h_flag) HIPCHK(c, hipHostMalloc((void**)&c->h_flag, 64, hipHostMallocDefault));
  int32_t* tabs = (int32_t*)c->gpart.p;
  int32_t* part = tabs + tab_ints;
  int32_t* d_flag = part + (size_t)L.ps * npairs;
  const uint32_t* codes = (const uint32_t*)c->codes.p;
  const uint8_t* wcand = wide ? d_cand : nullptr;
  CHK(ensure(c, c->gkv, sizeof(uint2) * (size_t)g.bs.recs * npairs));
  uint2* kv = (uint2*)c->gkv.p;
  dim3 cgrid(L.nchunk, 2, npairs);
  const size_t plan_lds = sizeof(int32_t) * 2 * ((size_t)L.pmax + 1);
  {
    Timed t(c, KID_GLOBAL_KEYS);
    HIPCHK(c, hipMemsetAsync(part, 0, plan_bytes, c->stream));
    if (L.nbins > 1024)
      hipLaunchKernelGGL((gpc::k_gp_hist<false, 2048>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H, g.bs.codes,
                         tabs, L, make_divw(W));
    else if (L.nbins > 256)
      hipLaunchKernelGGL((gpc::k_gp_hist<false, 1024>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H, g.bs.codes,
                         tabs, L, make_divw(W));
    else
      hipLaunchKernelGGL((gpc::k_gp_hist<false, 256>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H, g.bs.codes,
                         tabs, L, make_divw(W));
    hipLaunchKernelGGL(gpc::k_g_scan, dim3(1, 2 * npairs), dim3(1024), 0, c->stream, tabs, L.nbins * L.nchunk,
                       (long)L.nbins * L.nchunk);
    CHK(allow_dyn_lds(c, reinterpret_cast<const void*>(gpc::k_gp_plan), plan_lds));
    hipLaunchKernelGGL(gpc::k_gp_plan, dim3(npairs), dim3(GP_THREADS), plan_lds, c->stream, (const int32_t*)tabs,
                       (const int32_t*)c->stats.p, part, L, d_flag);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipMemcpyAsync(c->h_flag, d_flag, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  // (the scatter goes out before the host looks at the plan's words, as in the hash-table matcher below: the device works
  // through the host's round trip; a batch that is sorted instead has scattered once for nothing)
  CHK(ensure_flag_event(c));
  HIPCHK(c, hipEventRecord(c->e_flag, c->stream));
  {
    Timed t(c, KID_GLOBAL_SORT);
    if (L.nbins > 1024)
      hipLaunchKernelGGL((gpc::k_gp_scatter<false, 2048>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H,
                         g.bs.codes, (const int32_t*)tabs, L, make_divw(W), kv, g.bs.recs);
    else if (L.nbins > 256)
      hipLaunchKernelGGL((gpc::k_gp_scatter<false, 1024>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H,
                         g.bs.codes, (const int32_t*)tabs, L, make_divw(W), kv, g.bs.recs);
    else
      hipLaunchKernelGGL((gpc::k_gp_scatter<false, 256>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H,
                         g.bs.codes, (const int32_t*)tabs, L, make_divw(W), kv, g.bs.recs);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipEventSynchronize(c->e_flag));
  if (c->h_flag[0]) return GPC_OK;  // a single bin beyond 8192 records: the caller sorts instead
  const bool big_bins = c->h_flag[2] > GP_NB;  // some partitions (single bins) need the 8192-record join
  const int maxparts = c->h_flag[1] > 0 ? c->h_flag[1] : 1;
  {
    Timed t(c, KID_GLOBAL_MATCH);
    gpc::RjVirt v;
    v.kv = kv;
    v.part = part;
    v.staged = (uint32_t*)c->staged.p;
    v.recs = g.bs.recs;
    v.ps = L.ps;
    v.o_off = L.o_off;
    v.o_rowcnt = L.o_rowcnt;
    v.o_misc = L.o_misc;
    v.pmax = L.pmax;
    v.dw = make_divw(W);
    v.vtol = s->vertical_tolerance;
    v.min_recs = -1;
    v.use_list = 0;
    // 8192 slots for up to 4096 left records: 64 KiB, two workgroups = 32 waves per CU; 16384 for up to 8192: one workgroup.
    // Every partition goes to the 4096-record launch except the bins that are larger by themselves: those few (19 of
    // 1024 code ranges of a 1920x1080 image with the Tau forest) get a launch of the 8192-record instantiation, in which
    // all other workgroups return at once.  (Planning the whole batch for the larger kernel, as before, ran every
    // partition at one workgroup per CU.)
    int log2s = 13;
    size_t lds = ((size_t)8 * ((1u << log2s) + 1) + 15) / 16 * 16;
    const int apply_filter = (mode == 0);
    const dim3 jgrid(maxparts, npairs);
    dim3 jgrid_(jgrid);
#define LAUNCH_VJOIN(SPT, WIDE, STREAM)                                                                                 \
  do {                                                                                                                  \
    const void* fn_ = reinterpret_cast<const void*>(gpc::k_row_join<SPT, 1024, WIDE, true>);                            \
    CHK(allow_dyn_lds(c, fn_, lds));                                                                                    \
    hipLaunchKernelGGL((gpc::k_row_join<SPT, 1024, WIDE, true>), jgrid_, dim3(1024), lds, STREAM, (const uint32_t*)nullptr, \
                       (const uint8_t*)nullptr, W, H, s->disp_high, apply_filter, (const int32_t*)nullptr,              \
                       (uint32_t*)nullptr, (int32_t*)nullptr, log2s, 1, v);                                             \
  } while (0)
    if (big_bins) {
      // The few over-large partitions first, on a stream of their own: their 8192-record workgroups (one per CU, 44-48 us
      // per 8 pairs of 1920x1080 as a launch by itself) run beside the 4096-record launch instead of after it.
      CHK(ensure_aux_stream(c));
      HIPCHK(c, hipEventRecord(c->e_fork, c->stream));
      HIPCHK(c, hipStreamWaitEvent(c->s_aux, c->e_fork, 0));
      gpc::RjVirt v4 = v;
      v.min_recs = GP_NB;
      if (c->h_flag[3] <= GP_BIGCAP) {  // the plan's work list is the grid
        v.use_list = 1;
        jgrid_ = dim3(c->h_flag[3] > 0 ? c->h_flag[3] : 1, npairs);
      }
      log2s = 14;
      lds = ((size_t)8 * ((1u << log2s) + 1) + 15) / 16 * 16;
      if (wide) LAUNCH_VJOIN(8, true, c->s_aux); else LAUNCH_VJOIN(8, false, c->s_aux);
      HIPCHK(c, hipEventRecord(c->e_join, c->s_aux));
      v = v4;
      jgrid_ = jgrid;
      log2s = 13;
      lds = ((size_t)8 * ((1u << log2s) + 1) + 15) / 16 * 16;
    }
    if (wide) LAUNCH_VJOIN(4, true, c->stream); else LAUNCH_VJOIN(4, false, c->stream);
    if (big_bins) HIPCHK(c, hipStreamWaitEvent(c->stream, c->e_join, 0));
#undef LAUNCH_VJOIN
    hipLaunchKernelGGL(gpc::k_gp_gather, dim3((maxparts + GPG_PARTS - 1) / GPG_PARTS, npairs), dim3(RM_THREADS), 0, c->stream,
                       (const uint32_t*)c->staged.p, (const int32_t*)part, L, g.bs.recs, make_divw(W),
                       mode, d_out, g.bs.out, cap, d_counts, (const int32_t*)c->stats.p, d_ncand);
    HIPCHK(c, hipGetLastError());
  }
  *done = true;
  return GPC_OK;
}

int run_global_match(gpc_hip_ctx* c, int W, int H, int npairs, const gpc_settings* s, int mode, const uint8_t* d_cand,
                     void* d_out, int cap, int32_t* d_counts, int32_t* d_ncand) {
  GlobalPlan g;
  CHK(plan_global(c, W, H, npairs, mode, cap, false, g));
  if (!c->no_partition) {
    bool done = false;
    CHK(run_partition_match(c, g, W, H, npairs, s, mode, d_cand, d_out, cap, d_counts, d_ncand, &done));
    if (done) return GPC_OK;
  }
  const int apply_filter = (mode == 0);
  const uint32_t* codes = (const uint32_t*)c->codes.p;
  const uint8_t* wcand = wide_codes(c) ? d_cand : nullptr;  // 32-bit codes: 0xFFFFFFFF is told from the sentinel by the candidate byte
  const int32_t* stats = (const int32_t*)c->stats.p;
  uint32_t* keys[2] = {(uint32_t*)c->gkeys[0].p, (uint32_t*)c->gkeys[1].p};
  uint32_t* vals[2] = {(uint32_t*)c->gvals[0].p, (uint32_t*)c->gvals[1].p};
  dim3 rgrid(H - 2 * GPC_R, 2, npairs);
  {
    Timed t(c, KID_GLOBAL_KEYS);
    hipLaunchKernelGGL(gpc::k_g_rowcount, rgrid, dim3(RM_THREADS), 0, c->stream, codes, wcand, W, H, g.rowcnt, stats,
                       g.gmisc, g.bs);
    hipLaunchKernelGGL(gpc::k_g_build, rgrid, dim3(RM_THREADS), 0, c->stream, codes, wcand, W, H,
                       (const int32_t*)g.rowcnt, stats, keys[0], vals[0], g.gmisc, g.bs);
    HIPCHK(c, hipGetLastError());
  }
  {
    Timed t(c, KID_GLOBAL_SORT);
    CHK(radix_passes(c, g, npairs, keys, vals, 4));  // all 32 code bits
  }
  {
    Timed t(c, KID_GLOBAL_MATCH);
    const int ngm = (g.nmax + GMT_TILE - 1) / GMT_TILE;  // <= nmblk, so the match-block counters are large enough
    hipLaunchKernelGGL((gpc::k_g_match<false>), dim3(ngm, npairs), dim3(RM_THREADS), 0, c->stream,
                       (const uint32_t*)keys[0], (const uint32_t*)vals[0], (const int32_t*)g.gmisc, make_divw(W),
                       s->disp_high, s->vertical_tolerance, apply_filter, g.blkcnt, mode, (void*)nullptr, 0,
                       (int32_t*)nullptr, stats, (int32_t*)nullptr, g.bs);
    hipLaunchKernelGGL(gpc::k_g_scan, dim3(1, npairs), dim3(1024), 0, c->stream, g.blkcnt, ngm, (long)g.bs.blk);
    hipLaunchKernelGGL((gpc::k_g_match<true>), dim3(ngm, npairs), dim3(RM_THREADS), 0, c->stream,
                       (const uint32_t*)keys[0], (const uint32_t*)vals[0], (const int32_t*)g.gmisc, make_divw(W),
                       s->disp_high, s->vertical_tolerance, apply_filter, g.blkcnt, mode, d_out, cap, d_counts, stats,
                       d_ncand, g.bs);
    HIPCHK(c, hipGetLastError());
  }
  return GPC_OK;
}

// useHashtable mode by partition into bins of 1024 buckets + one workgroup per bin (k_htjoin.h).  *done = false when a bin
// holds more records than one workgroup takes (the overflow word, read back after the histogram: one stream
// synchronisation per call); nothing has been written then and the caller takes the radix path.
int run_hashtable_partition(gpc_hip_ctx* c, const GlobalPlan& g, int W, int H, int npairs, const gpc_settings* s, int mode,
                            const uint8_t* d_cand, void* d_out, int cap, int32_t* d_counts, int32_t* d_ncand, bool* done) {
  *done = false;
  if (H >= HTJ_MAXH) return GPC_OK;  // positions are packed y << 14 | x
  gpc::GpLayout L = {};
  // Bins of 1024 buckets (210 of them) up to ~1 M record slots per pair, 512 / 256 / 128 buckets per bin for larger
  // images: Hashmatch's buckets are `state % 214673`, so bins fill evenly and the estimate (70 % of the pixels are
  // candidates) may aim at 3500 of the 4096 records a workgroup takes.  The histogram then tells the truth: k_ht_check
  // reports the batch's largest bin, and a bin that does not fit sends the planning round again with as many fewer
  // buckets per bin as that takes (one more histogram, ~0.1 ms per 8 pairs of 1920x1080) instead of the whole batch to the
  // radix path; only a bin that stays too large at 128 buckets -- one state repeated thousands of times (striped images)
  // or more than ~6 M candidates -- goes on to the 8192-record kernel (one workgroup per CU) or, beyond that, the radix path.
  int lbits = HTJ_LBITS;
  while (lbits > 7 && (double)g.nmax * 0.7 / (double)((HM_BUCKETS >> lbits) + 1) > 3500.0) --lbits;
  if (c->ht_hint_w == W && c->ht_hint_h == H && c->ht_hint_lbits > 0 && c->ht_hint_lbits < lbits) lbits = c->ht_hint_lbits;
  if (c->ht_lbits > 0) lbits = c->ht_lbits;
  int rpt = 4;
  L.epi = s->epipolar_mode ? 1 : 0;
  const int rows = H - 2 * GPC_R;
  L.rows_per_chunk = rows >= 64 ? c->rows_per_chunk : (rows + 3) / 4;
  L.nchunk = (rows + L.rows_per_chunk - 1) / L.rows_per_chunk;
  CHK(ensure(c, c->staged, sizeof(uint2) * (size_t)(g.nmax / 2) * npairs));
  if (!c->h_flag) HIPCHK(c, hipHostMalloc((void**)&c->h_flag, 64, hipHostMallocDefault));
  const uint32_t* codes = (const uint32_t*)c->codes.p;
  const uint8_t* wcand = wide_codes(c) ? d_cand : nullptr;
  CHK(ensure(c, c->gkv, sizeof(uint2) * (size_t)g.bs.recs * npairs));
  uint2* kv = (uint2*)c->gkv.p;
  dim3 cgrid(L.nchunk, 2, npairs);
  int32_t *tabs = nullptr, *bincnt = nullptr, *biglist = nullptr;
  for (int attempt = 0;; ++attempt) {
    L.bshift = lbits;
    L.nbins = (int)((HM_BUCKETS + (1u << lbits) - 1) >> lbits);  // 210 / 420 / 839 / 1678
    const size_t tab_ints = (size_t)2 * npairs * L.nbins * L.nchunk;
    const size_t cnt_ints = (size_t)npairs * L.nbins;
    const size_t big_ints = (size_t)npairs * (HTJ_BIGCAP + 1);
    CHK(ensure(c, c->gpart, sizeof(int32_t) * (tab_ints + cnt_ints + 4 + big_ints)));
    tabs = (int32_t*)c->gpart.p;
    bincnt = tabs + tab_ints;
    int32_t* d_flag = bincnt + cnt_ints;
    biglist = d_flag + 4;
    {
      Timed t(c, KID_GLOBAL_KEYS);
      HIPCHK(c, hipMemsetAsync(d_flag, 0, sizeof(int32_t) * (4 + big_ints), c->stream));
      if (L.nbins > 1024)
        hipLaunchKernelGGL((gpc::k_gp_hist<true, 2048>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H, g.bs.codes,
                           tabs, L, make_divw(W));
      else if (L.nbins > 256)
        hipLaunchKernelGGL((gpc::k_gp_hist<true, 1024>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H, g.bs.codes,
                           tabs, L, make_divw(W));
      else
        hipLaunchKernelGGL((gpc::k_gp_hist<true, 256>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H, g.bs.codes,
                           tabs, L, make_divw(W));
      hipLaunchKernelGGL(gpc::k_g_scan, dim3(1, 2 * npairs), dim3(1024), 0, c->stream, tabs, L.nbins * L.nchunk,
                         (long)L.nbins * L.nchunk);
      hipLaunchKernelGGL(gpc::k_ht_check, dim3(npairs), dim3(256), 0, c->stream, (const int32_t*)tabs,
                         (const int32_t*)c->stats.p, L.nbins, L.nchunk, HTJ_THREADS * 4, d_flag, 512 * 4, biglist);
      HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipMemcpyAsync(c->h_flag, d_flag, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    // The scatter goes out BEFORE the host looks at the plan's words (an event marks them): it does not depend on them --
    // only whether its result is used does -- and the device then works through the host's round trip instead of idling
    // (~15 us per call; a batch that has to be planned again, or sorted instead, has scattered once for nothing).
    CHK(ensure_flag_event(c));
    HIPCHK(c, hipEventRecord(c->e_flag, c->stream));
  {
    Timed t(c, KID_GLOBAL_SORT);
    if (L.nbins > 1024)
      hipLaunchKernelGGL((gpc::k_gp_scatter<true, 2048>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H,
                         g.bs.codes, (const int32_t*)tabs, L, make_divw(W), kv, g.bs.recs);
    else if (L.nbins > 256)
      hipLaunchKernelGGL((gpc::k_gp_scatter<true, 1024>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H,
                         g.bs.codes, (const int32_t*)tabs, L, make_divw(W), kv, g.bs.recs);
    else
      hipLaunchKernelGGL((gpc::k_gp_scatter<true, 256>), cgrid, dim3(GPS_THREADS), 0, c->stream, codes, wcand, W, H,
                         g.bs.codes, (const int32_t*)tabs, L, make_divw(W), kv, g.bs.recs);
    HIPCHK(c, hipGetLastError());
  }
    HIPCHK(c, hipEventSynchronize(c->e_flag));
    if (!c->h_flag[0]) {  // every bin fits the 4096-record kernel
      c->ht_hint_w = W;
      c->ht_hint_h = H;
      c->ht_hint_lbits = lbits;
      break;
    }
    const int maxbin = c->h_flag[1];
    // halving the buckets per bin halves an evenly filled bin: how many halvings bring the largest one under 3900?
    int want = lbits;
    while (want > 7 && (maxbin >> (lbits - want)) > 3900) --want;
    if (want < lbits && c->ht_lbits == 0 && attempt < 3) {
      lbits = want;
      continue;
    }
    if (maxbin <= HTJ_THREADS * 8) {  // the 8192-record kernel takes what is left (one workgroup per CU)
      rpt = 8;
      break;
    }
    return GPC_OK;  // the caller sorts instead
  }
  {
    Timed t(c, KID_GLOBAL_MATCH);
    gpc::HtjArgs a;
    a.kv = kv;
    a.tabs = tabs;
    a.stats = (const int32_t*)c->stats.p;
    a.staged = (uint2*)c->staged.p;
    a.bincnt = bincnt;
    a.recs = g.bs.recs;
    a.nbins = L.nbins;
    a.nchunk = L.nchunk;
    a.lbits = lbits;
    // Ten-record lists that fill up (k_htjoin.h): measured per-record handling against one wave per bucket --
    // 1920x1080 x8 (13 records per bucket): k_ht_join 787 -> 527 us; 1024x436 x32 (2.7 per bucket): 239 -> 278 us.  The
    // batch's largest record count (k_ht_check reports it with the largest bin) decides: from 8 records per bucket on.
    a.mid = c->ht_mid > 0 ? c->ht_mid : ((double)c->h_flag[2] / (double)HM_BUCKETS >= 8.0 ? 32 : HM_CAP);
    a.epi = L.epi;
    a.disp_high = s->disp_high;
    a.vtol = s->vertical_tolerance;
    a.apply_filter = (mode == 0);
    a.dw = make_divw(W);
    // 512 threads where the bins are small (k_htjoin.h): at most 512 buckets, and bins of at most 2048 records; the few
    // larger ones (a bucket of repeated states can double a bin) go to a 1024-thread launch on a stream of its own beside it
    const bool half = rpt == 4 && lbits <= 9 && !c->ht_no_half &&
                      (double)c->h_flag[2] / (double)L.nbins <= 0.9 * 512 * 4;  // the average bin fits comfortably
    const bool split = half && c->h_flag[1] > 512 * 4;
    a.min_recs = -1;
    a.use_list = 0;
    a.biglist = biglist;
    if (c->debug_plan)
      fprintf(stderr, "[ht plan] lbits %d bins %d largest bin %d largest pair %d rpt %d half %d mid %d\n", lbits, L.nbins, c->h_flag[1],
              c->h_flag[2], rpt, (int)half, a.mid);
    dim3 hgrid(L.nbins, npairs);
#define LAUNCH_HTJ(RPT, NT, STREAM)                                                                                      \
  do {                                                                                                                   \
    const size_t lds_ = (size_t)8 * NT * RPT;                                                                            \
    CHK(allow_dyn_lds(c, reinterpret_cast<const void*>(gpc::k_ht_join<RPT, NT>), lds_));                                 \
    hipLaunchKernelGGL((gpc::k_ht_join<RPT, NT>), hgrid, dim3(NT), lds_, STREAM, a);                                     \
  } while (0)
    // (every bin is taken by exactly one launch: k_ht_check has seen that none exceeds the largest kernel's capacity)
    if (rpt == 8) {
      LAUNCH_HTJ(8, HTJ_THREADS, c->stream);
    } else if (half) {
      if (split) {
        CHK(ensure_aux_stream(c));
        HIPCHK(c, hipEventRecord(c->e_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->s_aux, c->e_fork, 0));
        a.min_recs = 512 * 4;
        if (c->h_flag[3] <= HTJ_BIGCAP) {  // the larger bins' list is the grid (else: every bin, nearly all returning at once)
          a.use_list = 1;
          hgrid = dim3(c->h_flag[3] > 0 ? c->h_flag[3] : 1, npairs);
        }
        LAUNCH_HTJ(4, HTJ_THREADS, c->s_aux);
        HIPCHK(c, hipEventRecord(c->e_join, c->s_aux));
        a.min_recs = -1;
        a.use_list = 0;
        hgrid = dim3(L.nbins, npairs);
      }
      LAUNCH_HTJ(4, 512, c->stream);
      if (split) HIPCHK(c, hipStreamWaitEvent(c->stream, c->e_join, 0));
    } else {
      LAUNCH_HTJ(4, HTJ_THREADS, c->stream);
    }
#undef LAUNCH_HTJ
    hipLaunchKernelGGL(gpc::k_ht_gather, dim3((L.nbins + HTG_BINS - 1) / HTG_BINS, npairs), dim3(RM_THREADS), 0, c->stream, a,
                       mode, d_out, g.bs.out, cap, d_counts, d_ncand);
    HIPCHK(c, hipGetLastError());
  }
  *done = true;
  return GPC_OK;
}

// useHashtable mode (hashmatch.hpp): stable radix sort by bucket id + one thread per bucket (k_hashtable.h)
int run_hashtable_match(gpc_hip_ctx* c, int W, int H, int npairs, const gpc_settings* s, int mode, const uint8_t* d_cand,
                        void* d_out, int cap, int32_t* d_counts, int32_t* d_ncand) {
  GlobalPlan g;
  CHK(plan_global(c, W, H, npairs, mode, cap, true, g));
  if (!c->no_partition) {
    bool done = false;
    CHK(run_hashtable_partition(c, g, W, H, npairs, s, mode, d_cand, d_out, cap, d_counts, d_ncand, &done));
    if (done) return GPC_OK;
  }
  const int apply_filter = (mode == 0);
  const int epi = s->epipolar_mode ? 1 : 0;
  const uint32_t* codes = (const uint32_t*)c->codes.p;
  const uint8_t* wcand = wide_codes(c) ? d_cand : nullptr;
  const int32_t* stats = (const int32_t*)c->stats.p;
  uint32_t* codes0 = (uint32_t*)c->gkeys[0].p;
  uint32_t* kv0 = (uint32_t*)c->gvals[0].p;
  uint32_t* keys[2] = {(uint32_t*)c->hkeys[0].p, (uint32_t*)c->hkeys[1].p};
  uint32_t* vals[2] = {(uint32_t*)c->hvals[0].p, (uint32_t*)c->hvals[1].p};
  dim3 rgrid(H - 2 * GPC_R, 2, npairs);
  {
    Timed t(c, KID_GLOBAL_KEYS);
    hipLaunchKernelGGL(gpc::k_g_rowcount, rgrid, dim3(RM_THREADS), 0, c->stream, codes, wcand, W, H, g.rowcnt, stats,
                       g.gmisc, g.bs);
    hipLaunchKernelGGL(gpc::k_g_build, rgrid, dim3(RM_THREADS), 0, c->stream, codes, wcand, W, H,
                       (const int32_t*)g.rowcnt, stats, codes0, kv0, g.gmisc, g.bs);
    hipLaunchKernelGGL(gpc::k_ht_bucket_ids, dim3(g.nmblk, npairs), dim3(256), 0, c->stream,
                       (const uint32_t*)codes0, (const uint32_t*)kv0, (const int32_t*)g.gmisc, make_divw(W), epi, keys[0],
                       vals[0], (uint2*)c->hrec.p, g.bs);
    HIPCHK(c, hipGetLastError());
  }
  {
    Timed t(c, KID_GLOBAL_SORT);
    CHK(radix_passes(c, g, npairs, keys, vals, 3));  // 214673 < 2^18: three 8-bit digits -> result in [1]
  }
  {
    Timed t(c, KID_GLOBAL_MATCH);
    const int nhp = (g.nmax + HP_TILE - 1) / HP_TILE;  // <= nmblk, so the match-block counters are large enough
    hipLaunchKernelGGL((gpc::k_ht_pairs<false>), dim3(nhp, npairs), dim3(HP_THREADS), 0, c->stream,
                       (const uint32_t*)keys[1], (const uint32_t*)vals[1], (const uint2*)c->hrec.p, keys[0],
                       vals[0], (const int32_t*)g.gmisc, make_divw(W), epi, s->disp_high, s->vertical_tolerance,
                       apply_filter, g.blkcnt, mode, (void*)nullptr, 0, (int32_t*)nullptr, stats, (int32_t*)nullptr,
                       g.bs);
    hipLaunchKernelGGL(gpc::k_g_scan, dim3(1, npairs), dim3(1024), 0, c->stream, g.blkcnt, nhp, (long)g.bs.blk);
    hipLaunchKernelGGL((gpc::k_ht_pairs<true>), dim3(nhp, npairs), dim3(HP_THREADS), 0, c->stream,
                       (const uint32_t*)keys[1], (const uint32_t*)vals[1], (const uint2*)c->hrec.p, keys[0],
                       vals[0], (const int32_t*)g.gmisc, make_divw(W), epi, s->disp_high, s->vertical_tolerance,
                       apply_filter, g.blkcnt, mode, d_out, cap, d_counts, stats, d_ncand, g.bs);
    HIPCHK(c, hipGetLastError());
  }
  return GPC_OK;
}

int check_settings(const gpc_settings* s) {
  if (!s) return GPC_E_INVALID;
  if (s->gradient_threshold < 0 || s->gradient_threshold > 255) return GPC_E_INVALID;
  return GPC_OK;
}

int forest_matches(gpc_hip_ctx* c, int W, int H) {
  if (!c->have_forest) return GPC_E_NO_FOREST;
  // the reference asserts FilterMask dims == image dims (inference.hpp:350-353)
  if (c->forest_w != W || c->forest_h != H) return GPC_E_INVALID;
  return GPC_OK;
}

// The device's address of page-locked host memory the GPU can write (hipHostMalloc / gpc_hip_host_alloc), or null.
void* device_view_of_host(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory: not an error of this library
    return nullptr;
  }
  if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
  return a.devicePointer;
}

// ---------------------------------------------------------------- device-resident preprocessed images

uint64_t fp_mix(uint64_t h, uint64_t v) {
  h ^= v;
  h *= 0x9E3779B97F4A7C15ull;
  return h ^ (h >> 29);
}

// words of `bytes` bytes at p: all of them (full), or 66 spread evenly, first and last included
uint64_t fp_array(uint64_t h, const void* p, size_t bytes, bool full) {
  const uint8_t* b = static_cast<const uint8_t*>(p);
  h = fp_mix(h, (uint64_t)bytes);
  if (bytes < 8) {
    for (size_t i = 0; i < bytes; ++i) h = fp_mix(h, b[i]);
    return h;
  }
  const size_t words = bytes / 8;
  auto word = [&](size_t byte_off) {
    uint64_t v;
    memcpy(&v, b + byte_off, 8);
    return v;
  };
  if (full) {
    uint64_t a0 = h, a1 = ~h, a2 = h * 3, a3 = h * 5;  // four chains: the multiply's latency is hidden
    size_t i = 0;
    for (; i + 4 <= words; i += 4) {
      a0 = fp_mix(a0, word(8 * i));
      a1 = fp_mix(a1, word(8 * i + 8));
      a2 = fp_mix(a2, word(8 * i + 16));
      a3 = fp_mix(a3, word(8 * i + 24));
    }
    for (; i < words; ++i) a0 = fp_mix(a0, word(8 * i));
    h = fp_mix(fp_mix(fp_mix(a0, a1), a2), a3);
  } else {
    const size_t samples = words < 65 ? words : 65;
    for (size_t k = 0; k < samples; ++k) h = fp_mix(h, word(8 * (samples > 1 ? k * (words - 1) / (samples - 1) : 0)));
  }
  return fp_mix(h, word(bytes - 8));  // the tail (bytes need not be a multiple of 8)
}

uint64_t fingerprint(const uint8_t* smooth, const uint8_t* grad, size_t n, const int32_t* mask, int n_mask, bool full) {
  uint64_t h = 0x6A09E667F3BCC908ull;
  h = fp_array(h, smooth, n, full);
  h = fp_array(h, grad, n, full);
  return fp_array(h, mask, sizeof(int32_t) * (size_t)n_mask, full);
}

bool ranges_overlap(const void* a, size_t na, const void* b, size_t nb) {
  const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
  return a && b && na && nb && x < y + nb && y < x + na;
}

// Host memory [p, p + bytes) is about to be (or has been) written by this library: records of ANY context that describe it
// are void.  Call with g_res_mu held.
void drop_overlapping(const void* p, size_t bytes) {
  for (gpc_hip_ctx* o : g_ctxs)
    for (auto& r : o->res) {
      if (!r.valid) continue;
      const size_t n = (size_t)r.W * r.H;
      if (ranges_overlap(p, bytes, r.smooth, n) || ranges_overlap(p, bytes, r.grad, n) ||
          ranges_overlap(p, bytes, r.mask, sizeof(int32_t) * (size_t)r.n_mask))
        r.valid = false;
    }
}

// The slot whose host copies these arrays are, or -1: same addresses, sizes and arithmetic, and the arrays still hold
// what was delivered (fingerprint).
int resident_slot(gpc_hip_ctx* c, const uint8_t* smooth, const uint8_t* grad, const int32_t* mask, int n_mask, int W, int H) {
  if (!c->resident_mode) return -1;
  std::lock_guard<std::mutex> g(g_res_mu);
  for (int k = 0; k < 2; ++k) {
    const gpc_hip_ctx::Resident& r = c->res[k];
    if (!r.valid || r.smooth != smooth || r.grad != grad || r.mask != mask || r.n_mask != n_mask || r.W != W || r.H != H ||
        r.naive != c->naive)
      continue;
    if (fingerprint(smooth, grad, (size_t)W * H, mask, n_mask, c->resident_mode == 2) == r.fp) return k;
  }
  return -1;
}

// Copies between page-locked staging and the caller's pageable arrays, shared among the expansion workers once they
// are worth waking (>= 512 KiB in all); host_copy_wait ends the group.
void host_copy(gpc_hip_ctx* c, void* dst, const void* src, size_t bytes, bool parallel) {
  if (!bytes) return;
  if (!parallel || c->pool.size() < 2) {
    memcpy(dst, src, bytes);
    return;
  }
  const size_t step = 192 * 1024;
  for (size_t at = 0; at < bytes; at += step) {
    ExpandJob j = {};
    j.slot = 6;  // a wait slot of its own (0 .. 3: landing slots of packed results, 7: the bounce buffer)
    j.copy_src = static_cast<const uint8_t*>(src) + at;
    j.copy_dst = static_cast<uint8_t*>(dst) + at;
    j.copy_bytes = at + step <= bytes ? step : bytes - at;
    c->pool.push(j);
  }
}
void host_copy_wait(gpc_hip_ctx* c) {
  if (c->pool.size() >= 1) c->pool.wait_slot(6);  // (whoever pushed: returns at once when nothing is pending)
}

// at least `bytes` of transfer arena; *dev receives the device's view of it
int xfer_reserve(gpc_hip_ctx* c, size_t bytes, uint8_t** dev) {
  if (bytes > c->h_xfer_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->h_xfer) HIPCHK(c, hipHostFree(c->h_xfer));
    c->h_xfer = nullptr;
    c->h_xfer_cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(c, hipHostMalloc((void**)&c->h_xfer, want, hipHostMallocDefault));
    c->h_xfer_cap = want;
  }
  HIPCHK(c, hipHostGetDevicePointer((void**)dev, c->h_xfer, 0));
  return GPC_OK;
}

// dst[0 .. bytes) = src[0 .. bytes) by a kernel (device memory or device views of page-locked host memory; both 16-byte
// aligned, bytes a multiple of 16): no copy-engine submission, no runtime staging
int dev_copy16(gpc_hip_ctx* c, void* dst, const void* src, size_t bytes) {
  if (!bytes) return GPC_OK;
  if ((bytes & 15u) || (((uintptr_t)dst | (uintptr_t)src) & 15u) || bytes / 16 >= (1ull << 32)) return GPC_E_INVALID;
  const unsigned n16 = (unsigned)(bytes / 16);
  hipLaunchKernelGGL(gpc::k_upload2, dim3((n16 + 255) / 256, 1), dim3(256), 0, c->stream, (const uint4*)src, (const uint4*)src,
                     (uint4*)dst, (uint4*)dst, n16);
  HIPCHK(c, hipGetLastError());
  return GPC_OK;
}

inline size_t pad16(size_t v) { return (v + 15) & ~(size_t)15; }

// The chunk pipeline is a few hundred HIP calls and event waits per batch: driven from a CPU of the OTHER socket every
// one of them crosses the inter-socket link on its way to the GPU, and the same 256-pair call takes 7.9 ms instead of
// 5.3 (profiles/r05_c_h2h_numa.txt; which socket a process's main thread lands on is the scheduler's choice, which is
// why one leg of the round-3 and round-4 bench records was slow and the other not).  So when the calling thread runs
// off the GPU's NUMA node -- and the process may use that node's CPUs -- the pipeline runs on a thread of its own, bound
// there like the expansion workers, and the caller waits for it.  The caller's own affinity is never touched.
template <class F>
int on_gpu_node(gpc_hip_ctx* c, int npairs, F&& body) {
  bool hop = c && !c->no_feeder && c->have_node_cpus && npairs >= 4;
  if (hop) {
    // ... or is held on one or two CPUs (a pinned worker of the caller's own pool): the runtime's helper threads are
    // created by whoever makes the first call, inherit that mask and then share the CPU with the pipeline -- the same
    // call took 9.8 ms from a thread pinned to one CPU of the GPU's own node
    const int cpu = sched_getcpu();
    cpu_set_t cur;
    const bool narrow = sched_getaffinity(0, sizeof cur, &cur) == 0 && CPU_COUNT(&cur) < 3 && CPU_COUNT(&c->node_cpus) >= 4;
    hop = narrow || (cpu >= 0 && cpu < CPU_SETSIZE && !CPU_ISSET(cpu, &c->node_cpus));
  }
  if (!hop) return body();
  int st = GPC_E_HIP;
  std::thread t([&] {
    (void)sched_setaffinity(0, sizeof c->node_cpus, &c->node_cpus);
    st = body();
  });
  t.join();
  ++c->fed_calls;
  return st;
}


// Swaps a lane's workspaces, join state and stream into the context's members (and back): the pipeline stages then run on
// the lane exactly as they run on the context.
struct LaneScope {
  gpc_hip_ctx* c;
  gpc_hip_ctx::Lane& l;
  hipStream_t user;
  LaneScope(gpc_hip_ctx* ctx, gpc_hip_ctx::Lane& lane) : c(ctx), l(lane), user(ctx->stream) { swap(); c->stream = l.s; }
  ~LaneScope() { swap(); c->stream = user; }
  void swap() {
    std::swap(c->smooth, l.smooth);
    std::swap(c->grad, l.grad);
    std::swap(c->codes, l.codes);
    std::swap(c->stats, l.stats);
    std::swap(c->jstate, l.jstate);
    std::swap(c->staged, l.staged);
    std::swap(c->rowcnt, l.rowcnt);
    std::swap(c->jstate_granules, l.jstate_granules);
    std::swap(c->join_epoch, l.join_epoch);
    std::swap(c->grad_is_bits, l.grad_is_bits);
  }
};

int ensure_lanes(gpc_hip_ctx* c) {
  for (auto& l : c->lanes) {
    if (l.s) continue;
    HIPCHK(c, hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&l.e_in, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&l.e_hash, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&l.e_join, hipEventDisableTiming));
  }
  return GPC_OK;
}

// every lane's queued work is done (host wait)
int drain_lanes(gpc_hip_ctx* c) {
  for (auto& l : c->lanes)
    if (l.s && l.used) HIPCHK(c, hipStreamSynchronize(l.s));
  return GPC_OK;
}

int ensure_pool(gpc_hip_ctx* c) {
  if (c->pool.size() == 0) {
    int nt = c->expand_threads > 0 ? c->expand_threads : default_expand_threads();
    nt = nt < 1 ? 1 : (nt > 32 ? 32 : nt);
    c->pool.start(nt, c->have_node_cpus ? &c->node_cpus : nullptr, &c->node_l3);
  }
  return GPC_OK;
}

}  // namespace

namespace gpc {
// per-image statistics as k_preprocess leaves them (candidates 0, last candidate row -1, OR of the codes 0): the
// match-from-resident-images path runs k_hash without a k_preprocess before it.  One thread per image.
__global__ void k_stats_init(int32_t* __restrict__ stats, int nimg) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nimg) return;
  stats[i * GPC_STAT_STRIDE + GPC_STAT_NCAND] = 0;
  stats[i * GPC_STAT_STRIDE + GPC_STAT_LASTROW] = -1;
  stats[i * GPC_STAT_STRIDE + GPC_STAT_CODEOR] = 0;
  stats[i * GPC_STAT_STRIDE + 3] = 0;
}
}  // namespace gpc

namespace {

}  // namespace

// =================================================================== C ABI

extern "C" {

int gpc_hip_abi_version(void) { return GPC_HIP_ABI_VERSION; }

const char* gpc_hip_status_string(int status) {
  switch (status) {
    case GPC_OK: return "ok";
    case GPC_E_INVALID: return "invalid argument";
    case GPC_E_NO_DEVICE: return "no usable HIP device (gfx950 required)";
    case GPC_E_HIP: return "HIP runtime error";
    case GPC_E_CAPACITY: return "output capacity too small";
    case GPC_E_NO_FOREST: return "no forest set";
    case GPC_E_FOREST_RANGE: return "forest test offset outside the 27x27 patch";
    case GPC_E_IO: return "forest file could not be read";
    case GPC_E_UNSUPPORTED: return "unsupported setting on the HIP path";
    default: return "unknown status";
  }
}

int gpc_hip_device_count(int* count) {
  if (!count) return GPC_E_INVALID;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    *count = 0;
    return GPC_E_NO_DEVICE;
  }
  *count = n;
  return GPC_OK;
}

int gpc_hip_create(int device, gpc_hip_ctx** out) {
  if (!out) return GPC_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return GPC_E_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return GPC_E_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GPC_E_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return GPC_E_NO_DEVICE;
  gpc_hip_ctx* c = new gpc_hip_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return GPC_E_NO_DEVICE;
  }
  c->stream = c->own_stream;
  const char* ht = getenv("GPC_HIP_HASH_TPW");
  if (ht && atoi(ht) > 0 && atoi(ht) <= 64) c->hash_tpw = atoi(ht);
  c->no_partition = getenv("GPC_HIP_NO_PARTITION") != nullptr;
  c->flat_chunks = getenv("GPC_HIP_FLAT_CHUNKS") != nullptr;
  if (const char* e = getenv("GPC_HIP_GP_TARGET")) {
    const int v = atoi(e);
    if (v >= 256 && v <= 3500) c->gp_target = v;
  }
  if (const char* e = getenv("GPC_HIP_GP_LOG2BINS")) {
    const int v = atoi(e);
    if (v >= 8 && v <= 11) c->gp_log2bins = v;
  }
  c->ht_no_half = getenv("GPC_HIP_HT_NO_HALF") != nullptr;
  if (const char* e = getenv("GPC_HIP_HT_MID")) {
    const int v = atoi(e);
    if (v >= HM_CAP && v <= 64) c->ht_mid = v;
  }
  if (const char* e = getenv("GPC_HIP_HT_LBITS")) {
    const int v = atoi(e);
    if (v >= 7 && v <= 10) c->ht_lbits = v;
  }
  if (const char* e = getenv("GPC_HIP_ROWS_PER_CHUNK")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 64) c->rows_per_chunk = v;
  }
  if (const char* e = getenv("GPC_HIP_DIRECT_MAX")) {
    const int v = atoi(e);
    if (v >= 0 && v <= 4096) c->direct_max = v;
  }
  const char* ck = getenv("GPC_HIP_CHUNK");
  if (ck && atoi(ck) > 0 && atoi(ck) <= 1024) c->chunk_pairs = atoi(ck);
  const char* et = getenv("GPC_HIP_EXPAND_THREADS");
  if (et && atoi(et) > 0 && atoi(et) <= 64) c->expand_threads = atoi(et);
  if (const char* e = getenv("GPC_HIP_UPLOAD")) c->upload_mode = atoi(e);
  c->no_grad_bits = getenv("GPC_HIP_NO_GRAD_BITS") != nullptr;
  c->no_feeder = getenv("GPC_HIP_NO_FEEDER") != nullptr;
  c->no_pair_packed = getenv("GPC_HIP_NO_PAIR_PACKED") != nullptr;
  c->no_fuse = getenv("GPC_HIP_NO_FUSE") != nullptr;
  if (const char* e = getenv("GPC_HIP_HASH_TALL")) c->hash_tall = atoi(e) ? 1 : 0;
  if (const char* e = getenv("GPC_HIP_PRE_ROWS")) {
    const int v = atoi(e);
    if (v == PP_ROWS || v == 10 || v == 7 || v == PP_ROWS_MID || v == 4 || v == PP_ROWS_SMALL) c->pre_rows = v;
  }
  c->fuse_always = getenv("GPC_HIP_FUSE_ALWAYS") != nullptr;
  if (const char* e = getenv("GPC_HIP_FUSE_WGS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 65536) c->fuse_wgs = v;
  }
  if (const char* e = getenv("GPC_HIP_FUSE_SHARDS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= RJ_SHARDS) c->fuse_shards = v;
  }
  if (const char* e = getenv("GPC_HIP_FUSE_MIN_PAIRS")) {
    const int v = atoi(e);
    if (v >= 1) c->fuse_min_pairs = v;
  }
  if (const char* e = getenv("GPC_HIP_RESIDENT")) {
    const int v = atoi(e);
    if (v >= 0 && v <= 2) c->resident_mode = v;
  }
  c->debug = getenv("GPC_HIP_DEBUG") != nullptr;
  c->debug_plan = getenv("GPC_HIP_DEBUG_PLAN") != nullptr;
  c->num_cus = prop.multiProcessorCount;
  find_gpu_node_cpus(c);
  const char* jn = getenv("GPC_HIP_JOIN_NT");
  if (jn && (atoi(jn) == 256 || atoi(jn) == 512 || atoi(jn) == 1024)) c->join_nt = atoi(jn);
  {
    std::lock_guard<std::mutex> g(g_res_mu);
    g_ctxs.push_back(c);
  }
  *out = c;
  return GPC_OK;
}

int gpc_hip_destroy(gpc_hip_ctx* c) {
  if (!c) return GPC_E_INVALID;
  {
    std::lock_guard<std::mutex> g(g_res_mu);
    for (size_t k = 0; k < g_ctxs.size(); ++k)
      if (g_ctxs[k] == c) { g_ctxs.erase(g_ctxs.begin() + k); break; }
  }
  (void)hipSetDevice(c->device);
  (void)drain_lanes(c);
  (void)hipStreamSynchronize(c->stream);
  // a fused-join time-out nobody has asked about (callers that only ever waited on their own stream): say so, once
  const int pending_err = check_join_err(c);
  if (pending_err != GPC_OK) fprintf(stderr, "gpc_hip_destroy: %s\n", c->err);
  DevBuf* bufs[] = {&c->raw, &c->smooth, &c->grad, &c->candmap, &c->codes, &c->staged, &c->rowcnt,
                    &c->stats, &c->out, &c->counts, &c->ncand, &c->mask, &c->gkeys[0], &c->gkeys[1],
                    &c->gvals[0], &c->gvals[1], &c->ghist, &c->gmisc, &c->hkeys[0], &c->hkeys[1],
                    &c->hvals[0], &c->hvals[1], &c->hrec, &c->forest_dev, &c->packed, &c->gpart, &c->jstate, &c->gkv,
                    &c->res_smooth, &c->res_grad};
  while (!c->train_sets.empty()) (void)gpc_hip_train_set_destroy(c, c->train_sets.back());
  for (DevBuf* b : bufs) release(*b);
  for (auto& s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  for (auto& s : c->free_spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  if (c->s_in) {
    (void)hipStreamDestroy(c->s_in);
    (void)hipStreamDestroy(c->s_out);
    (void)hipStreamDestroy(c->s_cnt);
    for (int i = 0; i < 4; ++i) {
      (void)hipEventDestroy(c->e_in[i]);
      (void)hipEventDestroy(c->e_comp[i]);
      (void)hipEventDestroy(c->e_cnt[i]);
      (void)hipEventDestroy(c->e_out[i]);
    }
  }
  if (c->e_flag) (void)hipEventDestroy(c->e_flag);
  if (c->e_pre) (void)hipEventDestroy(c->e_pre);
  for (auto& l : c->lanes) {
    if (l.s) {
      (void)hipStreamSynchronize(l.s);
      (void)hipStreamDestroy(l.s);
      (void)hipEventDestroy(l.e_in);
      (void)hipEventDestroy(l.e_hash);
      (void)hipEventDestroy(l.e_join);
    }
    DevBuf* lb[] = {&l.smooth, &l.grad, &l.codes, &l.stats, &l.jstate, &l.staged, &l.rowcnt};
    for (DevBuf* b : lb) release(*b);
  }
  if (c->s_aux) {
    (void)hipStreamDestroy(c->s_aux);
    (void)hipEventDestroy(c->e_fork);
    (void)hipEventDestroy(c->e_join);
  }
  c->pool.stop();
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->h_in) (void)hipHostFree(c->h_in);
  if (c->h_cnt) (void)hipHostFree(c->h_cnt);
  if (c->h_pre) (void)hipHostFree(c->h_pre);
  if (c->h_xfer) (void)hipHostFree(c->h_xfer);
  if (c->h_flag) (void)hipHostFree(c->h_flag);
  if (c->h_err) (void)hipHostFree(c->h_err);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
  return pending_err;
}

const char* gpc_hip_last_error(const gpc_hip_ctx* c) { return c ? c->err : "null context"; }

int gpc_hip_set_stream(gpc_hip_ctx* c, void* hip_stream) {
  if (!c) return GPC_E_INVALID;
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return GPC_OK;
}

int gpc_hip_synchronize(gpc_hip_ctx* c) {
  if (!c) return GPC_E_INVALID;
  CHK(drain_lanes(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return check_join_err(c);
}

int gpc_hip_reserve(gpc_hip_ctx* c, int W, int H, int max_pairs) {
  if (!c || max_pairs <= 0) return GPC_E_INVALID;
  CHK(check_dims(W, H));
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)W * H;
  const int nimg = 2 * max_pairs;
  CHK(ensure(c, c->raw, n * nimg));
  CHK(ensure(c, c->smooth, n * nimg));
  CHK(ensure(c, c->grad, n * nimg));
  CHK(ensure(c, c->codes, sizeof(uint32_t) * n * nimg));
  CHK(ensure(c, c->staged, sizeof(uint32_t) * n * max_pairs));
  CHK(ensure(c, c->rowcnt, sizeof(int32_t) * (size_t)H * nimg));
  CHK(ensure(c, c->stats, sizeof(int32_t) * GPC_STAT_STRIDE * nimg));
  CHK(ensure(c, c->counts, sizeof(int32_t) * nimg));
  CHK(ensure(c, c->ncand, sizeof(int32_t) * nimg));
  return GPC_OK;
}

int gpc_hip_set_arithmetic(gpc_hip_ctx* c, int mode) {
  if (!c || (mode != GPC_ARITH_SSE && mode != GPC_ARITH_NAIVE)) return GPC_E_INVALID;
  c->naive = (mode == GPC_ARITH_NAIVE);
  return GPC_OK;
}

int gpc_hip_host_alloc(gpc_hip_ctx* c, uint64_t bytes, void** ptr) {
  if (!c || !ptr || bytes == 0) return GPC_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipHostMalloc(ptr, (size_t)bytes, hipHostMallocDefault));
  return GPC_OK;
}

int gpc_hip_host_free(gpc_hip_ctx* c, void* ptr) {
  if (!c || !ptr) return GPC_E_INVALID;
  HIPCHK(c, hipHostFree(ptr));
  return GPC_OK;
}

// ------------------------------------------------------------------ forest

int gpc_hip_parse_forest(const char* text, int W, int H, gpc_filter_mask* fm) {
  if (!text || !fm) return GPC_E_INVALID;
  memset(fm, 0, sizeof(*fm));
  fm->width = W;
  fm->height = H;
  const char* p = text;
  int num_ferns = 0, nonzero = 0;
  if (!next_int(p, num_ferns)) return GPC_E_IO;
  for (int i = 0; i < num_ferns; ++i) {
    int fern_id, num_tests;
    std::string scale;
    if (!next_int(p, fern_id) || !next_tok(p, scale) || !next_int(p, num_tests)) return GPC_E_IO;
    for (int j = 0; j < num_tests; ++j) {
      int level, ix, iy, jx, jy, tau;
      if (!next_int(p, level) || !next_int(p, ix) || !next_int(p, iy) || !next_int(p, jx) ||
          !next_int(p, jy) || !next_int(p, tau))
        return GPC_E_IO;
      if (fm->num_tests < GPC_MAX_TESTS) {  // inference.hpp:426
        const int t = fm->num_tests++;
        // (two's-complement wrap where the reference's int arithmetic overflows -- undefined there; found by UBSan on a
        // forged file: offsets that large are refused by gpc_hip_set_forest's window check anyway)
        fm->mask[2 * t] = (int32_t)((uint32_t)ix + (uint32_t)iy * (uint32_t)W);
        fm->mask[2 * t + 1] = (int32_t)((uint32_t)jx + (uint32_t)jy * (uint32_t)W);
        fm->tau[t] = tau;
      } else {
        fm->discarded++;
      }
      if (tau != 0) nonzero++;  // discarded tests count too (inference.hpp:433)
    }
  }
  fm->type = nonzero ? 1 : 0;
  return GPC_OK;
}

int gpc_hip_read_forest(const char* path, int W, int H, gpc_filter_mask* fm) {
  if (!path || !fm) return GPC_E_INVALID;
  memset(fm, 0, sizeof(*fm));
  fm->width = W;
  fm->height = H;
  FILE* fp = fopen(path, "rb");
  if (!fp) return GPC_E_IO;
  std::string text;
  char buf[4096];
  size_t got;
  while ((got = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, got);
  fclose(fp);
  return gpc_hip_parse_forest(text.c_str(), W, H, fm);
}

int gpc_hip_set_forest(gpc_hip_ctx* c, const gpc_filter_mask* fm) {
  if (!c || !fm) return GPC_E_INVALID;
  if (fm->num_tests < 0 || fm->num_tests > GPC_MAX_TESTS) return GPC_E_INVALID;
  CHK(check_dims(fm->width, fm->height));
  // The header-only C++ API is stateless like the reference's Forest and hands the forest over with every match call:
  // the same tests again must not cost a stream synchronisation and a blocking copy (~30 us of a 0.2 ms call)
  if (c->have_forest && fm->num_tests == c->forest_src.num_tests && fm->type == c->forest_src.type &&
      fm->width == c->forest_src.width && fm->height == c->forest_src.height &&
      memcmp(fm->mask, c->forest_src.mask, sizeof(int32_t) * 2 * (size_t)fm->num_tests) == 0 &&
      memcmp(fm->tau, c->forest_src.tau, sizeof(int32_t) * (size_t)fm->num_tests) == 0)
    return GPC_OK;
  const int W = fm->width;
  GpcForestDev f, fn, ft;  // SSE order; reversed for the Naive arithmetic; SSE order with the tall tile's LDS offsets
  memset(&f, 0, sizeof f);
  memset(&fn, 0, sizeof fn);
  memset(&ft, 0, sizeof ft);
  for (int t = 0; t < fm->num_tests; ++t) {
    int d[4];
    for (int q = 0; q < 2; ++q) {
      // off = dx + dy*W with |dx| <= 13 < W/2: recover (dx, dy)
      const int off = fm->mask[2 * t + q];
      int dy = (off >= 0) ? (off + W / 2) / W : -((-off + W / 2) / W);
      int dx = off - dy * W;
      if (dx < -GPC_R || dx > GPC_R || dy < -GPC_R || dy > GPC_R) return GPC_E_FOREST_RANGE;
      d[2 * q] = dx;
      d[2 * q + 1] = dy;
    }
    // the hash kernel keeps 4 byte-shifted copies of the window: tap (dx,dy) is an aligned dword of copy dx&3
    int offs[2], offt[2];
    for (int q = 0; q < 2; ++q) {
      const int dx = d[2 * q], dy = d[2 * q + 1], sft = dx & 3;
      offs[q] = (sft * HT_COPY + dy * HT_STRIDE + (dx - sft)) / 4;
      offt[q] = (sft * ((HT_Y_TALL + 2 * GPC_R) * HT_STRIDE) + dy * HT_STRIDE + (dx - sft)) / 4;
    }
    f.off[t] = (offs[0] & 0xFFFF) | (offs[1] << 16);
    f.boff[2 * t] = offs[0] * 4;
    f.boff[2 * t + 1] = offs[1] * 4;
    ft.off[t] = (offt[0] & 0xFFFF) | (offt[1] << 16);   // (the byte offsets are what the kernels read)
    ft.boff[2 * t] = offt[0] * 4;
    ft.boff[2 * t + 1] = offt[1] * 4;
    f.tau[t] = (int)(int8_t)fm->tau[t];  // _mm_set1_epi8(tau) truncates (filter.hpp:651)
    if ((fm->tau[t] & 0xFF) == 0x80) f.tau_m128 = 1;
    // gpcFilterNaive shifts the code left per test: test t ends on bit T-1-t (filter.hpp:245-249)
    const int u = fm->num_tests - 1 - t;
    fn.off[u] = f.off[t];
    fn.boff[2 * u] = f.boff[2 * t];
    fn.boff[2 * u + 1] = f.boff[2 * t + 1];
    fn.tau[u] = fm->tau[t];              // gpcFilterTauNaive uses the int as is (:276)
  }
  memcpy(ft.tau, f.tau, sizeof f.tau);
  ft.tau_m128 = f.tau_m128;
  for (int t = 0; t < fm->num_tests; ++t) {
    const uint32_t tb = (uint32_t)fm->tau[t] & 0xFFu;
    f.tauk[t] = ft.tauk[t] = (int32_t)(tb == 0 ? 0u : f.tau_m128 ? tb * 0x01000100u : (((tb - 1u) & 0xFFu) * 0x01000100u) | 0x00FF00FFu);
  }
  f.num_tests = fn.num_tests = ft.num_tests = fm->num_tests;
  f.type = fn.type = ft.type = fm->type ? 1 : 0;
  c->forest = f;
  c->forest_naive = fn;
  HIPCHK(c, hipSetDevice(c->device));
  CHK(ensure(c, c->forest_dev, 3 * sizeof(GpcForestDev)));
  CHK(drain_lanes(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // a launch in flight may still read the previous tests
  const GpcForestDev both[3] = {f, fn, ft};
  HIPCHK(c, hipMemcpy(c->forest_dev.p, both, sizeof both, hipMemcpyHostToDevice));
  c->forest_w = fm->width;
  c->forest_h = fm->height;
  c->forest_src = *fm;
  c->have_forest = true;
  return GPC_OK;
}

// ------------------------------------------------------------------ host-buffer entry points

// Forest::preprocessImage in two steps (the one-call form below is both): _begin runs the kernels and leaves smooth,
// grad and the candidate list in page-locked staging memory of the context -- the device writes them there over the link
// itself -- and says how many candidates there are; _fetch copies them into the caller's arrays (which can be sized by
// then) and remembers those arrays as the host copies of an image that is still on the device (resident_slot).
static int preprocess_begin(gpc_hip_ctx* c, const uint8_t* raw, int W, int H, int thr, bool want_mask) {
  if (!c || !raw) return GPC_E_INVALID;
  if (thr < 0 || thr > 255) return GPC_E_INVALID;
  CHK(check_dims(W, H));
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)W * H;
  const size_t maxcand = (size_t)(W - 2 * GPC_R) * (H - 2 * GPC_R);
  c->pre_slot = -1;
  c->pend.active = false;
  if (n != c->res_n || !c->res_smooth.p) {  // another image size: both resident images go
    {
      std::lock_guard<std::mutex> g(g_res_mu);
      c->res[0].valid = c->res[1].valid = false;
    }
    CHK(ensure(c, c->res_smooth, 2 * n));
    CHK(ensure(c, c->res_grad, 2 * n));
    c->res_n = n;
  }
  if (!c->e_pre) HIPCHK(c, hipEventCreateWithFlags(&c->e_pre, hipEventDisableTiming));
  // staging: raw | smooth | grad | mask (every candidate a pixel can be) | count   (all offsets multiples of 16)
  const size_t mask_bytes = pad16(sizeof(int32_t) * maxcand);
  const size_t need = 3 * n + mask_bytes + 64;
  if (need > c->h_pre_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->h_pre) HIPCHK(c, hipHostFree(c->h_pre));
    c->h_pre = nullptr;
    c->h_pre_cap = 0;
    HIPCHK(c, hipHostMalloc((void**)&c->h_pre, need + need / 8, hipHostMallocDefault));
    c->h_pre_cap = need + need / 8;
  }
  uint8_t* d_stage = nullptr;  // the device's view of the staging block
  HIPCHK(c, hipHostGetDevicePointer((void**)&d_stage, c->h_pre, 0));
  const int slot = c->res_next;
  {
    std::lock_guard<std::mutex> g(g_res_mu);
    c->res[slot].valid = false;
  }
  uint8_t* d_sm = (uint8_t*)c->res_smooth.p + (size_t)slot * n;
  uint8_t* d_gr = (uint8_t*)c->res_grad.p + (size_t)slot * n;
  CHK(ensure(c, c->raw, n));
  CHK(ensure(c, c->stats, sizeof(int32_t) * GPC_STAT_STRIDE * 2));
  // the image: a page-locked one is read where it lies, a pageable one (ndb::Buffer, std::vector) passes through staging
  const uint8_t* v_raw = static_cast<const uint8_t*>(device_view_of_host(raw));
  if (!v_raw || ((uintptr_t)v_raw & 15u)) {
    memcpy(c->h_pre, raw, n);
    v_raw = d_stage;
  }
  CHK(dev_copy16(c, c->raw.p, v_raw, n));
  CHK(run_preprocess(c, (const uint8_t*)c->raw.p, nullptr, W, H, 1, 1, thr, false, d_sm, d_gr));
  // smooth and grad leave over the link while the candidate list is made
  const unsigned n16 = (unsigned)(n / 16);
  hipLaunchKernelGGL(gpc::k_upload2, dim3((n16 + 255) / 256, 2), dim3(256), 0, c->stream, (const uint4*)d_sm, (const uint4*)d_gr,
                     (uint4*)(d_stage + n), (uint4*)(d_stage + 2 * n), n16);
  HIPCHK(c, hipEventRecord(c->e_pre, c->stream));
  if (want_mask) {
    CHK(ensure(c, c->rowcnt, sizeof(int32_t) * (size_t)H * 2));
    dim3 grid(H - 2 * GPC_R, 1);
    Timed t(c, KID_MASK);
    hipLaunchKernelGGL(gpc::k_mask_count, grid, dim3(RM_THREADS), 0, c->stream, (const uint8_t*)d_gr, W, H, (int32_t*)c->rowcnt.p);
    hipLaunchKernelGGL(gpc::k_mask_write, grid, dim3(RM_THREADS), 0, c->stream, (const uint8_t*)d_gr, W, H,
                       (const int32_t*)c->rowcnt.p, reinterpret_cast<int32_t*>(d_stage + 3 * n), (int)maxcand,
                       reinterpret_cast<int32_t*>(d_stage + 3 * n + mask_bytes));
  }
  HIPCHK(c, hipGetLastError());
  c->pre_slot = slot;
  c->pre_W = W;
  c->pre_H = H;
  c->pre_have_mask = want_mask;
  return GPC_OK;
}

static int preprocess_fetch(gpc_hip_ctx* c, uint8_t* smooth, uint8_t* grad, int32_t* mask, int mask_cap, int* n_mask) {
  if (!c || c->pre_slot < 0 || mask_cap < 0) return GPC_E_INVALID;
  const int slot = c->pre_slot, W = c->pre_W, H = c->pre_H;
  c->pre_slot = -1;
  const size_t n = (size_t)W * H;
  const size_t maxcand = (size_t)(W - 2 * GPC_R) * (H - 2 * GPC_R);
  const size_t mask_bytes = pad16(sizeof(int32_t) * maxcand);
  const uint8_t* h_smooth = c->h_pre + n;
  const uint8_t* h_grad = h_smooth + n;
  const int32_t* h_mask = reinterpret_cast<const int32_t*>(h_grad + n);
  const int32_t* h_count = reinterpret_cast<const int32_t*>(h_grad + n + mask_bytes);
  {
    std::lock_guard<std::mutex> g(g_res_mu);  // whatever these arrays were the host copies of, they are no longer
    drop_overlapping(smooth, smooth ? n : 0);
    drop_overlapping(grad, grad ? n : 0);
    drop_overlapping(mask, sizeof(int32_t) * (size_t)(mask ? mask_cap : 0));
  }
  const bool par = n >= 128 * 1024;
  if (par) CHK(ensure_pool(c));
  // the two images first (they have landed when e_pre has passed), the candidate list when the stream is done
  HIPCHK(c, hipEventSynchronize(c->e_pre));
  if (smooth) host_copy(c, smooth, h_smooth, n, par);
  if (grad) host_copy(c, grad, h_grad, n, par);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int cnt = c->pre_have_mask ? h_count[0] : 0;
  if (n_mask) *n_mask = cnt;
  const int ncopy = (mask && c->pre_have_mask) ? (cnt < mask_cap ? cnt : mask_cap) : 0;
  if (ncopy) host_copy(c, mask, h_mask, sizeof(int32_t) * (size_t)ncopy, par);
  uint64_t fp = 0;
  const bool record = c->resident_mode && smooth && grad && c->pre_have_mask && (mask || cnt == 0) && cnt <= mask_cap;
  if (record) fp = fingerprint(h_smooth, h_grad, n, h_mask, cnt, c->resident_mode == 2);  // (the staging copy: the same bytes)
  if (par) host_copy_wait(c);
  if (record) {
    std::lock_guard<std::mutex> g(g_res_mu);
    gpc_hip_ctx::Resident& r = c->res[slot];
    r.smooth = smooth;
    r.grad = grad;
    r.mask = mask;
    r.n_mask = cnt;
    r.W = W;
    r.H = H;
    r.naive = c->naive;
    r.fp = fp;
    r.valid = true;
    c->res_next = slot ^ 1;
  }
  return (mask && c->pre_have_mask && cnt > mask_cap) ? GPC_E_CAPACITY : GPC_OK;
}

int gpc_hip_preprocess_begin(gpc_hip_ctx* c, const uint8_t* raw, int W, int H, int thr) {
  return preprocess_begin(c, raw, W, H, thr, true);
}

int gpc_hip_preprocess_fetch(gpc_hip_ctx* c, uint8_t* smooth, uint8_t* grad, int32_t* mask, int mask_cap, int* n_mask) {
  return preprocess_fetch(c, smooth, grad, mask, mask_cap, n_mask);
}

int gpc_hip_preprocess(gpc_hip_ctx* c, const uint8_t* raw, int W, int H, int thr, uint8_t* smooth,
                       uint8_t* grad, int32_t* mask, int mask_cap, int* n_mask) {
  CHK(preprocess_begin(c, raw, W, H, thr, n_mask || mask));
  return preprocess_fetch(c, smooth, grad, mask, mask ? mask_cap : 0, n_mask);
}

int gpc_hip_resident_hits(const gpc_hip_ctx* c) { return c ? c->resident_hits : 0; }

int gpc_hip_hash_codes(gpc_hip_ctx* c, const uint8_t* smooth, const uint8_t* grad, int W, int H,
                       uint32_t* codes) {
  if (!c || !smooth || !grad || !codes) return GPC_E_INVALID;
  CHK(check_dims(W, H));
  CHK(forest_matches(c, W, H));
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)W * H;
  CHK(ensure(c, c->smooth, n));
  CHK(ensure(c, c->grad, n));
  CHK(ensure(c, c->codes, sizeof(uint32_t) * n));
  CHK(ensure(c, c->stats, sizeof(int32_t) * GPC_STAT_STRIDE));
  // (the caller's arrays pass through the page-locked arena: gpc_hip_ctx::h_xfer)
  uint8_t* d_arena = nullptr;
  CHK(xfer_reserve(c, 4 * n, &d_arena));
  CHK(ensure_pool(c));
  host_copy(c, c->h_xfer, smooth, n, true);
  host_copy(c, c->h_xfer + n, grad, n, true);
  host_copy_wait(c);
  CHK(dev_copy16(c, c->smooth.p, d_arena, n));
  c->grad_is_bits = false;  // the caller's byte image
  CHK(dev_copy16(c, c->grad.p, d_arena + n, n));
  // the reference's zero-filled gpcstates buffer (inference.hpp:274): the kernel writes the candidate rows only
  HIPCHK(c, hipMemsetAsync(c->codes.p, 0, sizeof(uint32_t) * n, c->stream));
  CHK(run_hash(c, (const uint8_t*)c->smooth.p, (const uint8_t*)c->grad.p, nullptr, W, H, 1, true,
               (uint32_t*)c->codes.p));
  CHK(dev_copy16(c, d_arena, c->codes.p, sizeof(uint32_t) * n));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  host_copy(c, codes, c->h_xfer, sizeof(uint32_t) * n, true);
  host_copy_wait(c);
  return GPC_OK;
}

static int pinned_counts(gpc_hip_ctx* c, int npairs);

// Queues hash + match of two preprocessed images; the results go to the transfer arena (cap_dev records at most; the
// caller's page-locked array instead when direct_out is its device view) and are collected by match_fetch.
static int match_preprocessed_begin(gpc_hip_ctx* c, const uint8_t* smoothL, const uint8_t* gradL,
                                    const int32_t* maskL, int nL, const uint8_t* smoothR, const uint8_t* gradR,
                                    const int32_t* maskR, int nR, int W, int H, const gpc_settings* s, int mode,
                                    int cap_dev, void* direct_out) {
  if (!c || !smoothL || !gradL || !smoothR || !gradR || cap_dev < 0) return GPC_E_INVALID;
  if ((nL > 0 && !maskL) || (nR > 0 && !maskR) || nL < 0 || nR < 0) return GPC_E_INVALID;
  CHK(check_settings(s));
  CHK(check_dims(W, H));
  CHK(forest_matches(c, W, H));
  HIPCHK(c, hipSetDevice(c->device));
  c->pend.active = false;
  c->pre_slot = -1;
  const size_t n = (size_t)W * H;
  const size_t esz = mode == 0 ? sizeof(gpc_support) : sizeof(gpc_correspondence);
  if ((uint64_t)cap_dev * esz >= (1ull << 32)) return GPC_E_UNSUPPORTED;  // (beyond 2^30 pixels anyway)
  CHK(ensure(c, c->codes, sizeof(uint32_t) * 2 * n));
  CHK(ensure(c, c->stats, sizeof(int32_t) * GPC_STAT_STRIDE * 2));
  CHK(pinned_counts(c, 1));
  int32_t* d_cnt = nullptr;
  HIPCHK(c, hipHostGetDevicePointer((void**)&d_cnt, c->h_cnt, 0));
  // Supports of the epipolar sort-matcher cross the link PACKED (x | xR << 16 + the row counts: 4 bytes instead of 12 -- a
  // 1024x436 pair's 3.2 MB of records were 60 us of the link, a third of the call) and are expanded into the caller's
  // array by the worker threads, who read 1.1 MB where they copied 3.2
  const bool packed = mode == 0 && !direct_out && s->epipolar_mode && !s->use_hashtable && !c->no_pair_packed;
  const size_t rows_bytes = pad16(sizeof(int32_t) * (size_t)H);
  const size_t out_bytes = packed ? rows_bytes + pad16(sizeof(uint32_t) * (size_t)(cap_dev > 0 ? cap_dev : 1))
                                  : pad16(esz * (size_t)(cap_dev > 0 ? cap_dev : 1));
  const int sl = resident_slot(c, smoothL, gradL, maskL, nL, W, H);
  const int sr = sl >= 0 ? resident_slot(c, smoothR, gradR, maskR, nR, W, H) : -1;
  const bool resident = sl >= 0 && sr >= 0;
  // arena: [results | smooth L R | grad L R | mask L | mask R]; the inputs only on the upload path
  const size_t mL = pad16(sizeof(int32_t) * (size_t)nL), mR = pad16(sizeof(int32_t) * (size_t)nR);
  uint8_t* d_arena = nullptr;
  const size_t in_bytes = resident ? 0 : 4 * n + mL + mR;
  if (!direct_out || in_bytes) CHK(xfer_reserve(c, (direct_out ? 0 : out_bytes) + in_bytes, &d_arena));
  const size_t in_off = direct_out ? 0 : out_bytes;
  void* d_out = direct_out ? direct_out : d_arena;
  const uint8_t* d_sm = nullptr;
  const uint8_t* d_gr = nullptr;
  const uint8_t* d_cand = nullptr;
  if (resident) {
    // Both images are the host copies of images gpc_hip_preprocess left on the device: hash and match from there.  Their
    // candidates are the gradient image's (the mask list IS that image's non-zero pixels inside the margin), so the
    // pipeline is the batched one without its first kernel.
    d_sm = (const uint8_t*)c->res_smooth.p;
    d_gr = (const uint8_t*)c->res_grad.p;
    if (!(sl == 0 && sr == 1)) {  // right before left, or one image on both sides: put them in pair order
      CHK(ensure(c, c->smooth, 2 * n));
      CHK(ensure(c, c->grad, 2 * n));
      const int src[2] = {sl, sr};
      for (int k = 0; k < 2; ++k) {
        CHK(dev_copy16(c, (uint8_t*)c->smooth.p + k * n, d_sm + src[k] * n, n));
        CHK(dev_copy16(c, (uint8_t*)c->grad.p + k * n, d_gr + src[k] * n, n));
      }
      d_sm = (const uint8_t*)c->smooth.p;
      d_gr = (const uint8_t*)c->grad.p;
    }
    d_cand = d_gr;
    ++c->resident_hits;
  } else {
    CHK(ensure(c, c->smooth, 2 * n));
    CHK(ensure(c, c->grad, 2 * n));
    CHK(ensure(c, c->candmap, 2 * n));
    uint8_t* h_in = c->h_xfer + in_off;
    const bool par = in_bytes >= 512 * 1024;
    if (par) CHK(ensure_pool(c));
    host_copy(c, h_in, smoothL, n, par);
    host_copy(c, h_in + n, smoothR, n, par);
    host_copy(c, h_in + 2 * n, gradL, n, par);
    host_copy(c, h_in + 3 * n, gradR, n, par);
    host_copy(c, h_in + 4 * n, maskL, sizeof(int32_t) * (size_t)nL, par);
    host_copy(c, h_in + 4 * n + mL, maskR, sizeof(int32_t) * (size_t)nR, par);
    if (par) host_copy_wait(c);
    const uint8_t* d_in = d_arena + in_off;
    CHK(dev_copy16(c, c->smooth.p, d_in, 2 * n));
    CHK(dev_copy16(c, c->grad.p, d_in + 2 * n, 2 * n));
    uint8_t* d_cm = (uint8_t*)c->candmap.p;
    HIPCHK(c, hipMemsetAsync(d_cm, 0, 2 * n, c->stream));
    // the mask lists are read where they lie (over the link, once)
    if (nL > 0)
      hipLaunchKernelGGL(gpc::k_scatter_mask, dim3((nL + 255) / 256), dim3(256), 0, c->stream,
                         reinterpret_cast<const int32_t*>(d_in + 4 * n), nL, d_cm, W, H);
    if (nR > 0)
      hipLaunchKernelGGL(gpc::k_scatter_mask, dim3((nR + 255) / 256), dim3(256), 0, c->stream,
                         reinterpret_cast<const int32_t*>(d_in + 4 * n + mL), nR, d_cm + n, W, H);
    HIPCHK(c, hipGetLastError());
    d_sm = (const uint8_t*)c->smooth.p;
    d_gr = (const uint8_t*)c->grad.p;
    d_cand = d_cm;
  }
  c->grad_is_bits = false;  // byte images
  hipLaunchKernelGGL(gpc::k_stats_init, dim3(1), dim3(64), 0, c->stream, (int32_t*)c->stats.p, 2);
  CHK(run_hash(c, d_sm, d_gr, resident ? nullptr : d_cand, W, H, 2, false, (uint32_t*)c->codes.p));
  if (packed) {
    const PackedOut po = {reinterpret_cast<int32_t*>(d_arena), (long)cap_dev, (long)H};
    CHK(run_match(c, W, H, 1, s, 2, d_cand, d_arena + rows_bytes, cap_dev, d_cnt, nullptr, &po));
  } else {
    CHK(run_match(c, W, H, 1, s, mode, d_cand, d_out, cap_dev, d_cnt, nullptr));
  }
  c->pend.active = true;
  c->pend.direct = direct_out != nullptr;
  c->pend.have_ncand = false;
  c->pend.packed = packed;
  c->pend.esz = esz;
  c->pend.cap_dev = cap_dev;
  c->pend.H = H;
  return GPC_OK;
}

// Waits for what a *_begin queued and delivers min(count, cap) records; the true count always.  May be called again
// (a larger array after GPC_E_CAPACITY) until the next call on the context.
static int match_fetch(gpc_hip_ctx* c, void* out, int cap, int* n_out, int* ncl, int* ncr) {
  if (!c || !c->pend.active || !n_out || cap < 0 || (cap > 0 && !out)) return GPC_E_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  CHK(check_join_err(c));
  const int32_t cnt = c->h_cnt[0];
  *n_out = cnt;
  if (c->pend.have_ncand) {
    if (ncl) *ncl = c->h_cnt[1];
    if (ncr) *ncr = c->h_cnt[2];
  }
  int ncopy = cnt < cap ? cnt : cap;
  if (ncopy > c->pend.cap_dev) ncopy = c->pend.cap_dev;
  if (c->pend.packed && ncopy > 0) {
    const int H = c->pend.H;
    const int32_t* rows = reinterpret_cast<const int32_t*>(c->h_xfer);
    const uint32_t* words = reinterpret_cast<const uint32_t*>(c->h_xfer + pad16(sizeof(int32_t) * (size_t)H));
    const bool par = ncopy >= 32 * 1024;
    if (par) CHK(ensure_pool(c));
    const int parts = par ? (c->pool.size() > 1 ? c->pool.size() : 1) : 1;
    long first = 0;
    for (int q = 0; q < parts; ++q) {
      const int y0 = GPC_R + (int)((long)(H - 2 * GPC_R) * q / parts), y1 = GPC_R + (int)((long)(H - 2 * GPC_R) * (q + 1) / parts);
      if (first < ncopy && y1 > y0) {
        if (par) c->pool.push(ExpandJob{words, rows, H, y0, y1, (int)first, ncopy, static_cast<gpc_support*>(out), 6});
        else expand_rows(words, rows, y0, y1, first, ncopy, static_cast<gpc_support*>(out));
      }
      for (int y = y0; y < y1; ++y) first += rows[y];
    }
    if (par) host_copy_wait(c);
  } else if (!c->pend.direct && ncopy > 0) {
    const size_t bytes = c->pend.esz * (size_t)ncopy;
    const bool par = bytes >= 256 * 1024;
    if (par) CHK(ensure_pool(c));
    host_copy(c, out, c->h_xfer, bytes, par);
    if (par) host_copy_wait(c);
  }
  return (cnt > cap || cnt > c->pend.cap_dev) ? GPC_E_CAPACITY : GPC_OK;
}

static int match_preprocessed(gpc_hip_ctx* c, const uint8_t* smoothL, const uint8_t* gradL,
                              const int32_t* maskL, int nL, const uint8_t* smoothR, const uint8_t* gradR,
                              const int32_t* maskR, int nR, int W, int H, const gpc_settings* s, int mode,
                              void* out, int cap, int* n_out) {
  if (!n_out || cap < 0 || (cap > 0 && !out)) return GPC_E_INVALID;
  // a page-locked `out` is written by the matcher itself over the link
  const size_t esz = mode == 0 ? sizeof(gpc_support) : sizeof(gpc_correspondence);
  void* dv = (c && cap > 0 && (uint64_t)cap * esz < (1ull << 32)) ? device_view_of_host(out) : nullptr;
  CHK(match_preprocessed_begin(c, smoothL, gradL, maskL, nL, smoothR, gradR, maskR, nR, W, H, s, mode, cap, dv));
  const int st = match_fetch(c, out, cap, n_out, nullptr, nullptr);
  c->pend.active = false;
  return st;
}

int gpc_hip_rectified_match(gpc_hip_ctx* c, const uint8_t* smoothL, const uint8_t* gradL,
                            const int32_t* maskL, int nL, const uint8_t* smoothR, const uint8_t* gradR,
                            const int32_t* maskR, int nR, int W, int H, const gpc_settings* s,
                            gpc_support* out, int cap, int* n_out) {
  return match_preprocessed(c, smoothL, gradL, maskL, nL, smoothR, gradR, maskR, nR, W, H, s, 0, out, cap, n_out);
}

int gpc_hip_stereo_match(gpc_hip_ctx* c, const uint8_t* smoothL, const uint8_t* gradL,
                         const int32_t* maskL, int nL, const uint8_t* smoothR, const uint8_t* gradR,
                         const int32_t* maskR, int nR, int W, int H, const gpc_settings* s,
                         gpc_correspondence* out, int cap, int* n_out) {
  return match_preprocessed(c, smoothL, gradL, maskL, nL, smoothR, gradR, maskR, nR, W, H, s, 1, out, cap, n_out);
}

// the asynchronous forms: every match of two candidate lists has at most min(nL, nR) results
int gpc_hip_rectified_match_begin(gpc_hip_ctx* c, const uint8_t* smoothL, const uint8_t* gradL, const int32_t* maskL, int nL,
                                  const uint8_t* smoothR, const uint8_t* gradR, const int32_t* maskR, int nR, int W, int H,
                                  const gpc_settings* s) {
  return match_preprocessed_begin(c, smoothL, gradL, maskL, nL, smoothR, gradR, maskR, nR, W, H, s, 0, (nL < nR ? nL : nR) + 1, nullptr);
}

int gpc_hip_stereo_match_begin(gpc_hip_ctx* c, const uint8_t* smoothL, const uint8_t* gradL, const int32_t* maskL, int nL,
                               const uint8_t* smoothR, const uint8_t* gradR, const int32_t* maskR, int nR, int W, int H,
                               const gpc_settings* s) {
  return match_preprocessed_begin(c, smoothL, gradL, maskL, nL, smoothR, gradR, maskR, nR, W, H, s, 1, (nL < nR ? nL : nR) + 1, nullptr);
}

int gpc_hip_match_fetch(gpc_hip_ctx* c, void* out, int cap, int* n_out, int* n_cand_l, int* n_cand_r) {
  return match_fetch(c, out, cap, n_out, n_cand_l, n_cand_r);
}

// ------------------------------------------------------------------ batch entry points

int gpc_hip_match_batch_device(gpc_hip_ctx* c, const uint8_t* d_rawL, const uint8_t* d_rawR, int W, int H,
                               int npairs, const gpc_settings* s, gpc_support* d_out, int cap_per_pair,
                               int32_t* d_counts, int32_t* d_ncand) {
  if (!c || !d_rawL || !d_rawR || !d_out || !d_counts || npairs <= 0 || cap_per_pair <= 0) return GPC_E_INVALID;
  CHK(check_settings(s));
  CHK(check_dims(W, H));
  CHK(forest_matches(c, W, H));
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)W * H;
  if (c->pipeline > 1 && s->epipolar_mode && !s->use_hashtable) {
    // Two lanes: this call goes to the lane the previous one did not take.  Its inputs are what the context's stream has
    // produced so far (e_in); its k_preprocess starts when the other lane's k_hash is done -- beside that lane's join --
    // and its join is queued behind its own k_hash only: the other lane's join is draining by then and this one's
    // workgroups move in as places fall free.  Nothing is queued on the context's stream: gpc_hip_synchronize (or
    // gpc_hip_pipeline_join for a caller with a stream of its own) orders the results.
    CHK(ensure_lanes(c));
    gpc_hip_ctx::Lane& lane = c->lanes[c->next_lane];
    gpc_hip_ctx::Lane& other = c->lanes[c->next_lane ^ 1];
    c->next_lane ^= 1;
    HIPCHK(c, hipEventRecord(lane.e_in, c->stream));
    HIPCHK(c, hipStreamWaitEvent(lane.s, lane.e_in, 0));
    if (other.used) HIPCHK(c, hipStreamWaitEvent(lane.s, other.e_hash, 0));
    LaneScope in(c, lane);
    lane.used = true;
    CHK(ensure(c, c->codes, sizeof(uint32_t) * n * 2 * npairs));
    CHK(run_preprocess(c, d_rawL, d_rawR, W, H, npairs, 2, s->gradient_threshold, true));
    CHK(run_hash(c, (const uint8_t*)c->smooth.p, (const uint8_t*)c->grad.p, nullptr, W, H, 2 * npairs, false, (uint32_t*)c->codes.p));
    HIPCHK(c, hipEventRecord(lane.e_hash, c->stream));
    CHK(run_match(c, W, H, npairs, s, 0, (const uint8_t*)c->grad.p, d_out, cap_per_pair, d_counts, d_ncand));
    HIPCHK(c, hipEventRecord(lane.e_join, c->stream));
    return GPC_OK;
  }
  if (c->pipeline > 1) CHK(drain_lanes(c));  // (the device-wide matchers run on the context itself)
  CHK(ensure(c, c->codes, sizeof(uint32_t) * n * 2 * npairs));
  if (c->dbg_wait_pre) HIPCHK(c, hipStreamWaitEvent(c->stream, c->dbg_wait_pre, 0));
  CHK(run_preprocess(c, d_rawL, d_rawR, W, H, npairs, 2, s->gradient_threshold, true));
  if (c->dbg_wait_hash) HIPCHK(c, hipStreamWaitEvent(c->stream, c->dbg_wait_hash, 0));
  CHK(run_hash(c, (const uint8_t*)c->smooth.p, (const uint8_t*)c->grad.p, nullptr, W, H, 2 * npairs, false,
               (uint32_t*)c->codes.p));
  if (c->dbg_rec_hash) HIPCHK(c, hipEventRecord(c->dbg_rec_hash, c->stream));
  CHK(run_match(c, W, H, npairs, s, 0, (const uint8_t*)c->grad.p, d_out, cap_per_pair, d_counts, d_ncand));
  if (c->dbg_rec_join) HIPCHK(c, hipEventRecord(c->dbg_rec_join, c->stream));
  return GPC_OK;
}

// Experiment hook, not part of the C ABI (include/gpc_hip.h does not declare it): see gpc_hip_ctx::dbg_wait_pre.
int gpc_hip_set_pipeline(gpc_hip_ctx* c, int lanes) {
  if (!c || (lanes != 1 && lanes != 2)) return GPC_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  CHK(drain_lanes(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->pipeline = lanes;
  return GPC_OK;
}

int gpc_hip_pipeline_join(gpc_hip_ctx* c) {
  if (!c) return GPC_E_INVALID;
  for (auto& l : c->lanes)
    if (l.s && l.used) HIPCHK(c, hipStreamWaitEvent(c->stream, l.e_join, 0));
  return GPC_OK;
}

int gpc_hip_debug_pipeline_events(gpc_hip_ctx* c, void* wait_before_preprocess, void* wait_before_hash, void* record_after_hash,
                                  void* record_after_join) {
  if (!c) return GPC_E_INVALID;
  c->dbg_wait_pre = (hipEvent_t)wait_before_preprocess;
  c->dbg_wait_hash = (hipEvent_t)wait_before_hash;
  c->dbg_rec_hash = (hipEvent_t)record_after_hash;
  c->dbg_rec_join = (hipEvent_t)record_after_join;
  return GPC_OK;
}

// page-locked landing area for npairs counts + 2*npairs candidate counts
static int pinned_counts(gpc_hip_ctx* c, int npairs) {
  if ((size_t)npairs * 3 <= c->h_cnt_cap) return GPC_OK;
  if (c->h_cnt) HIPCHK(c, hipHostFree(c->h_cnt));
  c->h_cnt = nullptr;
  c->h_cnt_cap = 0;
  HIPCHK(c, hipHostMalloc((void**)&c->h_cnt, sizeof(int32_t) * 3 * (size_t)npairs, hipHostMallocDefault));
  c->h_cnt_cap = (size_t)npairs * 3;
  return GPC_OK;
}

static int batch_streams(gpc_hip_ctx* c) {
  if (c->s_in) return GPC_OK;
  HIPCHK(c, hipStreamCreateWithFlags(&c->s_in, hipStreamNonBlocking));
  HIPCHK(c, hipStreamCreateWithFlags(&c->s_out, hipStreamNonBlocking));
  HIPCHK(c, hipStreamCreateWithFlags(&c->s_cnt, hipStreamNonBlocking));  // the counts: never queued behind bulk copies
  for (int i = 0; i < 4; ++i) {
    HIPCHK(c, hipEventCreateWithFlags(&c->e_in[i], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->e_comp[i], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->e_cnt[i], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->e_out[i], hipEventDisableTiming));
  }
  return GPC_OK;
}

// Host buffers in, supports out.  The batch goes through the device in chunks: while chunk k is matched
// on the context's stream, chunk k+1 is uploaded on a second stream and the supports of chunk k-1 go
// back on a third (PCIe is full duplex), so the call costs about what the longer direction of the link
// costs -- the results' way back -- instead of upload + kernels + download one after the other.
// The only host waits are for a chunk's counts, which say how many supports of each pair to fetch.
static int match_batch_unpacked(gpc_hip_ctx* c, const uint8_t* rawL, const uint8_t* rawR, int W, int H, int npairs,
                                const gpc_settings* s, gpc_support* out, int cap, int32_t* counts, int32_t* ncand) {
  if (!c || !rawL || !rawR || !out || !counts || npairs <= 0 || cap <= 0) return GPC_E_INVALID;
  CHK(check_settings(s));
  CHK(check_dims(W, H));
  CHK(forest_matches(c, W, H));
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)W * H;
  const int chunk = npairs >= 8 ? (npairs >= 32 ? 8 : npairs / 4) : npairs;  // >= 4 chunks once there are 8 pairs
  const int nch = (npairs + chunk - 1) / chunk;
  CHK(ensure(c, c->raw, 2 * 2 * n * chunk));  // two slots x two sides
  CHK(ensure(c, c->out, sizeof(gpc_support) * (size_t)cap * npairs));
  CHK(ensure(c, c->counts, sizeof(int32_t) * npairs));
  CHK(ensure(c, c->ncand, sizeof(int32_t) * 2 * npairs));
  CHK(batch_streams(c));
  CHK(pinned_counts(c, npairs));  // (a copy straight into the caller's pageable arrays blocks the host, see gpc_hip_match_batch)
  // Pageable arrays never reach hipMemcpy (gpc_hip_ctx::h_xfer): images pass through the page-locked bounce buffer, results
  // through a page-locked landing area (two slots of a chunk each), copied by the worker threads
  const bool bounce = !device_view_of_host(rawL) || !device_view_of_host(rawR);
  const bool land = !device_view_of_host(out);
  if (bounce || land) CHK(ensure_pool(c));
  if (bounce) {
    const size_t need = 2 * 2 * n * (size_t)chunk;
    if (need > c->h_in_cap) {
      HIPCHK(c, hipStreamSynchronize(c->s_in));
      if (c->h_in) HIPCHK(c, hipHostFree(c->h_in));
      c->h_in = nullptr;
      c->h_in_cap = 0;
      HIPCHK(c, hipHostMalloc(&c->h_in, need, hipHostMallocDefault));
      c->h_in_cap = need;
    }
  }
  const size_t slot_bytes = sizeof(gpc_support) * (size_t)cap * chunk;
  if (land && 2 * slot_bytes > c->h_stage_cap) {
    c->pool.wait_all();
    HIPCHK(c, hipStreamSynchronize(c->s_out));
    if (c->h_stage) HIPCHK(c, hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->h_stage_cap = 0;
    HIPCHK(c, hipHostMalloc(&c->h_stage, 2 * slot_bytes, hipHostMallocDefault));
    c->h_stage_cap = 2 * slot_bytes;
  }
  int32_t* hc = c->h_cnt;
  int32_t* hn = c->h_cnt + npairs;
  int status = GPC_OK;
  // fetch the supports of chunk k (its counts are on their way: wait for them, then one copy per pair)
  auto collect = [&](int k) -> int {
    const int p0 = k * chunk, pc = (p0 + chunk <= npairs) ? chunk : npairs - p0;
    HIPCHK(c, hipEventSynchronize(c->e_cnt[k & 1]));
    memcpy(counts + p0, hc + p0, sizeof(int32_t) * pc);
    if (ncand) memcpy(ncand + 2 * p0, hn + 2 * p0, sizeof(int32_t) * 2 * pc);
    if (land) c->pool.wait_slot(k & 1);  // the copies of chunk k-2 have left this landing slot
    gpc_support* lslot = land ? reinterpret_cast<gpc_support*>((uint8_t*)c->h_stage + (size_t)(k & 1) * slot_bytes) : nullptr;
    for (int p = p0; p < p0 + pc; ++p) {
      const int ncopy = counts[p] < cap ? counts[p] : cap;
      if (counts[p] > cap) status = GPC_E_CAPACITY;
      if (ncopy > 0)
        HIPCHK(c, hipMemcpyAsync(land ? lslot + (size_t)(p - p0) * cap : out + (size_t)p * cap,
                                 (gpc_support*)c->out.p + (size_t)p * cap, sizeof(gpc_support) * (size_t)ncopy,
                                 hipMemcpyDeviceToHost, c->s_out));
    }
    if (land) HIPCHK(c, hipEventRecord(c->e_out[k & 1], c->s_out));
    return GPC_OK;
  };
  // landing slot of chunk k -> the caller's array, by the workers
  auto deliver = [&](int k) -> int {
    if (!land) return GPC_OK;
    const int p0 = k * chunk, pc = (p0 + chunk <= npairs) ? chunk : npairs - p0;
    HIPCHK(c, hipEventSynchronize(c->e_out[k & 1]));
    const gpc_support* lslot = reinterpret_cast<const gpc_support*>((uint8_t*)c->h_stage + (size_t)(k & 1) * slot_bytes);
    for (int p = p0; p < p0 + pc; ++p) {
      const int ncopy = counts[p] < cap ? counts[p] : cap;
      const size_t bytes = sizeof(gpc_support) * (size_t)ncopy, step = 256 * 1024;
      for (size_t at = 0; at < bytes; at += step) {
        ExpandJob j = {};
        j.slot = k & 1;
        j.copy_src = reinterpret_cast<const uint8_t*>(lslot + (size_t)(p - p0) * cap) + at;
        j.copy_dst = reinterpret_cast<uint8_t*>(out + (size_t)p * cap) + at;
        j.copy_bytes = at + step <= bytes ? step : bytes - at;
        c->pool.push(j);
      }
    }
    return GPC_OK;
  };
  for (int k = 0; k < nch; ++k) {
    const int slot = k & 1, p0 = k * chunk, pc = (p0 + chunk <= npairs) ? chunk : npairs - p0;
    uint8_t* d_l = (uint8_t*)c->raw.p + (size_t)slot * 2 * n * chunk;
    uint8_t* d_r = d_l + n * chunk;
    if (k >= 2) HIPCHK(c, hipStreamWaitEvent(c->s_in, c->e_comp[slot], 0));  // chunk k-2 has read this slot
    const uint8_t *srcL = rawL + (size_t)p0 * n, *srcR = rawR + (size_t)p0 * n;
    if (bounce) {
      uint8_t* bl = (uint8_t*)c->h_in + (size_t)slot * 2 * n * chunk;
      uint8_t* br = bl + n * chunk;
      if (k >= 2) HIPCHK(c, hipEventSynchronize(c->e_in[slot]));  // the upload of chunk k-2 has left this bounce slot
      const size_t bytes = n * pc, step = 256 * 1024;
      for (int side = 0; side < 2; ++side)
        for (size_t at = 0; at < bytes; at += step) {
          ExpandJob j = {};
          j.slot = 7;
          j.copy_src = (side ? srcR : srcL) + at;
          j.copy_dst = (side ? br : bl) + at;
          j.copy_bytes = at + step <= bytes ? step : bytes - at;
          c->pool.push(j);
        }
      c->pool.wait_slot(7);
      srcL = bl;
      srcR = br;
    }
    HIPCHK(c, hipMemcpyAsync(d_l, srcL, n * pc, hipMemcpyHostToDevice, c->s_in));
    HIPCHK(c, hipMemcpyAsync(d_r, srcR, n * pc, hipMemcpyHostToDevice, c->s_in));
    HIPCHK(c, hipEventRecord(c->e_in[slot], c->s_in));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->e_in[slot], 0));
    CHK(gpc_hip_match_batch_device(c, d_l, d_r, W, H, pc, s, (gpc_support*)c->out.p + (size_t)p0 * cap, cap,
                                   (int32_t*)c->counts.p + p0, (int32_t*)c->ncand.p + 2 * p0));
    HIPCHK(c, hipEventRecord(c->e_comp[slot], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->s_cnt, c->e_comp[slot], 0));
    HIPCHK(c, hipMemcpyAsync(hc + p0, (int32_t*)c->counts.p + p0, sizeof(int32_t) * pc, hipMemcpyDeviceToHost, c->s_cnt));
    HIPCHK(c, hipMemcpyAsync(hn + 2 * p0, (int32_t*)c->ncand.p + 2 * p0, sizeof(int32_t) * 2 * pc, hipMemcpyDeviceToHost, c->s_cnt));
    HIPCHK(c, hipEventRecord(c->e_cnt[slot], c->s_cnt));
    if (k >= 1) CHK(collect(k - 1));
    if (k >= 2) CHK(deliver(k - 2));
  }
  CHK(collect(nch - 1));
  if (nch >= 2) CHK(deliver(nch - 2));
  CHK(deliver(nch - 1));
  if (land) c->pool.wait_all();
  HIPCHK(c, hipStreamSynchronize(c->s_out));
  HIPCHK(c, hipStreamSynchronize(c->s_cnt));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  CHK(check_join_err(c));
  return status;
}

int gpc_hip_match_batch_device_packed(gpc_hip_ctx* c, const uint8_t* d_rawL, const uint8_t* d_rawR, int W, int H,
                                      int npairs, const gpc_settings* s, uint32_t* d_packed, int cap_per_pair,
                                      int32_t* d_rows, int32_t* d_counts, int32_t* d_ncand) {
  if (!c || !d_rawL || !d_rawR || !d_packed || !d_rows || !d_counts || npairs <= 0 || cap_per_pair <= 0) return GPC_E_INVALID;
  CHK(check_settings(s));
  if (!s->epipolar_mode || s->use_hashtable) return GPC_E_UNSUPPORTED;  // rows are the unit of the packed form
  CHK(check_dims(W, H));
  CHK(forest_matches(c, W, H));
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)W * H;
  CHK(ensure(c, c->codes, sizeof(uint32_t) * n * 2 * npairs));
  CHK(run_preprocess(c, d_rawL, d_rawR, W, H, npairs, 2, s->gradient_threshold, true));
  CHK(run_hash(c, (const uint8_t*)c->smooth.p, (const uint8_t*)c->grad.p, nullptr, W, H, 2 * npairs, false,
               (uint32_t*)c->codes.p));
  const PackedOut po = {d_rows, (long)cap_per_pair, (long)H};
  CHK(run_match(c, W, H, npairs, s, 2, (const uint8_t*)c->grad.p, d_packed, cap_per_pair, d_counts, d_ncand, &po));
  return GPC_OK;
}

int gpc_hip_expand_packed(const uint32_t* packed, const int32_t* rows, int H, int n, gpc_support* out) {
  if (!packed || !rows || !out || H < 2 * GPC_R || n < 0) return GPC_E_INVALID;
  expand_rows(packed, rows, GPC_R, H - GPC_R, 0, n, out);
  return GPC_OK;
}

// Host buffers in, supports out, for the epipolar sort-matcher (the reference's sparsematch settings): results
// cross the link PACKED (4 bytes per support + the row counts instead of 12 bytes per support) and worker threads
// expand them into the caller's ndb::Support arrays while the next chunks are uploaded, matched and downloaded.
// Per chunk k:  upload (s_in) -> kernels (stream) -> counts (s_cnt) | download of k-1 (s_out) | expansion of k-2 (pool).
// Device results rotate through 3 slots, the page-locked landing area through 4.
// ph != null: the results are LEFT packed in the caller's arrays (gpc_hip_match_batch_packed) instead of being expanded into `out`
struct PackedHost {
  uint32_t* packed;  // [npairs][cap] words xL | xR << 16
  int32_t* rows;     // [npairs][H] supports per row
};
static int match_batch_packed(gpc_hip_ctx* c, const uint8_t* rawL, const uint8_t* rawR, int W, int H, int npairs,
                              const gpc_settings* s, gpc_support* out, int cap, int32_t* counts, int32_t* ncand,
                              const PackedHost* ph = nullptr);

// One pair or two, page-locked `out` (BASELINE configs[1] taken literally: the reference's timed region for ONE pair).
// The chunk pipeline of match_batch_packed is built for the link's throughput: three streams, events, packed records
// and a pool of host threads that expand them -- for one pair that machinery IS the latency (0.20-0.24 ms against
// ~36 us of kernels and ~35 us of link time).  Here everything is queued on the context's stream and the join writes
// the 12-byte supports, the counts and the candidate counts straight into host memory over the link (consecutive
// lanes write consecutive records: whole PCIe write bursts) while it runs; one stream synchronisation ends the call.
static int match_batch_direct(gpc_hip_ctx* c, const uint8_t* rawL, const uint8_t* rawR, int W, int H, int npairs,
                              const gpc_settings* s, gpc_support* d_out_host, int cap, int32_t* counts, int32_t* ncand) {
  const size_t n = (size_t)W * H;
  CHK(ensure(c, c->raw, 2 * n * npairs));
  CHK(ensure(c, c->codes, sizeof(uint32_t) * n * 2 * npairs));
  CHK(pinned_counts(c, npairs));
  int32_t* d_cnt = nullptr;
  HIPCHK(c, hipHostGetDevicePointer((void**)&d_cnt, c->h_cnt, 0));
  uint8_t* d_l = (uint8_t*)c->raw.p;
  uint8_t* d_r = d_l + n * npairs;
  // Page-locked images are fetched by a kernel (one launch, both sides) instead of two copy-engine submissions; the
  // byte count is a multiple of 16 (W is), the pointers must be 16-byte aligned (gpc_hip_host_alloc's are)
  const void* vL = c->upload_mode ? device_view_of_host(rawL) : nullptr;
  const void* vR = vL ? device_view_of_host(rawR) : nullptr;
  const bool aligned = (((uintptr_t)vL | (uintptr_t)vR | (uintptr_t)d_l | (uintptr_t)d_r) & 15u) == 0 && (n * npairs) % 16 == 0 &&
                       n * npairs / 16 < (1ull << 31);
  if (vL && vR && aligned && c->upload_mode == 2) {  // the preprocess kernel reads the host's pages itself
    d_l = (uint8_t*)vL;
    d_r = (uint8_t*)vR;
  } else if (vL && vR && aligned) {
    const unsigned n16 = (unsigned)(n * npairs / 16);
    hipLaunchKernelGGL(gpc::k_upload2, dim3((n16 + 255) / 256, 2), dim3(256), 0, c->stream, (const uint4*)vL, (const uint4*)vR,
                       (uint4*)d_l, (uint4*)d_r, n16);
    HIPCHK(c, hipGetLastError());
  } else {  // pageable images (or unaligned ones): CPU copy into the page-locked arena, one upload launch from there
    uint8_t* d_arena = nullptr;
    const size_t side = pad16(n * npairs);
    CHK(xfer_reserve(c, 2 * side, &d_arena));
    memcpy(c->h_xfer, rawL, n * npairs);
    memcpy(c->h_xfer + side, rawR, n * npairs);
    const unsigned n16 = (unsigned)(side / 16);
    CHK(ensure(c, c->raw, 2 * side));
    d_l = (uint8_t*)c->raw.p;
    d_r = d_l + side;
    hipLaunchKernelGGL(gpc::k_upload2, dim3((n16 + 255) / 256, 2), dim3(256), 0, c->stream, (const uint4*)d_arena,
                       (const uint4*)(d_arena + side), (uint4*)d_l, (uint4*)d_r, n16);
    HIPCHK(c, hipGetLastError());
  }
  CHK(run_preprocess(c, d_l, d_r, W, H, npairs, 2, s->gradient_threshold, true));
  CHK(run_hash(c, (const uint8_t*)c->smooth.p, (const uint8_t*)c->grad.p, nullptr, W, H, 2 * npairs, false, (uint32_t*)c->codes.p));
  CHK(run_match(c, W, H, npairs, s, 0, (const uint8_t*)c->grad.p, d_out_host, cap, d_cnt, d_cnt + npairs));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  CHK(check_join_err(c));
  int status = GPC_OK;
  for (int p = 0; p < npairs; ++p) {
    counts[p] = c->h_cnt[p];
    if (counts[p] > cap) status = GPC_E_CAPACITY;
    if (ncand) {
      ncand[2 * p] = c->h_cnt[npairs + 2 * p];
      ncand[2 * p + 1] = c->h_cnt[npairs + 2 * p + 1];
    }
  }
  return status;
}

int gpc_hip_fed_calls(const gpc_hip_ctx* c) { return c ? c->fed_calls : 0; }

int gpc_hip_match_pair_begin(gpc_hip_ctx* c, const uint8_t* rawL, const uint8_t* rawR, int W, int H, const gpc_settings* s);

int gpc_hip_match_batch(gpc_hip_ctx* c, const uint8_t* rawL, const uint8_t* rawR, int W, int H, int npairs,
                        const gpc_settings* s, gpc_support* out, int cap, int32_t* counts, int32_t* ncand) {
  // One pair into a pageable array: the two-step form (results packed over the link, expanded by the workers): 0.20 ms
  // where the chunk pipeline's machinery took 0.27.  (A page-locked array is written by the kernels themselves: below.)
  if (c && npairs == 1 && rawL && rawR && out && counts && cap > 0 && s && !device_view_of_host(out)) {
    CHK(gpc_hip_match_pair_begin(c, rawL, rawR, W, H, s));
    int n = 0, nl = 0, nr = 0;
    const int st1 = match_fetch(c, out, cap, &n, &nl, &nr);
    c->pend.active = false;
    counts[0] = n;
    if (ncand) { ncand[0] = nl; ncand[1] = nr; }
    return st1;
  }
  const int st = on_gpu_node(c, npairs, [&] { return match_batch_packed(c, rawL, rawR, W, H, npairs, s, out, cap, counts, ncand); });
  if (c && st != GPC_OK && st != GPC_E_CAPACITY && st != GPC_E_INVALID) {
    // An error left the chunk pipeline half way: expansion jobs may still write into `out`, copies may still
    // target the staging slots.  Nothing of this call may be in flight when the caller gets its buffers back.
    c->pool.wait_all();
    if (c->s_in) {
      (void)hipStreamSynchronize(c->s_in);
      (void)hipStreamSynchronize(c->s_out);
      (void)hipStreamSynchronize(c->s_cnt);
    }
    if (c->s_aux) (void)hipStreamSynchronize(c->s_aux);
    (void)hipStreamSynchronize(c->stream);
  }
  return st;
}

static int match_batch_packed(gpc_hip_ctx* c, const uint8_t* rawL, const uint8_t* rawR, int W, int H, int npairs,
                              const gpc_settings* s, gpc_support* out, int cap, int32_t* counts, int32_t* ncand,
                              const PackedHost* ph) {
  if (!c || !rawL || !rawR || (!out && !ph) || !counts || npairs <= 0 || cap <= 0) return GPC_E_INVALID;
  if (ph && (!ph->packed || !ph->rows)) return GPC_E_INVALID;
  const double t_entry = host_ms();
  c->stage_ms[0] = c->stage_ms[1] = c->stage_ms[2] = c->stage_ms[3] = 0.f;
  CHK(check_settings(s));
  if (!s->epipolar_mode || s->use_hashtable) {
    if (ph) return GPC_E_UNSUPPORTED;  // rows are the unit of the packed format: the epipolar sort-matcher's
    return match_batch_unpacked(c, rawL, rawR, W, H, npairs, s, out, cap, counts, ncand);
  }
  CHK(check_dims(W, H));
  CHK(forest_matches(c, W, H));
  HIPCHK(c, hipSetDevice(c->device));
  if (!ph && npairs <= c->direct_max && (uint64_t)cap * sizeof(gpc_support) < (1ull << 32))
    if (void* dv = device_view_of_host(out))
      return match_batch_direct(c, rawL, rawR, W, H, npairs, s, (gpc_support*)dv, cap, counts, ncand);
  const size_t n = (size_t)W * H;
  int chunk = npairs < 4 ? npairs : (npairs / 8 < 1 ? 1 : (npairs / 8 > 16 ? 16 : npairs / 8));
  if (c->chunk_pairs > 0) chunk = c->chunk_pairs < npairs ? c->chunk_pairs : npairs;
  // Chunk schedule: full chunks in the middle; the FIRST one is split 1/4 + 3/4 (kernels start after a quarter of an
  // upload) and the LAST one 1/2 + 1/4 + 1/4 (the tail after the last upload -- its kernels, download and expansion --
  // covers 4 pairs instead of 16).  Slots are sized for a full chunk.
  std::vector<int> c_start, c_size;
  {
    int rem = npairs, at = 0;
    auto put = [&](int m) {
      if (m <= 0) return;
      c_start.push_back(at);
      c_size.push_back(m);
      at += m;
      rem -= m;
    };
    // (chunks of 8 split further cost more per-chunk overhead than they hide: 64 pairs 2.05 vs 1.67 ms; 256 pairs, chunks of
    // 16: 5.33 vs 5.87 ms)
    const bool shaped = !c->flat_chunks && chunk >= 16 && npairs >= 4 * chunk;
    if (shaped) {
      put(chunk / 4);
      put(chunk - chunk / 4);
    }
    while (rem > chunk) put(chunk);
    if (shaped && rem >= 4) {
      const int r = rem;
      put(r / 2);
      put(r / 4);
      put(r - r / 2 - r / 4);
    } else {
      put(rem);
    }
  }
  const int nch = (int)c_size.size();
  // (three-byte records x | (x - xR + dispHigh) << xbits were tried for the link: byte stores on the device and a
  // byte shuffle on the host made the call slower, 7.5 vs 6.2 ms per 256 pairs; the link is not the limit any more)
  const size_t rec = 4;
  const size_t hpad = ((size_t)H + 3) & ~(size_t)3;
  // a chunk's results: [row counts: chunk x hpad words | the pairs' records back to back | pad] -- one copy per chunk
  const size_t cb = 4 * hpad * chunk + ((rec * (size_t)cap * chunk + 15) & ~(size_t)15) + 16;
  CHK(ensure(c, c->raw, 2 * 2 * n * chunk));  // two slots x two sides
  CHK(ensure(c, c->packed, 3 * cb + sizeof(int32_t) * 3 * chunk));
  CHK(ensure(c, c->counts, sizeof(int32_t) * npairs));
  CHK(ensure(c, c->ncand, sizeof(int32_t) * 2 * npairs));
  const size_t stage_bytes = 4 * cb;
  if (stage_bytes > c->h_stage_cap) {
    c->pool.wait_all();
    if (c->h_stage) HIPCHK(c, hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->h_stage_cap = 0;
    HIPCHK(c, hipHostMalloc(&c->h_stage, stage_bytes, hipHostMallocDefault));
    c->h_stage_cap = stage_bytes;
  }
  CHK(pinned_counts(c, npairs));
  int32_t* hc = c->h_cnt;               // counts
  int32_t* hn = c->h_cnt + npairs;      // candidate counts
  CHK(batch_streams(c));
  // Pageable input images (malloc, std::vector, ndb::Buffer): a copy engine cannot read them, and hipMemcpyAsync then stages
  // them itself, synchronously, on the calling thread -- which serialised the whole chunk pipeline (32 pairs: 7.5 ms
  // against 1.1 ms from page-locked memory).  The workers copy a chunk into a page-locked bounce buffer instead (two
  // slots) while the device works on the chunks before it, and the upload runs from there.
  // (from four pairs on: for one pair waking the workers costs more than the runtime's own staging)
  // (every pageable batch, one pair included: memory hipMemcpy has seen stalls the queues when its owner frees it --
  // gpc_hip_ctx::h_xfer)
  const bool bounce = !device_view_of_host(rawL) || !device_view_of_host(rawR);
  if (bounce) {
    const size_t need = 2 * 2 * n * (size_t)chunk;
    if (need > c->h_in_cap) {
      HIPCHK(c, hipStreamSynchronize(c->s_in));
      if (c->h_in) HIPCHK(c, hipHostFree(c->h_in));
      c->h_in = nullptr;
      c->h_in_cap = 0;
      HIPCHK(c, hipHostMalloc(&c->h_in, need, hipHostMallocDefault));
      c->h_in_cap = need;
    }
  }
  {  // numThreads_ of the reference's settings asks for that many workers; otherwise what the process may use
    // (measured, 256 pairs: 3 .. 10 workers all keep up with the link -- 8.7 .. 8.9 ms per call, the link's 57 GB/s
    // shared by both directions being the limit; 14 workers on a 16-CPU share: 10.9 ms)
    int nt = c->expand_threads > 0 ? c->expand_threads : (s->num_threads > 1 ? s->num_threads : default_expand_threads());
    nt = nt < 1 ? 1 : (nt > 32 ? 32 : nt);
    c->pool.start(nt, c->have_node_cpus ? &c->node_cpus : nullptr, &c->node_l3);
  }
  uint8_t* d_pk = (uint8_t*)c->packed.p;
  uint8_t* h_pk = (uint8_t*)c->h_stage;
  int status = GPC_OK;
  int32_t* d_tot = reinterpret_cast<int32_t*>(d_pk + 3 * cb);  // [3][chunk] scratch of k_pair_totals
  // the counts of chunk k are on their way: wait for them, then fetch the chunk's row counts and records in one copy
  auto download = [&](int k) -> int {
    const int p0 = c_start[k], pc = c_size[k];
    HIPCHK(c, hipEventSynchronize(c->e_cnt[k & 3]));
    memcpy(counts + p0, hc + p0, sizeof(int32_t) * pc);
    if (ncand) memcpy(ncand + 2 * p0, hn + 2 * p0, sizeof(int32_t) * 2 * pc);
    c->pool.wait_slot(k & 3);  // the expansion of chunk k-4 has left this landing slot
    size_t recs = 0;
    for (int i = 0; i < pc; ++i) {
      const int cnt = counts[p0 + i];
      if (cnt > cap) status = GPC_E_CAPACITY;
      recs += (size_t)(cnt < cap ? cnt : cap);
    }
    const size_t bytes = 4 * hpad * chunk + rec * recs;
    HIPCHK(c, hipMemcpyAsync(h_pk + (size_t)(k & 3) * cb, d_pk + (size_t)(k % 3) * cb, bytes, hipMemcpyDeviceToHost, c->s_out));
    HIPCHK(c, hipEventRecord(c->e_out[k & 3], c->s_out));
    return GPC_OK;
  };
  auto expand = [&](int k) -> int {
    const int p0 = c_start[k], pc = c_size[k];
    HIPCHK(c, hipEventSynchronize(c->e_out[k & 3]));
    const int parts = c->pool.size() >= 8 ? 4 : 2;
    const uint8_t* slot = h_pk + (size_t)(k & 3) * cb;
    const uint8_t* recs = slot + 4 * hpad * chunk;
    for (int i = 0; i < pc; ++i) {
      const int32_t* rows = reinterpret_cast<const int32_t*>(slot) + (size_t)i * hpad;
      const int cnt = counts[p0 + i];
      const long limit = cnt < cap ? cnt : cap;
      long first = 0;
      if (ph) {  // the pair's words and row counts as they came over the link: two plain copies instead of the expansion
        ExpandJob j = {};
        j.slot = k & 3;
        int32_t* rdst = ph->rows + (size_t)(p0 + i) * H;   // rows 13 .. H-14 are the device's; the margins hold no support
        memset(rdst, 0, sizeof(int32_t) * GPC_R);
        memset(rdst + H - GPC_R, 0, sizeof(int32_t) * GPC_R);
        j.copy_src = rows + GPC_R;
        j.copy_dst = rdst + GPC_R;
        j.copy_bytes = sizeof(int32_t) * (size_t)(H - 2 * GPC_R);
        c->pool.push(j);
        if (limit > 0) {
          j.copy_src = recs;
          j.copy_dst = ph->packed + (size_t)(p0 + i) * cap;
          j.copy_bytes = rec * (size_t)limit;
          c->pool.push(j);
        }
        recs += rec * (size_t)limit;
        continue;
      }
      for (int q = 0; q < parts; ++q) {
        const int y0 = GPC_R + (int)((long)(H - 2 * GPC_R) * q / parts), y1 = GPC_R + (int)((long)(H - 2 * GPC_R) * (q + 1) / parts);
        if (first < limit && y1 > y0)
          c->pool.push(ExpandJob{reinterpret_cast<const uint32_t*>(recs), rows, H, y0, y1, (int)first, (int)limit,
                                 out + (size_t)(p0 + i) * cap, k & 3});
        for (int y = y0; y < y1; ++y) first += rows[y];
      }
      recs += rec * (size_t)limit;  // the next pair's records follow directly
    }
    return GPC_OK;
  };
  for (int k = 0; k < nch; ++k) {
    const int ev = k & 3, p0 = c_start[k], pc = c_size[k];
    uint8_t* d_l = (uint8_t*)c->raw.p + (size_t)(k & 1) * 2 * n * chunk;
    uint8_t* d_r = d_l + n * chunk;
    if (k >= 2) HIPCHK(c, hipStreamWaitEvent(c->s_in, c->e_comp[(k - 2) & 3], 0));  // chunk k-2 has read this slot
    const uint8_t *srcL = rawL + (size_t)p0 * n, *srcR = rawR + (size_t)p0 * n;
    if (bounce) {
      uint8_t* bl = (uint8_t*)c->h_in + (size_t)(k & 1) * 2 * n * chunk;
      uint8_t* br = bl + n * chunk;
      if (k >= 2) HIPCHK(c, hipEventSynchronize(c->e_in[(k - 2) & 3]));  // the upload of chunk k-2 has left this bounce slot
      const int pieces = c->pool.size() > 1 ? c->pool.size() : 1;
      const size_t bytes = n * pc, step = (bytes / pieces + 4095) & ~(size_t)4095;
      for (int side = 0; side < 2; ++side)
        for (size_t at = 0; at < bytes; at += step) {
          ExpandJob j = {};
          j.slot = 7;  // a wait slot of its own: the landing slots of the results use 0 .. 3
          j.copy_src = (side ? srcR : srcL) + at;
          j.copy_dst = (side ? br : bl) + at;
          j.copy_bytes = at + step <= bytes ? step : bytes - at;
          c->pool.push(j);
        }
      c->pool.wait_slot(7);
      srcL = bl;
      srcR = br;
    }
    HIPCHK(c, hipMemcpyAsync(d_l, srcL, n * pc, hipMemcpyHostToDevice, c->s_in));
    HIPCHK(c, hipMemcpyAsync(d_r, srcR, n * pc, hipMemcpyHostToDevice, c->s_in));
    HIPCHK(c, hipEventRecord(c->e_in[ev], c->s_in));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->e_in[ev], 0));
    if (k >= 3) HIPCHK(c, hipStreamWaitEvent(c->stream, c->e_out[(k - 3) & 3], 0));  // chunk k-3 has left this result slot
    uint8_t* slot = d_pk + (size_t)(k % 3) * cb;
    CHK(ensure(c, c->codes, sizeof(uint32_t) * n * 2 * pc));
    CHK(run_preprocess(c, d_l, d_r, W, H, pc, 2, s->gradient_threshold, true));
    CHK(run_hash(c, (const uint8_t*)c->smooth.p, (const uint8_t*)c->grad.p, nullptr, W, H, 2 * pc, false, (uint32_t*)c->codes.p));
    const PackedOut po = {reinterpret_cast<int32_t*>(slot), 0l, (long)hpad, d_tot + (size_t)(k % 3) * chunk};
    CHK(run_match(c, W, H, pc, s, 2, (const uint8_t*)c->grad.p, slot + 4 * hpad * chunk, cap, (int32_t*)c->counts.p + p0,
                  (int32_t*)c->ncand.p + 2 * p0, &po));
    HIPCHK(c, hipEventRecord(c->e_comp[ev], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->s_cnt, c->e_comp[ev], 0));
    HIPCHK(c, hipMemcpyAsync(hc + p0, (int32_t*)c->counts.p + p0, sizeof(int32_t) * pc, hipMemcpyDeviceToHost, c->s_cnt));
    HIPCHK(c, hipMemcpyAsync(hn + 2 * p0, (int32_t*)c->ncand.p + 2 * p0, sizeof(int32_t) * 2 * pc, hipMemcpyDeviceToHost, c->s_cnt));
    HIPCHK(c, hipEventRecord(c->e_cnt[ev], c->s_cnt));
    if (k >= 1) CHK(download(k - 1));
    if (k >= 2) CHK(expand(k - 2));
  }
  HIPCHK(c, hipEventSynchronize(c->e_in[(nch - 1) & 3]));  // (queued long before the kernels that follow it: no wait is added)
  c->stage_ms[0] = (float)(host_ms() - t_entry);
  CHK(download(nch - 1));
  c->stage_ms[1] = (float)(host_ms() - t_entry);
  if (nch >= 2) CHK(expand(nch - 2));
  CHK(expand(nch - 1));
  c->stage_ms[2] = (float)(host_ms() - t_entry);
  c->pool.wait_all();
  HIPCHK(c, hipStreamSynchronize(c->s_cnt));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  CHK(check_join_err(c));
  c->stage_ms[3] = (float)(host_ms() - t_entry);
  return status;
}

int gpc_hip_match_batch_packed(gpc_hip_ctx* c, const uint8_t* rawL, const uint8_t* rawR, int W, int H, int npairs,
                               const gpc_settings* s, uint32_t* packed, int cap, int32_t* rows, int32_t* counts, int32_t* ncand) {
  const PackedHost ph = {packed, rows};
  const int st = on_gpu_node(c, npairs, [&] { return match_batch_packed(c, rawL, rawR, W, H, npairs, s, nullptr, cap, counts, ncand, &ph); });
  if (c && st != GPC_OK && st != GPC_E_CAPACITY && st != GPC_E_INVALID && st != GPC_E_UNSUPPORTED) {
    c->pool.wait_all();   // (as in gpc_hip_match_batch: nothing of a failed call may still be in flight)
    if (c->s_in) {
      (void)hipStreamSynchronize(c->s_in);
      (void)hipStreamSynchronize(c->s_out);
      (void)hipStreamSynchronize(c->s_cnt);
    }
    (void)hipStreamSynchronize(c->stream);
  }
  return st;
}

int gpc_hip_host_threads(const gpc_hip_ctx* c) { return c ? c->pool.size() : 0; }

int gpc_hip_batch_stages(const gpc_hip_ctx* c, float* ms4) {
  if (!c || !ms4) return GPC_E_INVALID;
  for (int i = 0; i < 4; ++i) ms4[i] = c->stage_ms[i];
  return GPC_OK;
}

int gpc_hip_host_worker_cpus(gpc_hip_ctx* c, int* cpus, int cap) {
  if (!c || (cap > 0 && !cpus) || cap < 0) return 0;
  return c->pool.worker_cpus(cpus, cap);
}
int gpc_hip_host_numa_node(const gpc_hip_ctx* c) { return (c && c->have_node_cpus) ? c->numa_node : -1; }

// The whole timed region of samples/sparsematch.cpp:45-52 for one pair, queued: images through the arena (or read where
// they lie when page-locked), the batched pipeline for one pair, results into the arena for gpc_hip_match_fetch.
int gpc_hip_match_pair_begin(gpc_hip_ctx* c, const uint8_t* rawL, const uint8_t* rawR, int W, int H, const gpc_settings* s) {
  if (!c || !rawL || !rawR) return GPC_E_INVALID;
  CHK(check_settings(s));
  CHK(check_dims(W, H));
  CHK(forest_matches(c, W, H));
  HIPCHK(c, hipSetDevice(c->device));
  c->pend.active = false;
  c->pre_slot = -1;
  const size_t n = (size_t)W * H;
  const int cap_dev = (W - 2 * GPC_R) * (H - 2 * GPC_R) + 1;  // no pair has more supports than an image has candidates
  const bool packed = s->epipolar_mode && !s->use_hashtable && !c->no_pair_packed;  // (as in match_preprocessed_begin)
  const size_t rows_bytes = pad16(sizeof(int32_t) * (size_t)H);
  const size_t out_bytes = packed ? rows_bytes + pad16(sizeof(uint32_t) * (size_t)cap_dev) : pad16(sizeof(gpc_support) * (size_t)cap_dev);
  CHK(ensure(c, c->raw, 2 * n));
  CHK(ensure(c, c->codes, sizeof(uint32_t) * n * 2));
  CHK(pinned_counts(c, 1));
  int32_t* d_cnt = nullptr;
  HIPCHK(c, hipHostGetDevicePointer((void**)&d_cnt, c->h_cnt, 0));
  uint8_t* d_arena = nullptr;
  CHK(xfer_reserve(c, out_bytes + 2 * n, &d_arena));
  const uint8_t* vL = static_cast<const uint8_t*>(device_view_of_host(rawL));
  const uint8_t* vR = vL ? static_cast<const uint8_t*>(device_view_of_host(rawR)) : nullptr;
  if (!vL || !vR || (((uintptr_t)vL | (uintptr_t)vR) & 15u)) {
    memcpy(c->h_xfer + out_bytes, rawL, n);
    memcpy(c->h_xfer + out_bytes + n, rawR, n);
    vL = d_arena + out_bytes;
    vR = vL + n;
  }
  uint8_t* d_l = (uint8_t*)c->raw.p;
  uint8_t* d_r = d_l + n;
  const unsigned n16 = (unsigned)(n / 16);
  hipLaunchKernelGGL(gpc::k_upload2, dim3((n16 + 255) / 256, 2), dim3(256), 0, c->stream, (const uint4*)vL, (const uint4*)vR,
                     (uint4*)d_l, (uint4*)d_r, n16);
  HIPCHK(c, hipGetLastError());
  CHK(run_preprocess(c, d_l, d_r, W, H, 1, 2, s->gradient_threshold, true));
  CHK(run_hash(c, (const uint8_t*)c->smooth.p, (const uint8_t*)c->grad.p, nullptr, W, H, 2, false, (uint32_t*)c->codes.p));
  if (packed) {
    const PackedOut po = {reinterpret_cast<int32_t*>(d_arena), (long)cap_dev, (long)H};
    CHK(run_match(c, W, H, 1, s, 2, (const uint8_t*)c->grad.p, d_arena + rows_bytes, cap_dev, d_cnt, d_cnt + 1, &po));
  } else {
    CHK(run_match(c, W, H, 1, s, 0, (const uint8_t*)c->grad.p, d_arena, cap_dev, d_cnt, d_cnt + 1));
  }
  c->pend.active = true;
  c->pend.direct = false;
  c->pend.have_ncand = true;
  c->pend.packed = packed;
  c->pend.esz = sizeof(gpc_support);
  c->pend.cap_dev = cap_dev;
  c->pend.H = H;
  return GPC_OK;
}

int gpc_hip_match_pair(gpc_hip_ctx* c, const uint8_t* rawL, const uint8_t* rawR, int W, int H,
                       const gpc_settings* s, gpc_support* out, int cap, int* n_out, int* n_cand_l,
                       int* n_cand_r) {
  if (!n_out) return GPC_E_INVALID;
  int32_t cnt = 0, nc[2] = {0, 0};
  const int st = gpc_hip_match_batch(c, rawL, rawR, W, H, 1, s, out, cap, &cnt, nc);
  *n_out = cnt;
  if (n_cand_l) *n_cand_l = nc[0];
  if (n_cand_r) *n_cand_r = nc[1];
  return st;
}


// ------------------------------------------------------------------ warm-up

// Everything a first call would otherwise pay for inside the caller's timed region (the reference's sample starts its
// clock AFTER readForest, samples/sparsematch.cpp:42-45): the module's code objects, the workspaces of one pair of
// width x height, page-locked staging, streams, events and worker threads -- by running the host entry points once on a
// synthetic textured pair of that size (SURVEY 8d's generator) and throwing the results away.
// settings == NULL: the four matcher modes (epipolar x hashtable) with the reference's sparsematch thresholds.
int gpc_hip_warmup(gpc_hip_ctx* c, int W, int H, const gpc_settings* settings) {
  if (!c) return GPC_E_INVALID;
  CHK(check_dims(W, H));
  CHK(forest_matches(c, W, H));
  if (settings) CHK(check_settings(settings));
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)W * H;
  auto mix = [](uint32_t a, uint32_t b) {
    uint32_t h = a * 73856093u ^ b * 19349663u;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
  };
  // page-locked images and results: the single-pair path of gpc_hip_match_batch that Forest::matchPair takes
  uint8_t* pin = nullptr;
  const size_t cap = n / 2 + 1;
  HIPCHK(c, hipHostMalloc((void**)&pin, 2 * n + sizeof(gpc_support) * cap, hipHostMallocDefault));
  uint8_t *L = pin, *R = pin + n;
  gpc_support* pout = reinterpret_cast<gpc_support*>(pin + 2 * n);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W + 8; ++x) {
      const uint8_t v = (uint8_t)((((mix((uint32_t)x >> 2, (uint32_t)y >> 2) & 0xFF) * 3 + (mix((uint32_t)x, (uint32_t)y) & 0x3F)) >> 2));
      if (x < W) R[(size_t)y * W + x] = v;          // R(x) = P(x), L(x) = P(x + 8): disparity 8 everywhere
      if (x >= 8) L[(size_t)y * W + x - 8] = v;
    }
  int st = GPC_OK;
  {
    std::vector<uint8_t> sm[2] = {std::vector<uint8_t>(n), std::vector<uint8_t>(n)}, gr[2] = {std::vector<uint8_t>(n), std::vector<uint8_t>(n)};
    std::vector<int32_t> mk[2];
    int nm[2] = {0, 0};
    const uint8_t* raw[2] = {L, R};
    const int thr = settings ? settings->gradient_threshold : 5;
    const int maxcand = (W - 2 * GPC_R) * (H - 2 * GPC_R);
    for (int side = 0; side < 2 && st == GPC_OK; ++side) {
      st = gpc_hip_preprocess_begin(c, raw[side], W, H, thr);
      mk[side].resize((size_t)maxcand + 1);
      if (st == GPC_OK) st = gpc_hip_preprocess_fetch(c, sm[side].data(), gr[side].data(), mk[side].data(), maxcand, &nm[side]);
    }
    gpc_settings modes[4];
    int nmodes = 0;
    if (settings) modes[nmodes++] = *settings;
    else
      for (int k = 0; k < 4; ++k) modes[nmodes++] = gpc_settings{5, 128, 0, (k & 1) ? 0 : 1, (k >> 1) & 1, 1};
    std::vector<gpc_support> out((size_t)(nm[0] < nm[1] ? nm[0] : nm[1]) + 1);
    const int keep = c->resident_mode, hits = c->resident_hits;
    for (int k = 0; k < nmodes && st == GPC_OK; ++k) {
      int ns = 0;
      for (int pass = 0; pass < 2 && st == GPC_OK; ++pass) {  // from the resident images, then the upload path
        c->resident_mode = pass == 0 ? keep : 0;
        st = gpc_hip_rectified_match(c, sm[0].data(), gr[0].data(), mk[0].data(), nm[0], sm[1].data(), gr[1].data(), mk[1].data(),
                                     nm[1], W, H, &modes[k], out.data(), (int)out.size(), &ns);
      }
      c->resident_mode = keep;
      int32_t cnt = 0, nc[2];
      if (st == GPC_OK) st = gpc_hip_match_batch(c, L, R, W, H, 1, &modes[k], pout, (int)cap, &cnt, nc);                 // page-locked
      if (st == GPC_OK || st == GPC_E_CAPACITY) st = gpc_hip_match_batch(c, sm[0].data(), sm[1].data(), W, H, 1, &modes[k], out.data(), (int)out.size(), &cnt, nc);  // pageable
      if (st == GPC_OK || st == GPC_E_CAPACITY) st = gpc_hip_match_pair_begin(c, sm[0].data(), sm[1].data(), W, H, &modes[k]);   // Forest::matchPair
      if (st == GPC_OK) st = gpc_hip_match_fetch(c, out.data(), (int)out.size(), &ns, nullptr, nullptr);
      if (st == GPC_E_CAPACITY) st = GPC_OK;
    }
    c->resident_mode = keep;
    c->resident_hits = hits;
    std::lock_guard<std::mutex> g(g_res_mu);  // the host copies die with this scope
    c->res[0].valid = c->res[1].valid = false;
  }
  (void)hipHostFree(pin);
  return st;
}

// ------------------------------------------------------------------ fern training (k_train.h)

namespace {

void split_stats(int tp, int fp, int fn, int tot, double w1, gpc_split_stats* s) {
  // Fern.hpp:255-261, same expressions in the same order
  s->tp = tp;
  s->fp = fp;
  s->fn = fn;
  s->tot = tot;
  const double w2 = 1. - w1;
  s->prec = ((tp + fp) == 0) ? 0. : double(tp) / (tp + fp);
  s->rec = ((tp + fn) == 0) ? 0. : double(tp) / (tp + fn);
  s->hmean = (s->prec + s->rec == 0.) ? 0. : s->prec * s->rec / ((1. - w2) * s->prec + w2 * s->rec);
  s->convcomb = (1. - w2) * s->prec + w2 * s->rec;
}

int train_check(gpc_hip_ctx* c, gpc_hip_train_set* t) {
  if (!c || !t || t->owner != c) return GPC_E_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  return GPC_OK;
}

int train_scratch(gpc_hip_ctx* c, gpc_hip_train_set* t, size_t ncounts, size_t ncand) {
  if (ncounts > t->counts_cap) {
    if (t->counts) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(t->counts)); t->counts = nullptr; }
    HIPCHK(c, hipMalloc((void**)&t->counts, sizeof(int32_t) * ncounts));
    t->counts_cap = ncounts;
  }
  if (ncand > t->cand_cap) {
    if (t->d_cand) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(t->d_cand)); t->d_cand = nullptr; }
    HIPCHK(c, hipMalloc((void**)&t->d_cand, sizeof(gpc::GpcSplit) * ncand));
    t->cand_cap = ncand;
  }
  return GPC_OK;
}

bool split_ok(const gpc_split& p) { return p.i >= 0 && p.i < TS_PATCH && p.j >= 0 && p.j < TS_PATCH; }

}  // namespace

int gpc_hip_train_set_create(gpc_hip_ctx* c, const uint8_t* triplets, int n, gpc_hip_train_set** out) {
  if (!c || !triplets || n <= 0 || !out) return GPC_E_INVALID;
  *out = nullptr;
  HIPCHK(c, hipSetDevice(c->device));
  gpc_hip_train_set* t = new gpc_hip_train_set();
  t->owner = c;
  t->n = n;
  t->np = ((long)n + 255) / 256 * 256;
  uint8_t* d_aos = nullptr;
  const size_t bytes = (size_t)n * 3 * TS_PATCH;
  auto fail = [&](int st) {
    if (d_aos) (void)hipFree(d_aos);
    if (t->planes) (void)hipFree(t->planes);
    if (t->flags) (void)hipFree(t->flags);
    delete t;
    return st;
  };
  if (hipMalloc((void**)&d_aos, bytes) != hipSuccess || hipMalloc((void**)&t->planes, (size_t)3 * TS_PATCH * t->np) != hipSuccess ||
      hipMalloc((void**)&t->flags, (size_t)t->np) != hipSuccess) {
    snprintf(c->err, sizeof(c->err), "training set of %d triplets: device allocation failed", n);
    return fail(GPC_E_HIP);
  }
  if (hipMemcpyAsync(d_aos, triplets, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) return fail(GPC_E_HIP);
  hipLaunchKernelGGL(gpc::k_ts_transpose, dim3((unsigned)(t->np / 64), 3, 3), dim3(TS_THREADS), 0, c->stream,
                     (const uint8_t*)d_aos, n, t->np, t->planes);
  if (hipMemsetAsync(t->flags, 0, (size_t)t->np, c->stream) != hipSuccess) return fail(GPC_E_HIP);
  hipLaunchKernelGGL(gpc::k_ts_begin, dim3((unsigned)(t->np / TS_THREADS)), dim3(TS_THREADS), 0, c->stream, t->flags, n,
                     t->np, 1);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return fail(GPC_E_HIP);
  (void)hipFree(d_aos);
  c->train_sets.push_back(t);
  *out = t;
  return GPC_OK;
}

int gpc_hip_train_set_destroy(gpc_hip_ctx* c, gpc_hip_train_set* t) {
  if (!c || !t || t->owner != c) return GPC_E_INVALID;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (size_t k = 0; k < c->train_sets.size(); ++k)
    if (c->train_sets[k] == t) { c->train_sets.erase(c->train_sets.begin() + k); break; }
  if (t->planes) (void)hipFree(t->planes);
  if (t->flags) (void)hipFree(t->flags);
  if (t->counts) (void)hipFree(t->counts);
  if (t->d_cand) (void)hipFree(t->d_cand);
  delete t;
  return GPC_OK;
}

int gpc_hip_train_set_size(const gpc_hip_train_set* t) { return t ? t->n : 0; }

int gpc_hip_train_set_marks(gpc_hip_ctx* c, gpc_hip_train_set* t, const uint8_t* marks_in, uint8_t* marks_out) {
  CHK(train_check(c, t));
  uint8_t* d_m = nullptr;
  HIPCHK(c, hipMalloc((void**)&d_m, (size_t)t->n));
  const dim3 grid((unsigned)((t->n + TS_THREADS - 1) / TS_THREADS));
  int st = GPC_OK;
  if (marks_in) {
    if (hipMemcpyAsync(d_m, marks_in, (size_t)t->n, hipMemcpyHostToDevice, c->stream) != hipSuccess) st = GPC_E_HIP;
    hipLaunchKernelGGL(gpc::k_ts_set_marks, grid, dim3(TS_THREADS), 0, c->stream, t->flags, (const uint8_t*)d_m, t->n);
  }
  if (marks_out && st == GPC_OK) {
    hipLaunchKernelGGL(gpc::k_ts_get_marks, grid, dim3(TS_THREADS), 0, c->stream, (const uint8_t*)t->flags, d_m, t->n);
    if (hipMemcpyAsync(marks_out, d_m, (size_t)t->n, hipMemcpyDeviceToHost, c->stream) != hipSuccess) st = GPC_E_HIP;
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess) st = GPC_E_HIP;
  (void)hipFree(d_m);
  if (st != GPC_OK) snprintf(c->err, sizeof(c->err), "training set marks: HIP copy failed");
  return st;
}

int gpc_hip_train_eval_split(gpc_hip_ctx* c, gpc_hip_train_set* t, const gpc_split* params, int score_until_level,
                             double w1, gpc_split_stats* stats) {
  CHK(train_check(c, t));
  if (!params || !stats || score_until_level < 0) return GPC_E_INVALID;
  const int np_ = score_until_level + 1;
  if (np_ > 64) return GPC_E_UNSUPPORTED;
  for (int l = 0; l < np_; ++l)
    if (!split_ok(params[l])) return GPC_E_INVALID;
  CHK(train_scratch(c, t, 4, 64));
  HIPCHK(c, hipMemcpyAsync(t->d_cand, params, sizeof(gpc_split) * np_, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(t->counts, 0, sizeof(int32_t) * 4, c->stream));
  hipLaunchKernelGGL((gpc::k_ts_eval_split<false>), dim3((unsigned)(t->np / TS_THREADS)), dim3(TS_THREADS), 0, c->stream,
                     (const uint8_t*)t->planes, t->flags, t->n, t->np, (const gpc::GpcSplit*)t->d_cand, np_, t->counts);
  HIPCHK(c, hipGetLastError());
  int32_t h[4];
  HIPCHK(c, hipMemcpyAsync(h, t->counts, sizeof h, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  split_stats(h[0], h[1], h[2], h[3], w1, stats);
  return GPC_OK;
}

int gpc_hip_train_mark_split_samples(gpc_hip_ctx* c, gpc_hip_train_set* t, const gpc_split* params, int num_params) {
  CHK(train_check(c, t));
  if (num_params < 0 || (num_params > 0 && !params)) return GPC_E_INVALID;
  if (num_params > 64) return GPC_E_UNSUPPORTED;
  for (int l = 0; l < num_params; ++l)
    if (!split_ok(params[l])) return GPC_E_INVALID;
  CHK(train_scratch(c, t, 4, 64));
  if (num_params) HIPCHK(c, hipMemcpyAsync(t->d_cand, params, sizeof(gpc_split) * num_params, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL((gpc::k_ts_eval_split<true>), dim3((unsigned)(t->np / TS_THREADS)), dim3(TS_THREADS), 0, c->stream,
                     (const uint8_t*)t->planes, t->flags, t->n, t->np, (const gpc::GpcSplit*)t->d_cand, num_params,
                     (int32_t*)nullptr);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));  // params may be freed by the caller
  return GPC_OK;
}

int gpc_hip_train_begin_fern(gpc_hip_ctx* c, gpc_hip_train_set* t, int reset_marks) {
  CHK(train_check(c, t));
  hipLaunchKernelGGL(gpc::k_ts_begin, dim3((unsigned)(t->np / TS_THREADS)), dim3(TS_THREADS), 0, c->stream, t->flags, t->n,
                     t->np, reset_marks);
  HIPCHK(c, hipGetLastError());
  return GPC_OK;
}

int gpc_hip_train_eval_level(gpc_hip_ctx* c, gpc_hip_train_set* t, const gpc_split* cand, int ncand, int taulo, int tauhi,
                             int32_t* tp, int32_t* fp, int32_t* tot) {
  CHK(train_check(c, t));
  const int ntau = tauhi - taulo;
  if (!cand || ncand <= 0 || !tp || !fp || !tot) return GPC_E_INVALID;
  if (ntau <= 0 || ntau > TS_MAXTAU) return GPC_E_UNSUPPORTED;
  for (int k = 0; k < ncand; ++k)
    if (!split_ok(cand[k])) return GPC_E_INVALID;
  const size_t nc = (size_t)ncand * ntau;
  CHK(train_scratch(c, t, 2 * nc + 1, (size_t)ncand));
  HIPCHK(c, hipMemcpyAsync(t->d_cand, cand, sizeof(gpc_split) * ncand, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemsetAsync(t->counts, 0, sizeof(int32_t) * (2 * nc + 1), c->stream));
  {
    Timed tm(c, KID_TRAIN_EVAL);
    // longest chunks that still leave >= 4096 workgroups (16 per CU); measured on 1 Mi triplets: 16 Ki-triplet
    // chunks stream 1000 candidates at 7.3 TB/s (4 Ki: 6.4), but starve a 10-candidate launch
    int iters = 64;
    while (iters > 4 && ((t->np + (long)iters * TS_ITER - 1) / ((long)iters * TS_ITER)) * ncand < 4096) iters >>= 1;
    const unsigned chunks = (unsigned)((t->np + (long)iters * TS_ITER - 1) / ((long)iters * TS_ITER));
    hipLaunchKernelGGL(gpc::k_ts_eval_level, dim3(chunks, ncand), dim3(TS_THREADS), 0,
                       c->stream, (const uint8_t*)t->planes, (const uint8_t*)t->flags, t->np,
                       (const gpc::GpcSplit*)t->d_cand, taulo, ntau, iters, t->counts, t->counts + nc);
  }
  hipLaunchKernelGGL(gpc::k_ts_tot, dim3((unsigned)((t->np / 4 + TS_THREADS - 1) / TS_THREADS)), dim3(TS_THREADS), 0,
                     c->stream, (const uint8_t*)t->flags, t->np, t->counts + 2 * nc);
  HIPCHK(c, hipGetLastError());
  // one copy back: [tp | fp | tot] are contiguous on the device
  t->h_counts.resize(2 * nc + 1);
  HIPCHK(c, hipMemcpyAsync(t->h_counts.data(), t->counts, sizeof(int32_t) * (2 * nc + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(tp, t->h_counts.data(), sizeof(int32_t) * nc);
  memcpy(fp, t->h_counts.data() + nc, sizeof(int32_t) * nc);
  *tot = t->h_counts[2 * nc];
  return GPC_OK;
}

int gpc_hip_train_commit_level(gpc_hip_ctx* c, gpc_hip_train_set* t, const gpc_split* best, int mark_split) {
  CHK(train_check(c, t));
  if (!best || !split_ok(*best)) return GPC_E_INVALID;
  gpc::GpcSplit b = {best->i, best->j, best->tau};
  hipLaunchKernelGGL(gpc::k_ts_commit, dim3((unsigned)(t->np / TS_THREADS)), dim3(TS_THREADS), 0, c->stream,
                     (const uint8_t*)t->planes, t->flags, t->n, t->np, b, mark_split);
  HIPCHK(c, hipGetLastError());
  return GPC_OK;
}

int gpc_hip_train_fern(gpc_hip_ctx* c, gpc_hip_train_set* t, int max_depth, const gpc_split* cand, int num_resamples,
                       int taulo, int tauhi, int only_score_non_split, double w1, gpc_split* fernparams,
                       gpc_split_stats* level_stats) {
  CHK(train_check(c, t));
  if (max_depth <= 0 || !cand || num_resamples < 0 || !fernparams || !level_stats) return GPC_E_INVALID;
  if (max_depth > 64 || tauhi - taulo > TS_MAXTAU) return GPC_E_UNSUPPORTED;
  const int ntau = tauhi > taulo ? tauhi - taulo : 0;
  std::vector<int32_t> tp((size_t)num_resamples * ntau + 1), fp((size_t)num_resamples * ntau + 1);
  gpc_split_stats stats;
  memset(&stats, 0, sizeof stats);
  gpc_split best = {0, 0, 0};                       // SplitParams_t bestParams;            Fern.hpp:316
  for (int l = 0; l < max_depth; ++l) fernparams[l] = gpc_split{0, 0, 0};  // fernparams.resize(maxDepth)  :318
  CHK(gpc_hip_train_begin_fern(c, t, only_score_non_split));                // resetMarkOnSamples        :333
  for (int level = 0; level < max_depth; ++level) {
    float max_score = 0.f;
    int32_t tot = 0;
    if (num_resamples > 0 && ntau > 0)
      CHK(gpc_hip_train_eval_level(c, t, cand + (size_t)level * num_resamples, num_resamples, taulo, tauhi, tp.data(),
                                   fp.data(), &tot));
    for (int k = 0; k < num_resamples; ++k) {
      fernparams[level] = cand[(size_t)level * num_resamples + k];          // sampleHyperplane          :339
      for (int q = 0; q < ntau; ++q) {
        fernparams[level].tau = taulo + q;
        const int a = tp[(size_t)k * ntau + q], b = fp[(size_t)k * ntau + q];
        split_stats(a, b, tot - a - b, tot, w1, &stats);                   // evalSplit                  :343
        if (stats.hmean > max_score) {                                      // double against float       :346
          best = fernparams[level];
          max_score = (float)stats.hmean;
        }
      }
    }
    fernparams[level] = best;                                               // :352 (also when nothing scored above 0)
    CHK(gpc_hip_train_commit_level(c, t, &best, only_score_non_split));     // markSplitSamples(level)    :355
    level_stats[level] = stats;                                             // what train() prints        :357
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return GPC_OK;
}

// ------------------------------------------------------------------ measurement

// Host-only test hook (tests/cpp/sanitize_host.cpp: AddressSanitizer / UBSan / ThreadSanitizer builds of this file run on
// the CPU, no device): the expansion pool exactly as gpc_hip_match_batch drives it -- `npairs` pairs of packed records back
// to back (pair i: counts[i] supports, cut at `cap`; rows[i * hpad + y] of them in row y), every pair split into `parts`
// row ranges, `threads` workers, landing slots rotating like the chunks' -- plus the copy jobs of the pageable path.
// Not part of the C ABI (include/gpc_hip.h does not declare it).
extern "C" int gpc_hip_debug_expand_pool(const uint32_t* packed, const int32_t* rows, int hpad, int H, int npairs,
                                         const int32_t* counts, int cap, int threads, int parts, gpc_support* out,
                                         const uint8_t* copy_src, uint8_t* copy_dst, size_t copy_bytes) {
  if (!packed || !rows || !counts || !out || threads < 1 || parts < 1 || H < 2 * GPC_R + 1) return GPC_E_INVALID;
  ExpandPool pool;
  pool.start(threads, nullptr);
  const uint32_t* recs = packed;
  for (int i = 0; i < npairs; ++i) {
    const int32_t* r = rows + (size_t)i * hpad;
    const long limit = counts[i] < cap ? counts[i] : cap;
    long first = 0;
    if ((i & 3) == 3) pool.wait_slot((i + 1) & 3);  // (a landing slot is reused only when its jobs are done)
    for (int q = 0; q < parts; ++q) {
      const int y0 = GPC_R + (int)((long)(H - 2 * GPC_R) * q / parts), y1 = GPC_R + (int)((long)(H - 2 * GPC_R) * (q + 1) / parts);
      if (first < limit && y1 > y0)
        pool.push(ExpandJob{recs, r, H, y0, y1, (int)first, (int)limit, out + (size_t)i * cap, i & 3});
      for (int y = y0; y < y1; ++y) first += r[y];
    }
    recs += limit;
  }
  if (copy_bytes) {
    const size_t step = (copy_bytes / (size_t)threads + 4095) & ~(size_t)4095;
    for (size_t at = 0; at < copy_bytes; at += step) {
      ExpandJob j = {};
      j.slot = 7;
      j.copy_src = copy_src + at;
      j.copy_dst = copy_dst + at;
      j.copy_bytes = at + step <= copy_bytes ? step : copy_bytes - at;
      pool.push(j);
    }
    pool.wait_slot(7);
  }
  pool.wait_all();
  pool.stop();
  return GPC_OK;
}

// Host-only test hook (tests/cpp/sanitize_host.cpp, sanitizer builds on the CPU; not part of the C ABI): the fingerprint the
// resident-image records are held against (sampled or every byte) and the overlap test that drops them.
extern "C" int gpc_hip_debug_fingerprint(const uint8_t* smooth, const uint8_t* grad, size_t n, const int32_t* mask, int n_mask,
                                         int full, uint64_t* fp, const void* a, size_t na, const void* b, size_t nb, int* overlap) {
  if (!smooth || !grad || !fp || n_mask < 0 || (n_mask > 0 && !mask)) return GPC_E_INVALID;
  *fp = fingerprint(smooth, grad, n, mask, n_mask, full != 0);
  if (overlap) *overlap = ranges_overlap(a, na, b, nb) ? 1 : 0;
  return GPC_OK;
}

#ifdef GPC_WGLIFE
// diagnostic build only (tools/exp/join_wg_lives.py): per workgroup of the last k_row_join_fused launch start, end, rows | place
extern "C" int gpc_hip_debug_join_workgroups(gpc_hip_ctx* c, unsigned long long* out, int n_wg) {
  if (!c || !out || n_wg < 1 || n_wg > 4096) return GPC_E_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(gpc::g_rjf_wg), 3 * sizeof(unsigned long long) * (size_t)n_wg));
  return GPC_OK;
}
#endif
#ifdef GPC_STAMPS
// diagnostic build only: read and clear the s_memtime phase sums of k_row_join
extern "C" int gpc_hip_debug_stamps(gpc_hip_ctx* c, unsigned long long* out16) {
  if (!c || !out16) return GPC_E_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out16, HIP_SYMBOL(gpc::g_rj_stamps), 16 * sizeof(unsigned long long)));
  unsigned long long zero[16] = {0};
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(gpc::g_rj_stamps), zero, sizeof zero));
  return GPC_OK;
}
// the same for the phases of k_hash
extern "C" int gpc_hip_debug_htjoin_stamps(gpc_hip_ctx* c, unsigned long long* out16) {
  if (!c || !out16) return GPC_E_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out16, HIP_SYMBOL(gpc::g_hj_stamps), 16 * sizeof(unsigned long long)));
  const unsigned long long zero[16] = {0};
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(gpc::g_hj_stamps), zero, sizeof zero));
  return GPC_OK;
}
// per workgroup of the last k_hash launches: start, end (s_memrealtime), HW_ID | XCC_ID << 32 (3 words each, 8192 workgroups)
extern "C" int gpc_hip_debug_hash_workgroups(gpc_hip_ctx* c, unsigned long long* out, int n_wg) {
  if (!c || !out || n_wg < 1 || n_wg > 8192) return GPC_E_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(gpc::g_ht_wg), 3 * sizeof(unsigned long long) * (size_t)n_wg));
  return GPC_OK;
}
extern "C" int gpc_hip_debug_hash_stamps(gpc_hip_ctx* c, unsigned long long* out16) {
  if (!c || !out16) return GPC_E_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpyFromSymbol(out16, HIP_SYMBOL(gpc::g_ht_stamps), 16 * sizeof(unsigned long long)));
  unsigned long long zero[16] = {0};
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(gpc::g_ht_stamps), zero, sizeof zero));
  return GPC_OK;
}
#endif

int gpc_hip_enable_kernel_timing(gpc_hip_ctx* c, int enable) {
  if (!c) return GPC_E_INVALID;
  c->timing = enable != 0;
  return GPC_OK;
}

int gpc_hip_set_kernel_timing_mask(gpc_hip_ctx* c, unsigned mask) {
  if (!c) return GPC_E_INVALID;
  c->timing_mask = mask;
  return GPC_OK;
}

int gpc_hip_reset_kernel_timing(gpc_hip_ctx* c) {
  if (!c) return GPC_E_INVALID;
  CHK(drain_lanes(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (auto& s : c->spans) c->free_spans.push_back(s);
  c->spans.clear();
  return GPC_OK;
}

int gpc_hip_kernel_count(void) { return KID_COUNT; }

const char* gpc_hip_kernel_name(int index) {
  return (index >= 0 && index < KID_COUNT) ? kKernelNames[index] : "";
}

const char* gpc_hip_kernel_launch_name(const gpc_hip_ctx* c, int index) {
  return (c && index >= 0 && index < KID_COUNT) ? c->launch_name[index] : "";
}

int gpc_hip_kernel_time(gpc_hip_ctx* c, int index, float* total_ms, int* launches) {
  if (!c || index < 0 || index >= KID_COUNT) return GPC_E_INVALID;
  CHK(drain_lanes(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float tot = 0.f;
  int cnt = 0;
  for (auto& s : c->spans) {
    if (s.kid != index) continue;
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, s.a, s.b));
    tot += ms;
    ++cnt;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = cnt;
  return GPC_OK;
}

}  // extern "C"
