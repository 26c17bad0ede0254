// k_rowjoin.h -- epipolar-mode collision matching: per-row LDS hash join + counting rank.
//
// Replaces, for settings.epipolarMode_ == true, the descriptor build + `state |= y<<32`
// (inference.hpp:189-197), Forest::findCorrespondences (std::sort x2 + merge scan,
// inference.hpp:227-254) and the disparity filter of rectifiedMatch (inference.hpp:384-391).
// One image row per workgroup: with the row in the upper 32 state bits the reference's global
// sort is a per-row sort by code, and a source code can only meet target codes of its own row.
//
// What the reference's sort+merge decides for a row is, per code c:  cntL(c) == 1 and
// cntR(c) == 1  (with the tail-quirk variant cntR == 2 for the largest right code of the
// last populated right row); the sort is needed only for the ORDER of the output (ascending
// code).  So, entirely in LDS and without any compare-and-swap or sorting network:
//   1. the left row's codes are inserted into an ordered open-addressing table with
//      ds_max_rtn (Amble-Knuth ordered linear probing; a wave-level LDS CAS measured ~72
//      cycles of LDS pipe on MI355X, a returning ds_max ~8);
//   2. both rows look their code up (plain reads) and mark the slot: a returning ds_or sets the
//      side's SEEN flag, and a record that finds it already set adds the side's DUP flag; a right
//      record ORs its x into the (zeroed) low half of the same word with the same atomic: if the
//      right code is unique there was one writer, otherwise the value is not used;
//   3. every left record reads its slot: match iff neither side is DUP and the right side was
//      SEEN (+ disparity filter);
//   4. output position = rank of the code among the row's matches, by COUNTING on the top
//      bits the image's codes really use (k_hash ORs them into img_stats: bits a forest leaves
//      clear cost no resolution): one returning ds_add per match, an exclusive scan over the NT*SPT bucket counters
//      (DPP wave scan), and a look at the < 1 other matches sharing the bucket.
//      The thread still holds xL and xR, so it writes the packed support straight to its place.
//
// This header is the ONE-ROW-PER-WORKGROUP form: small launches, rows wider than 4096 pixels, the host entry point's gap-free
// packing (join + k_gather_rows as two launches) and, with VIRT, the partitions of the non-epipolar matcher.  Batched launches
// take k_rowjoin_fused.h: the same join as a persistent kernel that also writes the supports.
//
// 32-bit codes (WIDE): the SSE=OFF arithmetic with a 32-test forest sets bit 31, and the code
// 0xFFFFFFFF then collides with both sentinels (GPC_NOCAND in the code image, key 0 = code + 1
// in the table).  The WIDE instantiations read the candidate byte of such pixels to tell them
// apart and keep the one code that has no table key in three shared counters instead.
#pragma once
#include "gpc_device.h"

namespace gpc {

#define RJ_EMPTY 0xFFFFFFFFu
// flags of a table slot (high half of its word; the low half holds a right record's x)
#define RJ_LSEEN 0x00010000u
#define RJ_LDUP 0x00020000u
#define RJ_RSEEN 0x00040000u
#define RJ_RDUP 0x00080000u

// Diagnostic build only (-DGPC_STAMPS, tools/stamp_profile.py): s_memtime at phase boundaries,
// summed per phase into a debug buffer nothing else reads.  No stamp executes in the product build.
#ifdef GPC_STAMPS
__device__ unsigned long long g_rj_stamps[16];
#define RJ_STAMP(i)                                                                         \
  do {                                                                                      \
    unsigned long long t_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    rj_acc[i] = t_ - rj_t0;                                                                 \
    rj_t0 = t_;                                                                             \
  } while (0)
// one workgroup in 64 reports (uncontended atomics, issued after the last stamp)
#define RJ_STAMP_FLUSH()                                                                    \
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 5)                                           \
    for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_rj_stamps[i_], rj_acc[i_])
#define RJ_STAMP_INIT()                                                                     \
  unsigned long long rj_t0;                                                                 \
  unsigned long long rj_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                  \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rj_t0)::"memory")
#else
#define RJ_STAMP(i)
#define RJ_STAMP_INIT()
#define RJ_STAMP_FLUSH()
#endif

__device__ __forceinline__ uint32_t rj_hash(uint32_t key, int shift) { return (key * 0x9E3779B1u) >> shift; }

// ---------------------------------------------------------------- the join table
// Measured on MI355X (profiles/r01_ubench_lds_valu_calibration.txt): a wave-level LDS
// compare-and-swap costs ~72 cycles of the CU's LDS pipe, a returning ds_max/ds_add ~8, a plain
// random ds_read ~3.  So the table is built WITHOUT CAS: ordered linear probing (Amble & Knuth)
// with ds_max_rtn -- a probe writes max(slot, key); if it displaced a smaller key it carries
// that key on to the next slot.  Within one insert phase this converges to the unique ordered
// table whatever the interleaving; lookups (after the barrier) are read-only and stop at the
// first slot holding a smaller key.  Stored key = code + 1 (0 = empty slot).
// Only LEFT codes are inserted; right records merely look their code up.  A slot costs 8 bytes
// (key + one word of flags and x): 16 KiB per 1024-px row and 64 VGPRs, so EIGHT workgroups = 32
// waves share a CU.
//
// The two probe loops are written in gfx950 assembly: the lanes still probing narrow EXEC with
// v_cmpx and the loop ends on s_cbranch_execnz -- one scalar instruction per round where the
// compiler's structurised form of the same divergent loop spends five or six on exit masks
// (the kernel issued 0.72 scalar instructions per vector instruction before; PMC r01_k).

// continues the ordered insert of key `cur` whose first probe at slot h returned `o`
// (o == 0: slot was empty -> done; o == cur: already present -> done; o < cur: displaced o, carry
// it on; o > cur: keep cur and move on).  keys_lds = LDS byte offset of the key table.
__device__ __forceinline__ void rj_insert_chain(uint32_t keys_lds, uint32_t cur, uint32_t o, uint32_t h, uint32_t smask) {
  uint32_t addr;
  unsigned long long sv;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "v_cmpx_ne_u32_e32 vcc, 0, %[cur]\n\t"       // lanes without a record never probe on
      "v_cmpx_ne_u32_e32 vcc, 0, %[o]\n\t"
      "v_cmpx_ne_u32_e32 vcc, %[o], %[cur]\n\t"
      "s_cbranch_execz 2f\n"
      "1:\n\t"
      "v_min_u32_e32 %[cur], %[o], %[cur]\n\t"
      "v_add_u32_e32 %[h], 1, %[h]\n\t"
      "v_and_b32_e32 %[h], %[smask], %[h]\n\t"
      "v_lshl_add_u32 %[addr], %[h], 2, %[base]\n\t"
      "ds_max_rtn_u32 %[o], %[addr], %[cur]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmpx_ne_u32_e32 vcc, 0, %[o]\n\t"
      "v_cmpx_ne_u32_e32 vcc, %[o], %[cur]\n\t"
      "s_cbranch_execnz 1b\n"
      "2:\n\t"
      "s_mov_b64 exec, %[sv]"
      : [cur] "+v"(cur), [o] "+v"(o), [h] "+v"(h), [addr] "=&v"(addr), [sv] "=&s"(sv)
      : [smask] "s"(smask), [base] "s"(keys_lds)
      : "vcc", "memory");
}

// slot of key k on its probe path: larger keys sit in front of it.  kk = value of the first probe
// (pass 0 for a lane without a record).  Returns the slot where the walk stopped and, in kk, what
// it holds there: kk == k -> found.
__device__ __forceinline__ uint32_t rj_find_chain(uint32_t keys_lds, uint32_t k, uint32_t& kk, uint32_t h, uint32_t smask) {
  uint32_t addr;
  unsigned long long sv;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "v_cmpx_gt_u32_e32 vcc, %[kk], %[k]\n\t"
      "s_cbranch_execz 2f\n"
      "1:\n\t"
      "v_add_u32_e32 %[h], 1, %[h]\n\t"
      "v_and_b32_e32 %[h], %[smask], %[h]\n\t"
      "v_lshl_add_u32 %[addr], %[h], 2, %[base]\n\t"
      "ds_read_b32 %[kk], %[addr]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmpx_gt_u32_e32 vcc, %[kk], %[k]\n\t"
      "s_cbranch_execnz 1b\n"
      "2:\n\t"
      "s_mov_b64 exec, %[sv]"
      : [kk] "+v"(kk), [h] "+v"(h), [addr] "=&v"(addr), [sv] "=&s"(sv)
      : [k] "v"(k), [smask] "s"(smask), [base] "s"(keys_lds)
      : "vcc", "memory");
  return h;
}

// Occupancy the register allocator aims for: 256- and 512-thread rows keep 8 waves per SIMD (64
// VGPRs); 1024-thread rows with 8 / 16 pixel slots per thread (W > 4096: one workgroup per CU) get
// the 128 registers their 16 waves leave them.
template <int SPT, int NT>
struct RjOcc {
  static constexpr int kWaves = (SPT >= 8) ? 4 : 8;
};

// VIRT (k_partition.h): the "rows" are partitions of the non-epipolar matcher -- dense record arrays (code, pixel
// index) of a contiguous code range per side instead of two image rows; a record's "x" is its position in the
// partition, the disparity filter looks the pixel indices up, and results go to the partition's stretch of v.staged.
struct RjVirt {
  const uint2* kv;        // [npairs][recs]: records (code, pixel index y * W + x), side s of a pair at + s * (recs / 2).  One
                          // 8-byte element per record: the scatter that writes them leaves runs of ~5 records per bin and
                          // tile, and two 4-byte arrays made that twice as many partial cache lines (k_partition.h)
  int32_t* part;          // per pair (stride ps ints): cursors, partition offsets, match counts, misc (GpLayout)
  uint32_t* staged;       // [npairs][recs / 2] uint2: (left, right) pixel index of a partition's matches at its left offset
  long recs, ps;
  int o_off, o_rowcnt, o_misc, pmax;
  GpcDivW dw;
  int vtol;
  int use_list;           // 1: workgroup b takes partition part[o_misc + 8 + b] (the plan's list of over-large partitions: the grid
                          // of the 8192-record launch is that list, not every partition of which nearly all return at once --
                          // dispatching ~8000 workgroups of 128 KiB of LDS each cost 44-48 us per 8 pairs of 1920x1080)
  int min_recs;           // this launch takes the partitions with more than min_recs records on a side (and at most NT*SPT):
                          // the few bins that are large by themselves go to the 8192-slot instantiation, the rest to the 4096 one
};

// (The persistent variant that also writes the supports -- join + output in one launch -- is a kernel of its own:
// k_rowjoin_fused.h.)

// codes:   [npairs*2][H][W]   (image 2p = left, 2p+1 = right)
// cand:    [npairs*2][H][W]   candidate bytes (grad, or the caller's scattered mask): WIDE only
// staged:  [npairs][H][W]     packed (xL | xR<<16), first rowcnt entries of each row valid
// rowcnt:  [npairs][H]
// grid: (ceil((H - 26) / rpw), npairs); NT threads, NB = NT*SPT >= W; table of S = 1 << log2s slots,
//       S >= NB, S >= 2*(W-26) where the LDS allows it (only left codes are inserted: load factor
//       <= 0.5; rows beyond 8218 px fill a 16384-slot table up to W-26 / 16384 < 1)
// dynamic LDS: 8*(S+1) bytes  (16 KiB for W = 1024: 8 workgroups per CU = 32 waves, 64 VGPRs)
// Wide rows use more threads per row before more pixel slots per thread, so that the one
// or two workgroups that fit a CU (98 KiB of table at W = 3840) still fill its SIMDs.
template <int SPT, int NT, bool WIDE, bool VIRT = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(RjOcc<SPT, NT>::kWaves, 8))) void k_row_join(
    const uint32_t* __restrict__ codes, const uint8_t* __restrict__ cand, int W, int H, int disp_high, int apply_filter,
    const int32_t* __restrict__ img_stats, uint32_t* __restrict__ staged, int32_t* __restrict__ rowcnt,
    int log2s, int rpw, RjVirt v) {
#ifndef RJ_KEEP_RPW
  rpw = 1;  // (rows per workgroup with next-row prefetch measured within the noise of one row, docs/HISTORY.md 7, and its eight
            // prefetch registers put scratch into the 1024-thread instantiation: one row per workgroup it is)
#endif
  // flags of a table slot (one word per slot, x of a right record in the low half)
  constexpr uint32_t F_LSEEN = RJ_LSEEN, F_LDUP = RJ_LDUP, F_RSEEN = RJ_RSEEN, F_RDUP = RJ_RDUP, F_XMASK = 0xFFFFu;
  constexpr int NB = NT * SPT;
  extern __shared__ __attribute__((aligned(16))) uint32_t rj_lds[];
  __shared__ uint32_t s_max_key;
  __shared__ int s_tail_cnt;
  __shared__ unsigned s_tail_minx;
  __shared__ int s_sp_l, s_sp_r;   // WIDE: left / right candidates of this row whose code is 0xFFFFFFFF
  __shared__ unsigned s_sp_minx;   //       smallest x among the right ones
  __shared__ uint32_t s_w[NT / 64];
  __shared__ uint32_t s_cmin, s_cmax;      // smallest / largest matched code of the row: the rank buckets span that range
  __shared__ unsigned s_tail_xv, s_sp_xv;  // VIRT: position of the tail / key-less right record with the smallest pixel index
  const int S = 1 << log2s;
  uint32_t* t_key = rj_lds;               // [S]   stored key = code + 1, 0 = empty
  uint32_t* t_w = rj_lds + S + 1;         // [S]   per slot: seen / duplicate flags of either side, x of a right record in the low bits
  uint32_t* r_cnt = t_key;                // [NB+1] bucket counters -> starts   (reuses t_key, dead after step 2)
  uint32_t* r_key = t_w;                  // [NB]   matched codes, bucket-contiguous (reuses t_w, dead after step 3)
  // one returning OR on a slot's flag word; returns what the slot held
  auto mark = [&](uint32_t h, uint32_t val) -> uint32_t { return atomicOr(&t_w[h], val); };
  auto mark_noret = [&](uint32_t h, uint32_t val) { atomicOr(&t_w[h], val); };
  const uint32_t keys_lds = (uint32_t)(uintptr_t)t_key;  // low half of the flat address = LDS offset

  const int tid = threadIdx.x, lane = tid & 63;
  const int pair = (int)blockIdx.y;
  const int hshift = 32 - log2s;
  const uint32_t smask = (uint32_t)S - 1u;
  // VIRT: this workgroup's partition
  int32_t* vblk = nullptr;
  const uint2 *vrl = nullptr, *vrr = nullptr;  // the partition's records per side
  int v_nl = 0, v_nr = 0, v_offl = 0, v_p = 0;
  if (VIRT) {
    vblk = v.part + pair * v.ps;
    int p = blockIdx.x;
    if (v.use_list) {
      if (p >= vblk[v.o_misc + 3]) return;  // GP_NBIG
      p = vblk[v.o_misc + 8 + p];
    }
    v_p = p;
    if (p >= vblk[v.o_misc + 0]) return;  // GP_NPARTS
    v_offl = vblk[v.o_off + p];
    const int offr = vblk[v.o_off + v.pmax + 1 + p];
    v_nl = vblk[v.o_off + p + 1] - v_offl;
    v_nr = vblk[v.o_off + v.pmax + 1 + p + 1] - offr;
    if (v_nl > NB || v_nr > NB) return;  // another launch's partition (or k_gp_plan has raised the overflow flag: the host takes the radix path)
    if (max(v_nl, v_nr) <= v.min_recs) return;
    vrl = v.kv + pair * v.recs + v_offl;
    vrr = v.kv + pair * v.recs + v.recs / 2 + offr;
  }
  int last_r = 0;
  last_r = VIRT ? vblk[v.o_misc + 2] /* GP_LASTR */ : img_stats[(pair * 2 + 1) * GPC_STAT_STRIDE + GPC_STAT_LASTROW];
  // A workgroup handles `rpw` consecutive rows; the NEXT row's codes are fetched into registers
  // while the current row is joined, so only the first row's load latency is exposed.
  const int row0 = VIRT ? v_p : GPC_R + blockIdx.x * rpw;
  uint32_t ncl[SPT], ncr[SPT];
  uint32_t nspl = 0u, nspr = 0u;  // WIDE: bit j = pixel slot j is a candidate whose code is 0xFFFFFFFF
  auto fetch_row = [&](int yy) {
    nspl = nspr = 0u;
    if (VIRT) {
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
        const int x = j * NT + tid;
        ncl[j] = (x < v_nl) ? vrl[x].x : RJ_EMPTY;
        ncr[j] = (x < v_nr) ? vrr[x].x : RJ_EMPTY;
        if (WIDE) {  // every record is a candidate: 0xFFFFFFFF is the key-less code
          if (x < v_nl && ncl[j] == RJ_EMPTY) nspl |= 1u << j;
          if (x < v_nr && ncr[j] == RJ_EMPTY) nspr |= 1u << j;
        }
      }
      return;
    }
    const long ro = ((long)(pair * 2) * H + yy) * W;
    const uint32_t* rl = codes + ro;
    const uint32_t* rr_ = rl + (long)H * W;
    {
      // no branch around a load: with one the compiler sinks the key arithmetic into the branch and waits for every
      // pair of loads before it issues the next (four round trips per row instead of one)
      // (nor a clamp: pixel slots beyond W read into the next row -- the code image is allocated with that slack, the
      // host sees to it -- and are masked afterwards: one lane offset + an immediate per load)
      uint32_t tl[SPT], tr[SPT];
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
#ifdef RJ_DBG_NOLOAD     // (experiment, with RJ_DBG_EMPTY only: what the rows' loads cost)
        tl[j] = (uint32_t)(j * NT + tid) * 3u;
        tr[j] = (uint32_t)(j * NT + tid) * 5u;
#else
        tl[j] = rl[(uint32_t)(j * NT + tid)];
        tr[j] = rr_[(uint32_t)(j * NT + tid)];
#endif
      }
#pragma unroll
      for (int j = 0; j < SPT; ++j) {
        ncl[j] = (j * NT + tid < W) ? tl[j] : RJ_EMPTY;
        ncr[j] = (j * NT + tid < W) ? tr[j] : RJ_EMPTY;
      }
    }
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      const int x = j * NT + tid;
      if (WIDE) {
        // the hash kernel's candidate rule (k_hash.h): candidate byte set, inside the margin (the row is)
        const bool inm = x >= GPC_R && x < W - GPC_R;
        if (inm && ncl[j] == RJ_EMPTY && cand[ro + x]) nspl |= 1u << j;
        if (inm && ncr[j] == RJ_EMPTY && cand[ro + (long)H * W + x]) nspr |= 1u << j;
      }
    }
  };
  fetch_row(row0);
#pragma unroll 1
  for (int ri = 0; ri < rpw && (VIRT ? ri == 0 : row0 + ri < H - GPC_R); ++ri) {
  const int y = row0 + ri;
  RJ_STAMP_INIT();
#ifdef RJ_DBG_PADVALU  // calibration: how much of a row's time is VALU issue?  RJ_DBG_PADVALU x 8 dependent-free adds per wave and row
  {
    uint32_t p0 = tid, p1 = tid + 1, p2 = tid + 2, p3 = tid + 3;
    for (int q = 0; q < RJ_DBG_PADVALU; ++q)
      asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0\n\t"
                   "v_add_u32 %0, %0, %2\n\tv_add_u32 %1, %1, %3\n\tv_add_u32 %2, %2, %0\n\tv_add_u32 %3, %3, %1"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
    if ((p0 ^ p1 ^ p2 ^ p3) == 0x12345u && W < 0) rj_lds[0] = p0;  // (keeps the adds)
  }
#endif
#ifdef RJ_DBG_PADSALU
  {
    uint32_t q0 = (uint32_t)W, q1 = (uint32_t)H;
    for (int q = 0; q < RJ_DBG_PADSALU; ++q)
      asm volatile("s_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, %0\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, %0\n\t"
                   "s_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, %0\n\ts_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, %0"
                   : "+s"(q0), "+s"(q1) : : "scc");
    if ((q0 ^ q1) == 0x12345u && W < 0) rj_lds[0] = q0;
  }
#endif
  // ---- 0. this row's codes (already in flight), table clear, next row's loads
  uint32_t kl[SPT], kr[SPT];  // (the code itself is kl - 1 wherever it is needed: one register per slot less)
  uint32_t spl = 0u, spr = 0u;
  {
#pragma unroll
    for (int j = 0; j < SPT; ++j) {  // stored key = code + 1 (0 = no record in this pixel slot)
      kl[j] = ncl[j] + 1u;
      kr[j] = ncr[j] + 1u;
#ifdef RJ_DBG_EMPTY   // experiment: every pixel slot without a record -- what a row costs before it holds anything
      kl[j] = kr[j] = 0u * (ncl[j] + ncr[j]);
#endif
#ifdef RJ_DBG_LEFTONLY  // experiment: no right records (inserts and left lookups only)
      kr[j] = 0u * ncr[j];
#endif
    }
    if (WIDE) {
      spl = nspl;
      spr = nspr;
    }
    if (!VIRT && ri + 1 < rpw && y + 1 < H - GPC_R) fetch_row(y + 1);
    {  // 16-byte stores; the host rounds the allocation up to a multiple of 16 bytes
      uint4* z = reinterpret_cast<uint4*>(rj_lds);
      // the zeros are made HERE: as a plain constant the compiler keeps them in four registers across the whole row
      // and, at 64 VGPRs, spills them to scratch (one 16-byte store + load per thread and row)
      uint32_t z0;
      asm volatile("v_mov_b32 %0, 0" : "=v"(z0));
      const uint4 zero = make_uint4(z0, z0, z0, z0);
      const int nclear = (8 * (S + 1) + 15) / 16;
#ifndef RJ_DBG_NOCLEAR   // (experiment, with RJ_DBG_EMPTY only: what the table clear costs)
      for (int i = tid; i < nclear; i += NT) z[i] = zero;
#endif
    }
    if (tid == 0) {
      // (constants made here for the same reason as the zeros above: hoisted out of the row loop they are spilled to
      // scratch in the persistent instantiations, and the reload waits for every load in flight)
      uint32_t c0, c1;
      asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, -1" : "=v"(c0), "=v"(c1));
      s_cmin = c1;
      s_cmax = c0;
      s_max_key = c0;
      s_tail_cnt = (int)c0;
      s_tail_minx = c1;
      if (WIDE) {
        s_sp_l = (int)c0;
        s_sp_r = (int)c0;
        s_sp_minx = c1;
      }
    }
  }
  // Tail quirks of the reference's merge scan concern only the largest right
  // code of the last right row that has candidates: it matches iff it occurs exactly TWICE on
  // the right (then with the first of the two in mask order) and once on the left.
  const bool tail_row = (y == last_r);  // block-uniform
  __syncthreads();
  RJ_STAMP(0);

  // ---- 1. build the ordered table from the left codes (a slot without a record inserts key 0: a no-op)
  uint32_t h0l[SPT];
  {
    uint32_t old[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      // a pixel slot without a record inserts key 0 (a no-op) -- into a slot of its own: the atomics of
      // lanes that share an address are served one after the other
      h0l[j] = kl[j] ? rj_hash(kl[j], hshift) : ((uint32_t)(j * NT + tid) & smask);
      old[j] = atomicMax(&t_key[h0l[j]], kl[j]);
    }
#pragma unroll
    for (int j = 0; j < SPT; ++j) rj_insert_chain(keys_lds, kl[j], old[j], h0l[j], smask);
  }
  if (tail_row) {  // the largest right key of this row
    uint32_t max_k = 0;
#pragma unroll
    for (int j = 0; j < SPT; ++j) max_k = max(max_k, kr[j]);
    for (int o = 32; o > 0; o >>= 1) max_k = max(max_k, (uint32_t)__shfl_xor((int)max_k, o));
    if (lane == 0 && max_k) atomicMax(&s_max_key, max_k);
  }
  if (WIDE) {  // the code without a key: count its records on either side
    if (__ballot(spl != 0u) | __ballot(spr != 0u)) {
      if (spl) atomicAdd(&s_sp_l, __popc(spl));
      if (spr) {
        atomicAdd(&s_sp_r, __popc(spr));
        if (VIRT) {  // positions carry no order here: the smallest PIXEL INDEX is the first in mask order
#pragma unroll
          for (int j = 0; j < SPT; ++j)
            if ((spr >> j) & 1u) atomicMin(&s_sp_minx, vrr[j * NT + tid].y);
        } else {
          atomicMin(&s_sp_minx, (unsigned)((__ffs((int)spr) - 1) * NT + tid));
        }
      }
    }
  }
  __syncthreads();
  RJ_STAMP(1);
  if (WIDE && VIRT && spr) {  // which position holds the key-less right record with the smallest pixel index (read after the next barrier)
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (((spr >> j) & 1u) && vrr[j * NT + tid].y == s_sp_minx) s_sp_xv = (unsigned)(j * NT + tid);
  }

  // ---- 2. every record finds its code's slot (read-only) and marks it.  The marks of a side go out
  //      together (one LDS round trip for SPT returning atomics): a record without a slot ORs 0 into
  //      wherever its walk stopped, which changes nothing.
  uint32_t hl[SPT];
  {
    uint32_t h0r[SPT], f0l[SPT], f0r[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {  // first probes of all records together
      h0r[j] = kr[j] ? rj_hash(kr[j], hshift) : ((uint32_t)(j * NT + tid) & smask);
      f0l[j] = t_key[h0l[j]];
      f0r[j] = t_key[h0r[j]];
    }
    uint32_t seen[SPT];
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      uint32_t kk = kl[j] ? f0l[j] : 0u;
      hl[j] = rj_find_chain(keys_lds, kl[j], kk, h0l[j], smask);  // a left code is always found
    }
#pragma unroll
    for (int j = 0; j < SPT; ++j) seen[j] = mark(hl[j], kl[j] ? F_LSEEN : 0u);
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (kl[j] && (seen[j] & F_LSEEN)) mark_noret(hl[j], F_LDUP);  // a second left record of this code
    uint32_t hr[SPT], fr = 0u;
#pragma unroll
    for (int j = 0; j < SPT; ++j) {
      uint32_t kk = kr[j] ? f0r[j] : 0u;
      hr[j] = rj_find_chain(keys_lds, kr[j], kk, h0r[j], smask);
      if (kr[j] && kk == kr[j]) fr |= 1u << j;
    }
    // x goes into the zeroed low half with the same atomic: several writers only when the code is
    // not unique on the right, and then x is not used
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      seen[j] = mark(hr[j], ((fr >> j) & 1u) ? (F_RSEEN | (uint32_t)(j * NT + tid)) : 0u);
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (((fr >> j) & 1u) && (seen[j] & F_RSEEN)) mark_noret(hr[j], F_RDUP);
  }
  // the key the tail rule applies to; none when the row's largest right code is the key-less 0xFFFFFFFF
  uint32_t tail_key = 0u;
  bool tail_sp = false;
  if (tail_row) {  // block-uniform
    tail_sp = WIDE && s_sp_r > 0;
    tail_key = tail_sp ? 0u : s_max_key;
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if (kr[j] && kr[j] == tail_key) {
        atomicAdd(&s_tail_cnt, 1);
        atomicMin(&s_tail_minx, VIRT ? vrr[j * NT + tid].y : (unsigned)(j * NT + tid));
      }
    if (VIRT) {  // positions carry no order: find where the tail record with the smallest pixel index sits
      __syncthreads();
#pragma unroll
      for (int j = 0; j < SPT; ++j)
        if (kr[j] && kr[j] == tail_key && vrr[j * NT + tid].y == s_tail_minx) s_tail_xv = (unsigned)(j * NT + tid);
    }
  }
  __syncthreads();
  RJ_STAMP(2);
  // ---- 3. decide every left candidate; the key table is dead already: it becomes the rank counters
  {  // NB + 1 counters: SPT consecutive ones per thread (16-byte stores where SPT is 4: one LDS instruction instead of five)
    uint32_t z0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z0));
    if (SPT % 4 == 0) {
#pragma unroll
      for (int q = 0; q < SPT / 4; ++q) reinterpret_cast<uint4*>(r_cnt)[tid * (SPT / 4) + q] = make_uint4(z0, z0, z0, z0);
    } else {
#pragma unroll
      for (int q = 0; q < SPT; ++q) r_cnt[tid * SPT + q] = z0;
    }
    if (tid == 0) r_cnt[NB] = z0;
  }
  uint32_t okm = 0u;  // bit j = pixel slot j is a match
  uint32_t xr[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    xr[j] = 0u;
    bool good = false;
    if (kl[j]) {
      const uint32_t w = t_w[hl[j]];
      const bool tail = tail_row && kl[j] == tail_key;
      good = !(w & F_LDUP) && (tail ? (s_tail_cnt == 2) : ((w & (F_RSEEN | F_RDUP)) == F_RSEEN));
      xr[j] = tail ? (VIRT ? s_tail_xv : s_tail_minx) : (w & F_XMASK);
    } else if (WIDE && ((spl >> j) & 1u)) {
      good = (s_sp_l == 1) && (s_sp_r == (tail_sp ? 2 : 1));
      xr[j] = VIRT ? s_sp_xv : s_sp_minx;
    }
    if (good && apply_filter) {
      if (VIRT) {  // rectifiedMatch's filter on the two pixels (inference.hpp:384-391)
        const uint32_t pl = vrl[j * NT + tid].y, pr = vrr[xr[j]].y;
        const int yl = divw(pl, v.dw), yr = divw(pr, v.dw);
        good = abs(yl - yr) <= v.vtol && abs(((int)pl - yl * v.dw.W) - ((int)pr - yr * v.dw.W)) <= disp_high;
      } else {
        good = abs((int)(j * NT + tid) - (int)xr[j]) <= disp_high;
      }
    }
    if (good) okm |= 1u << j;
  }
  if (VIRT) {  // range of the partition's matched codes (one pair of LDS atomics per wave that has a match)
    uint32_t cmax = 0u, cmin = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < SPT; ++j)
      if ((okm >> j) & 1u) {
        cmax = max(cmax, kl[j] - 1u);
        cmin = min(cmin, kl[j] - 1u);
      }
    if (__ballot(okm != 0u)) {  // wave-uniform
      cmax = wave_max_u32(cmax);
      cmin = ~wave_max_u32(~cmin);
      if (lane == 0) {
        atomicMax(&s_cmax, cmax);
        atomicMin(&s_cmin, cmin);
      }
    }
  }
  __syncthreads();  // the flag words are dead from here on: their LDS is reused
  RJ_STAMP(3);
  // ---- 4. output position = rank of the code among the row's matches (counting rank)
  // Measured on one box and NOT adopted (546 / 514 us per 256 pairs as it stands):
  //   * matches alone in their bucket written straight from the scan, only shared buckets walked: 553 us;
  //   * matches first appended to a dense list (a wave reserving its stretch with one atomic) and ranked from
  //     there, one match per thread, with 16-bit counters beside the table: 522 us.
  // The NB buckets divide the range the codes really span, so that bits the forest leaves constant cost no resolution:
  // rows: [0, 2^bits) with bits from the OR of every code k_hash computed for the left image (a scalar load; tests
  // that never hold leave their bit clear there); partitions (VIRT): [smallest, largest matched code] of this
  // partition, reduced above (all its codes share a prefix).  The reduction costs a row kernel 18 us per 256 pairs.
  uint32_t cbase = 0u;
  int csh = 0;
  {
    uint32_t span;
    if (VIRT) {
      cbase = s_cmin;
      span = s_cmax >= cbase ? s_cmax - cbase : 0u;
    } else {
      span = (uint32_t)img_stats[(pair * 2) * GPC_STAT_STRIDE + GPC_STAT_CODEOR];
    }
    int lnb = 0;
    while ((1 << lnb) < NB) ++lnb;
    csh = (span ? 32 - __builtin_clz(span) : 0) - lnb;  // bits of the span beyond the log2(NB) a bucket index has
    if (csh < 0) csh = 0;
  }
  uint32_t rb[SPT], rs[SPT];
#pragma unroll
  for (int j = 0; j < SPT; ++j) {
    rb[j] = (kl[j] - 1u - cbase) >> csh;
    rs[j] = 0u;
    if ((okm >> j) & 1u) rs[j] = atomicAdd(&r_cnt[rb[j]], 1u);
  }
  __syncthreads();
  RJ_STAMP(4);
#ifndef RJ_OLD_RANK
  // How many matches share the bucket, read BEFORE the scan turns the counters into starts (every such read is done
  // before block_exscan's first barrier, the scan's stores come after it): four matches in five are alone in theirs and
  // need neither a place in r_key nor the two reads of neighbouring starts -- their rank is their bucket's start.
  // rs[j] becomes (arrival order | bucket count << 16): both are at most NB <= 4096... 16384 (VIRT) < 2^16.
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if ((okm >> j) & 1u) rs[j] |= r_cnt[rb[j]] << 16;
#endif
#ifndef RJ_DBG_NOSCAN    // (experiment, with RJ_DBG_EMPTY only)
  block_exscan<SPT, NT>(r_cnt, s_w, tid);  // r_cnt[b] = first rank of bucket b, r_cnt[NB] = number of matches
#endif
#ifdef RJ_OLD_RANK
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if ((okm >> j) & 1u) r_key[r_cnt[rb[j]] + rs[j]] = kl[j] - 1u;  // the code (WIDE: the key-less 0xFFFFFFFF ranks last)
#else
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if ((okm >> j) & 1u) {
      rb[j] = r_cnt[rb[j]];  // the bucket's first rank (the bucket index is not needed again)
      if ((rs[j] >> 16) > 1u) r_key[rb[j] + (rs[j] & 0xFFFFu)] = kl[j] - 1u;  // the code (WIDE: the key-less 0xFFFFFFFF ranks last)
    }
#endif
  __syncthreads();
  RJ_STAMP(5);
  const long rowbase = (long)pair * H + y;
  uint32_t* dst = (VIRT ? v.staged + pair * (v.recs / 2) + v_offl : staged + rowbase * W);
#pragma unroll
  for (int j = 0; j < SPT; ++j)
    if ((okm >> j) & 1u) {
      // (two plain ds_read_b32: for neighbouring words the compiler emits ds_read2_b32, which issues 3.6 times slower than
      // one ds_read_b32 on gfx950 -- profiles/r03_ubench2_issue_rates.txt -- and it unrolls the walk 16-fold with them)
#ifdef RJ_OLD_RANK
      uint32_t bidx = rb[j];
      const uint32_t s0 = r_cnt[bidx];
      asm volatile("" : "+v"(bidx));
      const uint32_t e0 = r_cnt[bidx + 1];
#else
      const uint32_t s0 = rb[j], e0 = s0 + (rs[j] >> 16);
#endif
      uint32_t rank = s0;
      const uint32_t cj = kl[j] - 1u;
#ifdef RJ_WALK_ALWAYS
      if (true) {
#else
      if (e0 - s0 > 1u) {  // a match alone in its bucket (four of five) has its rank already
#endif
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (uint32_t i = s0; i < e0; ++i) rank += (r_key[i] < cj);
      }
      if (VIRT) {
        // the two PIXEL INDICES, not the positions inside the partition: the partition's records were read by this
        // workgroup a moment ago (L2), where k_gp_gather fetched the same two words per match at random from memory
        // (83 -> 4x us per 8 pairs of 1920x1080 for that kernel)
        uint2* d2 = reinterpret_cast<uint2*>(v.staged) + pair * (v.recs / 2) + v_offl;
        d2[rank] = make_uint2(vrl[j * NT + tid].y, vrr[xr[j]].y);
      } else {
        dst[rank] = (uint32_t)(j * NT + tid) | (xr[j] << 16);
      }
    }
  if (tid == 0) {
    if (VIRT) vblk[v.o_rowcnt + y] = (int32_t)r_cnt[NB];
    else rowcnt[rowbase] = (int32_t)r_cnt[NB];
  }
#ifdef GPC_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RJ_STAMP(6);
  RJ_STAMP_FLUSH();
  if (ri + 1 < rpw && y + 1 < H - GPC_R) __syncthreads();  // the table is cleared again for the next row
  }  // rows of this workgroup
}

}  // namespace gpc
