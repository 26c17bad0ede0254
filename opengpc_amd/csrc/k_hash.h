// k_hash.h -- fern hash codes: T <= 32 pixel-pair tests in a 27x27 window of `smooth`.
//
// Replaces ndb::gpcFilter / ndb::gpcFilterTau (filter.hpp:547-606, 619-683) plus the
// zero-filled code buffer and descriptor gather of Forest::evalFastMaskOnSubsetSSE
// (inference.hpp:274-290).
//
// One 256-thread workgroup owns a 64x16 output tile.  The (64+32) x (16+26) smooth window
// is staged once into LDS with 16-byte coalesced loads (252 chunks, one per thread); the
// tests (LDS byte offsets, tau) arrive as a by-value kernel argument, i.e. in SGPRs, so
// every tap is one ds_read_u8 at `lane base + scalar offset`.  A wave covers one image row
// of 64 pixels per step: all lanes read the same LDS row -> conflict-free (4 lanes share a
// dword, broadcast).  Per test: v_sub + v_alignbit.  No MFMA: this is gather/compare.
//
// Output is a dense code image (u32 per pixel): the code for candidates, GPC_NOCAND for
// everything else (DENSE=false), or exactly the reference's gpcstates buffer (DENSE=true,
// used by gpc_hip_hash_codes for parity checks).
#pragma once
#include "gpc_device.h"

namespace gpc {

template <bool TAU>
__device__ __forceinline__ uint32_t fern_code(const uint8_t* __restrict__ tile, int base,
                                              const GpcForestDev& f, int x) {
  // Evaluate tests last-to-first so that test t lands on bit t of `acc`.
  uint32_t acc = 0;
#pragma unroll 4
  for (int t = f.num_tests - 1; t >= 0; --t) {
    const int a = tile[base + f.off_a[t]];
    int b = tile[base + f.off_b[t]];
    if (TAU) {
      // _mm_subs_epi8(b, tau): signed saturating byte subtract, result reinterpreted as
      // unsigned for the compare (filter.hpp:647-652)
      int sb = (int)(int8_t)b - f.tau[t];
      sb = min(max(sb, -128), 127);
      b = sb & 0xFF;
    }
    // (b - a) is negative iff a > b; shift its sign bit into acc
    acc = __builtin_amdgcn_alignbit(acc, (uint32_t)(b - a), 31);
  }
  // Bit placement of the reference's four byte planes (filter.hpp:574-595): tests 0..7 ->
  // bits 0..7; test 8 -> bit 0 unless x % 8 == 0 (64-bit-lane carry of bitMask += bitMask);
  // tests 9..31 -> bits 8..30.
  uint32_t code = (acc & 0xFFu) | ((acc >> 9) << 8);
  if ((acc & 0x100u) && (x & 7)) code |= 1u;
  return code;
}

// smooth, grad, candmap: [nimg][H][W]; codes: [nimg][H][W] u32
// candmap == nullptr: candidate <=> grad != 0 inside the margin (preprocessImage's mask).
template <bool TAU, bool DENSE>
__global__ __launch_bounds__(256) void k_hash(const uint8_t* __restrict__ smooth,
                                              const uint8_t* __restrict__ grad,
                                              const uint8_t* __restrict__ candmap,
                                              uint32_t* __restrict__ codes, int W, int H,
                                              GpcForestDev f, int32_t* __restrict__ img_stats) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[HT_ROWS * HT_STRIDE];
  __shared__ int s_cnt, s_last;

  const int img = blockIdx.z;
  const long n = (long)W * H;
  const uint8_t* sm = smooth + (long)img * n;
  const uint8_t* gr = grad + (long)img * n;
  const uint8_t* cm = candmap ? candmap + (long)img * n : nullptr;
  uint32_t* out = codes + (long)img * n;
  const int tx0 = blockIdx.x * HT_X, ty0 = blockIdx.y * HT_Y;
  const int tid = threadIdx.x;

  if (tid == 0) { s_cnt = 0; s_last = -1; }

  // ---- stage the smooth window; linear addressing like the reference's unaligned loads,
  //      bytes outside the buffer read as 0
  for (int c = tid; c < HT_ROWS * (HT_STRIDE / 16); c += 256) {
    const int r = c / (HT_STRIDE / 16), q = c - r * (HT_STRIDE / 16);
    const long k = (long)(ty0 - GPC_R + r) * W + (tx0 - HT_APRON + q * 16);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k >= 0 && k + 16 <= n) v = *reinterpret_cast<const uint4*>(sm + k);
    *reinterpret_cast<uint4*>(tile + r * HT_STRIDE + q * 16) = v;
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const int x = tx0 + lane;
  int cnt = 0, last = -1;
#pragma unroll 1
  for (int rr = 0; rr < HT_Y / 4; ++rr) {
    const int ly = wave * (HT_Y / 4) + rr;
    const int y = ty0 + ly;
    const bool inimg = (x < W) && (y < H);
    const long k = (long)y * W + x;
    const int g = inimg ? gr[k] : 0;
    const bool margin = x >= GPC_R && x < W - GPC_R && y >= GPC_R && y < H - GPC_R;
    const bool cand = margin && ((cm ? (int)cm[k] : g) != 0);
    // the reference skips 16-pixel groups without any gradient byte (filter.hpp:566)
    const unsigned long long gm = __ballot(g != 0);
    const bool group_any = ((gm >> (lane & 48)) & 0xFFFFull) != 0;
    const bool rows_ok = y >= GPC_R && y < H - 15;  // gpcFilterSegment(13, height-15) :602
    const bool compute = DENSE ? (inimg && rows_ok && group_any) : (cand && rows_ok && group_any);
    uint32_t code = 0;
    if (compute) {
      const int base = (ly + GPC_R) * HT_STRIDE + lane + HT_APRON;
      code = fern_code<TAU>(tile, base, f, x);
    }
    if (inimg) out[k] = DENSE ? code : (cand ? code : GPC_NOCAND);
    const unsigned long long cmask = __ballot(cand);
    if (cmask) { cnt += __popcll(cmask); last = y; }
  }
  if (!DENSE) {
    if (lane == 0 && cnt) { atomicAdd(&s_cnt, cnt); atomicMax(&s_last, last); }
    __syncthreads();
    if (tid == 0 && s_cnt) {
      atomicAdd(&img_stats[img * GPC_STAT_STRIDE + GPC_STAT_NCAND], s_cnt);
      atomicMax(&img_stats[img * GPC_STAT_STRIDE + GPC_STAT_LASTROW], s_last);
    }
  }
}

// candmap[img][k] = 1 for every k of the caller's mask list (inside the margin)
__global__ void k_scatter_mask(const int32_t* __restrict__ mask, int n_mask, uint8_t* __restrict__ candmap,
                               int W, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_mask) return;
  const int k = mask[i];
  if (k < 0 || k >= W * H) return;
  const int x = k % W, y = k / W;
  if (x >= GPC_R && x < W - GPC_R && y >= GPC_R && y < H - GPC_R) candmap[k] = 1;
}

}  // namespace gpc
