// ubench2.hip -- issue-rate calibration of the integer / SWAR VALU ops and LDS reads the hash and
// join kernels are made of, on gfx950.  Every body is inline asm (nothing for the compiler to
// re-associate); cycles come from s_memtime inside the kernel, so no clock assumption is needed.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench2 ubench2.hip ; run: ./ubench2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X

// 8 independent chains; one asm block = 8 instructions
#define OP2(name) \
  asm volatile(name " %0, %8, %0\n\t" name " %1, %8, %1\n\t" name " %2, %8, %2\n\t" name " %3, %8, %3\n\t" \
               name " %4, %8, %4\n\t" name " %5, %8, %5\n\t" name " %6, %8, %6\n\t" name " %7, %8, %7"      \
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k));
#define OP3(name) \
  asm volatile(name " %0, %8, %0, %9\n\t" name " %1, %8, %1, %9\n\t" name " %2, %8, %2, %9\n\t" name " %3, %8, %3, %9\n\t" \
               name " %4, %8, %4, %9\n\t" name " %5, %8, %5, %9\n\t" name " %6, %8, %6, %9\n\t" name " %7, %8, %7, %9"      \
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "v"(k2));
#define OPS(name, suffix) \
  asm volatile(name " %0, %8, %0 " suffix "\n\t" name " %1, %8, %1 " suffix "\n\t" name " %2, %8, %2 " suffix "\n\t" \
               name " %3, %8, %3 " suffix "\n\t" name " %4, %8, %4 " suffix "\n\t" name " %5, %8, %5 " suffix "\n\t" \
               name " %6, %8, %6 " suffix "\n\t" name " %7, %8, %7 " suffix                                          \
               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k));
// LDS reads: 8 reads in flight, then one wait
#define LDSRD(name, regs)                                                                                   \
  asm volatile(name " %0, %8\n\t" name " %1, %8 offset:256\n\t" name " %2, %8 offset:512\n\t"               \
               name " %3, %8 offset:768\n\t" name " %4, %8 offset:1024\n\t" name " %5, %8 offset:1280\n\t" \
               name " %6, %8 offset:1536\n\t" name " %7, %8 offset:1792\n\ts_waitcnt lgkmcnt(0)"           \
               : "=" regs(q0), "=" regs(q1), "=" regs(q2), "=" regs(q3), "=" regs(q4), "=" regs(q5), "=" regs(q6), "=" regs(q7) : "v"(addr));

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_bench(uint32_t* out, unsigned long long* cyc, int iters, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[8192];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) lds[i] = i * 2654435761u;
  __syncthreads();
  uint32_t r0 = tid * 7 + seed, r1 = tid * 13 + 1, r2 = tid ^ 0x55, r3 = tid + 99, r4 = tid * 3, r5 = tid * 5, r6 = tid * 11, r7 = ~tid;
  uint32_t k = seed * 0x01010101u + tid, k2 = 0x80808080u;
  uint32_t acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) { REP8(OP2("v_add_u32")) }
    else if (MODE == 1) { REP8(OP2("v_and_b32")) }
    else if (MODE == 2) { REP8(OP2("v_xor_b32")) }
    else if (MODE == 3) { REP8(OP2("v_sub_u32")) }
    else if (MODE == 4) { REP8(OP2("v_lshrrev_b32")) }
    else if (MODE == 5) { REP8(OP3("v_bfi_b32")) }
    else if (MODE == 6) { REP8(OP3("v_perm_b32")) }
    else if (MODE == 7) { REP8(OP3("v_alignbyte_b32")) }
    else if (MODE == 8) { REP8(OP3("v_and_or_b32")) }
    else if (MODE == 9) { REP8(OP3("v_lshl_or_b32")) }
    else if (MODE == 10) { REP8(OP2("v_pk_sub_u16")) }
    else if (MODE == 11) { REP8(OP3("v_fma_f32")) }
    else if (MODE == 12) { REP8(OPS("v_add_u32_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD")) }
    else if (MODE == 13) { REP8(OPS("v_add_u32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf")) }
    else if (MODE == 17) { REP8(OP3("v_lerp_u8")) }
    else if (MODE == 18) { REP8(OP3("v_lshl_add_u32")) }
    else if (MODE == 19) { REP8(OP3("v_sad_u8")) }
    else if (MODE == 40) { REP8(OP3("v_add3_u32")) }
    else if (MODE == 41) { REP8(OP3("v_xad_u32")) }
    else if (MODE == 42) { REP8(OP3("v_msad_u8")) }
    else if (MODE == 14) {
      REP8(asm volatile("v_bitop3_b32 %0, %8, %0, %9 bitop3:0xd8\n\tv_bitop3_b32 %1, %8, %1, %9 bitop3:0xd8\n\t"
                        "v_bitop3_b32 %2, %8, %2, %9 bitop3:0xd8\n\tv_bitop3_b32 %3, %8, %3, %9 bitop3:0xd8\n\t"
                        "v_bitop3_b32 %4, %8, %4, %9 bitop3:0xd8\n\tv_bitop3_b32 %5, %8, %5, %9 bitop3:0xd8\n\t"
                        "v_bitop3_b32 %6, %8, %6, %9 bitop3:0xd8\n\tv_bitop3_b32 %7, %8, %7, %9 bitop3:0xd8"
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "v"(k2));)
    } else if (MODE == 15) {  // compare into VCC + consume with addc (serial through vcc)
      REP8(asm volatile("v_cmp_gt_u32 vcc, %8, %0\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\t"
                        "v_cmp_gt_u32 vcc, %8, %2\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc\n\t"
                        "v_cmp_gt_u32 vcc, %8, %4\n\tv_addc_co_u32 %5, vcc, %5, %5, vcc\n\t"
                        "v_cmp_gt_u32 vcc, %8, %6\n\tv_addc_co_u32 %7, vcc, %7, %7, vcc"
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k) : "vcc");)
    } else if (MODE == 16) {  // mixed SWAR test body as in k_hash: and, or, sub, bitop3, lshr, bfi  (x8 / 6 ops = 1.33 tests)
      REP8(asm volatile("v_and_b32 %0, %9, %1\n\tv_or_b32 %2, %9, %3\n\tv_sub_u32 %4, %2, %0\n\t"
                        "v_bitop3_b32 %5, %1, %3, %4 bitop3:0xd8\n\tv_lshrrev_b32 %6, 1, %7\n\tv_bfi_b32 %7, %9, %5, %6\n\t"
                        "v_and_b32 %0, %9, %3\n\tv_or_b32 %2, %9, %1"
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "v"(k2));)
    } else if (MODE >= 25 && MODE <= 27) {  // wide reads at addresses that are only 4-byte (25, 26, 28) or 8-byte (27) aligned
      const uint32_t addr = (uint32_t)(uintptr_t)lds + ((tid & 63) * (MODE == 25 ? 8 : 16)) + ((it & 3) << 11) + (MODE == 27 ? 8 : 4);
      if (MODE == 25) {
        u32x2 q0, q1, q2, q3, q4, q5, q6, q7;
        REP8(LDSRD("ds_read_b64", "v"))
        r0 += q0.x + q1.y + q2.x + q3.y + q4.x + q5.y + q6.x + q7.y;
      } else {
        u32x4 q0, q1, q2, q3, q4, q5, q6, q7;
        REP8(LDSRD("ds_read_b128", "v"))
        r0 += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w;
      }
    } else if (MODE >= 20 && MODE <= 23) {
      const uint32_t addr = (uint32_t)(uintptr_t)lds + ((tid & 63) * (MODE == 20 ? 4 : MODE == 21 ? 8 : MODE == 22 ? 16 : 4)) + ((it & 3) << 11);
      if (MODE == 20) {
        uint32_t q0, q1, q2, q3, q4, q5, q6, q7;
        REP8(LDSRD("ds_read_b32", "v"))
        r0 += q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
      } else if (MODE == 21) {
        u32x2 q0, q1, q2, q3, q4, q5, q6, q7;
        REP8(LDSRD("ds_read_b64", "v"))
        r0 += q0.x + q1.y + q2.x + q3.y + q4.x + q5.y + q6.x + q7.y;
      } else if (MODE == 22) {
        u32x4 q0, q1, q2, q3, q4, q5, q6, q7;
        REP8(LDSRD("ds_read_b128", "v"))
        r0 += q0.x + q1.y + q2.z + q3.w + q4.x + q5.y + q6.z + q7.w;
      } else {
        uint32_t q0, q1, q2, q3, q4, q5, q6, q7;
        REP8(LDSRD("ds_read_u8", "v"))
        r0 += q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
      }
    } else if (MODE == 28 || MODE == 29) {  // two adjacent dwords per lane with ONE instruction, 8-byte stride; 29: only 4-byte aligned
      const uint32_t addr = (uint32_t)(uintptr_t)lds + (tid & 63) * 8 + ((it & 3) << 11) + (MODE == 29 ? 4 : 0);
      u32x2 q0, q1, q2, q3, q4, q5, q6, q7;
      REP8(asm volatile("ds_read2_b32 %0, %8 offset0:0 offset1:1\n\tds_read2_b32 %1, %8 offset0:64 offset1:65\n\t"
                        "ds_read2_b32 %2, %8 offset0:128 offset1:129\n\tds_read2_b32 %3, %8 offset0:192 offset1:193\n\t"
                        "ds_read2_b32 %4, %8 offset0:16 offset1:17\n\tds_read2_b32 %5, %8 offset0:80 offset1:81\n\t"
                        "ds_read2_b32 %6, %8 offset0:144 offset1:145\n\tds_read2_b32 %7, %8 offset0:208 offset1:209\n\ts_waitcnt lgkmcnt(0)"
                        : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7) : "v"(addr));)
      r0 += q0.x + q1.y + q2.x + q3.y + q4.x + q5.y + q6.x + q7.y;
    } else if (MODE >= 30 && MODE <= 35) {
      // one fern test over a wave's 4 rows x 4 px: 24 VALU beside the LDS reads of both taps --
      // 30: 8 ds_read_b32 (rows stored one after the other: k_hash as it is); 31 / 32 / 33: 5 / 4 / 6 ds_read_b64 (rows stored
      // in interleaved pairs: a tap whose first row is even takes 2 reads for 4 rows, an odd one 3)
      const uint32_t addr = (uint32_t)(uintptr_t)lds + (tid & 63) * (MODE == 30 ? 4 : 8) + ((it & 3) << 11);
#define VALU24_K15                                                                                                            \
  "v_and_b32 %2, %15, %2\n\tv_or_b32 %3, %15, %3\n\tv_sub_u32 %4, %15, %4\n\tv_xor_b32 %5, %15, %5\n\tv_lshrrev_b32 %6, 1, %6\n\t"   \
  "v_bitop3_b32 %7, %15, %7, %7 bitop3:0xd8\n\tv_and_b32 %2, %15, %2\n\tv_or_b32 %3, %15, %3\n\tv_sub_u32 %4, %15, %4\n\t"          \
  "v_xor_b32 %5, %15, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_bitop3_b32 %7, %15, %7, %7 bitop3:0xd8\n\t"                                \
  "v_and_b32 %2, %15, %2\n\tv_or_b32 %3, %15, %3\n\tv_sub_u32 %4, %15, %4\n\tv_xor_b32 %5, %15, %5\n\tv_lshrrev_b32 %6, 1, %6\n\t"   \
  "v_bitop3_b32 %7, %15, %7, %7 bitop3:0xd8\n\tv_and_b32 %2, %15, %2\n\tv_or_b32 %3, %15, %3\n\tv_sub_u32 %4, %15, %4\n\t"          \
  "v_xor_b32 %5, %15, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_bitop3_b32 %7, %15, %7, %7 bitop3:0xd8\n\ts_waitcnt lgkmcnt(0)"
#define VALU24_K13                                                                                                            \
  "v_and_b32 %2, %13, %2\n\tv_or_b32 %3, %13, %3\n\tv_sub_u32 %4, %13, %4\n\tv_xor_b32 %5, %13, %5\n\tv_lshrrev_b32 %6, 1, %6\n\t"   \
  "v_bitop3_b32 %7, %13, %7, %7 bitop3:0xd8\n\tv_and_b32 %2, %13, %2\n\tv_or_b32 %3, %13, %3\n\tv_sub_u32 %4, %13, %4\n\t"          \
  "v_xor_b32 %5, %13, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_bitop3_b32 %7, %13, %7, %7 bitop3:0xd8\n\t"                                \
  "v_and_b32 %2, %13, %2\n\tv_or_b32 %3, %13, %3\n\tv_sub_u32 %4, %13, %4\n\tv_xor_b32 %5, %13, %5\n\tv_lshrrev_b32 %6, 1, %6\n\t"   \
  "v_bitop3_b32 %7, %13, %7, %7 bitop3:0xd8\n\tv_and_b32 %2, %13, %2\n\tv_or_b32 %3, %13, %3\n\tv_sub_u32 %4, %13, %4\n\t"          \
  "v_xor_b32 %5, %13, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_bitop3_b32 %7, %13, %7, %7 bitop3:0xd8\n\ts_waitcnt lgkmcnt(0)"
      if (MODE == 35) {  // k_hash as the compiler emits it: two rows per ds_read2_b32 (row stride 72 dwords), 4 per test
        u32x2 q0, q1, a0, a1;
        const uint32_t addr4 = (uint32_t)(uintptr_t)lds + (tid & 63) * 4 + ((it & 3) << 11);
        REP8(asm volatile("ds_read2_b32 %0, %12 offset1:72\n\tds_read2_b32 %1, %12 offset0:144 offset1:216\n\t"
                          "ds_read2_b32 %8, %12 offset0:3 offset1:75\n\tds_read2_b32 %9, %12 offset0:147 offset1:219\n\t" VALU24_K13
                          : "=&v"(q0), "=&v"(q1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(a0), "=&v"(a1),
                            "=&v"(r0), "=&v"(r1)
                          : "v"(addr4), "v"(k));
             r2 += q0.x + a0.y; r3 += q1.y + a1.x;)
      } else if (MODE == 30) {
        uint32_t q0, q1, a0, a1, a2, a3, a4, a5;
        REP8(asm volatile("ds_read_b32 %0, %14\n\tds_read_b32 %1, %14 offset:288\n\tds_read_b32 %8, %14 offset:576\n\t"
                          "ds_read_b32 %9, %14 offset:864\n\tds_read_b32 %10, %14 offset:1152\n\tds_read_b32 %11, %14 offset:1440\n\t"
                          "ds_read_b32 %12, %14 offset:1728\n\tds_read_b32 %13, %14 offset:2016\n\t" VALU24_K15
                          : "=&v"(q0), "=&v"(q1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(a0), "=&v"(a1),
                            "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5)
                          : "v"(addr), "v"(k));
             r0 += q0 + a0 + a2 + a4; r1 += q1 + a1 + a3 + a5;)
      } else {
        u32x2 q0, q1, a0, a1, a2, a3;
        if (MODE == 31) {
          REP8(asm volatile("ds_read_b64 %0, %12\n\tds_read_b64 %1, %12 offset:576\n\tds_read_b64 %8, %12 offset:1152\n\t"
                            "ds_read_b64 %9, %12 offset:1728\n\tds_read_b64 %10, %12 offset:2304\n\t" VALU24_K13
                            : "=&v"(q0), "=&v"(q1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(a0), "=&v"(a1),
                              "=&v"(a2), "=&v"(a3)
                            : "v"(addr), "v"(k));
               r0 += q0.x + a0.x + a2.y; r1 += q1.y + a1.x + a3.y;)
        } else if (MODE == 32) {
          REP8(asm volatile("ds_read_b64 %0, %12\n\tds_read_b64 %1, %12 offset:576\n\tds_read_b64 %8, %12 offset:1152\n\t"
                            "ds_read_b64 %9, %12 offset:1728\n\t" VALU24_K13
                            : "=&v"(q0), "=&v"(q1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(a0), "=&v"(a1),
                              "=&v"(a2), "=&v"(a3)
                            : "v"(addr), "v"(k));
               r0 += q0.x + a0.x + a2.y; r1 += q1.y + a1.x + a3.y;)
        } else if (MODE == 34) {  // what the compiler makes of three adjacent row pairs per tap: ds_read2_b64 + ds_read_b64
          u32x4 w0, w1;
          REP8(asm volatile("ds_read2_b64 %8, %12 offset1:72\n\tds_read_b64 %0, %12 offset:1152\n\t"
                            "ds_read2_b64 %9, %12 offset0:1 offset1:73\n\tds_read_b64 %1, %12 offset:1160\n\t" VALU24_K13
                            : "=&v"(q0), "=&v"(q1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(w0), "=&v"(w1),
                              "=&v"(a2), "=&v"(a3)
                            : "v"(addr), "v"(k));
               r0 += q0.x + w0.x + w0.w; r1 += q1.y + w1.y + w1.z;)
        } else {
          REP8(asm volatile("ds_read_b64 %0, %12\n\tds_read_b64 %1, %12 offset:576\n\tds_read_b64 %8, %12 offset:1152\n\t"
                            "ds_read_b64 %9, %12 offset:1728\n\tds_read_b64 %10, %12 offset:2304\n\tds_read_b64 %11, %12 offset:2880\n\t" VALU24_K13
                            : "=&v"(q0), "=&v"(q1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "=&v"(a0), "=&v"(a1),
                              "=&v"(a2), "=&v"(a3)
                            : "v"(addr), "v"(k));
               r0 += q0.x + a0.x + a2.y; r1 += q1.y + a1.x + a3.y;)
        }
      }
    } else if (MODE == 24) {  // LDS reads beside VALU (the hash kernel's mix: 2 ds_read_b32 per 6 VALU)
      const uint32_t addr = (uint32_t)(uintptr_t)lds + (tid & 63) * 4 + ((it & 3) << 11);
      uint32_t q0, q1, q2, q3, q4, q5, q6, q7;
      REP8(asm volatile("ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:256\n\t"
                        "v_and_b32 %2, %9, %2\n\tv_or_b32 %3, %9, %3\n\tv_sub_u32 %4, %9, %4\n\t"
                        "v_xor_b32 %5, %9, %5\n\tv_lshrrev_b32 %6, 1, %6\n\tv_bfi_b32 %7, %9, %7, %7\n\ts_waitcnt lgkmcnt(0)"
                        : "=v"(q0), "=v"(q1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(addr), "v"(k));
           r0 += q0; r1 += q1;)
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  acc = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
  out[blockIdx.x * 256 + tid] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Res { float ms; double cyc; };

template <int MODE>
Res run(uint32_t* d_out, unsigned long long* d_cyc, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_bench<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_bench<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, iters, 2u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  hipEventDestroy(e0); hipEventDestroy(e1);
  return {ms, s / blocks};
}

int main() {
  uint32_t* d_out; unsigned long long* d_cyc;
  if (hipMalloc(&d_out, 256 * 8 * 256 * 4 * 2) != hipSuccess || hipMalloc(&d_cyc, 8 * 256 * 8 * 2) != hipSuccess) return 1;
  const int iters = 500;
  struct Case { const char* name; int mode; int per_iter; };  // per_iter = wave-instructions of the kind counted per loop iteration
  const Case cases[] = {
    {"v_add_u32", 0, 64}, {"v_and_b32", 1, 64}, {"v_xor_b32", 2, 64}, {"v_sub_u32", 3, 64}, {"v_lshrrev_b32", 4, 64},
    {"v_bfi_b32", 5, 64}, {"v_perm_b32", 6, 64}, {"v_alignbyte_b32", 7, 64}, {"v_and_or_b32", 8, 64}, {"v_lshl_or_b32", 9, 64},
    {"v_pk_sub_u16", 10, 64}, {"v_fma_f32", 11, 64}, {"v_add_u32_sdwa", 12, 64}, {"v_add_u32_dpp", 13, 64}, {"v_bitop3_b32", 14, 64},
    {"v_lerp_u8", 17, 64}, {"v_lshl_add_u32", 18, 64}, {"v_sad_u8", 19, 64}, {"v_add3_u32", 40, 64}, {"v_xad_u32", 41, 64}, {"v_msad_u8", 42, 64},
    {"v_cmp+v_addc (pairs)", 15, 64}, {"swar test mix (8 valu)", 16, 64},
    {"ds_read_b32 linear", 20, 64}, {"ds_read_b64 linear", 21, 64}, {"ds_read_b128 linear", 22, 64}, {"ds_read_u8 linear", 23, 64},
    {"2 ds_read_b32 + 6 valu", 24, 64}, {"ds_read_b64 addr%8==4", 25, 64}, {"ds_read_b128 addr%16==4", 26, 64},
    {"ds_read_b128 addr%16==8", 27, 64}, {"ds_read2_b32 adjacent", 28, 64}, {"ds_read2_b32 adj addr%8==4", 29, 64},
    // per_iter = 8 tests: the columns read "per fern test over 4 rows x 4 px"
    {"test: 8 ds_read_b32 + 24 valu", 30, 8}, {"test: 5 ds_read_b64 + 24 valu", 31, 8}, {"test: 4 ds_read_b64 + 24 valu", 32, 8},
    {"test: 6 ds_read_b64 + 24 valu", 33, 8}, {"test: 2 (read2_b64 + read_b64) + 24 valu", 34, 8},
    {"test: 4 ds_read2_b32 (rows 72 dwords apart) + 24 valu", 35, 8}};
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    const int blocks = 256 * wg_per_cu;
    printf("== %d workgroups of 256 threads per CU (%d waves/SIMD)\n", wg_per_cu, wg_per_cu);
    for (const Case& c : cases) {
      Res r;
      switch (c.mode) {
#define RUN(M) case M: r = run<M>(d_out, d_cyc, blocks, iters); break;
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(40) RUN(41) RUN(42)
        RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28) RUN(29) RUN(30) RUN(31) RUN(32) RUN(33) RUN(34) RUN(35)
        default: continue;
      }
      // cycles per instruction as one wave sees it, and per SIMD (divide by the waves sharing the SIMD);
      // LDS rows: per CU = per-wave figure / (4 * waves per SIMD)
      const double per_wave = r.cyc / ((double)iters * c.per_iter);
      // the s_memtime figures are in ticks of a clock that is NOT the shader clock on this box; the wall-clock column
      // (ns per wave-instruction per SIMD, launch overhead included) is the one to read
      printf("  %-26s %8.3f ms  %7.2f ticks/instr/wave  %6.3f ns/instr/SIMD  %6.3f ns/instr/CU\n", c.name, r.ms, per_wave,
             r.ms * 1e6 / ((double)iters * c.per_iter * wg_per_cu), r.ms * 1e6 / ((double)iters * c.per_iter * wg_per_cu * 4.0));
    }
  }
  return 0;
}
