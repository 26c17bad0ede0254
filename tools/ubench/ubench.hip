// ubench.hip -- calibration micro-benchmarks for the row/hash kernels' building blocks on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip ; run: ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k_bench(uint32_t* out, int iters, uint32_t seed) {
  __shared__ uint32_t lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 4096; i += 256) lds[i] = i * 2654435761u;
  __syncthreads();
  uint32_t a = tid * 7 + seed, b = tid * 13 + 1, c = tid ^ 0x55, d = tid + 99;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {  // 8 independent VALU ops
      a = a * 3 + 1; b = b * 5 + 2; c = c * 7 + 3; d = d * 9 + 4;
      a ^= b; c ^= d; b += c; d += a;
    } else if (MODE == 1) {  // 4 random ds_read_b32
      a = lds[(a >> 7) & 4095] + it; b = lds[(b >> 9) & 4095] ^ it; c = lds[(c >> 5) & 4095] + 3; d = lds[(d >> 11) & 4095] ^ a;
    } else if (MODE == 2) {  // 4 linear ds_read_b32
      a += lds[(tid + it) & 4095]; b += lds[(tid + 64 + it) & 4095]; c += lds[(tid + 128 + it) & 4095]; d += lds[(tid + 192 + it) & 4095];
    } else if (MODE == 3) {  // 4 random returning ds_add
      a += atomicAdd(&lds[(a * 2654435761u >> 20) & 4095], 1u); b += atomicAdd(&lds[(b * 2654435761u >> 20) & 4095], 1u);
      c += atomicAdd(&lds[(c * 2654435761u >> 20) & 4095], 1u); d += atomicAdd(&lds[(d * 2654435761u >> 20) & 4095], 1u);
    } else if (MODE == 4) {  // 4 random non-returning ds_or
      atomicOr(&lds[(a * 2654435761u >> 20) & 4095], 1u); atomicOr(&lds[(b * 2654435761u >> 20) & 4095], 2u);
      atomicOr(&lds[(c * 2654435761u >> 20) & 4095], 4u); atomicOr(&lds[(d * 2654435761u >> 20) & 4095], 8u);
      a += 3; b += 5; c += 7; d += 11;
    } else if (MODE == 5) {  // 4 random ds_cmpst_rtn
      a += atomicCAS(&lds[(a * 2654435761u >> 20) & 4095], 0xFFFFFFFFu, a); b += atomicCAS(&lds[(b * 2654435761u >> 20) & 4095], 0xFFFFFFFFu, b);
      c += atomicCAS(&lds[(c * 2654435761u >> 20) & 4095], 0xFFFFFFFFu, c); d += atomicCAS(&lds[(d * 2654435761u >> 20) & 4095], 0xFFFFFFFFu, d);
    } else if (MODE == 6) {  // 4 DPP movs + min
      a = min(a, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a, 0xB1, 0xF, 0xF, true)) + 1;
      b = min(b, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, true)) + 1;
      c = min(c, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x141, 0xF, 0xF, true)) + 1;
      d = min(d, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x140, 0xF, 0xF, true)) + 1;
    } else if (MODE == 7) {  // 4 ds_swizzle
      a = min(a, (uint32_t)__builtin_amdgcn_ds_swizzle((int)a, 0x401F)) + 1; b = min(b, (uint32_t)__builtin_amdgcn_ds_swizzle((int)b, 0x101F)) + 1;
      c = min(c, (uint32_t)__builtin_amdgcn_ds_swizzle((int)c, 0x7C1F)) + 1; d = min(d, (uint32_t)__builtin_amdgcn_ds_swizzle((int)d, 0x401F)) + 1;
    } else if (MODE == 8) {  // 4 ds_bpermute (shfl_xor 32)
      a = min(a, (uint32_t)__shfl_xor((int)a, 32)) + 1; b = min(b, (uint32_t)__shfl_xor((int)b, 32)) + 1;
      c = min(c, (uint32_t)__shfl_xor((int)c, 63)) + 1; d = min(d, (uint32_t)__shfl_xor((int)d, 32)) + 1;
    } else if (MODE == 9) {  // 4 linear ds_write_b32 + barrier-free
      lds[(tid + it) & 4095] = a; lds[(tid + 256 + it) & 4095] = b; lds[(tid + 512 + it) & 4095] = c; lds[(tid + 768 + it) & 4095] = d;
      a += 1; b += 2; c += 3; d += 4;
    } else if (MODE == 10) {  // same-address returning atomic (all lanes one address)
      a += atomicAdd(&lds[it & 4095], 1u); b += atomicAdd(&lds[(it + 1) & 4095], 1u); c += atomicAdd(&lds[(it + 2) & 4095], 1u); d += atomicAdd(&lds[(it + 3) & 4095], 1u);
    } else if (MODE == 11) {  // __syncthreads
      __syncthreads(); a += 1; __syncthreads(); b += 1; __syncthreads(); c += 1; __syncthreads(); d += 1;
    } else if (MODE == 12) {  // 4 random ds_read_u16
      const uint16_t* l16 = (const uint16_t*)lds;
      a = l16[(a >> 7) & 8191] + it; b = l16[(b >> 9) & 8191] ^ it; c = l16[(c >> 5) & 8191] + 3; d = l16[(d >> 11) & 8191] ^ a;
    }
  }
  acc = a ^ b ^ c ^ d;
  out[blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
float run(uint32_t* d_out, int blocks, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_bench<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_bench<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 2u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4 * 2));
  const char* names[] = {"8 VALU ops", "4 random ds_read_b32", "4 linear ds_read_b32", "4 random ds_add_rtn", "4 random ds_or (no rtn)",
                         "4 random ds_cmpst_rtn", "4 DPP mov+min+add", "4 ds_swizzle+min+add", "4 ds_bpermute+min+add", "4 linear ds_write_b32",
                         "4 same-address ds_add_rtn", "4 __syncthreads", "4 random ds_read_u16"};
  const int iters = 2000;
  for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
    const int blocks = 256 * wg_per_cu;
    printf("== %d workgroups of 256 threads per CU (%d waves/SIMD)\n", wg_per_cu, wg_per_cu);
    float ms[13];
    ms[0] = run<0>(d_out, blocks, iters); ms[1] = run<1>(d_out, blocks, iters); ms[2] = run<2>(d_out, blocks, iters);
    ms[3] = run<3>(d_out, blocks, iters); ms[4] = run<4>(d_out, blocks, iters); ms[5] = run<5>(d_out, blocks, iters);
    ms[6] = run<6>(d_out, blocks, iters); ms[7] = run<7>(d_out, blocks, iters); ms[8] = run<8>(d_out, blocks, iters);
    ms[9] = run<9>(d_out, blocks, iters); ms[10] = run<10>(d_out, blocks, iters); ms[11] = run<11>(d_out, blocks, iters);
    ms[12] = run<12>(d_out, blocks, iters);
    for (int m = 0; m < 13; ++m) {
      // cycles per loop iteration per CU at 2.4 GHz, normalised per workgroup-iteration
      double cyc = ms[m] * 1e-3 * 2.4e9 / iters / wg_per_cu;
      printf("  %-28s %8.3f ms   %7.1f cycles per WG-iteration (4 waves)\n", names[m], ms[m], cyc);
    }
  }
  return 0;
}
