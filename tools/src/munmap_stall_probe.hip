// munmap_stall_probe -- why does the GPU submission after a free() sometimes take 20-30 ms?  Times a trivial kernel +
// stream synchronisation after each of: (A) free of a malloc'd buffer no HIP call ever saw, (B) free of a buffer that was
// the destination of a pageable hipMemcpy D2H, (C) the same as source of a H2D copy, (D) hipHostFree of page-locked
// memory, (E) B without the free, (F) vector-sized churn.  usage: munmap_stall_probe [MiB]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ void k_nop(int* p) { if (threadIdx.x == 0 && p) p[0] += 1; }

static double now_ms() {
  using namespace std::chrono;
  return duration_cast<duration<double, std::milli>>(steady_clock::now().time_since_epoch()).count();
}
static hipStream_t s;
static int* d_flag;
static double submit() {
  const double t0 = now_ms();
  hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, d_flag);
  hipStreamSynchronize(s);
  return now_ms() - t0;
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)((argc > 1 ? atof(argv[1]) : 3.4) * 1024 * 1024);
  hipSetDevice(0);
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipMalloc(&d_flag, 64);
  hipMemset(d_flag, 0, 64);
  void* d_buf;
  hipMalloc(&d_buf, bytes);
  for (int i = 0; i < 5; ++i) submit();
  printf("buffer %.1f MiB; idle submit %.3f ms\n", bytes / 1048576.0, submit());
  for (int rep = 0; rep < 3; ++rep) {
    {  // A
      char* p = (char*)malloc(bytes);
      memset(p, 1, bytes);
      const double t0 = now_ms();
      free(p);
      const double tf = now_ms() - t0;
      const double s1 = submit(), s2 = submit();
      printf("A  malloc/touch/free (no HIP)           : free %.3f ms, next submit %.3f ms, then %.3f ms\n", tf, s1, s2);
    }
    {  // B
      char* p = (char*)malloc(bytes);
      double t0 = now_ms();
      hipMemcpy(p, d_buf, bytes, hipMemcpyDeviceToHost);
      const double tc = now_ms() - t0;
      t0 = now_ms();
      free(p);
      const double tf = now_ms() - t0;
      const double s1 = submit(), s2 = submit();
      printf("B  pageable D2H (%.3f ms) then free      : free %.3f ms, next submit %.3f ms, then %.3f ms\n", tc, tf, s1, s2);
    }
    {  // C
      char* p = (char*)malloc(bytes);
      memset(p, 2, bytes);
      double t0 = now_ms();
      hipMemcpyAsync(d_buf, p, bytes, hipMemcpyHostToDevice, s);
      hipStreamSynchronize(s);
      const double tc = now_ms() - t0;
      t0 = now_ms();
      free(p);
      const double tf = now_ms() - t0;
      const double s1 = submit(), s2 = submit();
      printf("C  pageable H2D (%.3f ms) then free      : free %.3f ms, next submit %.3f ms, then %.3f ms\n", tc, tf, s1, s2);
    }
    {  // D
      void* p = nullptr;
      double t0 = now_ms();
      hipHostMalloc(&p, bytes, hipHostMallocDefault);
      const double ta = now_ms() - t0;
      memset(p, 3, bytes);
      t0 = now_ms();
      hipHostFree(p);
      const double tf = now_ms() - t0;
      const double s1 = submit(), s2 = submit();
      printf("D  hipHostMalloc (%.3f ms) / hipHostFree : free %.3f ms, next submit %.3f ms, then %.3f ms\n", ta, tf, s1, s2);
    }
    {  // E
      static char* keep[8];
      char* p = (char*)malloc(bytes);
      hipMemcpy(p, d_buf, bytes, hipMemcpyDeviceToHost);
      keep[rep] = p;
      const double s1 = submit(), s2 = submit();
      printf("E  pageable D2H, buffer kept            : next submit %.3f ms, then %.3f ms\n", s1, s2);
    }
    {  // F: vector-sized churn
      double worst = 0;
      for (int k = 0; k < 8; ++k) {
        char* p = (char*)malloc(bytes / 8);
        hipMemcpy(p, d_buf, bytes / 8, hipMemcpyDeviceToHost);
        free(p);
        const double t = submit();
        if (t > worst) worst = t;
      }
      printf("F  8 x (D2H into %.2f MiB, free, submit)  : worst submit %.3f ms\n", bytes / 8 / 1048576.0, worst);
    }
  }
  return 0;
}
