// streams_bench.hip -- how does the achievable HBM bandwidth of MI355X depend on the NUMBER of concurrent streams of a
// kernel?  The transform passes with fused prologues / epilogues read 2-4 arrays and write 1-2 (first pass of the metric:
// d, xi -> work; final pass: work, xi, d, out -> out), the plain copy ceiling (copy_bench.hip: 5.3 TB/s) is a 1R + 1W number.
// Every workgroup owns one contiguous 64 KiB chunk of every array (the fastest plain copy pattern).
// build: hipcc --offload-arch=gfx950 -O3 -o streams_bench streams_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef float __attribute__((ext_vector_type(4))) f4;

struct Ptrs { const f4* r[6]; f4* w[3]; };

// NR read streams, NW write streams; RMW: the first write stream is also read (read-modify-write in place)
template <int NR, int NW, int RMW, int U>
__global__ void __launch_bounds__(256) k_streams(Ptrs p) {
  const size_t lo = (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int u0 = 0; u0 < 16; u0 += U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = lo + threadIdx.x + (u0 + u) * 256;
      v[u] = p.r[0][i];
#pragma unroll
      for (int s = 1; s < NR; ++s) v[u] += p.r[s][i];
      if (RMW) v[u] += p.w[0][i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = lo + threadIdx.x + (u0 + u) * 256;
#pragma unroll
      for (int s = 0; s < NW; ++s) __builtin_nontemporal_store(v[u] * (float)(s + 1), p.w[s] + i);
    }
  }
}

template <int NR, int NW, int RMW, int U>
static void run(const Ptrs& p, size_t bytes) {
  const unsigned blocks = (unsigned)(bytes / 65536);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_streams<NR, NW, RMW, U>), dim3(blocks), dim3(256), 0, 0, p);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_streams<NR, NW, RMW, U>), dim3(blocks), dim3(256), 0, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double moved = (double)(NR + NW + RMW) * bytes;
  printf("%d read + %d write streams%s, %d x 16 B in flight per stream and thread: %7.3f ms  %7.1f GB/s\n", NR, NW,
         RMW ? " (first written array read-modify-written)" : "", U, ms, moved / ms / 1e6);
}

int main() {
  const size_t G = (size_t)4 << 30;  // one 1024^3 fp32 field
  Ptrs p;
  for (int i = 0; i < 6; ++i) { void* q; CK(hipMalloc(&q, G)); CK(hipMemset(q, 0, G)); p.r[i] = (const f4*)q; }
  for (int i = 0; i < 3; ++i) { void* q; CK(hipMalloc(&q, G)); CK(hipMemset(q, 0, G)); p.w[i] = (f4*)q; }
#define ROW(NR, NW, RMW) run<NR, NW, RMW, 4>(p, G); run<NR, NW, RMW, 8>(p, G);
  ROW(1, 1, 0) ROW(2, 1, 0) ROW(3, 1, 0) ROW(4, 1, 0) ROW(6, 1, 0)
  ROW(1, 2, 0) ROW(2, 2, 0) ROW(3, 2, 0)
  ROW(1, 1, 1) ROW(2, 1, 1) ROW(3, 1, 1)
  return 0;
}
