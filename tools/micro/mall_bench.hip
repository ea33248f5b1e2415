// mall_bench.hip -- does a producer -> consumer hand-over through a small ring buffer stay in the 256 MiB Infinity
// Cache (memory-side L3) of MI355X?  If yes, two consecutive passes over a 4 GiB array that are chunked so that the
// consumer reads what the producer has just written cost ONE read + ONE write of HBM instead of two of each.
// build: hipcc --offload-arch=gfx950 -O3 -o mall_bench mall_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float __attribute__((ext_vector_type(4))) f4;

// every workgroup copies one contiguous 64 KiB chunk (the fastest plain copy measured by copy_bench.hip)
template <int NTS>
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ a, f4* __restrict__ b) {
  const size_t lo = (size_t)blockIdx.x * 4096;
  f4 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = a[lo + threadIdx.x + u * 256];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    if (NTS) __builtin_nontemporal_store(v[u], b + lo + threadIdx.x + u * 256); else b[lo + threadIdx.x + u * 256] = v[u];
  }
}

template <typename F>
static double timed_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const size_t G = (size_t)4 << 30;
  char *src, *dst, *tmp;
  CK(hipMalloc(&src, G)); CK(hipMalloc(&dst, G)); CK(hipMalloc(&tmp, G));
  CK(hipMemset(src, 1, G)); CK(hipMemset(dst, 0, G)); CK(hipMemset(tmp, 0, G));
  // 1. ping-pong between two buffers of S bytes
  for (size_t mib : {16, 32, 64, 96, 128, 192, 256, 512, 1024, 4096}) {
    const size_t S = mib << 20;
    const unsigned blocks = (unsigned)(S / 65536);
    for (int nts = 0; nts < 2; ++nts) {
      int flip = 0;
      const double ms = timed_ms([&] {
        const f4* a = (const f4*)(flip ? tmp : src); f4* b = (f4*)(flip ? src : tmp);
        if (nts) hipLaunchKernelGGL(k_copy<1>, dim3(blocks), dim3(256), 0, 0, a, b);
        else hipLaunchKernelGGL(k_copy<0>, dim3(blocks), dim3(256), 0, 0, a, b);
        flip ^= 1;
      }, 40);
      printf("ping-pong 2 x %5zu MiB  nt-store %d   %8.4f ms   %8.1f GB/s\n", mib, nts, ms, 2.0 * S / ms / 1e6);
    }
  }
  // 2. src (4 GiB) -> ring of S bytes -> dst (4 GiB), chunk by chunk; the reference is two full passes through a 4 GiB tmp
  {
    const unsigned blocks = (unsigned)(G / 65536);
    const double ms = timed_ms([&] {
      hipLaunchKernelGGL(k_copy<0>, dim3(blocks), dim3(256), 0, 0, (const f4*)src, (f4*)tmp);
      hipLaunchKernelGGL(k_copy<0>, dim3(blocks), dim3(256), 0, 0, (const f4*)tmp, (f4*)dst);
    }, 5);
    printf("two full passes through a 4 GiB intermediate         %8.3f ms   (algorithmic 4 x 4 GiB: %8.1f GB/s)\n", ms, 4.0 * G / ms / 1e6);
  }
  hipStream_t s1, s2;
  CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  for (size_t mib : {16, 32, 64, 128, 256}) {
    const size_t S = mib << 20;
    const unsigned blocks = (unsigned)(S / 65536);
    const size_t chunks = G / S;
    for (int ring = 1; ring <= 2; ++ring) {  // ring = number of S-sized slots in the ring buffer
      for (int nts = 0; nts < 2; ++nts) {
        const double ms = timed_ms([&] {
          for (size_t c = 0; c < chunks; ++c) {
            char* slot = tmp + (c % ring) * S;
            if (nts) hipLaunchKernelGGL(k_copy<1>, dim3(blocks), dim3(256), 0, 0, (const f4*)(src + c * S), (f4*)slot);
            else hipLaunchKernelGGL(k_copy<0>, dim3(blocks), dim3(256), 0, 0, (const f4*)(src + c * S), (f4*)slot);
            hipLaunchKernelGGL(k_copy<1>, dim3(blocks), dim3(256), 0, 0, (const f4*)slot, (f4*)(dst + c * S));
          }
        }, 3);
        printf("chunked chain, chunk %4zu MiB, ring %d slot(s), nt producer store %d   %8.3f ms   (as 2 x 4 GiB: %8.1f GB/s)\n",
               mib, ring, nts, ms, 2.0 * G / ms / 1e6);
      }
    }
    // two streams: producer of chunk c+1 overlaps the consumer of chunk c (ring of 2 slots, events for the hand-over)
    {
      std::vector<hipEvent_t> done(chunks), freed(chunks);
      for (auto& ev : done) CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      for (auto& ev : freed) CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      for (size_t c = 0; c < chunks; ++c) {
        char* slot = tmp + (c % 2) * S;
        if (c >= 2) CK(hipStreamWaitEvent(s1, freed[c - 2], 0));
        hipLaunchKernelGGL(k_copy<0>, dim3(blocks), dim3(256), 0, s1, (const f4*)(src + c * S), (f4*)slot);
        CK(hipEventRecord(done[c], s1));
        CK(hipStreamWaitEvent(s2, done[c], 0));
        hipLaunchKernelGGL(k_copy<1>, dim3(blocks), dim3(256), 0, s2, (const f4*)slot, (f4*)(dst + c * S));
        CK(hipEventRecord(freed[c], s2));
      }
      CK(hipStreamWaitEvent(s1, freed[chunks - 1], 0));
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("chunked chain, chunk %4zu MiB, 2 streams, ring 2                     %8.3f ms   (as 2 x 4 GiB: %8.1f GB/s)\n", mib, ms, 2.0 * G / ms / 1e6);
    }
  }
  return 0;
}
