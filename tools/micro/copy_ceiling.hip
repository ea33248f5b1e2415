// copy_ceiling.hip -- is the 6.29 TB/s float4 copy of MI355X_MICROARCH.md reachable on these boxes, and what separates the
// library's 5.1-5.4 TB/s copy from it?    Sweeps, for a device copy b <- a:
//   * array size 64 MiB .. 4 GiB per array (the 256 MiB Infinity Cache holds the small ones),
//   * grid-stride vs contiguous chunks per workgroup, workgroups per CU, float4 loads in flight per thread,
//   * store flavour: plain, non-temporal; load flavour: plain, non-temporal,
//   * read-only and write-only streams (which side is short of its share),
//   * hipMemcpyAsync / hipMemsetAsync of the runtime for reference.
// build: hipcc --offload-arch=gfx950 -O3 -o copy_ceiling copy_ceiling.hip ; run: ./copy_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float __attribute__((ext_vector_type(4))) f4;

template <int NTL>
__device__ __forceinline__ f4 ld(const f4* p) { return NTL ? __builtin_nontemporal_load(p) : *p; }
template <int NTS>
__device__ __forceinline__ void st(f4* p, f4 v) { if (NTS) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride, U float4 per thread in flight
template <int U, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_gs(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = i + u * stride < n4 ? ld<NTL>(a + i + u * stride) : f4{0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n4) st<NTS>(b + i + u * stride, v[u]);
  }
}
// contiguous chunk of chunk4 float4 per workgroup, workgroups walk the chunks block-cyclically (persistent grid)
template <int U, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_chunk(const f4* __restrict__ a, f4* __restrict__ b, size_t n4, size_t chunk4) {
  for (size_t lo = (size_t)blockIdx.x * chunk4; lo < n4; lo += (size_t)gridDim.x * chunk4) {
    const size_t hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
      f4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = i + u * 256 < hi ? ld<NTL>(a + i + u * 256) : f4{0, 0, 0, 0};
#pragma unroll
      for (int u = 0; u < U; ++u) if (i + u * 256 < hi) st<NTS>(b + i + u * 256, v[u]);
    }
  }
}
template <int U, int NTL>
__global__ void __launch_bounds__(256) k_read(const f4* __restrict__ a, float* out, size_t n4, size_t chunk4) {
  f4 s = {0, 0, 0, 0};
  for (size_t lo = (size_t)blockIdx.x * chunk4; lo < n4; lo += (size_t)gridDim.x * chunk4) {
    const size_t hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) if (i + u * 256 < hi) s += ld<NTL>(a + i + u * 256);
    }
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}
template <int U, int NTS>
__global__ void __launch_bounds__(256) k_write(f4* __restrict__ b, size_t n4, size_t chunk4) {
  const f4 v = {1, 2, 3, 4};
  for (size_t lo = (size_t)blockIdx.x * chunk4; lo < n4; lo += (size_t)gridDim.x * chunk4) {
    const size_t hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
#pragma unroll
      for (int u = 0; u < U; ++u) if (i + u * 256 < hi) st<NTS>(b + i + u * 256, v);
    }
  }
}
// the library's seven-stream CG update shape: x -= alpha d; r -= alpha q (reads x, r, d, q, b; writes x, r) -- no reductions
template <int U>
__global__ void __launch_bounds__(256) k_cg7(f4* __restrict__ x, f4* __restrict__ r, const f4* __restrict__ d, const f4* __restrict__ q,
                                             const f4* __restrict__ bb, float* out, size_t n4, size_t chunk4, float alpha) {
  f4 s = {0, 0, 0, 0};
  for (size_t lo = (size_t)blockIdx.x * chunk4; lo < n4; lo += (size_t)gridDim.x * chunk4) {
    const size_t hi = lo + chunk4 < n4 ? lo + chunk4 : n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
      f4 vx[U], vr[U], vd[U], vq[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t j = i + u * 256 < hi ? i + u * 256 : lo;
        vx[u] = x[j], vr[u] = r[j], vd[u] = d[j], vq[u] = q[j], vb[u] = bb[j];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) if (i + u * 256 < hi) {
        const f4 nx = vx[u] - alpha * vd[u], nr = vr[u] - alpha * vq[u];
        __builtin_nontemporal_store(nx, x + i + u * 256);
        __builtin_nontemporal_store(nr, r + i + u * 256);
        s += nx * nr + nx * vb[u];
      }
    }
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}

static float time_ms(const std::function<void()>& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs\n", prop.name, cus);
  const size_t maxbytes = (size_t)4 << 30;
  f4 *a, *b, *c, *d, *e5;
  float* out;
  CK(hipMalloc(&a, maxbytes));
  CK(hipMalloc(&b, maxbytes));
  CK(hipMalloc(&c, maxbytes));
  CK(hipMalloc(&d, maxbytes));
  CK(hipMalloc(&e5, maxbytes));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 1, maxbytes));
  CK(hipMemset(b, 1, maxbytes));
  CK(hipMemset(c, 0, maxbytes));
  CK(hipMemset(d, 0, maxbytes));
  CK(hipMemset(e5, 0, maxbytes));
  auto report = [](const char* name, size_t bytes_moved, float ms) {
    printf("%-64s %8.3f ms %9.1f GB/s\n", name, ms, bytes_moved / (ms * 1e-3) / 1e9);
    fflush(stdout);
  };
  char name[160];
  // ---- 1. size sweep of the plain grid-stride float4 copy (the guide's kernel shape) and of the chunked one
  for (size_t mb : {64, 128, 256, 512, 1024, 2048, 4096}) {
    const size_t bytes = mb << 20, n4 = bytes / 16;
    const int reps = mb <= 256 ? 50 : 10;
    for (int wgcu : {4, 8, 16, 32}) {
      const int blocks = cus * wgcu;
      snprintf(name, sizeof name, "copy %4zu MiB grid-stride U=1, %2d WG/CU", mb, wgcu);
      report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_gs<1, 0, 0>), dim3(blocks), dim3(256), 0, 0, a, b, n4); }, reps));
    }
    snprintf(name, sizeof name, "copy %4zu MiB grid-stride U=4 NT store, 8 WG/CU", mb);
    report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_gs<4, 0, 1>), dim3(cus * 8), dim3(256), 0, 0, a, b, n4); }, reps));
    snprintf(name, sizeof name, "copy %4zu MiB one thread per float4 (n4/256 WGs)", mb);
    report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_gs<1, 0, 0>), dim3((unsigned)(n4 / 256)), dim3(256), 0, 0, a, b, n4); }, reps));
    snprintf(name, sizeof name, "copy %4zu MiB chunk 64 KiB U=4 NT store, 8 WG/CU", mb);
    report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_chunk<4, 0, 1>), dim3(cus * 8), dim3(256), 0, 0, a, b, n4, (size_t)4096); }, reps));
  }
  // ---- 2. at 4 GiB: chunk size x loads in flight x workgroups per CU x store / load flavour
  const size_t bytes = maxbytes, n4 = bytes / 16;
  for (size_t ckib : {16, 64, 256, 1024}) {
    for (int wgcu : {2, 4, 8, 16}) {
      const int blocks = cus * wgcu;
      const size_t chunk4 = ckib * 64;
      snprintf(name, sizeof name, "4 GiB chunk %4zu KiB U=4 plain/plain      %2d WG/CU", ckib, wgcu);
      report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_chunk<4, 0, 0>), dim3(blocks), dim3(256), 0, 0, a, b, n4, chunk4); }, 10));
      snprintf(name, sizeof name, "4 GiB chunk %4zu KiB U=4 plain/NT store   %2d WG/CU", ckib, wgcu);
      report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_chunk<4, 0, 1>), dim3(blocks), dim3(256), 0, 0, a, b, n4, chunk4); }, 10));
      snprintf(name, sizeof name, "4 GiB chunk %4zu KiB U=4 NT load/NT store %2d WG/CU", ckib, wgcu);
      report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_chunk<4, 1, 1>), dim3(blocks), dim3(256), 0, 0, a, b, n4, chunk4); }, 10));
      snprintf(name, sizeof name, "4 GiB chunk %4zu KiB U=8 plain/NT store   %2d WG/CU", ckib, wgcu);
      report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_chunk<8, 0, 1>), dim3(blocks), dim3(256), 0, 0, a, b, n4, chunk4); }, 10));
      snprintf(name, sizeof name, "4 GiB chunk %4zu KiB U=2 plain/NT store   %2d WG/CU", ckib, wgcu);
      report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_chunk<2, 0, 1>), dim3(blocks), dim3(256), 0, 0, a, b, n4, chunk4); }, 10));
    }
  }
  // ---- 3. one-sided streams
  for (int wgcu : {4, 8, 16}) {
    const int blocks = cus * wgcu;
    snprintf(name, sizeof name, "4 GiB read only  U=4 plain   %2d WG/CU", wgcu);
    report(name, bytes, time_ms([&] { hipLaunchKernelGGL((k_read<4, 0>), dim3(blocks), dim3(256), 0, 0, a, out, n4, (size_t)4096); }, 10));
    snprintf(name, sizeof name, "4 GiB read only  U=4 NT      %2d WG/CU", wgcu);
    report(name, bytes, time_ms([&] { hipLaunchKernelGGL((k_read<4, 1>), dim3(blocks), dim3(256), 0, 0, a, out, n4, (size_t)4096); }, 10));
    snprintf(name, sizeof name, "4 GiB write only U=4 plain   %2d WG/CU", wgcu);
    report(name, bytes, time_ms([&] { hipLaunchKernelGGL((k_write<4, 0>), dim3(blocks), dim3(256), 0, 0, b, n4, (size_t)4096); }, 10));
    snprintf(name, sizeof name, "4 GiB write only U=4 NT      %2d WG/CU", wgcu);
    report(name, bytes, time_ms([&] { hipLaunchKernelGGL((k_write<4, 1>), dim3(blocks), dim3(256), 0, 0, b, n4, (size_t)4096); }, 10));
  }
  report("4 GiB hipMemsetAsync", bytes, time_ms([&] { CK(hipMemsetAsync(b, 0, bytes, 0)); }, 10));
  report("4 GiB hipMemcpyAsync D2D", 2 * bytes, time_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 10));
  // ---- 4. the seven streams of the CG update (5 reads, 2 writes) on five 4 GiB arrays
  for (int wgcu : {4, 8, 16}) {
    snprintf(name, sizeof name, "CG update 7 streams U=2 chunk 64 KiB %2d WG/CU", wgcu);
    report(name, 7 * bytes, time_ms([&] { hipLaunchKernelGGL((k_cg7<2>), dim3(cus * wgcu), dim3(256), 0, 0, a, b, c, d, e5, out, n4, (size_t)4096, 0.5f); }, 5));
    snprintf(name, sizeof name, "CG update 7 streams U=1 chunk 64 KiB %2d WG/CU", wgcu);
    report(name, 7 * bytes, time_ms([&] { hipLaunchKernelGGL((k_cg7<1>), dim3(cus * wgcu), dim3(256), 0, 0, a, b, c, d, e5, out, n4, (size_t)4096, 0.5f); }, 5));
  }
  return 0;
}
