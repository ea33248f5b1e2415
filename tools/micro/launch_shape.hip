// launch_shape.hip -- follow-up of copy_ceiling.hip: the plain "one float4 per thread, n/256 workgroups" copy reaches the
// guide's 6.2 TB/s while every persistent shape (grid-stride, block-cyclic chunks) stays at 4.5-5.5 TB/s.  Which NON-persistent
// shape keeps that rate with U vectors per thread (fewer workgroups -> fewer reduction partials), for a copy and for the
// seven streams of the CG update, with and without a per-workgroup partial-sum store?
// build: hipcc --offload-arch=gfx950 -O3 -o launch_shape launch_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1);} } while (0)
typedef float __attribute__((ext_vector_type(4))) f4;

// workgroup b owns the contiguous piece [b * 256 * U, (b + 1) * 256 * U) vectors; thread t takes t, t + 256, ...
template <int U, int NTL, int NTS>
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
  const size_t lo = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = NTL ? __builtin_nontemporal_load(a + lo + u * 256) : a[lo + u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(v[u], b + lo + u * 256); else b[lo + u * 256] = v[u]; }
}
// thread t takes U CONSECUTIVE vectors (64 B per thread for U = 4)
template <int U>
__global__ void __launch_bounds__(256) k_copy_consec(const f4* __restrict__ a, f4* __restrict__ b, size_t n4) {
  const size_t lo = ((size_t)blockIdx.x * 256 + threadIdx.x) * U;
  f4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = a[lo + u];
#pragma unroll
  for (int u = 0; u < U; ++u) b[lo + u] = v[u];
}
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double red[4];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
// seven streams of the CG update with three fp64 reductions stored per workgroup (PART) -- the library's FCgUpdate
template <int U, int NTL, int PART>
__global__ void __launch_bounds__(256) k_cg7(f4* __restrict__ x, f4* __restrict__ r, const f4* __restrict__ d, const f4* __restrict__ q,
                                             const f4* __restrict__ bb, double* part, size_t n4, float alpha) {
  const size_t lo = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 vx[U], vr[U], vd[U], vq[U], vb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t j = lo + u * 256;
    if (NTL) {
      vx[u] = __builtin_nontemporal_load(x + j), vr[u] = __builtin_nontemporal_load(r + j), vd[u] = __builtin_nontemporal_load(d + j);
      vq[u] = __builtin_nontemporal_load(q + j), vb[u] = __builtin_nontemporal_load(bb + j);
    } else {
      vx[u] = x[j], vr[u] = r[j], vd[u] = d[j], vq[u] = q[j], vb[u] = bb[j];
    }
  }
  double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const f4 nx = vx[u] - alpha * vd[u], nr = vr[u] - alpha * vq[u];
    __builtin_nontemporal_store(nx, x + lo + u * 256);
    __builtin_nontemporal_store(nr, r + lo + u * 256);
    for (int k = 0; k < 4; ++k) s0 += (double)nr[k] * nr[k], s1 += (double)nx[k] * nr[k], s2 += (double)nx[k] * vb[u][k];
  }
  if (PART) {
    const double t0 = block_sum(s0), t1 = block_sum(s1), t2 = block_sum(s2);
    if (threadIdx.x == 0) part[blockIdx.x] = t0, part[gridDim.x + blockIdx.x] = t1, part[2 * (size_t)gridDim.x + blockIdx.x] = t2;
  } else if (s0 + s1 + s2 == 12345.678) part[0] = s0;
}
// second stage: fixed-order sum of the partials (one workgroup per reduction)
__global__ void __launch_bounds__(1024) k_fold(const double* part, size_t nb, double* out) {
  __shared__ double red[16];
  double v = 0;
  for (size_t i = threadIdx.x; i < nb; i += 1024) v += part[blockIdx.x * nb + i];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { double s = 0; for (int w = 0; w < 16; ++w) s += red[w]; out[blockIdx.x] = s; }
}
// dot: 2 read streams + partials
template <int U>
__global__ void __launch_bounds__(256) k_dot(const f4* __restrict__ a, const f4* __restrict__ b, double* part, size_t n4) {
  const size_t lo = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  f4 va[U], vb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) va[u] = a[lo + u * 256], vb[u] = b[lo + u * 256];
  double s = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) for (int k = 0; k < 4; ++k) s += (double)va[u][k] * vb[u][k];
  const double t = block_sum(s);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
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
  const size_t bytes = (size_t)4 << 30, n4 = bytes / 16;
  f4 *a, *b, *c, *d, *e5;
  double *part, *out;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes)); CK(hipMalloc(&e5, bytes));
  CK(hipMalloc(&part, 3 * (n4 / 256) * sizeof(double)));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes)); CK(hipMemset(e5, 0, bytes));
  auto report = [](const char* name, size_t moved, float ms) { printf("%-72s %8.3f ms %9.1f GB/s\n", name, ms, moved / (ms * 1e-3) / 1e9); fflush(stdout); };
  char name[160];
#define COPY(U)                                                                                                               \
  snprintf(name, sizeof name, "copy 4 GiB, piece = 256 x %2d vectors (%3d KiB), plain / plain", U, U * 4);                      \
  report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_copy<U, 0, 0>), dim3((unsigned)(n4 / 256 / U)), dim3(256), 0, 0, a, b, n4); }, 10)); \
  snprintf(name, sizeof name, "copy 4 GiB, piece = 256 x %2d vectors (%3d KiB), plain / NT store", U, U * 4);                   \
  report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_copy<U, 0, 1>), dim3((unsigned)(n4 / 256 / U)), dim3(256), 0, 0, a, b, n4); }, 10)); \
  snprintf(name, sizeof name, "copy 4 GiB, piece = 256 x %2d vectors (%3d KiB), NT load / NT store", U, U * 4);                 \
  report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_copy<U, 1, 1>), dim3((unsigned)(n4 / 256 / U)), dim3(256), 0, 0, a, b, n4); }, 10)); \
  snprintf(name, sizeof name, "copy 4 GiB, piece = 256 x %2d vectors, NT load / plain store", U);                               \
  report(name, 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_copy<U, 1, 0>), dim3((unsigned)(n4 / 256 / U)), dim3(256), 0, 0, a, b, n4); }, 10));
  COPY(1) COPY(2) COPY(4) COPY(8) COPY(16)
  report("copy 4 GiB, thread takes 2 consecutive vectors", 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_copy_consec<2>), dim3((unsigned)(n4 / 512)), dim3(256), 0, 0, a, b, n4); }, 10));
  report("copy 4 GiB, thread takes 4 consecutive vectors", 2 * bytes, time_ms([&] { hipLaunchKernelGGL((k_copy_consec<4>), dim3((unsigned)(n4 / 1024)), dim3(256), 0, 0, a, b, n4); }, 10));
#define CG(U, NTL, PART)                                                                                                       \
  snprintf(name, sizeof name, "CG update 7 streams, piece = 256 x %d vectors, %s loads, %s", U, NTL ? "NT" : "plain",           \
           PART ? "3 partials per workgroup + fold" : "no reductions");                                                       \
  report(name, 7 * bytes, time_ms([&] {                                                                                        \
    hipLaunchKernelGGL((k_cg7<U, NTL, PART>), dim3((unsigned)(n4 / 256 / U)), dim3(256), 0, 0, a, b, c, d, e5, part, n4, 0.5f); \
    if (PART) hipLaunchKernelGGL(k_fold, dim3(3), dim3(1024), 0, 0, part, n4 / 256 / U, out);                                  \
  }, 5));
  CG(1, 0, 0) CG(1, 0, 1) CG(1, 1, 1) CG(2, 0, 0) CG(2, 0, 1) CG(2, 1, 1) CG(4, 0, 0) CG(4, 0, 1) CG(4, 1, 1)
#define DOT(U)                                                                                                                 \
  snprintf(name, sizeof name, "dot 2 streams, piece = 256 x %d vectors, partial per workgroup + fold", U);                      \
  report(name, 2 * bytes, time_ms([&] {                                                                                        \
    hipLaunchKernelGGL((k_dot<U>), dim3((unsigned)(n4 / 256 / U)), dim3(256), 0, 0, a, b, part, n4);                           \
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(1024), 0, 0, part, n4 / 256 / U, out);                                            \
  }, 10));
  DOT(1) DOT(2) DOT(4) DOT(8)
  return 0;
}
