// nk_rng.hip -- numpy's Generator(PCG64).normal stream on the device (algorithm and references: nk_rng.h).
#include <hip/hip_runtime.h>

#include "nk_rng.h"
#include "nk_util.h"

namespace {

constexpr int RNG_THREADS = 256;

__device__ const uint64_t NK_ZIG_KI_DEV[256] = NK_ZIG_KI_INIT;
__device__ const uint64_t NK_ZIG_WI_DEV[256] = NK_ZIG_WI_BITS_INIT;
__device__ const uint64_t NK_ZIG_FI_DEV[256] = NK_ZIG_FI_BITS_INIT;

struct RngLds {
  NkPcgJump jt;
  uint64_t ki[256];
  double wi[256], fi[256];
};

// jump table of this stream (depends on inc) -- one thread, 64 dependent squarings
__global__ void k_rng_table(NkU128 inc, NkPcgJump* out) {
  nk_pcg_jump_table(inc, *out);
}

__device__ __forceinline__ NkZig rng_load_lds(RngLds& l, const NkPcgJump* jt) {
  const uint64_t* src = (const uint64_t*)jt;
  uint64_t* dst = (uint64_t*)&l.jt;
  for (int i = threadIdx.x; i < (int)(sizeof(NkPcgJump) / 8); i += blockDim.x) dst[i] = src[i];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) {
    l.ki[i] = NK_ZIG_KI_DEV[i];
    l.wi[i] = __longlong_as_double((long long)NK_ZIG_WI_DEV[i]);
    l.fi[i] = __longlong_as_double((long long)NK_ZIG_FI_DEV[i]);
  }
  __syncthreads();
  return NkZig{l.ki, l.wi, l.fi};
}

// passes A + A2: per chunk the overrun and the number of normals it holds for its true entry; per block the sum
__global__ void __launch_bounds__(RNG_THREADS) k_rng_count(NkRngArgs a, const NkPcgJump* jt, uint8_t* over, uint8_t* cnt,
                                                           uint32_t* bsum, uint64_t* status) {
  __shared__ RngLds l;
  __shared__ int s_over[RNG_THREADS];
  __shared__ int s_sum[RNG_THREADS / 64];
  const NkZig z = rng_load_lds(l, jt);
  const int64_t k = (int64_t)blockIdx.x * RNG_THREADS + threadIdx.x;
  int c0 = 0, ov = 0;
  uint64_t m = 0;
  if (k < a.nchunks) nk_rng_pass_a(a, l.jt, z, k, c0, ov, m);
  s_over[threadIdx.x] = ov;
  __syncthreads();
  int c = 0;
  if (k < a.nchunks) {
    int e = 0;
    if (threadIdx.x > 0) {
      e = s_over[threadIdx.x - 1];
    } else if (k > 0) {  // the last chunk of the previous workgroup, once more (1/256 extra work, no second launch)
      int pc;
      uint64_t pm;
      nk_rng_pass_a(a, l.jt, z, k - 1, pc, e, pm);
    }
    unsigned err = 0;
    c = nk_rng_pass_a2(a, l.jt, z, k, e, c0, m, &err);
    if (err) atomicOr((unsigned long long*)&status[1], (unsigned long long)err);
    over[k] = (uint8_t)ov;
    cnt[k] = (uint8_t)c;
  }
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < RNG_THREADS / 64; ++w) t += s_sum[w];
    bsum[blockIdx.x] = (uint32_t)t;
  }
}

// pass S: exclusive prefix sum of the workgroup sums (one workgroup; 62 500 entries for 10^9 normals)
__global__ void __launch_bounds__(1024) k_rng_scan(const uint32_t* bsum, int64_t nblk, int64_t n, int64_t* boff, uint64_t* status) {
  __shared__ int64_t part[1024];
  const int64_t seg = (nblk + 1023) / 1024;
  const int64_t lo = (int64_t)threadIdx.x * seg, hi = lo + seg < nblk ? lo + seg : nblk;
  int64_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += bsum[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int64_t v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int64_t run = part[threadIdx.x] - s;
  for (int64_t i = lo; i < hi; ++i) {
    boff[i] = run;
    run += bsum[i];
  }
  if (threadIdx.x == 1023 && part[1023] < n) atomicOr((unsigned long long*)&status[1], (unsigned long long)NK_RNG_ERR_SHORT);
}

// pass B
template <typename T>
__global__ void __launch_bounds__(RNG_THREADS) k_rng_write(NkRngArgs a, const NkPcgJump* jt, const uint8_t* over, const uint8_t* cnt,
                                                           const int64_t* boff, T* out, uint64_t* status) {
  __shared__ RngLds l;
  __shared__ int s_scan[RNG_THREADS];
  const NkZig z = rng_load_lds(l, jt);
  const int64_t k = (int64_t)blockIdx.x * RNG_THREADS + threadIdx.x;
  const int c = k < a.nchunks ? cnt[k] : 0;
  s_scan[threadIdx.x] = c;
  __syncthreads();
  for (int d = 1; d < RNG_THREADS; d <<= 1) {
    const int v = threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0;
    __syncthreads();
    s_scan[threadIdx.x] += v;
    __syncthreads();
  }
  if (k >= a.nchunks) return;
  const int64_t off = boff[blockIdx.x] + s_scan[threadIdx.x] - c;
  const int entry = k > 0 ? over[k - 1] : 0;
  nk_rng_pass_b<T>(a, l.jt, z, k, entry, off, out, &status[0]);
}

// ---- bounded integers (nk_rng.h: NkIntArgs) ----------------------------------------------------------------------------------
// pass 1: accepted words per thread -> per-block sums (the scan is k_rng_scan's); pass 2: the accepted values behind the
// exclusive sum.  cnt[] holds at most 64 per thread.
__global__ void __launch_bounds__(RNG_THREADS) k_int_count(NkIntArgs a, const NkPcgJump* jt, int64_t nthreads, uint8_t* cnt, uint32_t* bsum) {
  __shared__ NkPcgJump l_jt;
  __shared__ int s_sum[RNG_THREADS / 64];
  for (int i = threadIdx.x; i < (int)(sizeof(NkPcgJump) / sizeof(uint64_t)); i += blockDim.x)
    reinterpret_cast<uint64_t*>(&l_jt)[i] = reinterpret_cast<const uint64_t*>(jt)[i];
  __syncthreads();
  const int64_t k = (int64_t)blockIdx.x * RNG_THREADS + threadIdx.x;
  int c = 0;
  if (k < nthreads) {
    c = nk_int_count(a, l_jt, k);
    cnt[k] = (uint8_t)c;
  }
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < RNG_THREADS / 64; ++w) t += s_sum[w];
    bsum[blockIdx.x] = (uint32_t)t;
  }
}
__global__ void __launch_bounds__(RNG_THREADS) k_int_write(NkIntArgs a, const NkPcgJump* jt, int64_t nthreads, const uint8_t* cnt,
                                                           const int64_t* boff, int64_t* out, uint64_t* status) {
  __shared__ NkPcgJump l_jt;
  __shared__ int s_scan[RNG_THREADS];
  for (int i = threadIdx.x; i < (int)(sizeof(NkPcgJump) / sizeof(uint64_t)); i += blockDim.x)
    reinterpret_cast<uint64_t*>(&l_jt)[i] = reinterpret_cast<const uint64_t*>(jt)[i];
  const int64_t k = (int64_t)blockIdx.x * RNG_THREADS + threadIdx.x;
  const int c = k < nthreads ? cnt[k] : 0;
  s_scan[threadIdx.x] = c;
  __syncthreads();
  for (int d = 1; d < RNG_THREADS; d <<= 1) {
    const int v = threadIdx.x >= d ? s_scan[threadIdx.x - d] : 0;
    __syncthreads();
    s_scan[threadIdx.x] += v;
    __syncthreads();
  }
  if (k >= nthreads) return;
  const int64_t off = boff[blockIdx.x] + s_scan[threadIdx.x] - c;
  if (off < a.n) nk_int_write(a, l_jt, k, off, out, &status[0]);
}

struct RngLayout {
  int64_t nchunks, nblk;
  int64_t o_over, o_cnt, o_bsum, o_boff, o_table, total;
};
RngLayout rng_layout(int64_t n, int attempt) {
  RngLayout L;
  L.nchunks = nk_rng_chunks_for(n, attempt);
  L.nblk = (L.nchunks + RNG_THREADS - 1) / RNG_THREADS;
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  L.o_over = 0;
  L.o_cnt = up(L.o_over + L.nchunks);
  L.o_bsum = up(L.o_cnt + L.nchunks);
  L.o_boff = up(L.o_bsum + 4 * L.nblk);
  L.o_table = up(L.o_boff + 8 * L.nblk);
  L.total = up(L.o_table + (int64_t)sizeof(NkPcgJump));
  return L;
}

template <typename T, int MODE>
__global__ void __launch_bounds__(256) k_rng_fixed(NkU128 state, NkU128 inc, const NkPcgJump* jt, int64_t n, double low,
                                                   double scale, T* out) {
  __shared__ NkPcgJump l_jt;
  for (int i = threadIdx.x; i < (int)(sizeof(NkPcgJump) / sizeof(uint64_t)); i += blockDim.x)
    reinterpret_cast<uint64_t*>(&l_jt)[i] = reinterpret_cast<const uint64_t*>(jt)[i];
  __syncthreads();
  nk_rng_fixed_body<T, MODE>(state, inc, l_jt, n, low, scale, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, out);
}

template <int MODE>
int rng_fixed_launch(const uint64_t* state, const uint64_t* inc, int64_t n, double low, double scale, void* out, int dtype,
                     void* scratch, hipStream_t stream, const char* what) {
  if (!state || !inc || n < 0 || (n > 0 && (!out || !scratch)) || (dtype != NK_F32 && dtype != NK_F64))
    return nk_set_error(NK_ERR_INVALID, "nk_pcg64 fixed-rate draw: bad argument");
  if (!(inc[1] & 1)) return nk_set_error(NK_ERR_INVALID, "nk_pcg64 fixed-rate draw: the increment of a PCG64 stream is odd");
  if (n == 0) return NK_OK;
  const NkU128 s{state[0], state[1]}, c{inc[0], inc[1]};
  NkPcgJump* jt = (NkPcgJump*)scratch;
  hipLaunchKernelGGL(k_rng_table, dim3(1), dim3(1), 0, stream, c, jt);
  const int64_t per = MODE == 0 ? NK_RNG_FIX : 2 * NK_RNG_FIX;
  const int64_t threads = (n + per - 1) / per;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  if (dtype == NK_F32)
    hipLaunchKernelGGL((k_rng_fixed<float, MODE>), dim3(blocks), dim3(256), 0, stream, s, c, jt, n, low, scale, (float*)out);
  else
    hipLaunchKernelGGL((k_rng_fixed<double, MODE>), dim3(blocks), dim3(256), 0, stream, s, c, jt, n, low, scale, (double*)out);
  return nk_check_launch(what);
}

}  // namespace

extern "C" int64_t nk_pcg64_fixed_scratch_bytes(void) { return (int64_t)sizeof(NkPcgJump); }

extern "C" int nk_pcg64_uniform(const uint64_t* state, const uint64_t* inc, int64_t n, double low, double high, void* out,
                                int dtype, void* scratch, void* stream) {
  return rng_fixed_launch<0>(state, inc, n, low, high - low, out, dtype, scratch, (hipStream_t)stream, "nk_pcg64_uniform");
}

extern "C" int nk_pcg64_pm1(const uint64_t* state, const uint64_t* inc, int64_t n, void* out, int dtype, int complex_units,
                            void* scratch, void* stream) {
  if (complex_units)
    return rng_fixed_launch<2>(state, inc, n, 0.0, 0.0, out, dtype, scratch, (hipStream_t)stream, "nk_pcg64_pm1");
  return rng_fixed_launch<1>(state, inc, n, 0.0, 0.0, out, dtype, scratch, (hipStream_t)stream, "nk_pcg64_pm1");
}

extern "C" int64_t nk_pcg64_integers_scratch_bytes(int64_t nthreads) {
  if (nthreads < 0) return 0;
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  const int64_t nblk = (nthreads + RNG_THREADS - 1) / RNG_THREADS;
  return up(nthreads) + up(4 * nblk) + up(8 * nblk) + up((int64_t)sizeof(NkPcgJump));
}

extern "C" int nk_pcg64_integers(const uint64_t* state, const uint64_t* inc, int64_t n, int64_t low, uint64_t rng, int64_t nthreads,
                                 int64_t* out, void* scratch, uint64_t* status, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!state || !inc || n < 0 || nthreads < 0 || !status || (n > 0 && (!out || !scratch || nthreads < 1)) || rng == 0)
    return nk_set_error(NK_ERR_INVALID, "nk_pcg64_integers: bad argument (rng = high - low > 0)");
  if (!(inc[1] & 1)) return nk_set_error(NK_ERR_INVALID, "nk_pcg64_integers: the increment of a PCG64 stream is odd");
  hipError_t e = hipMemsetAsync(status, 0, 2 * sizeof(uint64_t), stream);
  if (e != hipSuccess) return nk_set_hip_error(e, "nk_pcg64_integers: hipMemsetAsync");
  if (n == 0) return NK_OK;
  NkIntArgs a;
  a.state = NkU128{state[0], state[1]};
  a.inc = NkU128{inc[0], inc[1]};
  a.n = n;
  a.rng = rng;
  a.low = low;
  a.wide = rng > 0xFFFFFFFFull ? 1 : 0;
  if (a.wide) a.threshold = rng == ~0ull ? 0 : (~0ull - rng) % (rng + 1ull);
  else a.threshold = rng == 0xFFFFFFFFull ? 0 : (0xFFFFFFFFull - rng) % (rng + 1ull);
  auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
  const int64_t nblk = (nthreads + RNG_THREADS - 1) / RNG_THREADS;
  unsigned char* sc = (unsigned char*)scratch;
  uint8_t* cnt = sc;
  uint32_t* bsum = (uint32_t*)(sc + up(nthreads));
  int64_t* boff = (int64_t*)(sc + up(nthreads) + up(4 * nblk));
  NkPcgJump* jt = (NkPcgJump*)(sc + up(nthreads) + up(4 * nblk) + up(8 * nblk));
  hipLaunchKernelGGL(k_rng_table, dim3(1), dim3(1), 0, stream, a.inc, jt);
  hipLaunchKernelGGL(k_int_count, dim3((unsigned)nblk), dim3(RNG_THREADS), 0, stream, a, jt, nthreads, cnt, bsum);
  hipLaunchKernelGGL(k_rng_scan, dim3(1), dim3(1024), 0, stream, bsum, nblk, n, boff, status);
  hipLaunchKernelGGL(k_int_write, dim3((unsigned)nblk), dim3(RNG_THREADS), 0, stream, a, jt, nthreads, cnt, boff, out, status);
  return nk_check_launch("nk_pcg64_integers");
}

extern "C" int64_t nk_pcg64_normal_scratch_bytes(int64_t n, int attempt) {
  if (n < 0 || attempt < 0 || attempt > 8) return 0;
  return rng_layout(n, attempt).total;
}

extern "C" int nk_pcg64_normal(const uint64_t* state, const uint64_t* inc, int64_t n, double mean, double std, void* out,
                               int dtype, void* scratch, int64_t scratch_bytes, int attempt, uint64_t* status,
                               void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!state || !inc || n < 0 || !status || (n > 0 && (!out || !scratch)) || (dtype != NK_F32 && dtype != NK_F64) ||
      attempt < 0 || attempt > 8)
    return nk_set_error(NK_ERR_INVALID, "nk_pcg64_normal: bad argument");
  if (!(inc[1] & 1)) return nk_set_error(NK_ERR_INVALID, "nk_pcg64_normal: the increment of a PCG64 stream is odd");
  hipError_t e = hipMemsetAsync(status, 0, 2 * sizeof(uint64_t), stream);
  if (e != hipSuccess) return nk_set_hip_error(e, "nk_pcg64_normal: hipMemsetAsync");
  if (n == 0) return NK_OK;
  const RngLayout L = rng_layout(n, attempt);
  if (scratch_bytes < L.total) return nk_set_error(NK_ERR_INVALID, "nk_pcg64_normal: scratch smaller than nk_pcg64_normal_scratch_bytes");
  unsigned char* sc = (unsigned char*)scratch;
  NkRngArgs a;
  a.state = NkU128{state[0], state[1]};
  a.inc = NkU128{inc[0], inc[1]};
  a.n = n;
  a.nchunks = L.nchunks;
  a.mean = mean;
  a.std = std;
  NkPcgJump* jt = (NkPcgJump*)(sc + L.o_table);
  uint8_t* over = sc + L.o_over;
  uint8_t* cnt = sc + L.o_cnt;
  uint32_t* bsum = (uint32_t*)(sc + L.o_bsum);
  int64_t* boff = (int64_t*)(sc + L.o_boff);
  hipLaunchKernelGGL(k_rng_table, dim3(1), dim3(1), 0, stream, a.inc, jt);
  hipLaunchKernelGGL(k_rng_count, dim3((unsigned)L.nblk), dim3(RNG_THREADS), 0, stream, a, jt, over, cnt, bsum, status);
  hipLaunchKernelGGL(k_rng_scan, dim3(1), dim3(1024), 0, stream, bsum, L.nblk, n, boff, status);
  if (dtype == NK_F32)
    hipLaunchKernelGGL(k_rng_write<float>, dim3((unsigned)L.nblk), dim3(RNG_THREADS), 0, stream, a, jt, over, cnt, boff, (float*)out, status);
  else
    hipLaunchKernelGGL(k_rng_write<double>, dim3((unsigned)L.nblk), dim3(RNG_THREADS), 0, stream, a, jt, over, cnt, boff, (double*)out, status);
  return nk_check_launch("nk_pcg64_normal");
}
