// nk_vec.hip -- streaming vector kernels: reductions with wavefront (64-lane) shuffles and fp64
// accumulation, element-wise algebra, pointwise nonlinearities with derivative, bin gather / scatter,
// and the fused conjugate-gradient updates whose scalars never leave the device.
// All kernels are HBM-bound: 16-byte vector loads/stores when the pointers allow, grid-stride loops,
// at most 256 CUs x 8 workgroups.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "nk_core.h"
#include "nk_util.h"

static constexpr int NK_VEC_THREADS = 256;
static constexpr int NK_MAX_BLOCKS = 256 * 8;
static constexpr int NK_VEC_CHUNK = 4096;  // vectors (16 B each) a workgroup walks contiguously

// chunk length of a map functor in units of 256 vectors (F::CHUNK_UNITS, default 16 = 64 KiB per stream)
template <typename F, typename = void>
struct NkChunkUnits {
  static constexpr int value = NK_VEC_CHUNK / NK_VEC_THREADS;
};
template <typename F>
struct NkChunkUnits<F, std::void_t<decltype(F::CHUNK_UNITS)>> {
  static constexpr int value = F::CHUNK_UNITS;
};
static inline int nk_vec_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

static inline int nk_grid(int64_t nvec) {
  int64_t b = (nvec + NK_VEC_THREADS - 1) / NK_VEC_THREADS;
  if (b < 1) b = 1;
  if (b > NK_MAX_BLOCKS) b = NK_MAX_BLOCKS;
  return (int)b;
}

static inline int nk_grid_red(int64_t nvec) {
  int64_t b = (nvec + 32 * NK_VEC_THREADS - 1) / (32 * NK_VEC_THREADS);
  if (b > 512) b = std::max<int64_t>(512, (nvec + 64 * NK_VEC_THREADS - 1) / (64 * NK_VEC_THREADS));
  if (b < 1) b = 1;
  if (b > NK_MAX_BLOCKS) b = NK_MAX_BLOCKS;
  return (int)b;
}

// ---- grouping of the reductions (NkRedLayout, nk_util.h) -------------------------------------------------------------
// unit length of an array of n elements, vector width v: n / NK_RED_UNITS when every unit is a whole number of 256-vector
// rows and long enough to be worth its own sub-grid, else 0 (the array is ONE unit).  NK_RED_UNITS_OFF=1: always 0.
static inline int64_t nk_red_unit_of(int64_t n, int v) {
  static const int off = nk_vec_env_int("NK_RED_UNITS_OFF", 0);
  const int64_t row = (int64_t)v * NK_VEC_THREADS;
  if (off || n <= 0 || n % (NK_RED_UNITS * row) != 0 || n / (NK_RED_UNITS * row) < 2) return 0;
  return n / NK_RED_UNITS;
}
namespace {
thread_local NkRedLayout t_red_layout = {0, 0, 0, 0, 0, 0, nullptr};  // k_local == 0: none set (nk_red_layout)
}
static inline NkRedLayout nk_red_layout_for(int64_t n, int v) {
  const NkRedLayout& set = t_red_layout;
  if (set.k_local > 0 && set.unit_elems > 0 && n == set.k_local * set.unit_elems && set.unit_elems % ((int64_t)v * NK_VEC_THREADS) == 0)
    return set;  // a shard of whole units, as announced by the caller
  const int64_t unit = nk_red_unit_of(n, v);
  if (unit == 0) return NkRedLayout{1, 0, 1, 1, 1, 0, nullptr};
  return NkRedLayout{NK_RED_UNITS, unit, NK_RED_UNITS, NK_RED_UNITS, NK_RED_UNITS, 0, nullptr};
}

template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
  typedef float4 type;
  static constexpr int N = 4;
};
template <>
struct VecOf<double> {
  typedef double2 type;
  static constexpr int N = 2;
};

template <typename T>
__device__ __forceinline__ void nk_vload(const T* p, T (&v)[VecOf<T>::N]) {
  typename VecOf<T>::type t = *reinterpret_cast<const typename VecOf<T>::type*>(p);
  const T* q = reinterpret_cast<const T*>(&t);
#pragma unroll
  for (int i = 0; i < VecOf<T>::N; ++i) v[i] = q[i];
}
template <typename T>
__device__ __forceinline__ void nk_vstore(T* p, const T (&v)[VecOf<T>::N]) {
  typename VecOf<T>::type t;
  T* q = reinterpret_cast<T*>(&t);
#pragma unroll
  for (int i = 0; i < VecOf<T>::N; ++i) q[i] = v[i];
  *reinterpret_cast<typename VecOf<T>::type*>(p) = t;
}

static inline bool nk_aligned16(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// wave64 shuffle reduction + one LDS hop, result valid in thread 0
__device__ __forceinline__ double nk_block_sum(double v) {
  __shared__ double red[NK_VEC_THREADS / 64];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

// scratch of the deterministic reductions: per (device, stream), see nk_red_scratch (nk_util.h)
static_assert(NK_RED_MAX_BLOCKS >= NK_MAX_BLOCKS, "reduction scratch must cover the largest grid");

// ---- generic "map with up to 4 inputs, 3 outputs and 3 reductions" skeleton ---------------------------
// F::apply(const T* in[..] values, T* outs, double* red) is called per element.
template <typename T, typename F, bool VEC>
__device__ __forceinline__ void nk_map_body(int64_t n, const F& f, int cu_max, NkRedScratch rs, NkRedLayout lay) {
  static_assert(F::NRED <= NK_RED_MAX, "too many reductions for the scratch of nk_red_scratch");
  constexpr int V = VEC ? VecOf<T>::N : 1;
  double red[F::NRED > 0 ? F::NRED : 1];
#pragma unroll
  for (int r = 0; r < (F::NRED > 0 ? F::NRED : 1); ++r) red[r] = 0.0;
  const int64_t nvec = n / V;
  // the launch is a row of lay.k_local UNITS (NkRedLayout; one unit = the whole array for maps without reductions and
  // for short or odd lengths), every unit with gridDim.x / k_local workgroups of its own
  const int G = (int)gridDim.x / lay.k_local;
  const int unit = (int)blockIdx.x / G, ub = (int)blockIdx.x - unit * G;
  const int64_t uvec = lay.unit_elems > 0 ? lay.unit_elems / V : nvec;
  const int64_t u0 = (int64_t)unit * uvec, u1 = u0 + uvec;
  // block-cyclic over CONTIGUOUS chunks of NK_VEC_CHUNK vectors (64 KiB per stream): a workgroup that walks 64 KiB in
  // a row keeps its DRAM pages open -- a plain 1R+1W copy runs at 5.3 TB/s this way against 4.7-4.85 TB/s grid-stride
  // (tools/micro/copy_bench.hip, 4 GiB arrays)
  // (small arrays: shorter chunks so that every workgroup of the grid has work; grid and chunk depend on the unit only)
  int64_t cu = uvec / ((int64_t)G * NK_VEC_THREADS);
  cu = cu < 1 ? 1 : cu > cu_max ? cu_max : cu;
  const int64_t chunk = cu * NK_VEC_THREADS;
  const int64_t nchunk = (uvec + chunk - 1) / chunk;
  for (int64_t c = ub; c < nchunk; c += G) {
    const int64_t lo = u0 + c * chunk + threadIdx.x;
    if (cu == NK_VEC_CHUNK / NK_VEC_THREADS && lo + (NK_VEC_CHUNK - NK_VEC_THREADS) < u1) {  // the common, unrolled case
#pragma unroll 4
      for (int u = 0; u < NK_VEC_CHUNK / NK_VEC_THREADS; ++u) f.template run<V>((lo + u * NK_VEC_THREADS) * V, red);
    } else {
      const int64_t hi = u0 + (c + 1) * chunk < u1 ? u0 + (c + 1) * chunk : u1;
      for (int64_t i = lo; i < hi; i += NK_VEC_THREADS) f.template run<V>(i * V, red);
    }
  }
  if (VEC) {  // scalar tail (one-unit launches only: units are whole vectors)
    const int64_t tail0 = nvec * V;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < n - tail0) f.template run<1>(tail0 + gid, red);
  }
  if constexpr (F::NRED > 0) {
    // DETERMINISTIC reduction: every workgroup stores its partial sums; the workgroup that takes the last ticket adds
    // them up -- per unit over the unit's workgroups (a wavefront per unit: lanes stride the partials, then a shuffle
    // tree), then the units in order -- and updates result[].  Identical inputs therefore give identical bits on every
    // launch and on every GPU of a node (grid and grouping are functions of the length only), and a shard made of whole
    // units gives the unit sums of the full array (NkRedLayout) -- the replicated CG / line-search scalars of a multi-rank
    // KL minimisation agree without any exchange, and the sharded ones do not depend on the number of ranks.  The
    // scratch belongs to the (device, stream) pair of the launch; launches on one stream are serialised, and every launch
    // leaves the ticket at zero.
    __shared__ bool is_last;
    __shared__ double usum[NK_RED_MAX][NK_RED_UNITS];
#pragma unroll
    for (int r = 0; r < F::NRED; ++r) {
      const double s = nk_block_sum(red[r]);
      // (no __threadfence(): the partial goes to the device's coherence point by an atomic exchange, nk_util.h)
      if (threadIdx.x == 0) nk_publish_partial(&rs.partial[r * NK_RED_MAX_BLOCKS + blockIdx.x], s);
    }
    if (threadIdx.x == 0) is_last = nk_take_last_ticket(rs.ticket, gridDim.x);
    __syncthreads();
    if (is_last) {
      nk_acquire_partials();
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      for (int u = wave; u < lay.k_local; u += NK_VEC_THREADS / 64) {
#pragma unroll
        for (int r = 0; r < F::NRED; ++r) {
          double v = 0.0;
          for (int b = lane; b < G; b += 64) v += nk_read_partial(&rs.partial[r * NK_RED_MAX_BLOCKS + u * G + b]);
          for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
          if (lane == 0) usum[r][u] = v;
        }
      }
      __syncthreads();
      if (lay.units_out) {
        for (int g = threadIdx.x; g < lay.k_global; g += NK_VEC_THREADS) {
          const int seg = g / lay.seg_stride, w = g - seg * lay.seg_stride - lay.seg_off;
          const int lu = seg * lay.seg_units + w;
          const bool mine = w >= 0 && w < lay.seg_units && lu < lay.k_local;
#pragma unroll
          for (int r = 0; r < F::NRED; ++r) lay.units_out[r * lay.k_global + g] = mine ? usum[r][lu] : 0.0;
        }
      } else if (threadIdx.x == 0) {
#pragma unroll
        for (int r = 0; r < F::NRED; ++r) {
          double s = 0.0;
          for (int u = 0; u < lay.k_local; ++u) s += usum[r][u];
          f.result[r] += s;
        }
      }
      if (threadIdx.x == 0) nk_reset_ticket(rs.ticket);
    }
  }
}
template <typename T, typename F, bool VEC>
__global__ void __launch_bounds__(NK_VEC_THREADS) k_map(int64_t n, F f, int cu_max, NkRedScratch rs, NkRedLayout lay) {
  nk_map_body<T, F, VEC>(n, f, cu_max, rs, lay);
}
// BATCHED launch (include/niftyk.h, "batched launches"): blockIdx.y = member; every member has its own functor (pointers,
// scalars), its own block partials and ticket, and the grid / chunking of the single launch -- the same bits per member
template <typename F>
struct NkBatchOf {
  F f[NK_MAX_BATCH];
};
template <typename T, typename F, bool VEC>
__global__ void __launch_bounds__(NK_VEC_THREADS) k_map_b(int64_t n, NkBatchOf<F> fb, int cu_max, NkRedScratch rs, NkRedLayout lay) {
  const int m = blockIdx.y;
  rs.partial += (size_t)m * NK_RED_MEMBER_STRIDE;
  rs.ticket += m;
  nk_map_body<T, F, VEC>(n, fb.f[m], cu_max, rs, lay);
}

// Element-wise maps WITHOUT reductions, 16-byte vectors: one vector per thread, one 4 KiB piece per workgroup, as many
// workgroups as pieces (no persistent loop).  The dispatcher hands the workgroups out in order, so at any moment the chip works
// on ONE narrow window of each stream (~10 MiB) instead of 2048 chunks spread over the whole array: a 4 GiB copy runs at
// 6.2 TB/s in this shape -- the float4 copy figure of MI355X_MICROARCH.md -- against 5.1-5.3 TB/s for the block-cyclic 64 KiB
// chunks of k_map, 4.5-5.4 TB/s grid-stride, 5.7 / 5.2 TB/s for pieces of 8 / 64 KiB (tools/micro/copy_ceiling.hip,
// launch_shape.hip; profiles/r04_copy_ceiling.log).  Kernels with reductions keep k_map: one partial per piece would be 2^18
// partials per sum, and the seven-stream CG update gains nothing from the shape (5.57 vs 5.53 TB/s).
template <typename T, typename F>
__global__ void __launch_bounds__(NK_VEC_THREADS) k_map_flat(int64_t n, F f) {
  static_assert(F::NRED == 0, "k_map_flat: maps without reductions only");
  constexpr int V = VecOf<T>::N;
  const int64_t nvec = n / V;
  const int64_t i = (int64_t)blockIdx.x * NK_VEC_THREADS + threadIdx.x;
  if (i < nvec) f.template run<V>(i * V, nullptr);
  if (i < n - nvec * V) f.template run<1>(nvec * V + i, nullptr);  // scalar tail (< V elements)
}
template <typename T, typename F>
__global__ void __launch_bounds__(NK_VEC_THREADS) k_map_flat_b(int64_t n, NkBatchOf<F> fb) {
  static_assert(F::NRED == 0, "k_map_flat: maps without reductions only");
  constexpr int V = VecOf<T>::N;
  const F& f = fb.f[blockIdx.y];
  const int64_t nvec = n / V;
  const int64_t i = (int64_t)blockIdx.x * NK_VEC_THREADS + threadIdx.x;
  if (i < nvec) f.template run<V>(i * V, nullptr);
  if (i < n - nvec * V) f.template run<1>(nvec * V + i, nullptr);
}

template <typename T, int V>
__device__ __forceinline__ void nk_ld(const T* p, int64_t i, T (&v)[V]) {
  if constexpr (V == 1)
    v[0] = p[i];
  else
    nk_vload<T>(p + i, v);
}
template <typename T, int V>
__device__ __forceinline__ void nk_st(T* p, int64_t i, const T (&v)[V]) {
  if constexpr (V == 1)
    p[i] = v[0];
  else
    nk_vstore<T>(p + i, v);
}

template <typename T, typename F>
static int nk_launch_map(int64_t n, const F& f, bool aligned, hipStream_t st, const char* what) {
  if (n <= 0) return NK_OK;
  static const int cu_env = nk_vec_env_int("NK_VEC_CU", 0);  // developer sweep: chunk length in units of 256 vectors
  const int cu_max = cu_env > 0 ? cu_env : NkChunkUnits<F>::value;
  NkRedScratch rs{nullptr, nullptr};
  NkRedLayout lay{1, 0, 1, 1, 1, 0, nullptr};
  if (F::NRED > 0) {
    const int rc = nk_red_scratch(st, &rs);
    if (rc != NK_OK) return rc;
    if (aligned) lay = nk_red_layout_for(n, VecOf<T>::N);
  }
  // kernels with reductions pay a fixed tail per workgroup (block sums, fence, ticket, ~2 us each, a few rounds of them
  // per CU): below ~10^8 elements that tail, not the streaming part, sets the time.  Their grid therefore gives every
  // thread >= 32 vectors (>= 64 from 512 workgroups on) before it grows to the full 2048 -- 2^22 fp64: nk_cg_update
  // 121 -> 51 us, nk_vdot 77 -> 22 us; 2^24: 295 -> 208 us, 109 -> 60 us; 2^30 fp32 unchanged (tools/gpu_vec_probe.py).
  // The grid stays a function of n only, so the sums stay bit-reproducible.
  if (aligned) {
    const int64_t nvec = n / VecOf<T>::N;
    if constexpr (F::NRED == 0) {
      static const int flat_env = nk_vec_env_int("NK_VEC_FLAT", 1);  // 0: the block-cyclic chunks for every map (A/B)
      const int64_t pieces = (nvec + NK_VEC_THREADS - 1) / NK_VEC_THREADS;
      if (flat_env && pieces > NK_MAX_BLOCKS && pieces < ((int64_t)1 << 31)) {
        hipLaunchKernelGGL((k_map_flat<T, F>), dim3((unsigned)pieces), dim3(NK_VEC_THREADS), 0, st, n, f);
        return nk_check_launch(what);
      }
    }
    int grid = F::NRED > 0 ? nk_grid_red(nvec) : nk_grid(nvec > 0 ? nvec : 1);
    if (lay.unit_elems > 0) {  // (a unit has the same sub-grid whether the launch holds one unit or all of them)
      static const int unit_grid = std::min(std::max(nk_vec_env_int("NK_RED_UNIT_GRID", 64), 1), NK_RED_MAX_BLOCKS / NK_RED_UNITS);
      grid = lay.k_local * std::min(nk_grid_red(lay.unit_elems / VecOf<T>::N), unit_grid);
    }
    hipLaunchKernelGGL((k_map<T, F, true>), dim3(grid), dim3(NK_VEC_THREADS), 0, st, n, f, cu_max, rs, lay);
  } else {
    const int grid = F::NRED > 0 ? nk_grid_red(n) : nk_grid(n);
    hipLaunchKernelGGL((k_map<T, F, false>), dim3(grid), dim3(NK_VEC_THREADS), 0, st, n, f, cu_max, rs, lay);
  }
  return nk_check_launch(what);
}

// nk_launch_map for `count` members in one launch (grid.y = member).  The members of a batch run the kernel the single
// call would pick for each of them -- which needs every operand of every member 16-byte aligned; a batch with an unaligned
// member goes through the single launches instead (same bits, just not batched).  Reductions: no rank-sharded layouts
// (nk_red_layout) inside a batch.
template <typename T, typename F>
static int nk_launch_map_b(int64_t n, const F* fs, int count, bool aligned, hipStream_t st, const char* what) {
  if (n <= 0) return NK_OK;
  if (count < 1 || count > NK_MAX_BATCH) return nk_set_error(NK_ERR_INVALID, "batched launch: 1 <= count <= NK_MAX_BATCH");
  if (count == 1 || !aligned || t_red_layout.k_local > 0) {
    for (int m = 0; m < count; ++m) {
      const int rc = nk_launch_map<T>(n, fs[m], aligned, st, what);
      if (rc != NK_OK) return rc;
    }
    return NK_OK;
  }
  static const int cu_env = nk_vec_env_int("NK_VEC_CU", 0);
  const int cu_max = cu_env > 0 ? cu_env : NkChunkUnits<F>::value;
  NkRedScratch rs{nullptr, nullptr};
  NkRedLayout lay{1, 0, 1, 1, 1, 0, nullptr};
  if (F::NRED > 0) {
    const int rc = nk_red_scratch(st, &rs);
    if (rc != NK_OK) return rc;
    lay = nk_red_layout_for(n, VecOf<T>::N);
  }
  NkBatchOf<F> fb;
  for (int m = 0; m < NK_MAX_BATCH; ++m) fb.f[m] = fs[m < count ? m : 0];
  const int64_t nvec = n / VecOf<T>::N;
  if constexpr (F::NRED == 0) {
    static const int flat_env = nk_vec_env_int("NK_VEC_FLAT", 1);
    const int64_t pieces = (nvec + NK_VEC_THREADS - 1) / NK_VEC_THREADS;
    if (flat_env && pieces > NK_MAX_BLOCKS && pieces < ((int64_t)1 << 31)) {
      hipLaunchKernelGGL((k_map_flat_b<T, F>), dim3((unsigned)pieces, (unsigned)count), dim3(NK_VEC_THREADS), 0, st, n, fb);
      return nk_check_launch(what);
    }
  }
  int grid = F::NRED > 0 ? nk_grid_red(nvec) : nk_grid(nvec > 0 ? nvec : 1);
  if (lay.unit_elems > 0) {
    static const int unit_grid = std::min(std::max(nk_vec_env_int("NK_RED_UNIT_GRID", 64), 1), NK_RED_MAX_BLOCKS / NK_RED_UNITS);
    grid = lay.k_local * std::min(nk_grid_red(lay.unit_elems / VecOf<T>::N), unit_grid);
  }
  hipLaunchKernelGGL((k_map_b<T, F, true>), dim3((unsigned)grid, (unsigned)count), dim3(NK_VEC_THREADS), 0, st, n, fb, cu_max, rs, lay);
  return nk_check_launch(what);
}

// ---- functors ------------------------------------------------------------------------------------
template <typename T>
struct FDot {
  static constexpr int NRED = 1;
  const T *a, *b;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double* red) const {
    T x[V], y[V];
    nk_ld<T, V>(a, i, x);
    nk_ld<T, V>(b, i, y);
#pragma unroll
    for (int k = 0; k < V; ++k) red[0] += (double)x[k] * (double)y[k];
  }
};

template <typename T>
struct FSum {
  static constexpr int NRED = 1;
  const T* a;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double* red) const {
    T x[V];
    nk_ld<T, V>(a, i, x);
#pragma unroll
    for (int k = 0; k < V; ++k) red[0] += (double)x[k];
  }
};

// sum, sum of squares and the number of ignored entries (NaN or exactly 0) -- the statistics of extra.minisanity
template <typename T>
struct FStats {
  static constexpr int NRED = 3;
  const T* a;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double* red) const {
    T x[V];
    nk_ld<T, V>(a, i, x);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const double v = (double)x[k];
      const bool ign = (v != v) || v == 0.0;
      if (!ign) {
        red[0] += v;
        red[1] += v * v;
      }
      red[2] += ign ? 1.0 : 0.0;
    }
  }
};

template <typename T>
struct FBinary {
  static constexpr int NRED = 0;
  int op;
  const T *a, *b;
  double as, bs;
  T* out;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double*) const {
    T x[V], y[V], r[V];
    if (a) nk_ld<T, V>(a, i, x);
    if (b) nk_ld<T, V>(b, i, y);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const T xv = a ? x[k] : (T)as, yv = b ? y[k] : (T)bs;
      r[k] = op == NK_OP_ADD ? xv + yv : op == NK_OP_SUB ? xv - yv : op == NK_OP_MUL ? xv * yv : xv / yv;
    }
    nk_st<T, V>(out, i, r);
  }
};

template <typename T>
struct FAxpby {
  static constexpr int NRED = 0;
  double alpha, beta;
  const T *x, *y;
  T* out;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double*) const {
    T a[V], b[V], r[V];
    nk_ld<T, V>(x, i, a);
    if (y) nk_ld<T, V>(y, i, b);
#pragma unroll
    for (int k = 0; k < V; ++k) r[k] = (T)(alpha * (double)a[k] + (y ? beta * (double)b[k] : 0.0));
    nk_st<T, V>(out, i, r);
  }
};

// out = alpha x + beta y  AND  result += sum(out^2) of the values as STORED (rounded to T): the sample position p +- r of a
// KL evaluation together with its prior term 1/2 |x|^2 -- one pass instead of an axpby and a dot
template <typename T>
struct FAxpbySq {
  static constexpr int NRED = 1;
  double alpha, beta;
  const T *x, *y;
  T* out;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double* red) const {
    T a[V], b[V], r[V];
    nk_ld<T, V>(x, i, a);
    nk_ld<T, V>(y, i, b);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      r[k] = (T)(alpha * (double)a[k] + beta * (double)b[k]);
      red[0] += (double)r[k] * (double)r[k];
    }
    nk_st<T, V>(out, i, r);
  }
};

template <typename T>
struct FPointwise {
  static constexpr int NRED = 0;
  int fn;
  double param, param2;
  const T* x;
  T *fx, *dfx;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double*) const {
    T a[V], f[V], d[V];
    nk_ld<T, V>(x, i, a);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const double v = (double)a[k];
      double fv, dv;
      switch (fn) {
        case 0: fv = dv = exp(v); break;
        case 1: fv = log(v); dv = 1.0 / v; break;
        case 2: fv = sqrt(v); dv = 0.5 / fv; break;
        case 3: fv = tanh(v); dv = 1.0 - fv * fv; break;
        case 4: { const double th = tanh(v); fv = 0.5 + 0.5 * th; dv = 0.5 - 0.5 * th * th; } break;
        case 5: fv = 1.0 / v; dv = -fv * fv; break;
        case 6: fv = pow(v, param); dv = param * pow(v, param - 1.0); break;
        case 7: fv = fabs(v); dv = v == 0.0 ? (double)NAN : (v > 0.0 ? 1.0 : -1.0); break;
        case 8: fv = log1p(v); dv = 1.0 / (1.0 + v); break;
        case 9: fv = expm1(v); dv = fv + 1.0; break;
        case 10: fv = atan(v); dv = 1.0 / (1.0 + v * v); break;
        case 11: fv = sin(v); dv = cos(v); break;
        case 12: fv = cos(v); dv = -sin(v); break;
        case 13: { fv = tan(v); const double c = cos(v); dv = 1.0 / (c * c); } break;
        case 14: {  // sin(pi v) / (pi v); derivative (cos(pi v) - sinc v) / v, 0 at the origin
          const double y = M_PI * v;
          fv = v == 0.0 ? 1.0 : sin(y) / y;
          dv = v == 0.0 ? 0.0 : (cos(y) - fv) / v;
        } break;
        case 15: fv = log10(v); dv = (1.0 / M_LN10) / v; break;
        case 16: fv = sinh(v); dv = cosh(v); break;
        case 17: fv = cosh(v); dv = sinh(v); break;
        case 18: fv = v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : v); dv = v == 0.0 ? (double)NAN : 0.0; break;
        case 19:  // softplus log(1 + e^v): the identity above 33, zero below -33 (pointwise.py:99-122)
          if (v > 33.0) fv = v, dv = 1.0;
          else if (v < -33.0) fv = 0.0, dv = 0.0;
          else fv = log(1.0 + exp(v)), dv = 1.0 / (1.0 + exp(-v));
          break;
        case 20: fv = pow(param, v); dv = log(param) * fv; break;
        case 21: fv = v >= 0.0 ? 1.0 : 0.0; dv = 0.0; break;
        default: {  // 22 clip to [param, param2]; an absent bound is -inf / +inf.  Derivative 0 where the value sits on a bound
          fv = v < param ? param : (v > param2 ? param2 : v);
          dv = (fv == param || fv == param2) ? 0.0 : 1.0;
        } break;
      }
      f[k] = (T)fv;
      d[k] = (T)dv;
    }
    if (fx) nk_st<T, V>(fx, i, f);
    if (dfx) nk_st<T, V>(dfx, i, d);
  }
};

template <typename T>
struct FGather {
  static constexpr int NRED = 0;
  const T* table;
  const int32_t* pidx;
  T* out;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double*) const {
    T r[V];
    int32_t p[V];
    if constexpr (V == 4) {  // the streamed operands as one 16-byte access each, past the cache the table lives in
      typedef int32_t i4 __attribute__((ext_vector_type(4)));
      const i4 pv = __builtin_nontemporal_load(reinterpret_cast<const i4*>(pidx + i));
      p[0] = pv.x, p[1] = pv.y, p[2] = pv.z, p[3] = pv.w;
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) p[k] = pidx[i + k];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) r[k] = table[p[k]];
    if constexpr (V == 4 && sizeof(T) == 4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 o = {(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
      __builtin_nontemporal_store(o, reinterpret_cast<f4*>(out + i));
    } else {
      nk_st<T, V>(out, i, r);
    }
  }
};

template <typename T>
struct FScatter {
  static constexpr int NRED = 0;
  const T* in;
  const int32_t* pidx;
  double* bins;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double*) const {
    T a[V];
    nk_ld<T, V>(in, i, a);
#pragma unroll
    for (int k = 0; k < V; ++k) atomicAdd(bins + pidx[i + k], (double)a[k]);
  }
};

// CG: scal[1] = d.q
template <typename T>
struct FCgCurv {
  static constexpr int NRED = 1;
  const T *d, *q;
  double* result;  // = scal + 1
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double* red) const {
    T a[V], b[V];
    nk_ld<T, V>(d, i, a);
    nk_ld<T, V>(q, i, b);
#pragma unroll
    for (int k = 0; k < V; ++k) red[0] += (double)a[k] * (double)b[k];
  }
};

// CG: x -= alpha d; r -= alpha q; reductions r.r, x.r, x.b -> scal[2..4]
template <typename T>
struct FCgUpdate {
  static constexpr int NRED = 3;
  T *x, *r;
  const T *d, *q, *b;
  const double* scal;
  double* result;  // = scal + 2
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double* red) const {
    const double alpha = scal[0] / scal[1];
    T xv[V], rv[V], dv[V], qv[V], bv[V];
    nk_ld<T, V>(x, i, xv);
    nk_ld<T, V>(r, i, rv);
    nk_ld<T, V>(d, i, dv);
    nk_ld<T, V>(q, i, qv);
    if (b) nk_ld<T, V>(b, i, bv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const T xn = (T)((double)xv[k] - alpha * (double)dv[k]);
      const T rn = (T)((double)rv[k] - alpha * (double)qv[k]);
      xv[k] = xn;
      rv[k] = rn;
      red[0] += (double)rn * (double)rn;
      red[1] += (double)xn * (double)rn;
      if (b) red[2] += (double)xn * (double)bv[k];
    }
    nk_st<T, V>(x, i, xv);
    nk_st<T, V>(r, i, rv);
  }
};

// the same update without the linear term b: instead of x.r and x.b (from which the quadratic energy was recomputed) it
// returns d.r of the OLD residual, and the caller advances the energy by  dE = -alpha d.r + alpha^2/2 d.q  -- one
// input stream less (6 instead of 7), and the energy DIFFERENCE the stopping rule looks at is formed directly
template <typename T>
struct FCgUpdateDr {
  static constexpr int NRED = 2;
  T *x, *r;
  const T *d, *q;
  const double* scal;
  double* result;  // = scal + 2: gamma, d.r_old
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double* red) const {
    const double alpha = scal[0] / scal[1];
    T xv[V], rv[V], dv[V], qv[V];
    nk_ld<T, V>(x, i, xv);
    nk_ld<T, V>(r, i, rv);
    nk_ld<T, V>(d, i, dv);
    nk_ld<T, V>(q, i, qv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      red[1] += (double)dv[k] * (double)rv[k];
      const T xn = (T)((double)xv[k] - alpha * (double)dv[k]);
      const T rn = (T)((double)rv[k] - alpha * (double)qv[k]);
      xv[k] = xn;
      rv[k] = rn;
      red[0] += (double)rn * (double)rn;
    }
    nk_st<T, V>(x, i, xv);
    nk_st<T, V>(r, i, rv);
  }
};

// CG: d = max(0, gamma/gamma_prev) d + r
template <typename T>
struct FCgDir {
  static constexpr int NRED = 0;
  T* d;
  const T* r;
  const double* scal;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double*) const {
    double beta = scal[2] / scal[0];
    beta = beta > 0.0 ? beta : 0.0;
    T dv[V], rv[V];
    nk_ld<T, V>(d, i, dv);
    nk_ld<T, V>(r, i, rv);
#pragma unroll
    for (int k = 0; k < V; ++k) dv[k] = (T)(beta * (double)dv[k] + (double)rv[k]);
    nk_st<T, V>(d, i, dv);
  }
};

// after the direction update: gamma_prev <- gamma; the slots of the vector update's reductions (r.r, x.r | d.r, x.b) are
// left at zero, so that the next nk_cg_update[_dr] may be called with accumulate != 0 for every segment (one memset launch
// less).  The curvature slot is NOT touched by roll == 1: with the fused direction update the roll runs after the metric
// application has deposited d.q there.  roll == 2 (a caller whose roll precedes the next application) clears it as well.
__device__ __forceinline__ void nk_roll_scalars(double* scal, int roll) {
  scal[5] = scal[0] / scal[1];
  const double beta = scal[2] / scal[0];
  scal[6] = beta > 0.0 ? beta : 0.0;
  scal[0] = scal[2];
  scal[2] = scal[3] = scal[4] = 0.0;
  if (roll == 2) scal[1] = 0.0;
}
__global__ void k_cg_roll(double* scal, int roll) { nk_roll_scalars(scal, roll); }

// ---- C ABI --------------------------------------------------------------------------------------
#define NK_DISPATCH_DTYPE(dtype, ...)                                      \
  if ((dtype) == NK_F32) {                                                  \
    typedef float T;                                                        \
    __VA_ARGS__;                                                            \
  } else if ((dtype) == NK_F64) {                                           \
    typedef double T;                                                       \
    __VA_ARGS__;                                                            \
  } else {                                                                  \
    return nk_set_error(NK_ERR_INVALID, "dtype must be NK_F32 or NK_F64");  \
  }

static int nk_zero(double* p, int count, hipStream_t st) {
  hipError_t e = hipMemsetAsync(p, 0, sizeof(double) * count, st);
  if (e != hipSuccess) return nk_set_hip_error(e, "hipMemsetAsync");
  return NK_OK;
}

// ---- shards of unit-reduced arrays (NkRedLayout) ------------------------------------------------------------------------
extern "C" int64_t nk_red_unit(int64_t n, int dtype) {
  return dtype == NK_F32 ? nk_red_unit_of(n, VecOf<float>::N) : dtype == NK_F64 ? nk_red_unit_of(n, VecOf<double>::N) : 0;
}

extern "C" int nk_red_layout(int64_t unit_elems, int k_local, int k_global, int seg_units, int seg_stride, int seg_off,
                             double* units_out) {
  if (unit_elems == 0) {
    t_red_layout = NkRedLayout{0, 0, 0, 0, 0, 0, nullptr};
    return NK_OK;
  }
  if (unit_elems < 0 || k_local < 1 || k_local > NK_RED_UNITS || k_global < k_local || k_global > NK_RED_UNITS || seg_units < 1 ||
      seg_stride < seg_units || seg_off < 0 || seg_off + seg_units > seg_stride || k_local % seg_units != 0 ||
      (k_local / seg_units) * seg_stride > k_global)
    return nk_set_error(NK_ERR_INVALID, "nk_red_layout: bad unit layout");
  t_red_layout = NkRedLayout{k_local, unit_elems, k_global, seg_units, seg_stride, seg_off, units_out};
  return NK_OK;
}

__global__ void k_red_finish(const double* units, int k, int nred, double* result, int accumulate) {
  const int r = threadIdx.x;
  if (r >= nred) return;
  double s = 0.0;
  for (int u = 0; u < k; ++u) s += units[r * k + u];
  result[r] = accumulate ? result[r] + s : s;
}

extern "C" int nk_red_finish(const double* units, int k_global, int nred, double* result, int accumulate, void* stream) {
  if (!units || !result || k_global < 1 || nred < 1 || nred > NK_RED_MAX)
    return nk_set_error(NK_ERR_INVALID, "nk_red_finish: bad argument");
  hipLaunchKernelGGL(k_red_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, units, k_global, nred, result, accumulate);
  return nk_check_launch("nk_red_finish");
}

extern "C" int nk_vdot(int64_t n, const void* a, const void* b, int dtype, double* result, int accumulate,
                       void* stream) {
  if (n < 0 || !result || (n > 0 && (!a || !b))) return nk_set_error(NK_ERR_INVALID, "nk_vdot: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    int rc = nk_zero(result, 1, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FDot<T> f{(const T*)a, (const T*)b, result};
    return nk_launch_map<T>(n, f, nk_aligned16(a) && nk_aligned16(b), st, "nk_vdot");
  })
}

extern "C" int nk_sum(int64_t n, const void* a, int dtype, double* result, int accumulate, void* stream) {
  if (n < 0 || !result || (n > 0 && !a)) return nk_set_error(NK_ERR_INVALID, "nk_sum: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    int rc = nk_zero(result, 1, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FSum<T> f{(const T*)a, result};
    return nk_launch_map<T>(n, f, nk_aligned16(a), st, "nk_sum");
  })
}

extern "C" int nk_stats(int64_t n, const void* a, int dtype, double* result3, void* stream) {
  if (n < 0 || !result3 || (n > 0 && !a)) return nk_set_error(NK_ERR_INVALID, "nk_stats: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int rc = nk_zero(result3, 3, st);
  if (rc != NK_OK) return rc;
  NK_DISPATCH_DTYPE(dtype, {
    FStats<T> f{(const T*)a, result3};
    return nk_launch_map<T>(n, f, nk_aligned16(a), st, "nk_stats");
  })
}

extern "C" int nk_binary(int op, int64_t n, const void* a, double ascalar, const void* b, double bscalar, void* out,
                         int dtype, void* stream) {
  if (n < 0 || !out || op < 0 || op > 3 || (!a && !b)) return nk_set_error(NK_ERR_INVALID, "nk_binary: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FBinary<T> f{op, (const T*)a, (const T*)b, ascalar, bscalar, (T*)out, nullptr};
    return nk_launch_map<T>(n, f, nk_aligned16(a) && nk_aligned16(b) && nk_aligned16(out), (hipStream_t)stream,
                            "nk_binary");
  })
}

// ---- complex element-wise algebra on interleaved (re, im) arrays --------------------------------------------------
// out = a (*|/) b with b optionally conjugated; an operand is a complex array of n elements, a REAL array of n elements
// (kind 1) or a complex scalar (kind 2).  DiagonalOperator with a complex diagonal and its adjoint / inverse modes
// (diagonal_operator.py:194-214), complex residual weights of GaussianEnergy (energy_operators.py:517-595).
template <typename T>
struct FCplxMulDiv {
  static constexpr int NRED = 0;
  const T *a, *b;
  int akind, bkind;   // 0 complex array, 1 real array, 2 scalar
  double asr, asi, bsr, bsi;
  int conj_b, divide;
  T* out;
  double* result;
  __device__ __forceinline__ void one(int64_t c) const {  // complex element c
    double ar, ai, br, bi;
    if (akind == 0) ar = (double)a[2 * c], ai = (double)a[2 * c + 1];
    else if (akind == 1) ar = (double)a[c], ai = 0.0;
    else ar = asr, ai = asi;
    if (bkind == 0) br = (double)b[2 * c], bi = (double)b[2 * c + 1];
    else if (bkind == 1) br = (double)b[c], bi = 0.0;
    else br = bsr, bi = bsi;
    if (conj_b) bi = -bi;
    double re, im;
    if (divide) {
      const double den = br * br + bi * bi;
      re = (ar * br + ai * bi) / den, im = (ai * br - ar * bi) / den;
    } else {
      re = ar * br - ai * bi, im = ar * bi + ai * br;
    }
    out[2 * c] = (T)re, out[2 * c + 1] = (T)im;
  }
  // the map runs over the 2n reals of `out`: a work item of V reals covers V/2 complex elements (V = 1: the even index)
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double*) const {
    if constexpr (V == 1) {
      if ((i & 1) == 0) one(i >> 1);
    } else {
#pragma unroll
      for (int k = 0; k < V / 2; ++k) one((i >> 1) + k);
    }
  }
};

// complex pointwise functions: fn 0 exp, 1 log, 2 sqrt (principal branch), 3 reciprocal, 4 conjugate, 5 |z| (real output array)
template <typename T>
struct FCplxPointwise {
  static constexpr int NRED = 0;
  int fn;
  const T* x;
  T* out;
  double* result;
  __device__ __forceinline__ void one(int64_t c) const {
    const double re = (double)x[2 * c], im = (double)x[2 * c + 1];
    double fr = re, fi = -im;
    switch (fn) {
      case 0: { const double e = exp(re); fr = e * cos(im), fi = e * sin(im); } break;
      case 1: fr = 0.5 * log(re * re + im * im), fi = atan2(im, re); break;
      case 2: {
        const double r = hypot(re, im);
        fr = sqrt(0.5 * (r + re));
        fi = copysign(sqrt(0.5 * (r - re)), im);
      } break;
      case 3: { const double den = re * re + im * im; fr = re / den, fi = -im / den; } break;
      case 5: out[c] = (T)hypot(re, im); return;
      default: break;
    }
    out[2 * c] = (T)fr, out[2 * c + 1] = (T)fi;
  }
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double*) const {
    if constexpr (V == 1) {
      if ((i & 1) == 0) one(i >> 1);
    } else {
#pragma unroll
      for (int k = 0; k < V / 2; ++k) one((i >> 1) + k);
    }
  }
};

extern "C" int nk_cplx_muldiv(int64_t n, const void* a, int akind, double asr, double asi, const void* b, int bkind, double bsr,
                              double bsi, int conj_b, int divide, void* out, int dtype, void* stream) {
  if (n < 0 || !out || akind < 0 || akind > 2 || bkind < 0 || bkind > 2 || (akind != 2 && !a) || (bkind != 2 && !b))
    return nk_set_error(NK_ERR_INVALID, "nk_cplx_muldiv: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FCplxMulDiv<T> f{(const T*)a, (const T*)b, akind, bkind, asr, asi, bsr, bsi, conj_b, divide, (T*)out, nullptr};
    // (real-array operands are read with scalar loads inside `one`; the vector width only paces the index space)
    return nk_launch_map<T>(2 * n, f, nk_aligned16(out), (hipStream_t)stream, "nk_cplx_muldiv");
  })
}

extern "C" int nk_cplx_pointwise(int fn, int64_t n, const void* x, void* out, int dtype, void* stream) {
  if (n < 0 || !x || !out || fn < 0 || fn > 5) return nk_set_error(NK_ERR_INVALID, "nk_cplx_pointwise: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FCplxPointwise<T> f{fn, (const T*)x, (T*)out, nullptr};
    return nk_launch_map<T>(2 * n, f, nk_aligned16(out), (hipStream_t)stream, "nk_cplx_pointwise");
  })
}

extern "C" int nk_axpby(int64_t n, double alpha, const void* x, double beta, const void* y, void* out, int dtype,
                        void* stream) {
  if (n < 0 || !out || !x) return nk_set_error(NK_ERR_INVALID, "nk_axpby: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FAxpby<T> f{alpha, beta, (const T*)x, (const T*)y, (T*)out, nullptr};
    return nk_launch_map<T>(n, f, nk_aligned16(x) && nk_aligned16(y) && nk_aligned16(out), (hipStream_t)stream,
                            "nk_axpby");
  })
}

extern "C" int nk_axpby_sqnorm(int64_t n, double alpha, const void* x, double beta, const void* y, void* out, int dtype,
                               double* result, int accumulate, void* stream) {
  if (n < 0 || !out || !x || !y || !result) return nk_set_error(NK_ERR_INVALID, "nk_axpby_sqnorm: bad argument");
  if (!accumulate) {
    int rc = nk_zero(result, 1, (hipStream_t)stream);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FAxpbySq<T> f{alpha, beta, (const T*)x, (const T*)y, (T*)out, result};
    return nk_launch_map<T>(n, f, nk_aligned16(x) && nk_aligned16(y) && nk_aligned16(out), (hipStream_t)stream,
                            "nk_axpby_sqnorm");
  })
}

extern "C" int nk_pointwise(int fn, double param, int64_t n, const void* x, void* fx, void* dfx, int dtype,
                            void* stream) {
  if (n < 0 || !x || (!fx && !dfx) || fn < 0 || fn > 21) return nk_set_error(NK_ERR_INVALID, "nk_pointwise: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FPointwise<T> f{fn, param, 0.0, (const T*)x, (T*)fx, (T*)dfx, nullptr};
    return nk_launch_map<T>(n, f, nk_aligned16(x) && nk_aligned16(fx) && nk_aligned16(dfx), (hipStream_t)stream,
                            "nk_pointwise");
  })
}

extern "C" int nk_clip(double lo, double hi, int64_t n, const void* x, void* fx, void* dfx, int dtype, void* stream) {
  if (n < 0 || !x || (!fx && !dfx) || !(lo <= hi)) return nk_set_error(NK_ERR_INVALID, "nk_clip: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FPointwise<T> f{22, lo, hi, (const T*)x, (T*)fx, (T*)dfx, nullptr};
    return nk_launch_map<T>(n, f, nk_aligned16(x) && nk_aligned16(fx) && nk_aligned16(dfx), (hipStream_t)stream, "nk_clip");
  })
}

extern "C" int nk_gather(int64_t n, const void* table, const int32_t* pidx, void* out, int dtype, void* stream) {
  if (n < 0 || !table || !pidx || !out) return nk_set_error(NK_ERR_INVALID, "nk_gather: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FGather<T> f{(const T*)table, pidx, (T*)out, nullptr};
    return nk_launch_map<T>(n, f, nk_aligned16(out) && nk_aligned16(pidx), (hipStream_t)stream, "nk_gather");
  })
}

extern "C" int nk_scatter_add(int64_t n, const void* in, const int32_t* pidx, int64_t nbins, void* bins, int dtype,
                              void* stream) {
  if (n < 0 || !in || !pidx || !bins || nbins < 1) return nk_set_error(NK_ERR_INVALID, "nk_scatter_add: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FScatter<T> f{(const T*)in, pidx, (double*)bins, nullptr};
    return nk_launch_map<T>(n, f, nk_aligned16(in), (hipStream_t)stream, "nk_scatter_add");
  })
}

extern "C" int nk_cg_curv(int64_t n, const void* d, const void* q, int dtype, double* scal, int accumulate,
                          void* stream) {
  if (n < 0 || !d || !q || !scal) return nk_set_error(NK_ERR_INVALID, "nk_cg_curv: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    int rc = nk_zero(scal + 1, 1, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FCgCurv<T> f{(const T*)d, (const T*)q, scal + 1};
    return nk_launch_map<T>(n, f, nk_aligned16(d) && nk_aligned16(q), st, "nk_cg_curv");
  })
}

extern "C" int nk_cg_update(int64_t n, void* x, void* r, const void* d, const void* q, const void* b, int dtype,
                            double* scal, int accumulate, void* stream) {
  if (n < 0 || !x || !r || !d || !q || !scal) return nk_set_error(NK_ERR_INVALID, "nk_cg_update: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    int rc = nk_zero(scal + 2, 3, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FCgUpdate<T> f{(T*)x, (T*)r, (const T*)d, (const T*)q, (const T*)b, scal, scal + 2};
    return nk_launch_map<T>(
        n, f, nk_aligned16(x) && nk_aligned16(r) && nk_aligned16(d) && nk_aligned16(q) && nk_aligned16(b), st,
        "nk_cg_update");
  })
}

extern "C" int nk_cg_update_dr(int64_t n, void* x, void* r, const void* d, const void* q, int dtype, double* scal,
                               int accumulate, void* stream) {
  if (n < 0 || !x || !r || !d || !q || !scal) return nk_set_error(NK_ERR_INVALID, "nk_cg_update_dr: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    int rc = nk_zero(scal + 2, 2, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FCgUpdateDr<T> f{(T*)x, (T*)r, (const T*)d, (const T*)q, scal, scal + 2};
    return nk_launch_map<T>(n, f, nk_aligned16(x) && nk_aligned16(r) && nk_aligned16(d) && nk_aligned16(q), st,
                            "nk_cg_update_dr");
  })
}

extern "C" int nk_cg_direction(int64_t n, void* d, const void* r, int dtype, double* scal, int roll, void* stream) {
  if (n < 0 || (n > 0 && (!d || !r)) || !scal) return nk_set_error(NK_ERR_INVALID, "nk_cg_direction: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int rc;
  NK_DISPATCH_DTYPE(dtype, {
    FCgDir<T> f{(T*)d, (const T*)r, scal, nullptr};
    rc = nk_launch_map<T>(n, f, nk_aligned16(d) && nk_aligned16(r), st, "nk_cg_direction");
  })
  if (rc != NK_OK || !roll) return rc;
  hipLaunchKernelGGL(k_cg_roll, dim3(1), dim3(1), 0, st, scal, roll);
  return nk_check_launch("k_cg_roll");
}

// ---- batched launches (include/niftyk.h): the single entry points above for `count` members --------------------------
struct NkPtrs {
  void* p[NK_MAX_BATCH];
};
static inline bool nk_batch_count_ok(int count) { return count >= 1 && count <= NK_MAX_BATCH; }
template <typename P>
static inline bool nk_all_set(P const* ptrs, int count) {
  if (!ptrs) return false;
  for (int m = 0; m < count; ++m)
    if (!ptrs[m]) return false;
  return true;
}
template <typename P>
static inline bool nk_all_aligned16(P const* ptrs, int count) {
  if (!ptrs) return true;
  for (int m = 0; m < count; ++m)
    if (!nk_aligned16(ptrs[m])) return false;
  return true;
}
// cnt doubles at off of every member's scalar block <- 0 (one launch instead of `count` memsets)
__global__ void k_zero_ptrs(NkPtrs ptrs, int off, int cnt) {
  double* p = (double*)ptrs.p[blockIdx.x];
  if ((int)threadIdx.x < cnt) p[off + threadIdx.x] = 0.0;
}
static int nk_zero_b(double* const* ptrs, int count, int off, int cnt, hipStream_t st) {
  NkPtrs pp;
  for (int m = 0; m < NK_MAX_BATCH; ++m) pp.p[m] = ptrs[m < count ? m : 0];
  hipLaunchKernelGGL(k_zero_ptrs, dim3(count), dim3(64), 0, st, pp, off, cnt);
  return nk_check_launch("k_zero_ptrs");
}

extern "C" int nk_axpby_batch(int64_t n, int count, const double* alpha, const void* const* x, const double* beta,
                              const void* const* y, void* const* out, int dtype, void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || !alpha || !beta || !nk_all_set(x, count) || !nk_all_set(out, count) || !y)
    return nk_set_error(NK_ERR_INVALID, "nk_axpby_batch: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FAxpby<T> fs[NK_MAX_BATCH];
    for (int m = 0; m < count; ++m) fs[m] = FAxpby<T>{alpha[m], beta[m], (const T*)x[m], (const T*)y[m], (T*)out[m], nullptr};
    return nk_launch_map_b<T>(n, fs, count, nk_all_aligned16(x, count) && nk_all_aligned16(y, count) && nk_all_aligned16(out, count),
                              (hipStream_t)stream, "nk_axpby_batch");
  })
}

extern "C" int nk_axpby_sqnorm_batch(int64_t n, int count, const double* alpha, const void* const* x, const double* beta,
                                     const void* const* y, void* const* out, int dtype, double* const* result, int accumulate,
                                     void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || !alpha || !beta || !nk_all_set(x, count) || !nk_all_set(y, count) ||
      !nk_all_set(out, count) || !nk_all_set(result, count))
    return nk_set_error(NK_ERR_INVALID, "nk_axpby_sqnorm_batch: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    const int rc = nk_zero_b(result, count, 0, 1, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FAxpbySq<T> fs[NK_MAX_BATCH];
    for (int m = 0; m < count; ++m) fs[m] = FAxpbySq<T>{alpha[m], beta[m], (const T*)x[m], (const T*)y[m], (T*)out[m], result[m]};
    return nk_launch_map_b<T>(n, fs, count, nk_all_aligned16(x, count) && nk_all_aligned16(y, count) && nk_all_aligned16(out, count), st,
                              "nk_axpby_sqnorm_batch");
  })
}

extern "C" int nk_binary_batch(int op, int64_t n, int count, const void* const* a, const double* ascalar, const void* const* b,
                               const double* bscalar, void* const* out, int dtype, void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || op < 0 || op > 3 || !a || !b || !ascalar || !bscalar || !nk_all_set(out, count))
    return nk_set_error(NK_ERR_INVALID, "nk_binary_batch: bad argument");
  for (int m = 0; m < count; ++m)
    if (!a[m] && !b[m]) return nk_set_error(NK_ERR_INVALID, "nk_binary_batch: a member without an array operand");
  NK_DISPATCH_DTYPE(dtype, {
    FBinary<T> fs[NK_MAX_BATCH];
    for (int m = 0; m < count; ++m) fs[m] = FBinary<T>{op, (const T*)a[m], (const T*)b[m], ascalar[m], bscalar[m], (T*)out[m], nullptr};
    return nk_launch_map_b<T>(n, fs, count, nk_all_aligned16(a, count) && nk_all_aligned16(b, count) && nk_all_aligned16(out, count),
                              (hipStream_t)stream, "nk_binary_batch");
  })
}

extern "C" int nk_vdot_batch(int64_t n, int count, const void* const* a, const void* const* b, int dtype, double* const* result,
                             int accumulate, void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || !nk_all_set(a, count) || !nk_all_set(b, count) || !nk_all_set(result, count))
    return nk_set_error(NK_ERR_INVALID, "nk_vdot_batch: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    const int rc = nk_zero_b(result, count, 0, 1, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FDot<T> fs[NK_MAX_BATCH];
    for (int m = 0; m < count; ++m) fs[m] = FDot<T>{(const T*)a[m], (const T*)b[m], result[m]};
    return nk_launch_map_b<T>(n, fs, count, nk_all_aligned16(a, count) && nk_all_aligned16(b, count), st, "nk_vdot_batch");
  })
}

extern "C" int nk_gather_batch(int64_t n, int count, const void* const* table, const int32_t* pidx, void* const* out, int dtype,
                               void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || !pidx || !nk_all_set(table, count) || !nk_all_set(out, count))
    return nk_set_error(NK_ERR_INVALID, "nk_gather_batch: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FGather<T> fs[NK_MAX_BATCH];
    for (int m = 0; m < count; ++m) fs[m] = FGather<T>{(const T*)table[m], pidx, (T*)out[m], nullptr};
    return nk_launch_map_b<T>(n, fs, count, nk_all_aligned16(out, count) && nk_aligned16(pidx), (hipStream_t)stream, "nk_gather_batch");
  })
}

extern "C" int nk_cg_curv_batch(int64_t n, int count, const void* const* d, const void* const* q, int dtype, double* const* scal,
                                int accumulate, void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || !nk_all_set(d, count) || !nk_all_set(q, count) || !nk_all_set(scal, count))
    return nk_set_error(NK_ERR_INVALID, "nk_cg_curv_batch: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    const int rc = nk_zero_b(scal, count, 1, 1, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FCgCurv<T> fs[NK_MAX_BATCH];
    for (int m = 0; m < count; ++m) fs[m] = FCgCurv<T>{(const T*)d[m], (const T*)q[m], scal[m] + 1};
    return nk_launch_map_b<T>(n, fs, count, nk_all_aligned16(d, count) && nk_all_aligned16(q, count), st, "nk_cg_curv_batch");
  })
}

extern "C" int nk_cg_update_batch(int64_t n, int count, void* const* x, void* const* r, const void* const* d, const void* const* q,
                                  const void* const* b, int dtype, double* const* scal, int accumulate, void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || !nk_all_set(x, count) || !nk_all_set(r, count) || !nk_all_set(d, count) ||
      !nk_all_set(q, count) || !b || !nk_all_set(scal, count))
    return nk_set_error(NK_ERR_INVALID, "nk_cg_update_batch: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    const int rc = nk_zero_b(scal, count, 2, 3, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FCgUpdate<T> fs[NK_MAX_BATCH];
    for (int m = 0; m < count; ++m)
      fs[m] = FCgUpdate<T>{(T*)x[m], (T*)r[m], (const T*)d[m], (const T*)q[m], (const T*)b[m], scal[m], scal[m] + 2};
    return nk_launch_map_b<T>(n, fs, count,
                              nk_all_aligned16(x, count) && nk_all_aligned16(r, count) && nk_all_aligned16(d, count) &&
                                  nk_all_aligned16(q, count) && nk_all_aligned16(b, count),
                              st, "nk_cg_update_batch");
  })
}

extern "C" int nk_cg_update_dr_batch(int64_t n, int count, void* const* x, void* const* r, const void* const* d,
                                     const void* const* q, int dtype, double* const* scal, int accumulate, void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || !nk_all_set(x, count) || !nk_all_set(r, count) || !nk_all_set(d, count) ||
      !nk_all_set(q, count) || !nk_all_set(scal, count))
    return nk_set_error(NK_ERR_INVALID, "nk_cg_update_dr_batch: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (!accumulate) {
    const int rc = nk_zero_b(scal, count, 2, 2, st);
    if (rc != NK_OK) return rc;
  }
  NK_DISPATCH_DTYPE(dtype, {
    FCgUpdateDr<T> fs[NK_MAX_BATCH];
    for (int m = 0; m < count; ++m) fs[m] = FCgUpdateDr<T>{(T*)x[m], (T*)r[m], (const T*)d[m], (const T*)q[m], scal[m], scal[m] + 2};
    return nk_launch_map_b<T>(n, fs, count,
                              nk_all_aligned16(x, count) && nk_all_aligned16(r, count) && nk_all_aligned16(d, count) &&
                                  nk_all_aligned16(q, count),
                              st, "nk_cg_update_dr_batch");
  })
}

__global__ void k_cg_roll_b(NkPtrs scals, int roll) { nk_roll_scalars((double*)scals.p[blockIdx.x], roll); }

extern "C" int nk_cg_direction_batch(int64_t n, int count, void* const* d, const void* const* r, int dtype, double* const* scal,
                                     int roll, void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || (n > 0 && (!nk_all_set(d, count) || !nk_all_set(r, count))) || !nk_all_set(scal, count))
    return nk_set_error(NK_ERR_INVALID, "nk_cg_direction_batch: bad argument");
  hipStream_t st = (hipStream_t)stream;
  int rc = NK_OK;
  if (n > 0) {
    NK_DISPATCH_DTYPE(dtype, {
      FCgDir<T> fs[NK_MAX_BATCH];
      for (int m = 0; m < count; ++m) fs[m] = FCgDir<T>{(T*)d[m], (const T*)r[m], scal[m], nullptr};
      rc = nk_launch_map_b<T>(n, fs, count, nk_all_aligned16(d, count) && nk_all_aligned16(r, count), st, "nk_cg_direction_batch");
    })
  }
  if (rc != NK_OK || !roll) return rc;
  NkPtrs pp;
  for (int m = 0; m < NK_MAX_BATCH; ++m) pp.p[m] = scal[m < count ? m : 0];
  hipLaunchKernelGGL(k_cg_roll_b, dim3(count), dim3(1), 0, st, pp, roll);
  return nk_check_launch("k_cg_roll_b");
}

// the sum of `count` vectors in the order of parallel.pair_tree (reference utilities.py:349-414): neighbours at distance 1
// first, then 2, then 4; every partial sum rounded to T exactly like the stored result of nk_axpby(1, a, 1, b)
template <typename T>
struct FSumTree {
  static constexpr int NRED = 0;
  const T* term[NK_MAX_BATCH];
  int count;
  T* out;
  double* result;
  template <int V>
  __device__ __forceinline__ void run(int64_t i, double*) const {
    T v[NK_MAX_BATCH][V];
#pragma unroll
    for (int m = 0; m < NK_MAX_BATCH; ++m)
      if (m < count) nk_ld<T, V>(term[m], i, v[m]);
#pragma unroll
    for (int gap = 1; gap < NK_MAX_BATCH; gap *= 2) {
#pragma unroll
      for (int left = 0; left + gap < NK_MAX_BATCH; left += 2 * gap) {
        if (left + gap < count) {
#pragma unroll
          for (int k = 0; k < V; ++k) v[left][k] = (T)((double)v[left][k] + (double)v[left + gap][k]);
        }
      }
    }
    nk_st<T, V>(out, i, v[0]);
  }
};

extern "C" int nk_sum_tree(int64_t n, int count, const void* const* term, void* out, int dtype, void* stream) {
  if (n < 0 || !nk_batch_count_ok(count) || !nk_all_set(term, count) || !out)
    return nk_set_error(NK_ERR_INVALID, "nk_sum_tree: bad argument");
  NK_DISPATCH_DTYPE(dtype, {
    FSumTree<T> f;
    for (int m = 0; m < NK_MAX_BATCH; ++m) f.term[m] = (const T*)term[m < count ? m : 0];
    f.count = count;
    f.out = (T*)out;
    f.result = nullptr;
    return nk_launch_map<T>(n, f, nk_all_aligned16(term, count) && nk_aligned16(out), (hipStream_t)stream, "nk_sum_tree");
  })
}

// ---- power-bin index straight from integer k^2 (PowerSpace pindex for equal harmonic distances,
//      nifty/cl/domains/power_space.py:172-180 + rg_space.py:116-128 without the 8 N-byte int64 array) ----
struct NkShape3 {
  int64_t n0, n1, n2;
};
__global__ void k_pindex_k2(NkShape3 s, const int32_t* __restrict__ table, int32_t* __restrict__ pidx,
                            unsigned long long* __restrict__ rho, int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t i2 = i % s.n2, r = i / s.n2;
    const int64_t i1 = r % s.n1, i0 = r / s.n1;
    const int64_t a = i0 < s.n0 - i0 ? i0 : s.n0 - i0;
    const int64_t b = i1 < s.n1 - i1 ? i1 : s.n1 - i1;
    const int64_t c = i2 < s.n2 - i2 ? i2 : s.n2 - i2;
    const int32_t p = table[a * a + b * b + c * c];
    pidx[i] = p;
    if (rho) atomicAdd(rho + p, 1ULL);
  }
}

extern "C" int nk_pindex_from_k2(int ndim, const int64_t* shape, const int32_t* k2table, int32_t* pidx, int64_t* rho,
                                 void* stream) {
  if (ndim < 1 || ndim > 3 || !shape || !k2table || !pidx) return nk_set_error(NK_ERR_INVALID, "nk_pindex_from_k2: bad argument");
  NkShape3 s{1, 1, 1};
  if (ndim == 1) s.n2 = shape[0];
  if (ndim == 2) s.n1 = shape[0], s.n2 = shape[1];
  if (ndim == 3) s.n0 = shape[0], s.n1 = shape[1], s.n2 = shape[2];
  const int64_t total = s.n0 * s.n1 * s.n2;
  hipLaunchKernelGGL(k_pindex_k2, dim3(nk_grid(total)), dim3(NK_VEC_THREADS), 0, (hipStream_t)stream, s, k2table, pidx,
                     (unsigned long long*)rho, total);
  return nk_check_launch("k_pindex_k2");
}

// ---- row-wise complex helper of the chirp-z composition (see niftyk.h) ------------------------------------------
template <typename T>
__global__ void __launch_bounds__(NK_VEC_THREADS) k_cplx_rows(int64_t total, int64_t in_cols, int64_t out_cols, const T* __restrict__ a,
                                                              const T* __restrict__ w, T* __restrict__ out, int mode, T scale,
                                                              T sgn) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / out_cols, c = i - r * out_cols;
    T re = (T)0, im = (T)0;
    if (c < in_cols) {
      const int64_t j = r * in_cols + c;
      if (mode == 1) {
        re = a[j];
      } else {
        re = a[2 * j];
        im = a[2 * j + 1];
      }
    }
    if (mode == 2) {
      out[i] = scale * (re + sgn * im);
      continue;
    }
    if (w && c < in_cols) {
      const T wr = w[2 * c], wi = w[2 * c + 1];
      const T t = re * wr - im * wi;
      im = re * wi + im * wr;
      re = t;
    }
    out[2 * i] = scale * re;
    out[2 * i + 1] = scale * im;
  }
}

extern "C" int nk_cplx_rows(int64_t rows, int64_t in_cols, int64_t out_cols, const void* a, const void* w, void* out, int mode,
                            double scale, int sgn, int dtype, void* stream) {
  if (rows < 0 || in_cols < 0 || out_cols < 0 || mode < 0 || mode > 2 || !out || (!a && rows * in_cols > 0))
    return nk_set_error(NK_ERR_INVALID, "nk_cplx_rows: bad argument");
  if (mode == 2 && out_cols > in_cols) return nk_set_error(NK_ERR_INVALID, "nk_cplx_rows: mode 2 cannot pad");
  const int64_t total = rows * out_cols;
  if (total == 0) return NK_OK;
  NK_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL(k_cplx_rows<T>, dim3(nk_grid(total)), dim3(NK_VEC_THREADS), 0, (hipStream_t)stream, total, in_cols,
                       out_cols, (const T*)a, (const T*)w, (T*)out, mode, (T)scale, (T)(sgn < 0 ? -1 : 1));
    return nk_check_launch("k_cplx_rows");
  })
}

// ---- any-length c2c over the last axis in ONE launch (chirp-z / Bluestein; nk_bluestein_rows, niftyk.h) --------------------
// X[k] = w[k] sum_j (x[j] w[j]) conj(w)[k-j], w[j] = exp(-+ i pi j^2 / n): a cyclic convolution of power-of-two length
// m >= 2n - 1.  The composition in backend.py spends three c2c transforms and three element-wise launches of m-sized rows on
// it; here a workgroup keeps its padded rows in LDS for the whole trip: chirp on the way in, decimation in frequency in place
// (natural -> bit-reversed order), times the filter's spectrum stored in bit-reversed order, decimation in time in place
// (bit-reversed -> natural order, conjugate twiddles), chirp and 1/m on the way out -- no reordering pass, global traffic =
// the rows in, the rows out.  Two radix-2 levels per barrier (radix-4 butterflies in registers; one radix-2 level when
// log2 m is odd); short rows share a workgroup (R rows of m points, R m <= 4096).  The ends of an N-D transform ride along:
// real input rows (in_real) for the first axis, the Hartley combination Re + sgn Im (out_hartley) for the last.
template <typename T>
__device__ __forceinline__ C2<T> nk_cconj(C2<T> a) { return C2<T>{a.x, -a.y}; }
template <typename T, bool INV>
__device__ __forceinline__ C2<T> nk_blu_tw(const C2<T>* __restrict__ tw, int k) {
  const C2<T> t = tw[k];
  return INV ? nk_cconj(t) : t;
}
// rotate by -i (forward) / +i (inverse)
template <typename T, bool INV>
__device__ __forceinline__ C2<T> nk_blu_rot(C2<T> a) { return INV ? C2<T>{-a.y, a.x} : C2<T>{a.y, -a.x}; }

template <typename T>
__global__ void __launch_bounds__(256) k_bluestein_rows(int64_t rows, int n, int m, int R, const void* __restrict__ in_raw,
                                                        const C2<T>* __restrict__ w, const C2<T>* __restrict__ bhat_br,
                                                        const C2<T>* __restrict__ tw, void* __restrict__ out_raw, T scale, int in_real,
                                                        int out_hartley) {
  extern __shared__ __align__(16) unsigned char nk_blu_lds[];
  C2<T>* lds = reinterpret_cast<C2<T>*>(nk_blu_lds);
  const int tid = threadIdx.x;
  const int half_m = m >> 1, quarter_m = m >> 2;
  int lg = 0;
  while ((1 << lg) < m) ++lg;
  const int64_t groups = (rows + R - 1) / R;
  for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const int64_t row0 = g * R;
    const int nr = rows - row0 < R ? (int)(rows - row0) : R;
    for (int i = tid; i < nr * m; i += 256) {
      const int r = i / m, j = i - r * m;
      C2<T> v{(T)0, (T)0};
      if (j < n) {
        const int64_t src = (row0 + r) * n + j;
        v = in_real ? C2<T>{reinterpret_cast<const T*>(in_raw)[src], (T)0} : reinterpret_cast<const C2<T>*>(in_raw)[src];
        v = cmul(v, w[j]);
      }
      lds[i] = v;
    }
    __syncthreads();
    // ---- forward, decimation in frequency: spans m/2, m/4, ..., 1 -- two levels per trip
    int span = half_m;
    for (; span >= 2; span >>= 2) {
      const int q = span >> 1;  // the second level's span
      const int s1 = half_m / span, s2 = half_m / q;
      for (int i = tid; i < nr * quarter_m; i += 256) {
        const int r = i / quarter_m, b = i - r * quarter_m;
        const int j = b & (q - 1), base = ((b - j) << 2) + j;
        C2<T>* a = lds + r * m + base;
        const C2<T> x0 = a[0], x1 = a[q], x2 = a[span], x3 = a[span + q];
        const C2<T> t1 = nk_blu_tw<T, false>(tw, j * s1), t2 = nk_blu_tw<T, false>(tw, j * s2);
        const C2<T> y0 = cadd(x0, x2), y1 = cadd(x1, x3);
        const C2<T> y2 = cmul(csub(x0, x2), t1), y3 = cmul(nk_blu_rot<T, false>(csub(x1, x3)), t1);
        a[0] = cadd(y0, y1);
        a[q] = cmul(csub(y0, y1), t2);
        a[span] = cadd(y2, y3);
        a[span + q] = cmul(csub(y2, y3), t2);
      }
      __syncthreads();
    }
    if (span == 1) {  // log2 m odd: one radix-2 level is left (twiddle 1)
      for (int i = tid; i < nr * half_m; i += 256) {
        const int r = i / half_m, b = i - r * half_m;
        C2<T>* a = lds + r * m + (b << 1);
        const C2<T> u = a[0], v = a[1];
        a[0] = cadd(u, v);
        a[1] = csub(u, v);
      }
      __syncthreads();
    }
    for (int i = tid; i < nr * m; i += 256) lds[i] = cmul(lds[i], bhat_br[i % m]);
    __syncthreads();
    // ---- inverse, decimation in time from bit-reversed order: spans 1, 2, ..., m/2; conjugate twiddles
    span = 1;
    if (lg & 1) {
      for (int i = tid; i < nr * half_m; i += 256) {
        const int r = i / half_m, b = i - r * half_m;
        C2<T>* a = lds + r * m + (b << 1);
        const C2<T> u = a[0], v = a[1];
        a[0] = cadd(u, v);
        a[1] = csub(u, v);
      }
      __syncthreads();
      span = 2;
    }
    for (; span <= quarter_m; span <<= 2) {
      const int q = span, big = span << 1;  // levels with spans q and 2q
      const int s1 = half_m / q, s2 = half_m / big;
      for (int i = tid; i < nr * quarter_m; i += 256) {
        const int r = i / quarter_m, b = i - r * quarter_m;
        const int j = b & (q - 1), base = ((b - j) << 2) + j;
        C2<T>* a = lds + r * m + base;
        const C2<T> t1 = nk_blu_tw<T, true>(tw, j * s1), t2 = nk_blu_tw<T, true>(tw, j * s2);
        // level 1 (span q): pairs (0, q) and (2q, 3q), both with twiddle t1
        const C2<T> x0 = a[0], x1 = cmul(a[q], t1), x2 = a[big], x3 = cmul(a[big + q], t1);
        const C2<T> y0 = cadd(x0, x1), y1 = csub(x0, x1), y2 = cadd(x2, x3), y3 = csub(x2, x3);
        // level 2 (span 2q): pairs (0, 2q) with twiddle t2 and (q, 3q) with twiddle t2 * (+i)
        const C2<T> z2 = cmul(y2, t2), z3 = nk_blu_rot<T, true>(cmul(y3, t2));
        a[0] = cadd(y0, z2);
        a[big] = csub(y0, z2);
        a[q] = cadd(y1, z3);
        a[big + q] = csub(y1, z3);
      }
      __syncthreads();
    }
    for (int i = tid; i < nr * n; i += 256) {
      const int r = i / n, k = i - r * n;
      const C2<T> v = cmul(lds[r * m + k], w[k]);
      const int64_t dst = (row0 + r) * n + k;
      if (out_hartley) reinterpret_cast<T*>(out_raw)[dst] = scale * (v.x + (T)out_hartley * v.y);
      else reinterpret_cast<C2<T>*>(out_raw)[dst] = C2<T>{scale * v.x, scale * v.y};
    }
    __syncthreads();
  }
}

extern "C" int nk_bluestein_rows(int64_t rows, int n, int m, const void* in, const void* w, const void* bhat_br, const void* tw, void* out,
                                 double scale, int in_real, int out_hartley, int dtype, void* stream) {
  if (rows < 0 || n < 1 || m < 4 || (m & (m - 1)) != 0 || m < 2 * n - 1 || !w || !bhat_br || !tw || (rows > 0 && (!in || !out)) ||
      out_hartley < -1 || out_hartley > 1)
    return nk_set_error(NK_ERR_INVALID, "nk_bluestein_rows: bad argument (m: a power of two >= max(4, 2 n - 1))");
  if (rows == 0) return NK_OK;
  if ((in_real || out_hartley) && in == out) return nk_set_error(NK_ERR_INVALID, "nk_bluestein_rows: real ends need distinct arrays");
  NK_DISPATCH_DTYPE(dtype, {
    if (sizeof(C2<T>) * (size_t)m > 64 * 1024)
      return nk_set_error(NK_ERR_UNSUPPORTED, "nk_bluestein_rows: the padded row does not fit 64 KiB of LDS");
    int R = (int)std::max<int64_t>(1, std::min<int64_t>(4096 / m, rows));  // rows per workgroup: R m <= 4096 points
    const size_t lds = sizeof(C2<T>) * (size_t)m * R;
    const int64_t groups = (rows + R - 1) / R;
    const unsigned blocks = (unsigned)std::min<int64_t>(groups, 256 * 8);
    hipLaunchKernelGGL(k_bluestein_rows<T>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, rows, n, m, R, in, (const C2<T>*)w,
                       (const C2<T>*)bhat_br, (const C2<T>*)tw, out, (T)(scale / m), in_real, out_hartley);
    return nk_check_launch("k_bluestein_rows");
  })
}

// ---- fold the per-XCD private VJP accumulators ------------------------------------------------------------------
__global__ void k_fold_copies(int64_t n, int copies, int64_t stride, const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t gs = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gs) {
    double s = 0.0;
    for (int c = 0; c < copies; ++c) s += src[c * stride + i];
    dst[i] = s;
  }
}

extern "C" int nk_fold_copies(int64_t n, int copies, int64_t stride, const double* src, double* dst, void* stream) {
  if (n < 0 || copies < 1 || !src || !dst) return nk_set_error(NK_ERR_INVALID, "nk_fold_copies: bad argument");
  if (n == 0) return NK_OK;
  hipLaunchKernelGGL(k_fold_copies, dim3(nk_grid(n)), dim3(NK_VEC_THREADS), 0, (hipStream_t)stream, n, copies, stride, src, dst);
  return nk_check_launch("k_fold_copies");
}

// ---- octant expand / scatter ---------------------------------------------------------------------------------------
struct NkOct {
  int A, M, NL;     // grid axes, right aligned (1 for missing axes)
  int Ah, Mh, Ch;   // octant extents n/2 + 1
};
static int nk_make_oct(int ndim, const int64_t* shape, NkOct& o) {
  if (ndim < 1 || ndim > 3 || !shape) return nk_set_error(NK_ERR_INVALID, "octant: bad shape");
  o.A = ndim == 3 ? (int)shape[0] : 1;
  o.M = ndim >= 2 ? (int)shape[ndim - 2] : 1;
  o.NL = (int)shape[ndim - 1];
  o.Ah = o.A / 2 + 1, o.Mh = o.M / 2 + 1, o.Ch = o.NL / 2 + 1;
  return NK_OK;
}

// one workgroup per octant line (a, b); lanes run over c = k_last
template <typename T>
__global__ void k_octant_expand(NkOct o, int compact, const T* __restrict__ table, const int32_t* __restrict__ pidx,
                                T* __restrict__ field) {
  const int b = blockIdx.x % o.Mh, a = blockIdx.x / o.Mh;
  if (compact) {  // octant array only: field8[a][b][c]
    const int64_t src = ((int64_t)a * o.M + b) * o.NL, dst = ((int64_t)a * o.Mh + b) * o.Ch;
    for (int c = threadIdx.x; c < o.Ch; c += blockDim.x) field[dst + c] = table[pidx[src + c]];
    return;
  }
  const int am = a ? o.A - a : 0, bm = b ? o.M - b : 0;
  const int64_t r00 = ((int64_t)a * o.M + b) * o.NL, r01 = ((int64_t)a * o.M + bm) * o.NL;
  const int64_t r10 = ((int64_t)am * o.M + b) * o.NL, r11 = ((int64_t)am * o.M + bm) * o.NL;
  for (int c = threadIdx.x; c < o.Ch; c += blockDim.x) {
    const int cm = c ? o.NL - c : 0;
    const T v = table[pidx[r00 + c]];
    field[r00 + c] = v;
    field[r00 + cm] = v;
    field[r01 + c] = v;
    field[r01 + cm] = v;
    field[r10 + c] = v;
    field[r10 + cm] = v;
    field[r11 + c] = v;
    field[r11 + cm] = v;
  }
}

__global__ void k_octant_scatter(NkOct o, int swap_merge, const double* __restrict__ w8, const int32_t* __restrict__ pidx,
                                 double* __restrict__ abar) {
  const int b = blockIdx.x % o.Mh, a = blockIdx.x / o.Mh;
  if (swap_merge && a > b) return;  // (b, a) is folded into (a, b)
  const double* l0 = w8 + ((int64_t)a * o.Mh + b) * o.Ch;
  const double* l1 = w8 + ((int64_t)b * o.Mh + a) * o.Ch;
  const bool both = swap_merge && a < b;
  const int64_t r00 = ((int64_t)a * o.M + b) * o.NL;
  for (int c = threadIdx.x; c < o.Ch; c += blockDim.x) {
    double s = l0[c];
    if (both) s += l1[c];
    atomicAdd(abar + pidx[r00 + c], s);
  }
}

// ---- segmented gather-sum: dst[s] (+)= sum of src[perm[i]] over rowptr[s] <= i < rowptr[s+1] ---------------
// The scatter-add N -> nb of the power distributor's adjoint (distributors.py:106-127, utilities.py:222-246) turned
// inside out for a STATIC index map: perm lists the source points bin by bin (a stable sort of the bin index, made once),
// so every bin is summed by one thread in a fixed order -- no atomics, bit-reproducible.  Used for the quadrant sums of
// 2-D grids (a few points per bin: 2048^2: 47 us of global fp64 atomics -> see DESIGN 3.2); on 3-D grids a bin holds
// hundreds of points scattered over the octant and the shell-binned scatter above wins.
__global__ void __launch_bounds__(256) k_segment_sum(int64_t nseg, const int32_t* __restrict__ rowptr,
                                                     const int32_t* __restrict__ perm, const double* __restrict__ src,
                                                     double* __restrict__ dst, int accumulate) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const int lo = rowptr[s], hi = rowptr[s + 1];
  double v = 0.0;
  for (int i = lo; i < hi; ++i) v += src[perm[i]];
  dst[s] = accumulate ? dst[s] + v : v;
}

extern "C" int nk_segment_sum(int64_t nseg, const int32_t* rowptr, const int32_t* perm, const double* src, double* dst,
                              int accumulate, void* stream) {
  if (nseg < 0 || (nseg > 0 && (!rowptr || !perm || !src || !dst)))
    return nk_set_error(NK_ERR_INVALID, "nk_segment_sum: bad argument");
  if (nseg == 0) return NK_OK;
  hipLaunchKernelGGL(k_segment_sum, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, (hipStream_t)stream, nseg, rowptr,
                     perm, src, dst, accumulate);
  return nk_check_launch("k_segment_sum");
}

extern "C" int nk_octant_expand(int ndim, const int64_t* shape, const void* table, const int32_t* pidx, void* field,
                                int dtype, int compact, void* stream) {
  NkOct o;
  int rc = nk_make_oct(ndim, shape, o);
  if (rc != NK_OK) return rc;
  if (!table || !pidx || !field) return nk_set_error(NK_ERR_INVALID, "nk_octant_expand: null argument");
  const int64_t blocks = (int64_t)o.Ah * o.Mh;
  const int threads = o.Ch >= 256 ? 256 : 64;
  NK_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL(k_octant_expand<T>, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, o, compact,
                       (const T*)table,
                       pidx, (T*)field);
  })
  return nk_check_launch("k_octant_expand");
}

__device__ __forceinline__ int nk_isqrt_ceil(int x) {  // smallest n >= 0 with n*n >= x (x < 2^24)
  if (x <= 0) return 0;
  // sqrtf of an integer below 2^24 is off by less than one ulp: the candidate is right or off by one, one step each way
  // settles it (no loops: the shell kernels call this twice per octant line)
  int n = (int)ceilf(sqrtf((float)x));
  n += (n * n < x) ? 1 : 0;
  n -= (n > 0 && (n - 1) * (n - 1) >= x) ? 1 : 0;
  return n;
}

// ---- octant expansion for NATURAL binning without an index stream ----------------------------------------------------
// On a grid with equal harmonic distances the bin of a point is a function of the integer k^2 = a^2 + b^2 + c^2 alone
// (bins = the ascending distinct k^2, bin_k2[bin] = its k^2).  Step 1 spreads the nb table entries (fp64, as the
// amplitude kernels produce them) over a DENSE table indexed by k^2 in the field dtype (3 MiB fp32 at 1024^3: L2
// resident); step 2 writes field8[a][b][c] = dense[a^2 + b^2 + c^2] with the index computed from the coordinates: the only
// HBM stream left is the store (the plain gather also read one int32 bin index per octant point).
template <typename T>
__global__ void __launch_bounds__(256) k_k2_dense(int64_t nb, const int32_t* __restrict__ bin_k2, const double* __restrict__ table,
                                                  T* __restrict__ dense) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb) dense[bin_k2[i]] = (T)table[i];
}
// `order` (optional): the octant lines (a, b) sorted by a^2 + b^2.  The lines of one workgroup -- one wavefront each -- then
// read nearly the same stretch of the table (index a^2 + b^2 + c^2, offsets within a few entries of each other for equal c),
// so that all but the first of them hit the CU's vector cache: in natural order every lane of every request pulled its own
// 128-byte line out of L2 for 4 bytes (0.51 ms at 1024^3 fp32 for a 0.54 GB array; sorted, 4 / 8 / 16 lines per workgroup:
// 0.39 / 0.30 / 0.27 ms).
template <typename T>
__global__ void __launch_bounds__(1024) k_octant_expand_k2(NkOct o, const int32_t* __restrict__ order,
                                                           const T* __restrict__ dense, T* __restrict__ field8) {
  // one wavefront per octant line (a, b); lanes run over c
  const int lane = threadIdx.x & 63;
  const int64_t idx = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (idx >= (int64_t)o.Ah * o.Mh) return;
  const int64_t line = order ? order[idx] : idx;
  const int b = (int)(line % o.Mh), a = (int)(line / o.Mh);
  const int r2 = a * a + b * b;
  T* dst = field8 + line * o.Ch;
  // (eight gathers per lane issued ahead of the stores: 0.492 vs 0.497 ms in natural order; two / four / eight lines per
  // wavefront in sorted order: 0.33 / 0.44 / 0.50 ms against 0.27 -- more wavefronts in flight beat longer ones)
  for (int c = lane; c < o.Ch; c += 64) __builtin_nontemporal_store(dense[r2 + c * c], dst + c);
}

extern "C" int nk_octant_expand_k2(int ndim, const int64_t* shape, const double* table, const int32_t* bin_k2, int64_t nb,
                                   void* dense, void* field8, int dtype, const int32_t* line_order, void* stream) {
  NkOct o;
  int rc = nk_make_oct(ndim, shape, o);
  if (rc != NK_OK) return rc;
  if (!table || !bin_k2 || !dense || !field8 || nb < 1) return nk_set_error(NK_ERR_INVALID, "nk_octant_expand_k2: bad argument");
  if ((int64_t)(o.Ah - 1) * (o.Ah - 1) + (int64_t)(o.Mh - 1) * (o.Mh - 1) + (int64_t)(o.Ch - 1) * (o.Ch - 1) >= (1 << 24))
    return nk_set_error(NK_ERR_UNSUPPORTED, "nk_octant_expand_k2: k^2 range too large");
  hipStream_t st = (hipStream_t)stream;
  const int64_t lines = (int64_t)o.Ah * o.Mh;
  static const int lpw_env = nk_vec_env_int("NK_EXPAND_LINES", 0);  // developer sweep: lines (wavefronts) per workgroup
  const int lpw = lpw_env > 0 && lpw_env <= 16 ? lpw_env : (line_order ? 16 : 4);
  NK_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL(k_k2_dense<T>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, nb, bin_k2, table, (T*)dense);
    // (a shell-by-shell variant -- table window in LDS, runs of field8 written per line like the shell scatter walks them --
    // was measured in round 3: 0.84 ms against 0.51 ms at 1024^3 fp32; the short store runs and the integer square roots
    // cost more than the L2 gathers they replace)
    hipLaunchKernelGGL(k_octant_expand_k2<T>, dim3((unsigned)((lines + lpw - 1) / lpw)), dim3(64 * lpw), 0, st, o, line_order,
                       (const T*)dense, (T*)field8);
  })
  return nk_check_launch("k_octant_expand_k2");
}

extern "C" int nk_octant_scatter(int ndim, const int64_t* shape, const double* w8, const int32_t* pidx, double* abar,
                                 int merge_swapped_lines, void* stream) {
  NkOct o;
  int rc = nk_make_oct(ndim, shape, o);
  if (rc != NK_OK) return rc;
  if (!w8 || !pidx || !abar) return nk_set_error(NK_ERR_INVALID, "nk_octant_scatter: null argument");
  const int64_t blocks = (int64_t)o.Ah * o.Mh;
  const int threads = o.Ch >= 256 ? 256 : 64;
  const int swap_merge = (merge_swapped_lines && ndim == 3 && o.A == o.M) ? 1 : 0;
  hipLaunchKernelGGL(k_octant_scatter, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, o, swap_merge, w8, pidx,
                     abar);
  return nk_check_launch("k_octant_scatter");
}

// ---- shell-binned octant scatter (natural binning on equal-distance grids) -------------------------------
// Bins are the distinct integer k^2 = a^2 + b^2 + c^2 in ascending order, so a block of NK_SHELL_BINS consecutive
// bins is a spherical shell [klo, khi) in k^2.  One workgroup owns (shell j, split s): it walks the octant lines
// (a, b) that cut the shell -- their c-range follows from two integer square roots -- and accumulates the run
// w8[a][b][c_lo..c_hi) into an LDS copy of the shell's bins (LDS fp64 adds, no global atomics); every (j, s) writes
// its whole shell to partial[s][.], folded afterwards.
#ifndef NK_SHELL_BINS
#define NK_SHELL_BINS 4096
#endif
#ifndef NK_SHELL_SPLITS
#define NK_SHELL_SPLITS 16
#endif
#ifndef NK_SHELL_NU
#define NK_SHELL_NU 8
#endif


__global__ void __launch_bounds__(256)
    k_octant_scatter_k2(NkOct o, const double* __restrict__ w8, const int32_t* __restrict__ pidx,
                        const int32_t* __restrict__ bin_k2, int nb, int64_t pstride, double* __restrict__ partial) {
  __shared__ double acc[NK_SHELL_BINS];
  const int j = blockIdx.x / NK_SHELL_SPLITS, s = blockIdx.x % NK_SHELL_SPLITS;
  const int bin0 = j * NK_SHELL_BINS;
  const int nbin = min(NK_SHELL_BINS, nb - bin0);
  for (int i = threadIdx.x; i < NK_SHELL_BINS; i += blockDim.x) acc[i] = 0.0;
  __syncthreads();
  const int klo = bin_k2[bin0];
  const int khi = bin0 + NK_SHELL_BINS < nb ? bin_k2[bin0 + NK_SHELL_BINS] : 0x7fffffff;
  const int hc2 = (o.Ch - 1) * (o.Ch - 1);
  // 16 groups of 16 lanes; a group takes NU lines per trip and issues all their loads before the first LDS add:
  // the runs are short (~10 points), so the kernel lives on memory-level parallelism, not on bandwidth per request
  constexpr int NG = 16, NU = NK_SHELL_NU;
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const bool last = khi == 0x7fffffff;
  for (int a = s; a < o.Ah; a += NK_SHELL_SPLITS) {
    const int ra = a * a;
    if (ra >= khi) break;
    const int b_hi = last ? o.Mh : min(o.Mh, nk_isqrt_ceil(khi - ra));  // b^2 < khi - ra
    const int b_lo = nk_isqrt_ceil(klo - hc2 - ra);                      // b^2 + hc2 >= klo - ra
    for (int b0 = b_lo + grp; b0 < b_hi; b0 += NG * NU) {
      int cc[NU], ce[NU];
      const double* wl[NU];
      const int32_t* pl[NU];
      double v[NU];
      int32_t pb[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int b = b0 + u * NG;
        const int r2 = ra + b * b;
        const bool on = b < b_hi;
        cc[u] = nk_isqrt_ceil(klo - r2) + l16;
        ce[u] = on ? (last ? o.Ch : min(o.Ch, nk_isqrt_ceil(khi - r2))) : 0;
        wl[u] = w8 + ((int64_t)a * o.Mh + b) * o.Ch;
        pl[u] = pidx + ((int64_t)a * o.M + b) * o.NL;
        const bool in = cc[u] < ce[u];
        v[u] = in ? wl[u][cc[u]] : 0.0;
        pb[u] = in ? pl[u][cc[u]] : bin0;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u)
        if (cc[u] < ce[u]) atomicAdd(&acc[pb[u] - bin0], v[u]);
      // rare: runs longer than 16 points
#pragma unroll
      for (int u = 0; u < NU; ++u)
        for (int c = cc[u] + 16; c < ce[u]; c += 16) atomicAdd(&acc[pl[u][c] - bin0], wl[u][c]);
    }
  }
  __syncthreads();
  double* dst = partial + (int64_t)s * pstride + bin0;
  for (int i = threadIdx.x; i < nbin; i += blockDim.x) dst[i] = acc[i];
}


// FIXED-POINT variant (the default when the caller passes the scale): the same walk, but every contribution is rounded to a
// multiple of q = 2^(e - 44), 2^e >= max |w8| (the maximum comes from the launch that wrote w8: nk_fuse.w8max), and added
// as a 64-bit INTEGER -- integer addition is associative, so the LDS atomics of the 16 groups give the same bits in any
// order: bit-reproducible bin sums at the speed of the floating-point atomics.  A workgroup adds < 2^18 points of
// magnitude <= 2^e: |sum| < 2^(e + 18) = 2^62 q, no overflow.  Rounding: <= q / 2 = max |w8| * 2.8e-14 per point, i.e.
// the relative error of a bin sum is ~1e-14 * (max |w8| / typical |w8| of the bin) -- the price: bins whose contributions
// are many orders of magnitude below the largest one of the whole grid lose relative (not absolute) accuracy.
// The floating-point-atomic kernel above stays for callers without a scale (w8max == NULL).
// x rounded to the nearest integer (ties to even, like __double2ll_rn) for |x| < 2^51, as two's complement: adding
// 1.5 * 2^52 leaves the integer in the low mantissa bits -- one fp64 add and one 64-bit subtract instead of the eight
// fp64-rate instructions of the library conversion (the scaled octant sums are below 2^45 by construction)
__device__ __forceinline__ unsigned long long nk_fixed_rn(double x) {
  const double magic = 6755399441055744.0;
  return (unsigned long long)(__double_as_longlong(x + magic) - __double_as_longlong(magic));
}

__global__ void __launch_bounds__(256)
    k_octant_scatter_k2_fx(NkOct o, const double* __restrict__ w8, const int32_t* __restrict__ pidx,
                           const int32_t* __restrict__ bin_k2, int nb, int64_t pstride, double* __restrict__ partial,
                           const double* __restrict__ w8max) {
  __shared__ unsigned long long acc[NK_SHELL_BINS];
  __shared__ unsigned int npoints;
  const int j = blockIdx.x / NK_SHELL_SPLITS, s = blockIdx.x % NK_SHELL_SPLITS;
  const int bin0 = j * NK_SHELL_BINS;
  const int nbin = min(NK_SHELL_BINS, nb - bin0);
  for (int i = threadIdx.x; i < NK_SHELL_BINS; i += blockDim.x) acc[i] = 0ull;
  if (threadIdx.x == 0) npoints = 0u;
  __syncthreads();
  // scale 2^(44 - e) with 2^e >= max |w8| > 0 (frexp: max = m * 2^e, 0.5 <= m < 1); all-zero input: scale irrelevant.
  // A NaN or an infinity among the octant sums arrives here as a non-finite maximum (the final pass joins the maxima
  // with a NaN-propagating rule): the bins are then NaN -- loud -- instead of rounded garbage.
  const double gmax = *w8max;
  const bool finite = gmax <= 1.79769313486231570815e308;  // false for NaN and +inf
  int e = 0;
  if (finite && gmax > 0.0) (void)frexp(gmax, &e);
  const double scale = ldexp(1.0, 44 - e), inv = ldexp(1.0, e - 44);
  unsigned int mine = 0u;
  const int klo = bin_k2[bin0];
  const int khi = bin0 + NK_SHELL_BINS < nb ? bin_k2[bin0 + NK_SHELL_BINS] : 0x7fffffff;
  const int hc2 = (o.Ch - 1) * (o.Ch - 1);
  constexpr int NG = 16, NU = NK_SHELL_NU;
  const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
  const bool last = khi == 0x7fffffff;
  for (int a = s; a < o.Ah; a += NK_SHELL_SPLITS) {
    const int ra = a * a;
    if (ra >= khi) break;
    const int b_hi = last ? o.Mh : min(o.Mh, nk_isqrt_ceil(khi - ra));  // b^2 < khi - ra
    const int b_lo = nk_isqrt_ceil(klo - hc2 - ra);                      // b^2 + hc2 >= klo - ra
    for (int b0 = b_lo + grp; b0 < b_hi; b0 += NG * NU) {
      int cc[NU], ce[NU];
      const double* wl[NU];
      const int32_t* pl[NU];
      double v[NU];
      int32_t pb[NU];
      // rows of line b0 once, the NU - 1 others of the trip by adding the row steps (wave-uniform) -- a 64-bit product per
      // line and array otherwise; likewise b^2 -> (b + NG)^2
      const double* wl0 = w8 + ((int64_t)a * o.Mh + b0) * o.Ch;
      const int32_t* pl0 = pidx + ((int64_t)a * o.M + b0) * o.NL;
      const int64_t wstep = (int64_t)NG * o.Ch, pstep = (int64_t)NG * o.NL;
      int r2 = ra + b0 * b0, dr2 = 2 * NG * b0 + NG * NG;  // r2(b + NG) = r2(b) + 2 NG b + NG^2
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int b = b0 + u * NG;
        const bool on = b < b_hi;
        cc[u] = nk_isqrt_ceil(klo - r2) + l16;
        ce[u] = on ? (last ? o.Ch : min(o.Ch, nk_isqrt_ceil(khi - r2))) : 0;
        wl[u] = wl0 + u * wstep;
        pl[u] = pl0 + u * pstep;
        const bool in = cc[u] < ce[u];
        v[u] = in ? wl[u][cc[u]] : 0.0;
        pb[u] = in ? pl[u][cc[u]] : bin0;
        r2 += dr2;
        dr2 += 2 * NG * NG;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u)
        if (cc[u] < ce[u]) {
          atomicAdd(&acc[pb[u] - bin0], nk_fixed_rn(v[u] * scale));
          ++mine;
        }
      // rare: runs longer than 16 points
#pragma unroll
      for (int u = 0; u < NU; ++u)
        for (int c = cc[u] + 16; c < ce[u]; c += 16) {
          atomicAdd(&acc[pl[u][c] - bin0], nk_fixed_rn(wl[u][c] * scale));
          ++mine;
        }
    }
  }
  atomicAdd(&npoints, mine);
  __syncthreads();
  // overflow guard: < 2^18 points of magnitude <= 2^e per workgroup keep |sum| < 2^62 quanta.  Callers pick this kernel
  // only for grids whose busiest (shell, split) is below that (nifty_amd counts it exactly at model set-up); a grid that
  // breaks the bound anyway gets NaN, never a wrapped sum.
  const bool ok = finite && npoints < (1u << 18);
  double* dst = partial + (int64_t)s * pstride + bin0;
  for (int i = threadIdx.x; i < nbin; i += blockDim.x)
    dst[i] = ok ? (double)(long long)acc[i] * inv : __longlong_as_double(0x7ff8000000000000LL);
}

extern "C" int nk_octant_scatter_k2(int ndim, const int64_t* shape, const double* w8, const int32_t* pidx,
                                    const int32_t* bin_k2, int64_t nb, double* scratch, double* abar, const double* w8max,
                                    void* stream) {
  NkOct o;
  int rc = nk_make_oct(ndim, shape, o);
  if (rc != NK_OK) return rc;
  if (!w8 || !pidx || !bin_k2 || !scratch || !abar || nb < 1 || nb > 0x7fffffff)
    return nk_set_error(NK_ERR_INVALID, "nk_octant_scatter_k2: bad argument");
  if ((int64_t)(o.Ah - 1) * (o.Ah - 1) + (int64_t)(o.Mh - 1) * (o.Mh - 1) + (int64_t)(o.Ch - 1) * (o.Ch - 1) >= (1 << 24))
    return nk_set_error(NK_ERR_UNSUPPORTED, "nk_octant_scatter_k2: k^2 range too large");
  // a workgroup of the fixed-point kernel must add < 2^18 points (overflow bound).  The busiest (shell, split) holds 2.13x
  // the average at 1024^3 (125 694 points against 59 006, counted exactly; the multiplicity of k^2 peaks near the cube
  // face): the average is held below 2^16, a margin of 4 on that ratio.  Larger octants (beyond ~1100^3) take the
  // floating-point atomics (not reproducible in the last bit, never wrong).
  const int64_t shells = (nb + NK_SHELL_BINS - 1) / NK_SHELL_BINS;
  const int64_t pstride = (nb + 31) / 32 * 32;
  static const int fp_atomics = nk_vec_env_int("NK_SCATTER_FP_ATOMICS", 0);  // 1: floating-point LDS atomics even with a scale
  const bool fixed = w8max != nullptr && !fp_atomics &&
                     (int64_t)o.Ah * o.Mh * o.Ch / NK_SHELL_SPLITS < ((int64_t)1 << 16) * shells;
  if (fixed)
    hipLaunchKernelGGL(k_octant_scatter_k2_fx, dim3((unsigned)(shells * NK_SHELL_SPLITS)), dim3(256), 0, (hipStream_t)stream, o,
                       w8, pidx, bin_k2, (int)nb, pstride, scratch, w8max);
  else
    hipLaunchKernelGGL(k_octant_scatter_k2, dim3((unsigned)(shells * NK_SHELL_SPLITS)), dim3(256), 0, (hipStream_t)stream, o, w8,
                       pidx, bin_k2, (int)nb, pstride, scratch);
  rc = nk_check_launch("k_octant_scatter_k2");
  if (rc != NK_OK) return rc;
  return nk_fold_copies(nb, NK_SHELL_SPLITS, pstride, scratch, abar, stream);
}

// ---- sparse response (LOSResponse, reference library/los_response.py:144-253): CSR with int32 columns and
//      float32 weights (the reference stores float32 weights too, :196), fp64 accumulation ------------------
// y[i] = sum_j (wgt ? wgt[j] : 1) * x[col[j]]  over rowptr[i] <= j < rowptr[i+1].  LANES lanes share a row (1, 4, 16 or 64):
// lane l adds the entries l, l + LANES, ... in ascending order, then a fixed shuffle tree joins the lanes -- the
// summation order is a function of the matrix only: bit-reproducible, no atomics.  This one kernel serves
//   * LOSResponse TIMES (one wavefront per line of sight, thousands of pixels per row),
//   * LOSResponse ADJOINT_TIMES through the TRANSPOSED matrix built once at set-up (one thread per pixel, a few lines
//     each) instead of fp64 atomics into a zeroed image,
//   * every scatter-add of a STATIC index map (DOFDistributor / PowerDistributor ADJOINT_TIMES, partial contractions,
//     the octant sums of grids without the shell structure): rows = bins, col = the bin-sorted permutation, wgt = NULL.
template <typename T, int LANES>
__global__ void __launch_bounds__(256) k_csr_rowsum(int64_t nrows, const int64_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ col, const float* __restrict__ wgt,
                                                    const T* __restrict__ x, T* __restrict__ y) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid / LANES;
  const int lane = (int)(gid % LANES);
  int64_t lo = 0, hi = 0;
  if (row < nrows) lo = rowptr[row], hi = rowptr[row + 1];
  double acc = 0.0;
  if (wgt) {
    for (int64_t j = lo + lane; j < hi; j += LANES) acc += (double)wgt[j] * (double)x[col[j]];
  } else {
    for (int64_t j = lo + lane; j < hi; j += LANES) acc += (double)x[col[j]];
  }
#pragma unroll
  for (int off = LANES / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, LANES);
  if (lane == 0 && row < nrows) y[row] = (T)acc;
}

// x[col[j]] += wgt[j] * y[i]: adjoint by atomics on a pre-zeroed x (rows overlap arbitrarily).  Kept for matrices whose
// transpose the caller does not hold; the sums then depend on the order of the atomics in the last bit.
template <typename T>
__global__ void k_spmv_t(int64_t nrows, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                         const float* __restrict__ wgt, const T* __restrict__ y, double* __restrict__ x) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int64_t lo = rowptr[row], hi = rowptr[row + 1];
  const double yv = (double)y[row];
  for (int64_t j = lo + lane; j < hi; j += 64) atomicAdd(x + col[j], (double)wgt[j] * yv);
}


// SHORT rows (LANES == 1: the transposed response -- a few lines per pixel --, the bin sums of a static index map), STAGED.
// With one thread per row every thread walks a chain of dependent loads -- row pointer -> (column, weight) -> gathered
// operand -- a few entries long: the launch is bound by those latencies, not by its bytes (4096^2 transposed line-of-sight
// matrix, 2.6e7 entries: 0.24 ms for 0.48 GB).  Here a workgroup owns ROWS_PER_WG consecutive rows, i.e. ONE contiguous
// range of entries: all threads stream that range in windows of W entries -- coalesced, independent loads, the operands
// gathered right behind them -- into LDS, and only then does every thread add up the runs of ITS rows, reading LDS.  The
// additions of a row run in the order of k_csr_rowsum<T, 1> (ascending entry, one fused multiply-add each): same bits.
// MB members of a batched launch share the index and weight loads (k_csr_rowsum_b's MB).
// rows per thread / entries per window: one member 4 / 2048 (24 KiB of LDS); four members 1 / 1024 (36 KiB: the bin sums of a
// batch -- 3.3 entries per row -- then fit ONE window per workgroup; with 4 / 512 they took seven, and the launch was slower
// than one thread per row: 105 against 85 us for eight members at 2048^2)
template <int MB>
struct NkStaged {
  static constexpr int RPT = MB == 1 ? 4 : 1, W = MB == 1 ? 2048 : 1024, ROWS = 256 * RPT;
};
template <typename T, int MB>
__global__ void __launch_bounds__(256) k_csr_rowsum_staged(int64_t nrows, const int64_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col, const float* __restrict__ wgt,
                                                           NkPtrs xs, NkPtrs ys, int count) {
  constexpr int W = NkStaged<MB>::W, U = W / 256, NK_STAGED_RPT = NkStaged<MB>::RPT, NK_STAGED_ROWS = NkStaged<MB>::ROWS;
  __shared__ double sx[MB][W];
  __shared__ float sw[W];
  const int tid = threadIdx.x, m0 = blockIdx.y * MB;
  const int64_t r0 = (int64_t)blockIdx.x * NK_STAGED_ROWS;
  const int64_t r1 = r0 + NK_STAGED_ROWS < nrows ? r0 + NK_STAGED_ROWS : nrows;
  const int64_t e0 = rowptr[r0], e1 = rowptr[r1];
  int64_t lo[NK_STAGED_RPT], hi[NK_STAGED_RPT];
  double acc[NK_STAGED_RPT][MB];
#pragma unroll
  for (int k = 0; k < NK_STAGED_RPT; ++k) {
    const int64_t row = r0 + k * 256 + tid;
    lo[k] = hi[k] = e1;
    if (row < r1) lo[k] = rowptr[row], hi[k] = rowptr[row + 1];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[k][m] = 0.0;
  }
  for (int64_t base = e0; base < e1; base += W) {
    const int cnt = e1 - base < W ? (int)(e1 - base) : W;
    // stage: every load of the window is issued before the first one is used
    int32_t c[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = tid + u * 256;
      c[u] = j < cnt ? col[base + j] : -1;
      w[u] = (wgt && j < cnt) ? wgt[base + j] : 1.0f;
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
      if (m0 + m >= count) break;
      const T* __restrict__ x = (const T*)xs.p[m0 + m];
      T v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = c[u] >= 0 ? x[c[u]] : (T)0;
#pragma unroll
      for (int u = 0; u < U; ++u) sx[m][tid + u * 256] = (double)v[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) sw[tid + u * 256] = w[u];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NK_STAGED_RPT; ++k) {
      const int a = (int)((lo[k] > base ? lo[k] : base) - base);
      const int b = (int)((hi[k] < base + cnt ? hi[k] : base + cnt) - base);
      if (wgt) {
        for (int j = a; j < b; ++j) {
          const double wj = (double)sw[j];
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[k][m] = __builtin_fma(wj, sx[m][j], acc[k][m]);
        }
      } else {
        for (int j = a; j < b; ++j)
#pragma unroll
          for (int m = 0; m < MB; ++m) acc[k][m] += sx[m][j];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < NK_STAGED_RPT; ++k) {
    const int64_t row = r0 + k * 256 + tid;
    if (row < r1)
#pragma unroll
      for (int m = 0; m < MB; ++m)
        if (m0 + m < count) ((T*)ys.p[m0 + m])[row] = (T)acc[k][m];
  }
}
// NK_ROWSUM_STAGED=0: one thread per row for the short rows again (A/B; same bits)
static inline bool nk_rowsum_staged() {
  static const int on = nk_vec_env_int("NK_ROWSUM_STAGED", 1);
  return on != 0;
}
template <typename T, int MB>
static int nk_launch_rowsum_staged(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, int count,
                                   const void* const* x, void* const* y, hipStream_t st) {
  const int64_t blocks = (nrows + NkStaged<MB>::ROWS - 1) / NkStaged<MB>::ROWS;
  if (blocks > 0x7fffffffLL) return nk_set_error(NK_ERR_UNSUPPORTED, "nk_csr_rowsum: too many rows for one launch");
  NkPtrs xs, ys;
  for (int m = 0; m < NK_MAX_BATCH; ++m) xs.p[m] = const_cast<void*>(x[m < count ? m : 0]), ys.p[m] = y[m < count ? m : 0];
  hipLaunchKernelGGL((k_csr_rowsum_staged<T, MB>), dim3((unsigned)blocks, (unsigned)((count + MB - 1) / MB)), dim3(256), 0, st, nrows,
                     rowptr, col, wgt, xs, ys, count);
  return nk_check_launch("k_csr_rowsum_staged");
}

template <typename T, int LANES>
static int nk_launch_rowsum(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, const void* x, void* y,
                            hipStream_t st) {
  const int64_t blocks = (nrows * LANES + 255) / 256;
  if (blocks > 0x7fffffffLL) return nk_set_error(NK_ERR_UNSUPPORTED, "nk_csr_rowsum: too many rows for one launch");
  hipLaunchKernelGGL((k_csr_rowsum<T, LANES>), dim3((unsigned)blocks), dim3(256), 0, st, nrows, rowptr, col, wgt, (const T*)x,
                     (T*)y);
  return nk_check_launch("k_csr_rowsum");
}

extern "C" int nk_csr_rowsum(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, const void* x, void* y,
                             int dtype, int lanes, void* stream) {
  if (nrows < 0 || !rowptr || (nrows > 0 && (!x || !y || !col)))
    return nk_set_error(NK_ERR_INVALID, "nk_csr_rowsum: bad argument");
  if (lanes != 1 && lanes != 4 && lanes != 16 && lanes != 64)
    return nk_set_error(NK_ERR_INVALID, "nk_csr_rowsum: lanes must be 1, 4, 16 or 64");
  if (nrows == 0) return NK_OK;
  hipStream_t st = (hipStream_t)stream;
  NkProfScope ps(st, 8, lanes == 1 ? 0 : lanes == 4 ? 1 : lanes == 16 ? 2 : 3, wgt ? 0 : 1);
  NK_DISPATCH_DTYPE(dtype, {
    switch (lanes) {
      case 1:
        if (nk_rowsum_staged()) return nk_launch_rowsum_staged<T, 1>(nrows, rowptr, col, wgt, 1, &x, &y, st);
        return nk_launch_rowsum<T, 1>(nrows, rowptr, col, wgt, x, y, st);
      case 4: return nk_launch_rowsum<T, 4>(nrows, rowptr, col, wgt, x, y, st);
      case 16: return nk_launch_rowsum<T, 16>(nrows, rowptr, col, wgt, x, y, st);
      default: return nk_launch_rowsum<T, 64>(nrows, rowptr, col, wgt, x, y, st);
    }
  })
}

// One thread (group) walks ITS row's entries once and gathers from every member's x: the row pointers, column indices and
// weights are read once per launch instead of once per member (short rows -- the bin sums of a static index map, a few
// entries each -- spend most of their bytes on that index).  Per member the additions run in the order of k_csr_rowsum: the
// same bits.  MB = members handled by one launch slice (compile-time, the tail slice zero-pads).
template <typename T, int LANES, int MB>
__global__ void __launch_bounds__(256) k_csr_rowsum_b(int64_t nrows, const int64_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col, const float* __restrict__ wgt, NkPtrs xs,
                                                      NkPtrs ys, int count) {
  const int m0 = blockIdx.y * MB;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid / LANES;
  const int lane = (int)(gid % LANES);
  int64_t lo = 0, hi = 0;
  if (row < nrows) lo = rowptr[row], hi = rowptr[row + 1];
  double acc[MB];
#pragma unroll
  for (int k = 0; k < MB; ++k) acc[k] = 0.0;
  // (two loops, written like k_csr_rowsum's: the weighted one contracts to the same fused multiply-add)
  if (wgt) {
    for (int64_t j = lo + lane; j < hi; j += LANES) {
      const int32_t c = col[j];
      const double w = (double)wgt[j];
#pragma unroll
      for (int k = 0; k < MB; ++k)
        if (m0 + k < count) acc[k] += w * (double)((const T*)xs.p[m0 + k])[c];
    }
  } else {
    for (int64_t j = lo + lane; j < hi; j += LANES) {
      const int32_t c = col[j];
#pragma unroll
      for (int k = 0; k < MB; ++k)
        if (m0 + k < count) acc[k] += (double)((const T*)xs.p[m0 + k])[c];
    }
  }
#pragma unroll
  for (int k = 0; k < MB; ++k) {
#pragma unroll
    for (int off = LANES / 2; off > 0; off >>= 1) acc[k] += __shfl_down(acc[k], off, LANES);
    if (lane == 0 && row < nrows && m0 + k < count) ((T*)ys.p[m0 + k])[row] = (T)acc[k];
  }
}
template <typename T, int LANES>
static int nk_launch_rowsum_b(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, int count,
                              const void* const* x, void* const* y, hipStream_t st) {
  const int64_t blocks = (nrows * LANES + 255) / 256;
  if (blocks > 0x7fffffffLL) return nk_set_error(NK_ERR_UNSUPPORTED, "nk_csr_rowsum_batch: too many rows for one launch");
  NkPtrs xs, ys;
  for (int m = 0; m < NK_MAX_BATCH; ++m) xs.p[m] = const_cast<void*>(x[m < count ? m : 0]), ys.p[m] = y[m < count ? m : 0];
  // short rows (one lane per row: bin sums) take four members per thread; long rows keep one member per grid row (their time
  // is the gather itself, and four accumulators per lane would only cost registers)
  if constexpr (LANES == 1) {
    if (nk_rowsum_staged()) return nk_launch_rowsum_staged<T, 4>(nrows, rowptr, col, wgt, count, x, y, st);
    hipLaunchKernelGGL((k_csr_rowsum_b<T, LANES, 4>), dim3((unsigned)blocks, (unsigned)((count + 3) / 4)), dim3(256), 0, st, nrows,
                       rowptr, col, wgt, xs, ys, count);
  } else {
    hipLaunchKernelGGL((k_csr_rowsum_b<T, LANES, 1>), dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, st, nrows, rowptr, col,
                       wgt, xs, ys, count);
  }
  return nk_check_launch("k_csr_rowsum_b");
}

extern "C" int nk_csr_rowsum_batch(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, int count,
                                   const void* const* x, void* const* y, int dtype, int lanes, void* stream) {
  if (nrows < 0 || !rowptr || !nk_batch_count_ok(count) || (nrows > 0 && (!nk_all_set(x, count) || !nk_all_set(y, count) || !col)))
    return nk_set_error(NK_ERR_INVALID, "nk_csr_rowsum_batch: bad argument");
  if (lanes != 1 && lanes != 4 && lanes != 16 && lanes != 64)
    return nk_set_error(NK_ERR_INVALID, "nk_csr_rowsum_batch: lanes must be 1, 4, 16 or 64");
  if (nrows == 0) return NK_OK;
  hipStream_t st = (hipStream_t)stream;
  NkProfScope ps(st, 8, lanes == 1 ? 0 : lanes == 4 ? 1 : lanes == 16 ? 2 : 3, wgt ? 0 : 1, count);
  NK_DISPATCH_DTYPE(dtype, {
    switch (lanes) {
      case 1: return nk_launch_rowsum_b<T, 1>(nrows, rowptr, col, wgt, count, x, y, st);
      case 4: return nk_launch_rowsum_b<T, 4>(nrows, rowptr, col, wgt, count, x, y, st);
      case 16: return nk_launch_rowsum_b<T, 16>(nrows, rowptr, col, wgt, count, x, y, st);
      default: return nk_launch_rowsum_b<T, 64>(nrows, rowptr, col, wgt, count, x, y, st);
    }
  })
}

extern "C" int nk_spmv(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, const void* x, void* y,
                       int dtype, void* stream) {
  if (!wgt && nrows > 0) return nk_set_error(NK_ERR_INVALID, "nk_spmv: bad argument");
  return nk_csr_rowsum(nrows, rowptr, col, wgt, x, y, dtype, 64, stream);
}

extern "C" int nk_spmv_t(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, const void* y,
                         double* x, int dtype, void* stream) {
  if (nrows < 0 || !rowptr || (nrows > 0 && (!x || !y))) return nk_set_error(NK_ERR_INVALID, "nk_spmv_t: bad argument");
  if (nrows == 0) return NK_OK;
  const unsigned blocks = (unsigned)((nrows + 3) / 4);
  NK_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL(k_spmv_t<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, nrows, rowptr, col, wgt, (const T*)y, x);
  })
  return nk_check_launch("k_spmv_t");
}

// ---- LONG rows over a grid: the response re-ordered by TILES of the grid (nk_tiled_rowsum, include/niftyk.h) -----------
// A line of sight that runs across the rows of the image touches a new cache line per pixel: with one wavefront per line
// (k_csr_rowsum<T, 64>) every gathered operand costs a 64-byte sector of fabric traffic for 8 bytes used, and the launch runs
// at 1.4 TB/s of algorithmic bytes (4096^2, 1e4 lines: 0.29 ms).  Here the entries are sorted by (tile of th x tw grid
// points, row) once at set-up: a workgroup loads ITS tile of x into LDS with whole-line reads -- every point of x is read
// once per launch -- and streams the tile's entries (uint16 position inside the tile + float32 weight: 6 bytes instead of
// 8).  The entries of one row inside the tile, a SEGMENT, are padded to whole BLOCKS of 8: a lane loads one block with three
// 16-byte requests, multiplies and adds its eight products in order, and the lanes of a segment -- at most 16 neighbours --
// are joined by four shuffle steps among lanes with the same partial-sum slot; the first lane of the run stores the sum.  No
// LDS traffic besides the gathers, no barrier after the tile is loaded: every wavefront walks its own 64-block STEPS of the
// tile, the loads of its next step in flight while it reduces the current one.  (Measured on the way, same matrix: products
// staged through LDS with 16 lanes per segment, 30 narrow loads per 512 entries: 0.155 ms whatever the tile shape, barriers
// or bank conflicts -- bound by the instruction count per entry, not by bytes or latency.)  A second small launch adds a
// row's partial sums in slot order.  No atomics; the order of every addition is a function of the plan only.
static_assert(sizeof(uint4) == 8 * sizeof(uint16_t), "a block is eight uint16 positions = one 16-byte request");
constexpr int NK_TILED_WAVES = 4;      // wavefronts per workgroup, each walking its own steps of the tile
struct NkTiledArgs {
  nk_tiled_csr m;
  NkPtrs xs, ys;
  double* scratch;
};
struct NkTiledBlock {
  uint4 l;      // eight uint16 positions
  float4 w0, w1;
  int slot;
};
template <typename T>
__global__ void __launch_bounds__(64 * NK_TILED_WAVES) k_tiled_partials(NkTiledArgs a) {
  extern __shared__ __align__(16) unsigned char nk_tiled_lds[];
  const nk_tiled_csr& m = a.m;
  const int tile_elems = m.th * m.tw;
  // rows of the LDS tile are one element longer than tw when tw is a power of two: the positions of a line that runs ACROSS
  // the rows are tw apart -- a multiple of the bank period -- and would all hit one bank
  const int pad_shift = (m.tw & (m.tw - 1)) == 0 ? 31 - __clz(m.tw) : 31;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, item = blockIdx.x;
  T* tile = reinterpret_cast<T*>(nk_tiled_lds);
  const T* __restrict__ x = (const T*)a.xs.p[blockIdx.y];
  double* __restrict__ partial = a.scratch + (int64_t)blockIdx.y * m.n_slots;
  const int64_t b1 = m.item_blk[item + 1];
  int64_t blk = m.item_blk[item] + wave * 64 + lane;  // this lane's block of the wavefront's first step
  const uint4* __restrict__ locv = reinterpret_cast<const uint4*>(m.loc);
  const float4* __restrict__ wgtv = reinterpret_cast<const float4*>(m.wgt);
  auto request = [&](int64_t bb) {
    NkTiledBlock r;
    r.slot = -1 - lane;  // (distinct negative values: lanes beyond the tile join nobody)
    r.l = uint4{0, 0, 0, 0};
    r.w0 = r.w1 = float4{0.f, 0.f, 0.f, 0.f};
    if (bb < b1) r.l = locv[bb], r.w0 = wgtv[2 * bb], r.w1 = wgtv[2 * bb + 1], r.slot = m.blk_slot[bb];
    return r;
  };
  NkTiledBlock cur = request(blk);
  {  // the tile: (outer, ty, tx)
    const int t = m.item_tile[item];
    const int ntx = (m.nx + m.tw - 1) / m.tw, nty = (m.ny + m.th - 1) / m.th;
    const int tx = t % ntx, ty = (t / ntx) % nty;
    const int64_t o = t / (ntx * nty);
    const int64_t origin = (o * m.ny + (int64_t)ty * m.th) * m.nx + (int64_t)tx * m.tw;
    const int hh = m.ny - ty * m.th < m.th ? m.ny - ty * m.th : m.th;
    const int ww = m.nx - tx * m.tw < m.tw ? m.nx - tx * m.tw : m.tw;
    for (int i = tid; i < tile_elems; i += 64 * NK_TILED_WAVES) {
      const int ly = i / m.tw, lx = i - ly * m.tw;
      tile[i + (i >> pad_shift)] = (ly < hh && lx < ww) ? x[origin + (int64_t)ly * m.nx + lx] : (T)0;
    }
  }
  __syncthreads();
  const int64_t first = blk - lane;  // (uniform) first block of the wavefront's current step
  for (int64_t s = first; s < b1; s += 64 * NK_TILED_WAVES) {
    const NkTiledBlock nxt = request(s + lane + 64 * NK_TILED_WAVES);  // in flight while this step is reduced
    const unsigned lw[4] = {cur.l.x, cur.l.y, cur.l.z, cur.l.w};
    const float ww[8] = {cur.w0.x, cur.w0.y, cur.w0.z, cur.w0.w, cur.w1.x, cur.w1.y, cur.w1.z, cur.w1.w};
    double xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = (int)((lw[i >> 1] >> ((i & 1) * 16)) & 0xffffu);
      xv[i] = (double)tile[p + (p >> pad_shift)];
    }
    double acc = 0.0;
    {
#pragma clang fp contract(off)  // rounded products, added in order: what tiled_rowsum_host states (__dmul_rn is a plain `*` here)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double pr = (double)ww[i] * xv[i];
        acc = acc + pr;
      }
    }
    const int slot = cur.slot;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const double other = __shfl_down(acc, off, 64);
      const int oslot = __shfl_down(slot, off, 64);
      acc += (lane + off < 64 && oslot == slot) ? other : 0.0;
    }
    const int prev = __shfl_up(slot, 1, 64);
    if (slot >= 0 && (lane == 0 || prev != slot)) partial[slot] = acc;
    cur = nxt;
  }
}
// y[row] = the row's partial sums in slot order (16 lanes per row, the lanes' sums joined by a fixed tree)
template <typename T>
__global__ void __launch_bounds__(256) k_tiled_rows(NkTiledArgs a) {
  const nk_tiled_csr& m = a.m;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = gid >> 4;
  const int q = (int)(gid & 15);
  const double* __restrict__ partial = a.scratch + (int64_t)blockIdx.y * m.n_slots;
  int64_t lo = 0, hi = 0;
  if (row < m.n_rows) lo = m.row_slot[row], hi = m.row_slot[row + 1];
  double acc = 0.0;
  for (int64_t j = lo + q; j < hi; j += 16) acc += partial[j];
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) acc += __shfl_down(acc, off, 16);
  if (q == 0 && row < m.n_rows) ((T*)a.ys.p[blockIdx.y])[row] = (T)acc;
}

extern "C" int nk_tiled_rowsum(const nk_tiled_csr* m, int count, const void* const* x, void* const* y, double* scratch, int dtype,
                               void* stream) {
  if (!m || !nk_batch_count_ok(count) || m->n_rows < 0 || m->n_items < 0 || m->n_slots < 0)
    return nk_set_error(NK_ERR_INVALID, "nk_tiled_rowsum: bad argument");
  if (m->n_rows == 0) return NK_OK;
  if (!nk_all_set(x, count) || !nk_all_set(y, count) || !m->row_slot || (m->n_slots > 0 && !scratch))
    return nk_set_error(NK_ERR_INVALID, "nk_tiled_rowsum: bad argument");
  if (m->th < 1 || m->tw < 1 || m->th * m->tw > 32768 || m->ny < 1 || m->nx < 1)
    return nk_set_error(NK_ERR_INVALID, "nk_tiled_rowsum: a tile holds at most 32768 grid points");
  if (m->n_items > 0 && (!m->item_tile || !m->item_blk || !m->blk_slot || !m->loc || !m->wgt))
    return nk_set_error(NK_ERR_INVALID, "nk_tiled_rowsum: bad argument");
  if ((reinterpret_cast<uintptr_t>(m->loc) | reinterpret_cast<uintptr_t>(m->wgt)) & 15)
    return nk_set_error(NK_ERR_INVALID, "nk_tiled_rowsum: loc and wgt must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  NkTiledArgs a;
  a.m = *m;
  a.scratch = scratch;
  for (int k = 0; k < NK_MAX_BATCH; ++k) a.xs.p[k] = const_cast<void*>(x[k < count ? k : 0]), a.ys.p[k] = y[k < count ? k : 0];
  NkProfScope ps(st, 8, 3, 0, count);
  NK_DISPATCH_DTYPE(dtype, {
    const size_t lds = sizeof(T) * (size_t)m->th * (m->tw + 1);
    if (lds > 64 * 1024) return nk_set_error(NK_ERR_UNSUPPORTED, "nk_tiled_rowsum: tile too large for the LDS budget");
    if (m->n_items > 0) {
      hipLaunchKernelGGL(k_tiled_partials<T>, dim3((unsigned)m->n_items, (unsigned)count), dim3(64 * NK_TILED_WAVES), lds, st, a);
      const int rc = nk_check_launch("k_tiled_partials");
      if (rc != NK_OK) return rc;
    }
    const int64_t blocks = (m->n_rows * 16 + 255) / 256;
    hipLaunchKernelGGL(k_tiled_rows<T>, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, st, a);
    return nk_check_launch("k_tiled_rows");
  })
}

// ---- cyclic shift of a C-ordered array along any of its (<= 6) axes: out[(i_d + shift_d) mod n_d ...] = in[i ...]
//      (FFTShiftOperator, reference operators/harmonic_operators.py:383-423: numpy's fftshift / ifftshift).  Elements are moved
//      as opaque units of 4, 8 or 16 bytes (float, double / complex64, complex128); a thread walks elements of the LAST axis,
//      so loads are contiguous runs and stores are contiguous up to the one wrap-around of that axis.
struct NkRollArgs {
  int ndim;
  int64_t n[6], shift[6];  // shift already reduced to [0, n)
};
template <typename U>
__global__ void __launch_bounds__(256) k_roll(NkRollArgs a, int64_t total, const U* __restrict__ in, U* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t rem = i, dst = 0, stride = 1;
    for (int d = a.ndim - 1; d >= 0; --d) {
      const int64_t c = rem % a.n[d];
      rem /= a.n[d];
      int64_t t = c + a.shift[d];
      if (t >= a.n[d]) t -= a.n[d];
      dst += t * stride;
      stride *= a.n[d];
    }
    out[dst] = in[i];
  }
}
extern "C" int nk_roll(int ndim, const int64_t* shape, const int64_t* shift, int elem_bytes, const void* in, void* out, void* stream) {
  if (ndim < 1 || ndim > 6 || !shape || !shift || (elem_bytes != 4 && elem_bytes != 8 && elem_bytes != 16))
    return nk_set_error(NK_ERR_INVALID, "nk_roll: 1 <= ndim <= 6, elements of 4, 8 or 16 bytes");
  NkRollArgs a;
  a.ndim = ndim;
  int64_t total = 1;
  for (int d = 0; d < ndim; ++d) {
    if (shape[d] < 0) return nk_set_error(NK_ERR_INVALID, "nk_roll: negative axis length");
    a.n[d] = shape[d];
    a.shift[d] = shape[d] > 0 ? ((shift[d] % shape[d]) + shape[d]) % shape[d] : 0;
    total *= shape[d];
  }
  if (total == 0) return NK_OK;
  if (!in || !out || in == out) return nk_set_error(NK_ERR_INVALID, "nk_roll: in and out must be distinct arrays");
  hipStream_t st = (hipStream_t)stream;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 64);
  if (elem_bytes == 4) hipLaunchKernelGGL(k_roll<uint32_t>, dim3(blocks), dim3(256), 0, st, a, total, (const uint32_t*)in, (uint32_t*)out);
  else if (elem_bytes == 8) hipLaunchKernelGGL(k_roll<uint64_t>, dim3(blocks), dim3(256), 0, st, a, total, (const uint64_t*)in, (uint64_t*)out);
  else hipLaunchKernelGGL(k_roll<uint4>, dim3(blocks), dim3(256), 0, st, a, total, (const uint4*)in, (uint4*)out);
  return nk_check_launch("k_roll");
}

// ---- inclusive prefix sum of a vector (the two log-integrations of the generic amplitude graph, reference
//      library/correlated_fields.py:147-161).  nb-sized (<= ~1e6) and far off the hot path -- the fused amplitude
//      kernels of nk_amp.hip are what the fused engine uses -- so ONE workgroup walks the vector in tiles with a
//      running carry: deterministic, fp64 accumulation.  reverse != 0: suffix sums out[i] = sum_{j >= i} in[j].
template <typename T>
__global__ void __launch_bounds__(1024) k_cumsum(int64_t n, const T* __restrict__ in, T* __restrict__ out, int reverse) {
  __shared__ double wsum[16];
  __shared__ double carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0.0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int64_t src = reverse ? n - 1 - i : i;
    double v = i < n ? (double)in[src] : 0.0;
    for (int off = 1; off < 64; off <<= 1) {  // inclusive scan inside the wavefront
      const double t = __shfl_up(v, off, 64);
      if (lane >= off) v += t;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    double pre = carry_s;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    if (i < n) out[src] = (T)(v + pre);
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = v + pre;
    __syncthreads();
  }
}

extern "C" int nk_cumsum(int64_t n, const void* in, void* out, int reverse, int dtype, void* stream) {
  if (n < 0 || (n > 0 && (!in || !out))) return nk_set_error(NK_ERR_INVALID, "nk_cumsum: bad argument");
  if (n == 0) return NK_OK;
  NK_DISPATCH_DTYPE(dtype, {
    hipLaunchKernelGGL(k_cumsum<T>, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, (const T*)in, (T*)out, reverse);
  })
  return nk_check_launch("k_cumsum");
}
