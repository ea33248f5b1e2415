// nk_fft.hip -- gfx950 kernels + C ABI for the genuine N-D Hartley transform and the c2c FFT.
// See nk_fft_phases.h for the algorithm; nk_core.h for the LDS line FFT and fused prologue/epilogue.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cstdio>
#include <cstring>
#include <new>

#include "nk_fft_batch.h"

// ------------------------------------------------------------------------------------------------
// optional live profiling: HIP events around every pass-kernel launch, on the launch stream
// ------------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>
using ProfScope = NkProfScope;  // live per-kernel HIP events (nk_util.h)
thread_local const NkBatchCtx* t_batch = nullptr;  // set by nk_hartley_fused_batch around its launches

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void nk_run_stages(C2<T>* lds, int tid, int nthr, const NkLinePlan& lp, const NkTile& tl,
                                              const C2<T>* __restrict__ tw) {
  int L = lp.n;
  for (int s = 0; s < lp.nstage; ++s) {
    const int R = lp.radix[s];
    NK_STAGE_DISPATCH(R, lds, tid, nthr, lp, tl, L, tw, s)
    L /= R;
    __syncthreads();
  }
}

// generic LDS kernels: any thread count works (loops stride by blockDim.x).  fp64 is capped at 512 threads so that the
// compiler may use 256 VGPRs -- with the default 1024-thread bound (128 VGPRs) every fp64 kernel spilled 52-68 B / lane
template <typename T>
constexpr int nk_gen_max_threads() {
  return sizeof(T) == 8 ? 512 : 1024;
}
template <typename T>
static inline int nk_gen_threads(int t) {
  return t > nk_gen_max_threads<T>() ? nk_gen_max_threads<T>() : t;
}

template <typename T>
__global__ void __launch_bounds__(nk_gen_max_threads<T>()) k_passA(NkPassA p, NkFuse f, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ twr,
                        C2<T>* __restrict__ work) {
  extern __shared__ __align__(16) unsigned char smem[];
  C2<T>* lds = (C2<T>*)smem;
  nk_passA_load<T>(p, f, blockIdx.x, threadIdx.x, blockDim.x, lds);
  __syncthreads();
  nk_run_stages<T>(lds, threadIdx.x, blockDim.x, p.lp, p.tl, tw);
  nk_passA_store<T>(p, blockIdx.x, threadIdx.x, blockDim.x, lds, twr, work);
}

template <typename T>
__global__ void __launch_bounds__(nk_gen_max_threads<T>()) k_pass1d(NkPassA p, NkFuse f, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ twr) {
  extern __shared__ __align__(16) unsigned char smem[];
  C2<T>* lds = (C2<T>*)smem;
  nk_passA_load<T>(p, f, blockIdx.x, threadIdx.x, blockDim.x, lds);
  __syncthreads();
  nk_run_stages<T>(lds, threadIdx.x, blockDim.x, p.lp, p.tl, tw);
  double acc = 0.0;
  nk_pass1d_store<T>(p, f, blockIdx.x, threadIdx.x, blockDim.x, lds, twr, acc);
  nk_flush_energy(f, acc, smem);
}

template <typename T>
__global__ void __launch_bounds__(nk_gen_max_threads<T>()) k_passB(NkPassS p, const C2<T>* __restrict__ tw, C2<T>* __restrict__ work) {
  extern __shared__ __align__(16) unsigned char smem[];
  C2<T>* lds = (C2<T>*)smem;
  nk_passS_load<T>(p, blockIdx.x, threadIdx.x, blockDim.x, lds, work);
  __syncthreads();
  nk_run_stages<T>(lds, threadIdx.x, blockDim.x, p.lp, p.tl, tw);
  nk_passB_store<T>(p, blockIdx.x, threadIdx.x, blockDim.x, lds, work);
}

template <typename T>
__global__ void __launch_bounds__(nk_gen_max_threads<T>()) k_passC(NkPassS p, NkFuse f, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ work,
                        C2<T>* __restrict__ scratch) {
  extern __shared__ __align__(16) unsigned char smem[];
  C2<T>* lds = (C2<T>*)smem;
  nk_passS_load<T>(p, blockIdx.x, threadIdx.x, blockDim.x, lds, work);
  __syncthreads();
  nk_run_stages<T>(lds, threadIdx.x, blockDim.x, p.lp, p.tl, tw);
  double acc = 0.0;
  nk_passC_store<T>(p, f, blockIdx.x, threadIdx.x, blockDim.x, lds, scratch, acc);
  nk_flush_energy(f, acc, smem);
}

template <typename T>
__global__ void k_passD(NkGeom g, NkFuse f, const C2<T>* __restrict__ scratch, int64_t total) {
  __shared__ double red[16];
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  if (gid < total) nk_passD<T>(g, f, gid, scratch, acc);
  nk_flush_energy(f, acc, red);
}

// ---- fast path: register-resident passes (nk_fft2.h) ---------------------------------------------------
template <typename T, int H, bool IS_1D>
__global__ void __launch_bounds__((ContigTile<T, H>::THREADS))
    k2_contig(NkPassA p, NkFuse f, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ twr, C2<T>* __restrict__ work) {
  extern __shared__ __align__(16) unsigned char smem[];
  DeviceExec<T, Sched<T, H>::E> ex;
  double acc = 0.0;
  nk_contig_body<T, H, ContigTile<T, H>::TILE, IS_1D>(ex, p, f, blockIdx.x, (T*)smem, tw, twr, work, &acc);
  if (IS_1D) nk_flush_energy(f, acc, smem);
}

// first pass (MODE 3): two workgroups per CU (<= 128 VGPRs at 512 threads); the in-place pass keeps the whole
// register file for its loads in flight
template <typename T, int N, int MODE, int PC>
__global__ void __launch_bounds__((StridedTile<T, N, nk_strided_cx<MODE, PC>(), MODE>::THREADS),
                                  (StridedTile<T, N, nk_strided_cx<MODE, PC>(), MODE>::SC::E == 64
                                       ? 2  // 256 threads x 64 elements: two workgroups per CU
                                       : (MODE == 3 && NK_S1_TWO_WG && StridedTile<T, N>::LDS_BYTES <= 80 * 1024 &&
                                                  StridedTile<T, N>::THREADS <= 512
                                              ? 2 * StridedTile<T, N>::THREADS / 256
                                              : NK_S0_WAVES)))
    k2_strided(NkPassS p, NkFuse f, const C2<T>* __restrict__ tw, C2<T>* __restrict__ work, C2<T>* __restrict__ scratch, int xmap) {
  extern __shared__ __align__(16) unsigned char smem[];
  using ST = StridedTile<T, N, nk_strided_cx<MODE, PC>(), MODE>;
  DeviceExec<T, ST::SC::E> ex;
  double acc = 0.0;
  // the octant prologues bring their own XCD-aware order on 3-D grids (nk_oct_block_remap); in 2-D they take the
  // XCD-contiguous one like every other class: neighbouring column tiles -- 64-byte row segments at 4096 fp64 -- on ONE XCD,
  // whose L2 then fetches each 128-byte line once instead of two XCDs fetching it each
  constexpr bool OCT = MODE == 3 && (PC == 4 || PC == 5 || PC == 9);
  const int64_t blk = (xmap && (!OCT || p.g.ndim != 3)) ? nk_xcd_contig(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  C2<T>* tw_lds = ST::TWLDS ? reinterpret_cast<C2<T>*>(smem + ST::LDS_BYTES) : nullptr;
  nk_strided_body<T, N, ST::TILE, MODE, PC, nk_strided_cx<MODE, PC>()>(ex, p, f, blk, (T*)smem, tw, work, scratch, &acc, tw_lds);
  (void)acc;
}

// min waves / SIMD: 4 only for the light fp32 affine / multiply classes; everything else gets 168 VGPRs -- at 128 the fp64
// and likelihood kernels spilled (fp64 512^3 affine pass 0.61 -> 0.42 ms without the spills)
template <typename T, int NL, bool COUPLES, int EC, int PAIR>
__global__ void __launch_bounds__((FinalTile<T, NL, EC, COUPLES ? 2 : 1>::THREADS),
                                  (nk_final_waves<T, COUPLES, EC, FinalTile<T, NL, EC, COUPLES ? 2 : 1>::THREADS>()))
    k2_final(NkPassF p, NkFuse f, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ work, int xmap) {
  extern __shared__ __align__(16) unsigned char smem[];
  DeviceExec<T, SchedF<T, NL>::E> ex;
  double acc = 0.0;
  float wmax = 0.0f;
  const int64_t blk = xmap ? nk_xcd_contig(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x + p.blk0;
  nk_final_body<T, NL, FinalTile<T, NL, EC, COUPLES ? 2 : 1>::TILE, COUPLES, EC, PAIR>(ex, p, f, blk, (T*)smem, tw, work, &acc,
                                                                                      (COUPLES && (EC == 2 || EC == -1)) ? &wmax : nullptr);
  nk_flush_energy(f, acc, smem);
  if constexpr (COUPLES && (EC == 2 || EC == -1)) nk_flush_wmax(f, wmax);  // 3-D launches only (nk_final_with_slots)
}

template <typename T, int NL, bool COUPLES, int EC, int PAIR = 0>
static int nk_launch_final_c(NkPassF pf, const NkFuse& f, const C2<T>* tw, const C2<T>* work, hipStream_t st) {
  using CT = FinalTile<T, NL, EC, COUPLES ? 2 : 1>;
  static_assert(!COUPLES || CT::TILE >= 2, "the couple (b0, M - b0) must live in one workgroup");
  auto kern = k2_final<T, NL, COUPLES, EC, PAIR>;
  static unsigned long long attr_mask = 0;  // per-device attribute
  if (CT::LDS_BYTES > 64 * 1024 && nk_first_on_device(attr_mask)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CT::LDS_BYTES);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(k2_final)");
  }
  pf.tiles_per_a = (COUPLES && pf.A > 1 && CT::TILE >= 2) ? (pf.M / 2 + 1 + CT::TILE / 2 - 1) / (CT::TILE / 2)
                                               : (pf.M + CT::TILE - 1) / CT::TILE;
  int64_t blocks = (int64_t)pf.g.batch * (pf.A / 2 + 1) * pf.tiles_per_a;
  static const int xmap_env = nk_env_int("NK_XMAP", NK_XMAP_DEFAULT);
  NkFuse fs = f;
  pf.blk0 = 0;
  if (pf.a_cnt > 0) {
    // one stage of a pipelined sandwich: the workgroups of the pair indices a0 .. a0 + a_cnt - 1, their reduction slots
    // where the whole launch would have them (slot = workgroup * waves + wave)
    if (pf.g.batch != 1 || (xmap_env & 4)) return nk_set_error(NK_ERR_UNSUPPORTED, "staged final pass: batch 1, natural block order");
    pf.blk0 = (int64_t)pf.a0 * pf.tiles_per_a;
    blocks = (int64_t)pf.a_cnt * pf.tiles_per_a;
    const int64_t off = pf.blk0 * ((CT::THREADS + 63) / 64);
    if (fs.value_slots > 0) {
      if (fs.value) fs.value += off;
      if (fs.w8max) fs.w8max += off;
      fs.value_slots = fs.value_slots > off ? (int)(fs.value_slots - off) : 0;
      if (fs.value_slots == 0) return nk_set_error(NK_ERR_RUNTIME, "final pass: more wavefronts than reduction slots (nk_value_slot_count)");
    }
  }
  // every wavefront owns one slot of the energy / |w8| areas (nk_final_with_slots sized them): never drop a partial silently
  if (fs.value_slots > 0 && blocks * ((CT::THREADS + 63) / 64) > fs.value_slots)
    return nk_set_error(NK_ERR_RUNTIME, "final pass: more wavefronts than reduction slots (nk_value_slot_count)");
  if (t_batch != nullptr) {
    if constexpr (nk_twin_final<NL, COUPLES, EC, PAIR>()) {
      if (pf.a_cnt > 0 || pf.g.batch != 1) return nk_set_error(NK_ERR_UNSUPPORTED, "batched final pass: one unstaged grid per member");
      return nk_twin_launch_final<T, NL, COUPLES, EC>(pf, blocks, tw, xmap_env & 4, st);
    } else {
      return nk_set_error(NK_ERR_UNSUPPORTED, "batched final pass: no batched twin of this kernel class");
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(CT::THREADS), CT::LDS_BYTES, st, pf, fs, tw, work, xmap_env & 4);
  return nk_check_launch("k2_final");
}

// the scatter epilogue (VJP) runs on line couples (8 sign-flip images per atomic); everything else on plain pairs
template <typename T, int NL>
static int nk_launch_final(const NkPassF& pf, const NkFuse& f, const C2<T>* tw, const C2<T>* work, hipStream_t st) {
  static const int generic = nk_env_int("NK_EC_GENERIC", 0);
  if (generic && !f.field_octant) {
    if (f.epi == NK_EPI_VJP) return nk_launch_final_c<T, NL, true, -1>(pf, f, tw, work, st);
    return nk_launch_final_c<T, NL, false, -1>(pf, f, tw, work, st);
  }
  if (f.epi == NK_EPI_VJP && f.afield) {
    // 2-D grids have no couples of lines (A == 1: the slots are single pairs anyway), yet the couple tile of two pairs is what
    // the COUPLES kernels are built on -- at 4096 fp64 points 135 KiB of LDS, ONE workgroup per CU.  For such lines the
    // single-pair build runs instead (half the LDS, twice the resident workgroups): the same slots, lanes and reduction-slot
    // order, i.e. the same bits
    if constexpr (nk_final_single_2d<T, NL>())
      if (pf.A == 1) return nk_launch_final_c<T, NL, false, 2>(pf, f, tw, work, st);
    return nk_launch_final_c<T, NL, true, 2>(pf, f, tw, work, st);
  }
  if (f.epi == NK_EPI_VJP) return nk_launch_final_c<T, NL, true, -1>(pf, f, tw, work, st);
  if (f.epi == NK_EPI_AFFINE) return nk_launch_final_c<T, NL, false, 0>(pf, f, tw, work, st);
  if (f.epi == NK_EPI_MUL) return nk_launch_final_c<T, NL, false, 1>(pf, f, tw, work, st);
  if (f.epi == NK_EPI_LIKELIHOOD) {
    if constexpr (sizeof(T) == 8)  // float arrays at both ends of a double pipeline (nk_fuse.io32)
      if (f.io32) return nk_launch_final_c<T, NL, false, 5>(pf, f, tw, work, st);
    return nk_launch_final_c<T, NL, false, 3>(pf, f, tw, work, st);
  }
  return nk_launch_final_c<T, NL, false, -1>(pf, f, tw, work, st);
}

// final pass of the sandwich pipeline (row-mirror pairing): the hot classes are compile-time, the rest run-time
template <typename T, int NL>
static int nk_launch_final3(const NkPassF& pf, const NkFuse& f, const C2<T>* tw, const C2<T>* work, hipStream_t st) {
  if (f.epi == NK_EPI_VJP && f.afield) return nk_launch_final_c<T, NL, true, 2, 1>(pf, f, tw, work, st);
  if (f.epi == NK_EPI_VJP) return nk_launch_final_c<T, NL, true, -1, 1>(pf, f, tw, work, st);
  if (f.epi == NK_EPI_AFFINE) return nk_launch_final_c<T, NL, false, 0, 1>(pf, f, tw, work, st);
  return nk_launch_final_c<T, NL, false, -1, 1>(pf, f, tw, work, st);
}

template <typename T>
static int nk_dispatch_final3(int nl, const NkPassF& pf, const NkFuse& f, const C2<T>* tw, const C2<T>* work,
                              hipStream_t st) {
  switch (nl) {
#define NK_CASE(NN) \
  case NN:          \
    return nk_launch_final3<T, NN>(pf, f, tw, work, st);
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  return nk_set_error(NK_ERR_UNSUPPORTED, "no fast final pass for this length");
}

template <typename T>
static int nk_dispatch_final_pair(int nl, const NkPassF& pf, const NkFuse& fa, const NkFuse& fb, const C2<T>* tw,
                                  const C2<T>* worka, const C2<T>* workb, hipStream_t st) {
  switch (nl) {
#define NK_CASE(NN) \
  case NN:          \
    return nk_launch_final_pair<T, NN>(pf, fa, fb, tw, worka, workb, st);
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  return nk_set_error(NK_ERR_UNSUPPORTED, "no fast final pass for this length");
}

template <typename T>
static int nk_dispatch_final(int nl, const NkPassF& pf, const NkFuse& f, const C2<T>* tw, const C2<T>* work,
                             hipStream_t st) {
  switch (nl) {
#define NK_CASE(NN) \
  case NN:          \
    return nk_launch_final<T, NN>(pf, f, tw, work, st);
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  return nk_set_error(NK_ERR_UNSUPPORTED, "no fast final pass for this length");
}

template <typename T, int H, bool IS_1D>
static int nk_launch_contig(const NkPassA& pa, const NkFuse& f, const C2<T>* tw, const C2<T>* twr, C2<T>* work,
                            hipStream_t st) {
  using CT = ContigTile<T, H>;
  auto kern = k2_contig<T, H, IS_1D>;
  static unsigned long long attr_mask = 0;  // per-device attribute
  if (CT::LDS_BYTES > 64 * 1024 && nk_first_on_device(attr_mask)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CT::LDS_BYTES);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(k2_contig)");
  }
  const int64_t blocks = (pa.nlines + CT::TILE - 1) / CT::TILE;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(CT::THREADS), CT::LDS_BYTES, st, pa, f, tw, twr, work);
  return nk_check_launch("k2_contig");
}

template <typename T, int N, int MODE, int PC>
static int nk_launch_strided_pc(NkPassS ps, const NkFuse& f, const C2<T>* tw, C2<T>* work, C2<T>* scratch, hipStream_t st) {
  using ST = StridedTile<T, N, nk_strided_cx<MODE, PC>(), MODE>;
  auto kern = k2_strided<T, N, MODE, PC>;
  static unsigned long long attr_mask = 0;  // per-device attribute
  if (ST::LDS_TOTAL > 64 * 1024 && nk_first_on_device(attr_mask)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ST::LDS_TOTAL);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(k2_strided)");
  }
  ps.tl.tile = ST::TILE;
  ps.tl.dtile = nk_make_div(ST::TILE);
  ps.tiles_per_slab = (int)(ps.inner / ST::TILE);
  const int64_t blocks = ps.outer * ps.tiles_per_slab;
  static const int xmap_env = nk_env_int("NK_XMAP", NK_XMAP_DEFAULT);
  // in-place pass: bit 1 for every layout, bit 3 for the middle-axis pass of the sandwich only (blo > 0: 1.66 -> 1.62 ms at
  // 1024^3 fp32, while the in-place pass of the six-pass pipeline loses 8 % with it)
  const int xmap = MODE == 3 ? (xmap_env & 1) : ((xmap_env & 2) | ((xmap_env & 8) && ps.blo > 0 ? 2 : 0));
  if (t_batch != nullptr) {
    if constexpr (MODE != 3 || !nk_twin_strided<N, PC>()) {
      return nk_set_error(NK_ERR_UNSUPPORTED, "batched strided pass: no batched twin of this kernel class");
    } else {
      return nk_twin_launch_strided<T, N, PC>(ps, blocks, tw, xmap, st);
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(ST::THREADS), ST::LDS_TOTAL, st, ps, f, tw, work, scratch, xmap);
  return nk_check_launch("k2_strided");
}

template <typename T, bool IS_1D>
static int nk_dispatch_contig(int h, const NkPassA& pa, const NkFuse& f, const C2<T>* tw, const C2<T>* twr, C2<T>* work,
                              hipStream_t st) {
  switch (h) {
#define NK_CASE(NN) \
  case NN:          \
    return nk_launch_contig<T, NN, IS_1D>(pa, f, tw, twr, work, st);
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  return nk_set_error(NK_ERR_UNSUPPORTED, "no fast contiguous pass for this length");
}

// pick the compile-time prologue specialisation (first pass only)
template <typename T, int N, int MODE>
static int nk_launch_strided(const NkPassS& ps, const NkFuse& f, const C2<T>* tw, C2<T>* work, C2<T>* scratch, hipStream_t st) {
  if constexpr (MODE == 3) {
    if constexpr (sizeof(T) == 8)  // float excitations under a double pipeline (nk_fuse.io32)
      if (f.field_octant && f.pro == NK_PRO_AMP && f.io32) return nk_launch_strided_pc<T, N, MODE, 9>(ps, f, tw, work, scratch, st);
    if (f.field_octant && f.pro == NK_PRO_AMP) return nk_launch_strided_pc<T, N, MODE, 4>(ps, f, tw, work, scratch, st);
    if (f.field_octant && f.pro == NK_PRO_AMP_JVP) return nk_launch_strided_pc<T, N, MODE, 5>(ps, f, tw, work, scratch, st);
    if (f.pro == NK_PRO_PLAIN) return nk_launch_strided_pc<T, N, MODE, 0>(ps, f, tw, work, scratch, st);
    if (f.pro == NK_PRO_MUL) return nk_launch_strided_pc<T, N, MODE, 6>(ps, f, tw, work, scratch, st);
    if (f.pro == NK_PRO_AMP && f.afield) return nk_launch_strided_pc<T, N, MODE, 1>(ps, f, tw, work, scratch, st);
    if (f.pro == NK_PRO_AMP_JVP && f.afield && f.dafield)
      return nk_launch_strided_pc<T, N, MODE, 3>(ps, f, tw, work, scratch, st);
    if (f.pro == NK_PRO_AMP_JVP && f.afield && f.dampT)
      return nk_launch_strided_pc<T, N, MODE, 2>(ps, f, tw, work, scratch, st);
  }
  return nk_launch_strided_pc<T, N, MODE, -1>(ps, f, tw, work, scratch, st);
}

template <typename T, int MODE>
static int nk_dispatch_strided(int n, const NkPassS& ps, const NkFuse& f, const C2<T>* tw, C2<T>* work, C2<T>* scratch,
                               hipStream_t st) {
  switch (n) {
#define NK_CASE(NN) \
  case NN:          \
    return nk_launch_strided<T, NN, MODE>(ps, f, tw, work, scratch, st);
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  return nk_set_error(NK_ERR_UNSUPPORTED, "no fast strided pass for this length");
}

// ---- sandwich pipeline (nk_fft3.h) ------------------------------------------------------------------------
// NK_CONTIG3_WAVES (experiment): upper bound of the waves per SIMD the compiler should plan for -- the LDS tile limits the
// row kernels to ~5 waves per SIMD anyway, so a register budget of 512 / 5 VGPRs costs no occupancy and lets the scheduler
// keep more of the 64 prologue loads of a thread in flight
#ifndef NK_CONTIG3_WAVES
#define NK_CONTIG3_WAVES 0
#endif
#if NK_CONTIG3_WAVES > 0
#define NK_CONTIG3_ATTR __attribute__((amdgpu_waves_per_eu(1, NK_CONTIG3_WAVES)))
#else
#define NK_CONTIG3_ATTR
#endif
template <typename T, int H, int PC>
__global__ void __launch_bounds__((Contig3Tile<T, H>::THREADS)) NK_CONTIG3_ATTR
    k3_contig(NkPass3 p, NkFuse f, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ twr, C2<T>* __restrict__ work) {
  extern __shared__ __align__(16) unsigned char smem[];
  DeviceExec<T, Contig3Tile<T, H>::SC::E> ex;
  nk_contig3_body<T, H, Contig3Tile<T, H>::TILE, PC>(ex, p, f, blockIdx.x, (T*)smem, tw, twr, work);
}

// QUAD variant: one workgroup per octant row pair of pairs (nk_fft3.h)
// Occupancy the compiler aims at (= its register budget): a row pass lives on the loads it keeps in flight, and with a free
// hand the compiler trades registers for resident waves (k3_contig_quad<double,256,5>: 88 VGPRs, five waves per SIMD --
// 0.69 ms; capped at four waves it takes 144 VGPRs and 0.62 ms).  Per field type and prologue class, from measurements.
#ifndef NK_QUAD_MAXW_F64_5
#define NK_QUAD_MAXW_F64_5 4
#endif
#ifndef NK_QUAD_MAXW_F64_8
#define NK_QUAD_MAXW_F64_8 8
#endif
#ifndef NK_QUAD_MAXW_F32_5
#define NK_QUAD_MAXW_F32_5 8
#endif
#ifndef NK_QUAD_MAXW_F32_8
#define NK_QUAD_MAXW_F32_8 8
#endif
template <typename T, int PC>
constexpr int nk_quad_max_waves() {
  return sizeof(T) == 8 ? (PC == 8 ? NK_QUAD_MAXW_F64_8 : NK_QUAD_MAXW_F64_5) : (PC == 8 ? NK_QUAD_MAXW_F32_8 : NK_QUAD_MAXW_F32_5);
}
template <typename T, int H, int PC>
__global__ void __launch_bounds__((Contig3Tile<T, H>::QTHREADS))
    k3_contig_quad(NkPass3 p, NkFuse f, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ twr, C2<T>* __restrict__ work) {
  extern __shared__ __align__(16) unsigned char smem[];
  DeviceExec<T, Contig3Tile<T, H>::SC::E> ex;
  nk_contig3_body<T, H, 4, PC, true>(ex, p, f, (int64_t)blockIdx.x + p.blk0, (T*)smem, tw, twr, work, (int)blockIdx.y);
}
// the same kernel under an occupancy cap (only the classes that want one are launched through it: the attribute changes the
// compiler's scheduling even when its bound is not binding -- k3_contig_quad<double,256,8> 0.95 -> 1.06 ms under (1, 8))
template <typename T, int H, int PC, int MAXW>
__global__ void __launch_bounds__((Contig3Tile<T, H>::QTHREADS)) __attribute__((amdgpu_waves_per_eu(1, MAXW)))
    k3_contig_quad_w(NkPass3 p, NkFuse f, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ twr, C2<T>* __restrict__ work) {
  extern __shared__ __align__(16) unsigned char smem[];
  DeviceExec<T, Contig3Tile<T, H>::SC::E> ex;
  nk_contig3_body<T, H, 4, PC, true>(ex, p, f, (int64_t)blockIdx.x + p.blk0, (T*)smem, tw, twr, work, (int)blockIdx.y);
}
template <typename T, int H, int PC>
static auto nk_quad_kernel() {
  if constexpr (nk_quad_max_waves<T, PC>() < 8)
    return k3_contig_quad_w<T, H, PC, nk_quad_max_waves<T, PC>()>;
  else
    return k3_contig_quad<T, H, PC>;
}

template <typename T, int H, int PC>
static int nk_launch_contig3(const NkPass3& p3, const NkFuse& f, const C2<T>* tw, const C2<T>* twr, C2<T>* work, hipStream_t st) {
  using CT = Contig3Tile<T, H>;
  if constexpr ((PC == 4 || PC == 5 || PC == 7 || PC == 8) && CT::QUAD_OK) {
    // NK_CONTIG_QUAD: 1 (default) = QUAD workgroups for every launch on a 3-D grid, 2 = for the staged launches of a
    // pipelined sandwich only, 0 = never.  QUAD cuts the FETCH of the JVP class from 15.2 to 10.1 GB per launch at 1024^3
    // fp32 (the octant lines of a[pidx] / da[pidx] are read once per workgroup instead of once per row); on an otherwise idle
    // GPU the pass takes the same time either way (3.59-3.63 vs 3.60-3.61 ms per launch over a bench step, identical bits:
    // the extra fetches were served by L2 / Infinity Cache) -- the smaller footprint on the fabric is what a rank wants
    // while its RCCL exchange runs beside the pass.
    static const int quad = nk_env_int("NK_CONTIG_QUAD", 1);
    if ((quad == 1 || (quad == 2 && p3.nblk > 0)) && p3.g.ndim == 3) {
      auto qkern = nk_quad_kernel<T, H, PC>();
      static unsigned long long qattr_mask = 0;
      if (CT::QLDS_BYTES > 64 * 1024 && nk_first_on_device(qattr_mask)) {
        hipError_t e = hipFuncSetAttribute((const void*)qkern, hipFuncAttributeMaxDynamicSharedMemorySize, CT::QLDS_BYTES);
        if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(k3_contig_quad)");
      }
      const int64_t batch = p3.nlines / ((int64_t)p3.g.na * p3.g.nm);
      int64_t qblocks = batch * (p3.g.na / 2 + 1) * (p3.g.nm / 2 + 1);
      if (p3.nblk > 0) {  // one stage of a pipelined sandwich
        if (batch != 1 || p3.blk0 < 0 || p3.blk0 + p3.nblk > qblocks) return nk_set_error(NK_ERR_INVALID, "k3_contig_quad: bad stage range");
        qblocks = p3.nblk;
      }
      // grid: x = the (a8, b8) index inside a batch member (stage ranges: batch 1), y = the batch member
      const int64_t per = (int64_t)(p3.g.na / 2 + 1) * (p3.g.nm / 2 + 1);
      const int64_t gx = p3.nblk > 0 ? qblocks : per;
      if (gx > 0x7fffffffLL || batch > 65535) return nk_set_error(NK_ERR_UNSUPPORTED, "too many lines for one launch");
      NkPass3 pq = p3;
      pq.dmh = nk_make_div(p3.g.nm / 2 + 1);
      hipLaunchKernelGGL(qkern, dim3((unsigned)gx, (unsigned)(p3.nblk > 0 ? 1 : batch)), dim3(CT::QTHREADS), CT::QLDS_BYTES, st, pq, f,
                         tw, twr, work);
      return nk_check_launch("k3_contig_quad");
    }
  }
  if (p3.nblk > 0)
    return nk_set_error(NK_ERR_UNSUPPORTED, "nk_fuse.pipe_chunks: the staged first pass exists for the QUAD launches only (3-D plan, "
                                             "octant prologue classes, last axis <= 2048 fp32, NK_CONTIG_QUAD != 0)");
  auto kern = k3_contig<T, H, PC>;
  static unsigned long long attr_mask = 0;  // per-device attribute
  if (CT::LDS_BYTES > 64 * 1024 && nk_first_on_device(attr_mask)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CT::LDS_BYTES);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(k3_contig)");
  }
  const int64_t blocks = (p3.nlines + CT::TILE - 1) / CT::TILE;
  if (blocks > 0x7fffffffLL) return nk_set_error(NK_ERR_UNSUPPORTED, "too many lines for one launch");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(CT::THREADS), CT::LDS_BYTES, st, p3, f, tw, twr, work);
  return nk_check_launch("k3_contig");
}

template <typename T, int H>
static int nk_launch_contig3_pc(const NkPass3& p3, const NkFuse& f, const C2<T>* tw, const C2<T>* twr, C2<T>* work, hipStream_t st) {
  if (f.field_octant && f.pro == NK_PRO_AMP) return nk_launch_contig3<T, H, 4>(p3, f, tw, twr, work, st);
  if (f.field_octant && f.pro == NK_PRO_AMP_JVP && f.cg_r && f.dafield) return nk_launch_contig3<T, H, 8>(p3, f, tw, twr, work, st);
  if (f.field_octant && f.pro == NK_PRO_AMP_JVP && f.pidx_octant && f.dampT) return nk_launch_contig3<T, H, 7>(p3, f, tw, twr, work, st);
  if (f.field_octant && f.pro == NK_PRO_AMP_JVP) return nk_launch_contig3<T, H, 5>(p3, f, tw, twr, work, st);
  if (f.pro == NK_PRO_PLAIN) return nk_launch_contig3<T, H, 0>(p3, f, tw, twr, work, st);
  if (f.pro == NK_PRO_MUL) return nk_launch_contig3<T, H, 6>(p3, f, tw, twr, work, st);
  return nk_launch_contig3<T, H, -1>(p3, f, tw, twr, work, st);
}

template <typename T>
static int nk_dispatch_contig3(int h, const NkPass3& p3, const NkFuse& f, const C2<T>* tw, const C2<T>* twr, C2<T>* work,
                               hipStream_t st) {
  switch (h) {
#define NK_CASE(NN) \
  case NN:          \
    return nk_launch_contig3_pc<T, NN>(p3, f, tw, twr, work, st);
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  return nk_set_error(NK_ERR_UNSUPPORTED, "no fast contiguous pass for this length");
}

#ifndef NK_MID_CX
#define NK_MID_CX 1
#endif
#ifndef NK_MID_WAVES
#define NK_MID_WAVES 0  // > 0: min waves per SIMD the fused middle kernel must allow (register cap)
#endif
#ifndef NK_MID_PF
#define NK_MID_PF 0   // persistent workgroups that prefetch the next tile into registers
#endif
#ifndef NK_MID_WIDE_XM
#define NK_MID_WIDE_XM 2
#endif
#ifndef NK_MID_WIDE
#define NK_MID_WIDE 0  // 1: wide schedule (SchedW, two workgroups per CU) for the fused middle kernel -- spills 116..464 B per lane, 3.1 -> 5.0 ms
#endif
// tile / schedule / exchange mode of the fused middle kernel.
// TWO: the fp32 512-thread x 32-element configuration (1024-point lines) with a constant diagonal runs TWO workgroups per
// CU: split exchange (64 KiB of LDS), composed twiddles and a 128-VGPR cap (32 B / lane of spills).  One workgroup per CU
// serialises the ~11 us of VALU work of the two line transforms with the ~12 us its tile's HBM traffic takes; two
// independent ones overlap them: 3.07 -> 2.24 ms at 1024^3 fp32 (gpurun_out/r02r_probe.log).
#ifndef NK_MID_TWO
#define NK_MID_TWO 1
#endif
#ifndef NK_MID_TWO_MF
#define NK_MID_TWO_MF 0  // the two-workgroup configuration also for a field diagonal
#endif
template <typename T, int N, bool MF>
struct MidCfg {
  static constexpr bool WIDE = NK_MID_WIDE && SchedW<T, N>::E == 64;
  static constexpr bool TWO = NK_MID_TWO && !WIDE && (!MF || NK_MID_TWO_MF) && sizeof(T) == 4 && Sched<T, N>::E == 32 &&
                              StridedTile<T, N, false, 3>::THREADS == 512 && StridedTile<T, N, false, 3>::LDS_BYTES <= 64 * 1024;
  using ST = StridedTile<T, N, !WIDE && !TWO && NK_MID_CX != 0, WIDE ? 0 : 3>;
  using SC = typename ST::SC;
  static constexpr int XM = WIDE ? NK_MID_WIDE_XM : (ST::CPLX ? 1 : 0);
  static constexpr bool TWC = TWO;
  static constexpr int WAVES = NK_MID_WAVES > 0 ? NK_MID_WAVES : TWO ? 4 : ST::THREADS <= 256 ? 2 : 1;
};
template <typename T, int N, bool MF>
__global__ void __launch_bounds__((MidCfg<T, N, MF>::ST::THREADS), (MidCfg<T, N, MF>::WAVES))
    k3_mid(NkPassM pm, NkFuse f, const C2<T>* __restrict__ tw, C2<T>* __restrict__ work, int64_t nblk, int xmap) {
  extern __shared__ __align__(16) unsigned char smem[];
  using CF = MidCfg<T, N, MF>;
  using ST = typename CF::ST;
  DeviceExecR<MidRegs<T, CF::SC::E, NK_MID_PF != 0>> ex;
  C2<T>* tw_lds = ST::TWLDS ? reinterpret_cast<C2<T>*>(smem + ST::LDS_BYTES) : nullptr;
  nk_mid_body<T, N, ST::TILE, CF::XM, MF, NK_MID_PF != 0, typename CF::SC, CF::TWC>(ex, pm, f, (int64_t)blockIdx.x, (int64_t)gridDim.x,
                                                                                     nblk, xmap, (T*)smem, tw, work, tw_lds);
}

template <typename T, int N, bool MF>
static int nk_launch_mid(NkPassM pm, const NkFuse& f, const C2<T>* tw, C2<T>* work, hipStream_t st) {
  using ST = typename MidCfg<T, N, MF>::ST;
  auto kern = k3_mid<T, N, MF>;
  static unsigned long long attr_mask = 0;  // per-device attribute
  if (ST::LDS_TOTAL > 64 * 1024 && nk_first_on_device(attr_mask)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ST::LDS_TOTAL);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(k3_mid)");
  }
  pm.s.tl.tile = ST::TILE;
  pm.s.tl.dtile = nk_make_div(ST::TILE);
  pm.s.tiles_per_slab = (int)(pm.s.inner / ST::TILE);
  const int64_t blocks = pm.s.outer * pm.s.tiles_per_slab;
  static const int xmap_env = nk_env_int("NK_XMAP", NK_XMAP_DEFAULT);
  // persistent: as many workgroups as the device keeps resident (LDS decides: one or two per CU)
  int64_t grid = blocks;
  if (NK_MID_PF) {
    static int cus = 0;
    if (cus == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    static const int per_cu_env = nk_env_int("NK_MID_WG_PER_CU", 0);
    const int per_cu = per_cu_env > 0 ? per_cu_env : (ST::LDS_TOTAL <= 80 * 1024 && ST::THREADS <= 512 ? 2 : 1);
    if (grid > (int64_t)cus * per_cu) grid = (int64_t)cus * per_cu;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(ST::THREADS), ST::LDS_TOTAL, st, pm, f, tw, work, blocks, xmap_env & 2);
  return nk_check_launch("k3_mid");
}

template <typename T>
static int nk_dispatch_mid(int n, const NkPassM& pm, const NkFuse& f, const C2<T>* tw, C2<T>* work, hipStream_t st) {
  switch (n) {
#define NK_CASE(NN) \
  case NN:          \
    return f.mul ? nk_launch_mid<T, NN, true>(pm, f, tw, work, st) : nk_launch_mid<T, NN, false>(pm, f, tw, work, st);
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  return nk_set_error(NK_ERR_UNSUPPORTED, "no fast strided pass for this length");
}

static bool nk_fast_enabled() {
  static int v = -1;
  if (v < 0) v = nk_env_int("NK_FAST", 1) ? 1 : 0;
  return v == 1;
}

// ---- c2c passes -----------------------------------------------------------------------------------
// contiguous-axis c2c pass: lines of n complex, `swap` exchanges re/im on load and store (inverse
// transform through the forward kernel), result scaled.
struct NkPassCC {
  NkLinePlan lp;
  NkTile tl;
  int64_t nlines;
  int swap;
  double scale;
  NkDiv dn;  // / lp.n
};

template <typename T>
__global__ void __launch_bounds__(nk_gen_max_threads<T>()) k_c2c_contig(NkPassCC p, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ in,
                             C2<T>* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  C2<T>* lds = (C2<T>*)smem;
  const int n = p.lp.n, tile = p.tl.tile;
  const int64_t line0 = (int64_t)blockIdx.x * tile;
  for (int idx = threadIdx.x; idx < tile * n; idx += blockDim.x) {
    int j, t;
    nk_fdivmod((uint32_t)idx, p.dn, t, j);
    C2<T> z{(T)0, (T)0};
    if (line0 + t < p.nlines) z = in[(line0 + t) * n + j];
    if (p.swap) z = C2<T>{z.y, z.x};
    lds[nk_lds_addr(p.tl, j, t)] = z;
  }
  __syncthreads();
  nk_run_stages<T>(lds, threadIdx.x, blockDim.x, p.lp, p.tl, tw);
  const T sc = (T)p.scale;
  for (int idx = threadIdx.x; idx < tile * n; idx += blockDim.x) {
    int k, t;
    nk_fdivmod((uint32_t)idx, p.dn, t, k);
    if (line0 + t >= p.nlines) continue;
    C2<T> z = lds[nk_lds_addr(p.tl, nk_digit_reverse(p.lp, k), t)];
    if (p.swap) z = C2<T>{z.y, z.x};
    out[(line0 + t) * n + k] = C2<T>{z.x * sc, z.y * sc};
  }
}

// strided c2c pass in place on `data` viewed as [outer][n][inner]
template <typename T>
__global__ void __launch_bounds__(nk_gen_max_threads<T>()) k_c2c_strided(NkPassS p, int swap, const C2<T>* __restrict__ tw, C2<T>* __restrict__ data) {
  extern __shared__ __align__(16) unsigned char smem[];
  C2<T>* lds = (C2<T>*)smem;
  const int n = p.lp.n, tile = p.tl.tile;
  const int64_t o = blockIdx.x / p.tiles_per_slab;
  const int64_t c0 = (blockIdx.x % p.tiles_per_slab) * (int64_t)tile;
  C2<T>* base = data + o * n * p.inner + c0;
  const int valid = nk_tile_columns(p, c0);
  for (int idx = threadIdx.x; idx < tile * n; idx += blockDim.x) {
    int t, j;
    nk_fdivmod((uint32_t)idx, p.tl.dtile, j, t);
    C2<T> z = t < valid ? base[(int64_t)j * p.inner + t] : C2<T>{(T)0, (T)0};
    if (swap) z = C2<T>{z.y, z.x};
    lds[nk_lds_addr(p.tl, j, t)] = z;
  }
  __syncthreads();
  nk_run_stages<T>(lds, threadIdx.x, blockDim.x, p.lp, p.tl, tw);
  for (int idx = threadIdx.x; idx < tile * n; idx += blockDim.x) {
    int t, k;
    nk_fdivmod((uint32_t)idx, p.tl.dtile, k, t);
    C2<T> z = lds[nk_lds_addr(p.tl, nk_digit_reverse(p.lp, k), t)];
    if (swap) z = C2<T>{z.y, z.x};
    if (t < valid) base[(int64_t)k * p.inner + t] = z;
  }
}

// ------------------------------------------------------------------------------------------------
// plan object
// ------------------------------------------------------------------------------------------------
struct nk_plan {
  NkHostPlan hp;
  int64_t shape[3];
  int ndim;
  int64_t batch;
  void* d_tw_a = nullptr;   // tw (h) for pass A
  void* d_twr_a = nullptr;  // untangle twiddles
  void* d_tw_b = nullptr;
  void* d_tw_c = nullptr;
  void* d_tw_f = nullptr;   // full-length table of the last axis (final pass of the strided-first pipeline)
  void* d_tw_t64 = nullptr, *d_tw_t32 = nullptr;  // sub-line tables of the two-level first-axis pass (2-D plans)
  // c2c: full-length contiguous-axis plan
  NkPassCC cc{};
  int threads_cc = 256;
  size_t lds_cc = 0;
  void* d_tw_cc = nullptr;
  NkPassS c2c_mid{}, c2c_first{};
  int threads_cm = 256, threads_cf = 256;
  size_t lds_cm = 0, lds_cf = 0;
};

static int nk_upload_twiddle(void** dptr, const std::vector<double>& tw, int dtype) {
  const size_t n = tw.size();
  if (n == 0) {
    *dptr = nullptr;
    return NK_OK;
  }
  hipError_t e;
  if (dtype == NK_F64) {
    e = hipMalloc(dptr, n * sizeof(double));
    if (e != hipSuccess) return nk_set_hip_error(e, "hipMalloc(twiddle)");
    e = hipMemcpy(*dptr, tw.data(), n * sizeof(double), hipMemcpyHostToDevice);
  } else {
    std::vector<float> f(n);
    for (size_t i = 0; i < n; ++i) f[i] = (float)tw[i];
    e = hipMalloc(dptr, n * sizeof(float));
    if (e != hipSuccess) return nk_set_hip_error(e, "hipMalloc(twiddle)");
    e = hipMemcpy(*dptr, f.data(), n * sizeof(float), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) return nk_set_hip_error(e, "hipMemcpy(twiddle)");
  return NK_OK;
}

template <typename K>
static int nk_allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(max dynamic LDS)");
  }
  return NK_OK;
}

extern "C" int nk_plan_create(nk_plan** out, int ndim, const int64_t* shape, int dtype, int64_t batch) {
  if (!out || !shape) return nk_set_error(NK_ERR_INVALID, "nk_plan_create: null argument");
  nk_plan* P = new (std::nothrow) nk_plan();
  if (!P) return nk_set_error(NK_ERR_NOMEM, "nk_plan_create: out of host memory");
  const char* msg = "";
  int rc = nk_host_plan_init(P->hp, ndim, shape, dtype, batch, &msg);
  if (rc != NK_OK) {
    delete P;
    return nk_set_error(rc, msg);
  }
  P->ndim = ndim;
  P->batch = batch;
  for (int d = 0; d < 3; ++d) P->shape[d] = d < ndim ? shape[d] : 1;
  NkHostPlan& hp = P->hp;
  rc = nk_upload_twiddle(&P->d_tw_a, hp.tw_a, dtype);
  if (rc == NK_OK) rc = nk_upload_twiddle(&P->d_twr_a, hp.twr_a, dtype);
  if (rc == NK_OK && ndim == 3) rc = nk_upload_twiddle(&P->d_tw_b, hp.tw_b, dtype);
  if (rc == NK_OK && ndim >= 2) rc = nk_upload_twiddle(&P->d_tw_c, hp.tw_c, dtype);
  if (rc == NK_OK && ndim >= 2) rc = nk_upload_twiddle(&P->d_tw_f, hp.tw_f, dtype);
  if (rc == NK_OK && ndim == 2) rc = nk_upload_twiddle(&P->d_tw_t64, hp.tw_t64, dtype);
  if (rc == NK_OK && ndim == 2) rc = nk_upload_twiddle(&P->d_tw_t32, hp.tw_t32, dtype);
  // c2c plan pieces: contiguous last axis of full length nl, strided middle/first axes with inner = nl
  if (rc == NK_OK) {
    const NkGeom& g = hp.g;
    const size_t cs = hp.csize;
    if ((size_t)(g.nl + g.nl / 16 + 1) * cs > 144 * 1024) {
      P->cc.lp.n = 0;  // c2c unsupported for this length (Hartley still fine)
    } else {
      P->cc.lp = nk_make_line_plan(g.nl);
      P->cc.dn = nk_make_div(g.nl);
      const int lstride = g.nl + g.nl / 16 + 1;
      const size_t line_bytes = (size_t)lstride * cs;
      int64_t tile = (int64_t)(32 * 1024 / line_bytes);
      const int64_t want = (2048 + g.nl - 1) / g.nl;
      if (tile > want) tile = want;
      if (tile < 1) tile = 1;
      P->cc.nlines = batch * g.na * g.nm;
      if (tile > P->cc.nlines) tile = P->cc.nlines;
      P->cc.tl.tile = (int)tile;
      P->cc.tl.dtile = nk_make_div((int)tile);
      P->cc.tl.t_fastest = 0;
      P->cc.tl.lstride = lstride;
      P->lds_cc = (size_t)tile * line_bytes;
      P->threads_cc = nk_round_threads(tile * g.nl / 4);
      std::vector<double> tw;
      nk_fill_twiddle(tw, g.nl, g.nl);
      rc = nk_upload_twiddle(&P->d_tw_cc, tw, dtype);
      auto setup = [&](NkPassS& ps, int n, int64_t outer, int64_t inner, int& threads, size_t& lds) {
        ps.g = g;
        ps.lp = nk_make_line_plan(n);
        ps.outer = outer;
        ps.inner = inner;
        const int T = nk_pick_strided_tile(n, inner, cs, "NK_TILE_C2C", outer);
        ps.tl.tile = T;
        ps.tl.dtile = nk_make_div(T);
        ps.tl.t_fastest = 1;
        ps.tl.tstride = T;
        ps.tiles_per_slab = (int)((inner + T - 1) / T);
        lds = (size_t)n * T * cs;
        threads = nk_round_threads((int64_t)n * T / 4);
      };
      if (ndim == 3) setup(P->c2c_mid, g.nm, batch * g.na, g.nl, P->threads_cm, P->lds_cm);
      if (ndim >= 2) setup(P->c2c_first, g.na, batch, (int64_t)g.nm * g.nl, P->threads_cf, P->lds_cf);
    }
  }
  // opt in to > 64 KiB dynamic LDS where the tiles need it
  if (rc == NK_OK) {
    if (dtype == NK_F32) {
      rc = nk_allow_lds(k_passA<float>, hp.lds_a);
      if (rc == NK_OK) rc = nk_allow_lds(k_pass1d<float>, hp.lds_a);
      if (rc == NK_OK) rc = nk_allow_lds(k_passB<float>, hp.lds_b);
      if (rc == NK_OK) rc = nk_allow_lds(k_passC<float>, hp.lds_c);
      if (rc == NK_OK) rc = nk_allow_lds(k_c2c_contig<float>, P->lds_cc);
      if (rc == NK_OK) rc = nk_allow_lds(k_c2c_strided<float>, P->lds_cm > P->lds_cf ? P->lds_cm : P->lds_cf);
    } else {
      rc = nk_allow_lds(k_passA<double>, hp.lds_a);
      if (rc == NK_OK) rc = nk_allow_lds(k_pass1d<double>, hp.lds_a);
      if (rc == NK_OK) rc = nk_allow_lds(k_passB<double>, hp.lds_b);
      if (rc == NK_OK) rc = nk_allow_lds(k_passC<double>, hp.lds_c);
      if (rc == NK_OK) rc = nk_allow_lds(k_c2c_contig<double>, P->lds_cc);
      if (rc == NK_OK) rc = nk_allow_lds(k_c2c_strided<double>, P->lds_cm > P->lds_cf ? P->lds_cm : P->lds_cf);
    }
  }
  if (rc != NK_OK) {
    nk_plan_destroy(P);
    return rc;
  }
  *out = P;
  return NK_OK;
}

extern "C" int nk_plan_destroy(nk_plan* P) {
  if (!P) return NK_OK;
  void* ptrs[] = {P->d_tw_a, P->d_twr_a, P->d_tw_b, P->d_tw_c, P->d_tw_cc, P->d_tw_f, P->d_tw_t64, P->d_tw_t32};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete P;
  return NK_OK;
}

static bool nk_plan_uses_pipeline2(const nk_plan* P) {
  const NkHostPlan& hp = P->hp;
  const bool f32 = hp.dtype == NK_F32;
  if (!nk_fast_enabled() || nk_env_int("NK_PIPELINE", 2) != 2 || hp.g.ndim < 2 || !nk_fast_size(hp.g.nl)) return false;
  const bool first_ok = f32 ? nk_fast_strided_ok<float>(hp.g.na, hp.pc.inner) : nk_fast_strided_ok<double>(hp.g.na, hp.pc.inner);
  const bool mid_ok = hp.g.ndim == 2 || (f32 ? nk_fast_strided_ok<float>(hp.g.nm, hp.pb.inner)
                                             : nk_fast_strided_ok<double>(hp.g.nm, hp.pb.inner));
  return first_ok && mid_ok;
}

// energy / curvature sums of the final pass: ONE slot per wavefront at the end of the workspace, zeroed before the
// launch, then folded in a fixed order (k_fold_slots_a: 256 workgroups over contiguous ranges; k_fold_slots_b: their
// partials, fixed tree) -- the same bits on every run.  (Until round 2: atomics on 256 slots, order-dependent in the last
// bit, which the energy-based stopping rules turned into different iteration counts from run to run.)
#define NK_FOLD_BLOCKS 256
static inline int64_t nk_value_slot_count(const NkHostPlan& hp) {
  // one slot per WAVEFRONT of the final pass: P = nl / 16 threads per line (SchedF), whole wavefronts per workgroup, a
  // workgroup owns >= 1 line: <= lines * max(1, nl / 1024) wavefronts on a full tiling.  The couples of the 3-D pass tile
  // (A/2 + 1) x (M/2 + 1) line pairs instead of A x M / 4 (x (1 + 2/A)(1 + 2/M) <= 1.07 from 64^2 on, at one pair per
  // workgroup that is 0.53 lines), odd tile counts round up: 9/8 of the lines + 512 covers every launcher
  const int64_t lines = (int64_t)hp.g.batch * hp.g.na * hp.g.nm;
  const int64_t per_line = hp.g.nl > 1024 ? hp.g.nl / 1024 : 1;
  int64_t n = std::max<int64_t>(lines * per_line * 9 / 8 + 512, NK_FOLD_BLOCKS);
  // the generic LDS kernels (mixed-radix grids) own one slot per wavefront as well: pass C / pass 1-D with up to 16
  // wavefronts per workgroup, pass D with 4
  const int64_t tile_a = hp.pa.tl.tile > 0 ? hp.pa.tl.tile : 1;
  const int64_t gen = hp.pc.outer * hp.pc.tiles_per_slab * 16 + (lines + 255) / 256 * 4 + (hp.pa.nlines + tile_a - 1) / tile_a * 16;
  n = std::max(n, gen + 512);
  return (n + 255) / 256 * 256;
}
__device__ __forceinline__ double nk_fold_block_sum(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}
// maximum of non-negative values that propagates NaN (fmax would drop it)
__device__ __forceinline__ double nk_max_nan(double a, double b) { return (a != a || b != b) ? (a != a ? a : b) : fmax(a, b); }
__device__ __forceinline__ double nk_fold_block_max(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v = nk_max_nan(v, __shfl_down(v, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s = nk_max_nan(s, red[w]);
  return s;
}
__global__ void __launch_bounds__(256) k_fold_max_a(const double* __restrict__ slots, int64_t n, double* __restrict__ part) {
  __shared__ double red[4];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double v = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) v = nk_max_nan(v, slots[i]);
  const double s = nk_fold_block_max(v, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void __launch_bounds__(NK_FOLD_BLOCKS) k_fold_max_b(const double* __restrict__ part, double* __restrict__ value) {
  __shared__ double red[NK_FOLD_BLOCKS / 64];
  const double s = nk_fold_block_max(part[threadIdx.x], red);
  if (threadIdx.x == 0) *value = s;
}
__global__ void __launch_bounds__(256) k_fold_slots_a(const double* __restrict__ slots, int64_t n, double* __restrict__ part) {
  __shared__ double red[4];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double v = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) v += slots[i];
  const double s = nk_fold_block_sum(v, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void __launch_bounds__(NK_FOLD_BLOCKS) k_fold_slots_b(const double* __restrict__ part, double* __restrict__ value) {
  __shared__ double red[NK_FOLD_BLOCKS / 64];
  const double s = nk_fold_block_sum(part[threadIdx.x], red);
  if (threadIdx.x == 0) *value += s;
}
// both folds in one pair of launches (the scatter epilogue of a CG iteration's last sample wants the curvature sum AND
// max |w8|): workgroups [0, NK_FOLD_BLOCKS) sum the value slots, the rest take the maximum of the |w8| slots
__global__ void __launch_bounds__(256) k_fold_both_a(const double* __restrict__ vslots, const double* __restrict__ wslots, int64_t n,
                                                     double* __restrict__ vpart, double* __restrict__ wpart) {
  __shared__ double red[4];
  const bool is_max = blockIdx.x >= NK_FOLD_BLOCKS;
  const int b = is_max ? blockIdx.x - NK_FOLD_BLOCKS : blockIdx.x;
  const double* slots = is_max ? wslots : vslots;
  const int64_t per = (n + NK_FOLD_BLOCKS - 1) / NK_FOLD_BLOCKS;
  const int64_t lo = (int64_t)b * per, hi = lo + per < n ? lo + per : n;
  double v = 0.0;
  if (is_max) {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) v = nk_max_nan(v, slots[i]);
    const double s = nk_fold_block_max(v, red);
    if (threadIdx.x == 0) wpart[b] = s;
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) v += slots[i];
    const double s = nk_fold_block_sum(v, red);
    if (threadIdx.x == 0) vpart[b] = s;
  }
}
__global__ void __launch_bounds__(NK_FOLD_BLOCKS) k_fold_both_b(const double* __restrict__ vpart, const double* __restrict__ wpart,
                                                                double* __restrict__ value, double* __restrict__ wmax) {
  __shared__ double red[NK_FOLD_BLOCKS / 64];
  if (blockIdx.x == 0) {
    const double s = nk_fold_block_sum(vpart[threadIdx.x], red);
    if (threadIdx.x == 0) *value += s;
  } else {
    const double s = nk_fold_block_max(wpart[threadIdx.x], red);
    if (threadIdx.x == 0) *wmax = s;
  }
}
// the slot area of a workspace: [value slots: nk_value_slot_count][partials: NK_FOLD_BLOCKS][|w8| slots][partials]
static inline double* nk_value_slots(const NkHostPlan& hp, void* workspace) {
  return (double*)((char*)workspace + (hp.work_bytes + 255) / 256 * 256 + (hp.scratch_bytes + 255) / 256 * 256 + 256);
}
static inline double* nk_wmax_slots(const NkHostPlan& hp, void* workspace) {
  return nk_value_slots(hp, workspace) + nk_value_slot_count(hp) + NK_FOLD_BLOCKS;
}
static int nk_fold_value_slots(const NkHostPlan& hp, double* slots, double* value, hipStream_t st) {
  const int64_t n = nk_value_slot_count(hp);
  double* part = slots + n;
  hipLaunchKernelGGL(k_fold_slots_a, dim3(NK_FOLD_BLOCKS), dim3(256), 0, st, slots, n, part);
  hipLaunchKernelGGL(k_fold_slots_b, dim3(1), dim3(NK_FOLD_BLOCKS), 0, st, part, value);
  return nk_check_launch("k_fold_slots");
}
// final-pass launch with the per-workgroup slots the fuse record asks for (energy / curvature sum, max |w8|): zero them,
// run `launch(f2)`, fold them in a fixed order
template <typename Launch>
static int nk_final_with_slots(const NkHostPlan& hp, void* workspace, const NkFuse& f, hipStream_t st, Launch&& launch) {
  const bool want_value = f.value && (f.epi == NK_EPI_LIKELIHOOD || f.epi == NK_EPI_VJP);
  const bool want_wmax = f.w8max && f.w8 && f.epi == NK_EPI_VJP;
  if (want_wmax && hp.g.ndim != 3)
    return nk_set_error(NK_ERR_UNSUPPORTED, "nk_fuse.w8max: only the 3-D final pass reports max |w8| (it feeds nk_octant_scatter_k2)");
  if (!want_value && !want_wmax) {
    NkFuse f0 = f;
    f0.w8max = nullptr;
    return launch(f0);
  }
  const int64_t n = nk_value_slot_count(hp);
  NkFuse f2 = f;
  f2.value_slots = (int)std::min<int64_t>(n, 0x7fffffff);
  double* vslots = nk_value_slots(hp, workspace);
  double* wslots = nk_wmax_slots(hp, workspace);
  const bool wmax_on = want_wmax;
  {  // the two areas are adjacent: one memset covers whatever is wanted
    double* lo = want_value ? vslots : wslots;
    double* hi = wmax_on ? wslots + n : vslots + n;
    hipError_t e = hipMemsetAsync(lo, 0, (size_t)(hi - lo) * sizeof(double), st);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipMemsetAsync(reduction slots)");
  }
  f2.value = want_value ? vslots : nullptr;
  f2.w8max = wmax_on ? wslots : nullptr;
  int rc = launch(f2);
  if (rc != NK_OK) return rc;
  if (want_value && wmax_on) {
    hipLaunchKernelGGL(k_fold_both_a, dim3(2 * NK_FOLD_BLOCKS), dim3(256), 0, st, vslots, wslots, n, vslots + n, wslots + n);
    hipLaunchKernelGGL(k_fold_both_b, dim3(2), dim3(NK_FOLD_BLOCKS), 0, st, vslots + n, wslots + n, f.value, f.w8max);
    return nk_check_launch("k_fold_both");
  }
  if (want_value) return nk_fold_value_slots(hp, vslots, f.value, st);
  hipLaunchKernelGGL(k_fold_max_a, dim3(NK_FOLD_BLOCKS), dim3(256), 0, st, wslots, n, wslots + n);
  hipLaunchKernelGGL(k_fold_max_b, dim3(1), dim3(NK_FOLD_BLOCKS), 0, st, wslots + n, f.w8max);
  return nk_check_launch("k_fold_max");
}

extern "C" int nk_plan_octant_vjp(const nk_plan* P) { return P && nk_plan_uses_pipeline2(P) ? 1 : 0; }

extern "C" size_t nk_plan_workspace_bytes(const nk_plan* P) {
  if (!P) return 0;
  // [work | scratch], scratch aligned to 256 B
  size_t w = (P->hp.work_bytes + 255) / 256 * 256;
  return w + (P->hp.scratch_bytes + 255) / 256 * 256 + 256 + 2 * (nk_value_slot_count(P->hp) + NK_FOLD_BLOCKS) * sizeof(double);
}

// ------------------------------------------------------------------------------------------------
// execution
// ------------------------------------------------------------------------------------------------
template <typename T>
static int nk_run_hartley(const nk_plan* P, const NkFuse& f, int convention, void* workspace, hipStream_t st) {
  const NkHostPlan& hp = P->hp;
  NkPassA pa = hp.pa;
  pa.g.sign = convention == NK_HARTLEY_CANONICAL ? -1 : 1;
  const C2<T>* tw_a = (const C2<T>*)P->d_tw_a;
  const C2<T>* twr = (const C2<T>*)P->d_twr_a;
  const int64_t blocks_a = (pa.nlines + pa.tl.tile - 1) / pa.tl.tile;
  if (blocks_a > 0x7fffffffLL) return nk_set_error(NK_ERR_UNSUPPORTED, "too many lines for one launch");
  const bool fast = nk_fast_enabled();
  // Energy / curvature sums of the kernels below (everything but the strided-first pipeline, which brings its own slot
  // handling): one slot per WAVEFRONT in the workspace's slot area, folded in a fixed order afterwards -- bit-reproducible
  // like the fast path (round 3; until then one fp64 atomic per workgroup).  Without a workspace (1-D calls may omit it) or
  // if the area were too small the atomics remain.
  const bool want_value = f.value && (f.epi == NK_EPI_LIKELIHOOD || f.epi == NK_EPI_VJP);
  const int64_t slot_cap = nk_value_slot_count(hp);
  double* vslots = workspace ? nk_value_slots(hp, workspace) : nullptr;
  int64_t slot_used = 0;
  bool slots_on = want_value && vslots != nullptr;
  auto with_slots = [&](int64_t blocks, int threads) {
    NkFuse g = f;
    if (!slots_on) return g;
    const int64_t need = blocks * ((threads + 63) / 64);
    if (slot_used + need > slot_cap) {
      slots_on = false;
      return g;
    }
    g.value = vslots + slot_used;
    g.value_slots = (int)need;
    slot_used += need;
    return g;
  };
  auto fold_slots = [&]() -> int {
    if (slot_used == 0) return NK_OK;
    return nk_fold_value_slots(hp, vslots, f.value, st);
  };
  if (hp.g.ndim == 1) {
    ProfScope ps(st, 0, f.pro, f.epi);
    if (fast && nk_fast_contig_ok(hp.g.h))
      return nk_dispatch_contig<T, true>(hp.g.h, pa, f, tw_a, twr, (C2<T>*)nullptr, st);
    if (slots_on) {
      hipError_t e = hipMemsetAsync(vslots, 0, (size_t)slot_cap * sizeof(double), st);
      if (e != hipSuccess) return nk_set_hip_error(e, "hipMemsetAsync(reduction slots)");
    }
    const int thr = nk_gen_threads<T>(hp.threads_a);
    const NkFuse f1 = with_slots(blocks_a, thr);
    hipLaunchKernelGGL(k_pass1d<T>, dim3((unsigned)blocks_a), dim3(thr), hp.lds_a, st, pa, f1, tw_a, twr);
    const int rc1 = nk_check_launch("k_pass1d");
    return rc1 != NK_OK ? rc1 : fold_slots();
  }
  if (!workspace) return nk_set_error(NK_ERR_INVALID, "nk_hartley: workspace required for ndim >= 2");
  C2<T>* work = (C2<T>*)workspace;
  C2<T>* scratch = (C2<T>*)((char*)workspace + (hp.work_bytes + 255) / 256 * 256);
  int rc;
  // ---- strided-first pipeline (default when every axis has a specialised kernel): strided c2c passes on the
  //      real array viewed as complex pairs, then ONE contiguous final pass per line pair (k, -k)
  static const int pipeline = nk_env_int("NK_PIPELINE", 2);
  if (fast && pipeline == 2 && nk_fast_size(hp.g.nl) && nk_fast_strided_ok<T>(hp.g.na, hp.pc.inner) &&
      (hp.g.ndim == 2 || nk_fast_strided_ok<T>(hp.g.nm, hp.pb.inner))) {
    // 3-D work array: natural [batch][first] slabs whose stride is padded by NK_WORK_PAD elements -- the in-place pass
    // over the first axis otherwise walks an exact power-of-two stride (nm*nl/2 elements: HBM channel aliasing,
    // 2.9 -> 2.05 ms at 1024^3 fp32).  NK_WORK_BLO=1 selects the transposed slab order [batch][mid][first][last/2]
    // instead (second pass at stride nl/2, first-pass stores at the big stride): measured slower in total.
    static const int work_blo = nk_env_int("NK_WORK_BLO", 0), work_pad = nk_env_int("NK_WORK_PAD", 2080);
    NkPipe2 q = nk_pipe2_setup(hp, pa.g.sign, work_blo, work_pad);
    // per-thread address parts are 32-bit: (threads per line) * (row stride) must stay below 2^31 elements
    {
      const int64_t smax = q.s1.ss > q.s0.inner ? q.s1.ss : q.s0.inner;
      // ... and the per-thread BYTE offsets (nk_at32) below 2^32
      if (128 * (smax > q.s1.inner ? smax : q.s1.inner) * (int64_t)sizeof(C2<T>) >= ((int64_t)1 << 32))
        return nk_set_error(NK_ERR_UNSUPPORTED, "transform too large for the 32-bit thread offsets of the strided passes");
    }
    if (hp.g.ndim == 3) {
      {
        ProfScope ps(st, 1, f.pro, f.epi);
        rc = nk_dispatch_strided<T, 3>(hp.g.nm, q.s1, f, (const C2<T>*)P->d_tw_b, work, scratch, st);
      }
      if (rc != NK_OK) return rc;
      ProfScope ps(st, 2, f.pro, f.epi);
      rc = nk_dispatch_strided<T, 0>(hp.g.na, q.s0, f, (const C2<T>*)P->d_tw_c, work, scratch, st);
    } else {
      ProfScope ps(st, 1, f.pro, f.epi);  // (the two launches of a two-level pass count as ONE first-axis pass)
      int n1, n2;
      if (nk_tl_split<T>(hp.g, n1, n2))
        rc = nk_tl_first_axis<T>(q.s1, n1, n2, f, (const C2<T>*)P->d_tw_t64, (const C2<T>*)(n2 == 64 ? P->d_tw_t64 : P->d_tw_t32),
                                 (const C2<T>*)P->d_tw_c, work, st);
      else
        rc = nk_dispatch_strided<T, 3>(hp.g.na, q.s1, f, (const C2<T>*)P->d_tw_c, work, scratch, st);
    }
    if (rc != NK_OK) return rc;
    const NkPassF& pf = q.pf;
    ProfScope ps(st, 3, f.pro, f.epi);
    static const int skip_final = nk_env_int("NK_SKIP_FINAL", 0);  // debugging aid
    if (skip_final) return NK_OK;
    return nk_final_with_slots(hp, workspace, f, st, [&](const NkFuse& f2) {
      return nk_dispatch_final<T>(hp.g.nl, pf, f2, (const C2<T>*)P->d_tw_f, (const C2<T>*)work, st);
    });
  }
  {
    ProfScope ps(st, 1, f.pro, f.epi);
    if (fast && nk_fast_contig_ok(hp.g.h)) {
      rc = nk_dispatch_contig<T, false>(hp.g.h, pa, f, tw_a, twr, work, st);
    } else {
      hipLaunchKernelGGL(k_passA<T>, dim3((unsigned)blocks_a), dim3(nk_gen_threads<T>(hp.threads_a)), hp.lds_a, st, pa, f, tw_a, twr, work);
      rc = nk_check_launch("k_passA");
    }
  }
  if (rc != NK_OK) return rc;
  NkPassS pc = hp.pc;
  pc.g.sign = pa.g.sign;
  if (hp.g.ndim == 3) {
    const int64_t blocks_b = hp.pb.outer * hp.pb.tiles_per_slab;
    ProfScope ps(st, 2, f.pro, f.epi);
    if (fast && nk_fast_strided_ok<T>(hp.g.nm, hp.pb.inner)) {
      rc = nk_dispatch_strided<T, 0>(hp.g.nm, hp.pb, f, (const C2<T>*)P->d_tw_b, work, scratch, st);
    } else {
      hipLaunchKernelGGL(k_passB<T>, dim3((unsigned)blocks_b), dim3(nk_gen_threads<T>(hp.threads_b)), hp.lds_b, st, hp.pb,
                         (const C2<T>*)P->d_tw_b, work);
      rc = nk_check_launch("k_passB");
    }
    if (rc != NK_OK) return rc;
  }
  const int64_t blocks_c = pc.outer * pc.tiles_per_slab;
  if (slots_on) {
    hipError_t e = hipMemsetAsync(vslots, 0, (size_t)slot_cap * sizeof(double), st);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipMemsetAsync(reduction slots)");
  }
  {
    ProfScope ps(st, 3, f.pro, f.epi);
    const int thr = nk_gen_threads<T>(hp.threads_c);
    const NkFuse fc = with_slots(blocks_c, thr);
    hipLaunchKernelGGL(k_passC<T>, dim3((unsigned)blocks_c), dim3(thr), hp.lds_c, st, pc, fc, (const C2<T>*)P->d_tw_c,
                       (const C2<T>*)work, scratch);
    rc = nk_check_launch("k_passC");
  }
  if (rc != NK_OK) return rc;
  const int64_t total_d = (int64_t)hp.g.batch * hp.g.nm * hp.g.na;
  {
    ProfScope ps(st, 4, f.pro, f.epi);
    const int64_t blocks_d = (total_d + 255) / 256;
    const NkFuse fd = with_slots(blocks_d, 256);
    hipLaunchKernelGGL(k_passD<T>, dim3((unsigned)blocks_d), dim3(256), 0, st, pc.g, fd, (const C2<T>*)scratch, total_d);
  }
  rc = nk_check_launch("k_passD");
  return rc != NK_OK ? rc : fold_slots();
}

static int nk_fused_check(const nk_plan* P, const nk_fuse* fuse, int convention) {
  if (!P || !fuse) return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: null argument");
  if (!fuse->in || !fuse->out) return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: in/out must be set");
  if (convention != NK_HARTLEY_NON_CANONICAL && convention != NK_HARTLEY_CANONICAL)
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: unknown hartley convention");
  if ((fuse->pro == NK_PRO_AMP || fuse->pro == NK_PRO_AMP_JVP) && (!fuse->pidx || !fuse->amp))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: AMP prologue needs pidx and amp");
  if (fuse->pro == NK_PRO_AMP_JVP && (!fuse->damp || !fuse->in2))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: AMP_JVP prologue needs damp and in2");
  if (fuse->pro == NK_PRO_MUL && !fuse->in2)
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: MUL prologue needs in2");
  if (fuse->epi == NK_EPI_VJP && (!fuse->pidx || !fuse->amp || !fuse->xi || !fuse->abar))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: VJP epilogue needs pidx, amp, xi and abar");
  if (fuse->epi == NK_EPI_LIKELIHOOD && (!fuse->data || !fuse->value))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: LIKELIHOOD epilogue needs data and value");
  if (fuse->cg_r) return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: cg_r is a prologue of nk_hartley_sandwich only");
  if (fuse->io32 && !(P->hp.dtype == NK_F64 && nk_plan_uses_pipeline2(P) && fuse->field_octant && fuse->pro == NK_PRO_AMP &&
                      fuse->epi == NK_EPI_LIKELIHOOD))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: io32 needs an fp64 plan with the octant pipeline, the AMP prologue "
                                        "with an octant amplitude field and the LIKELIHOOD epilogue");
  if (fuse->field_octant) {
    if (!nk_plan_uses_pipeline2(P))
      return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: field_octant needs nk_plan_octant_vjp(plan) != 0");
    const bool amp_pro = fuse->pro == NK_PRO_AMP || fuse->pro == NK_PRO_AMP_JVP;
    if ((amp_pro && !fuse->afield) || (fuse->pro == NK_PRO_AMP_JVP && !fuse->dafield) ||
        (fuse->epi == NK_EPI_VJP && !fuse->afield))
      return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused: field_octant needs afield (and dafield for AMP_JVP)");
    const NkGeom& g = P->hp.g;
    if ((int64_t)(g.na / 2 + 1) * (g.nm / 2 + 1) * (g.nl / 2 + 1) >= ((int64_t)1 << 31))
      return nk_set_error(NK_ERR_UNSUPPORTED, "nk_hartley_fused: octant field too large (>= 2^31 elements)");
  }
  return NK_OK;
}

extern "C" int nk_hartley_fused(const nk_plan* P, const nk_fuse* fuse, int convention, void* workspace,
                                void* stream) {
  const int rc = nk_fused_check(P, fuse, convention);
  if (rc != NK_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (P->hp.dtype == NK_F32) return nk_run_hartley<float>(P, *fuse, convention, workspace, st);
  return nk_run_hartley<double>(P, *fuse, convention, workspace, st);
}

// ---- nk_hartley_fused for a batch of members (include/niftyk.h) ---------------------------------------------------------
struct NkSlotArr {
  double* slots[NK_MAX_BATCH];  // nullptr: the member has no reduction
  double* value[NK_MAX_BATCH];
};
__global__ void __launch_bounds__(256) k_zero_slots_b(NkSlotArr sa, int64_t n) {
  double* p = sa.slots[blockIdx.y];
  if (!p) return;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0.0;
}
__global__ void __launch_bounds__(256) k_fold_slots_a_b(NkSlotArr sa, int64_t n) {
  __shared__ double red[4];
  const double* slots = sa.slots[blockIdx.y];
  if (!slots) return;
  double* part = sa.slots[blockIdx.y] + n;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double v = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) v += slots[i];
  const double s = nk_fold_block_sum(v, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void __launch_bounds__(NK_FOLD_BLOCKS) k_fold_slots_b_b(NkSlotArr sa, int64_t n) {
  __shared__ double red[NK_FOLD_BLOCKS / 64];
  if (!sa.slots[blockIdx.y]) return;
  const double* part = sa.slots[blockIdx.y] + n;
  const double s = nk_fold_block_sum(part[threadIdx.x], red);
  if (threadIdx.x == 0) *sa.value[blockIdx.y] += s;
}

static bool nk_same_class(const nk_fuse& a, const nk_fuse& b) {
  return a.pro == b.pro && a.epi == b.epi && !a.afield == !b.afield && !a.dafield == !b.dafield && !a.dampT == !b.dampT &&
         a.field_octant == b.field_octant && !a.io32 == !b.io32 && !a.w8 == !b.w8 && !a.wfull == !b.wfull &&
         a.abar_copies == b.abar_copies && !a.pidx_octant == !b.pidx_octant;
}

extern "C" int nk_plan_batch_ok(const nk_plan* P) {
  if (!P || !nk_plan_uses_pipeline2(P) || !nk_fast_enabled()) return 0;
  const NkHostPlan& hp = P->hp;
  static const int pipeline = nk_env_int("NK_PIPELINE", 2);
  if (hp.g.ndim != 2 || hp.g.batch != 1 || pipeline != 2 || !nk_fast_size(hp.g.nl)) return 0;
  return P->hp.dtype == NK_F32 ? nk_fast_strided_ok<float>(hp.g.na, hp.pc.inner) : nk_fast_strided_ok<double>(hp.g.na, hp.pc.inner);
}

template <typename T>
static int nk_run_hartley_batch(const nk_plan* P, const nk_fuse* fuse, int count, int convention, void* const* workspace,
                                hipStream_t st) {
  const NkHostPlan& hp = P->hp;
  NkBatchCtx bc;
  bc.count = count;
  bc.fuse = fuse;
  for (int m = 0; m < NK_MAX_BATCH; ++m) {
    const int k = m < count ? m : 0;
    bc.wa.work[m] = workspace[k];
    bc.wa.scratch[m] = (char*)workspace[k] + (hp.work_bytes + 255) / 256 * 256;
  }
  static const int work_blo = nk_env_int("NK_WORK_BLO", 0), work_pad = nk_env_int("NK_WORK_PAD", 2080);
  NkPipe2 q = nk_pipe2_setup(hp, convention == NK_HARTLEY_CANONICAL ? -1 : 1, work_blo, work_pad);
  {
    const int64_t smax = q.s1.ss > q.s0.inner ? q.s1.ss : q.s0.inner;
    if (128 * (smax > q.s1.inner ? smax : q.s1.inner) * (int64_t)sizeof(C2<T>) >= ((int64_t)1 << 32))
      return nk_set_error(NK_ERR_UNSUPPORTED, "transform too large for the 32-bit thread offsets of the strided passes");
  }
  // reduction slots of the final pass, member by member (nk_final_with_slots: energy / curvature sums; 2-D plans have no
  // max |w8|)
  const int64_t n = nk_value_slot_count(hp);
  NkSlotArr sa;
  bool any = false;
  for (int m = 0; m < NK_MAX_BATCH; ++m) {
    sa.slots[m] = nullptr, sa.value[m] = nullptr;
    if (m >= count) {
      bc.final_fuse[m] = fuse[0];
      continue;
    }
    NkFuse f2 = fuse[m];
    if (f2.w8max && f2.w8 && f2.epi == NK_EPI_VJP)
      return nk_set_error(NK_ERR_UNSUPPORTED, "nk_fuse.w8max: only the 3-D final pass reports max |w8|");
    f2.w8max = nullptr;
    if (f2.value && (f2.epi == NK_EPI_LIKELIHOOD || f2.epi == NK_EPI_VJP)) {
      sa.slots[m] = nk_value_slots(hp, workspace[m]);
      sa.value[m] = f2.value;
      f2.value = sa.slots[m];
      f2.value_slots = (int)std::min<int64_t>(n, 0x7fffffff);
      any = true;
    }
    bc.final_fuse[m] = f2;
  }
  struct Scope {  // the launchers see the batch only inside this call
    Scope(const NkBatchCtx* b) { t_batch = b; }
    ~Scope() { t_batch = nullptr; }
  } scope(&bc);
  int rc;
  {
    NkProfScope ps(st, 1, fuse[0].pro, fuse[0].epi, count);
    int n1, n2;
    if (nk_tl_split<T>(hp.g, n1, n2))
      rc = nk_tl_first_axis<T>(q.s1, n1, n2, fuse[0], (const C2<T>*)P->d_tw_t64, (const C2<T>*)(n2 == 64 ? P->d_tw_t64 : P->d_tw_t32),
                               (const C2<T>*)P->d_tw_c, (C2<T>*)bc.wa.work[0], st);
    else
      rc = nk_dispatch_strided<T, 3>(hp.g.na, q.s1, fuse[0], (const C2<T>*)P->d_tw_c, (C2<T>*)bc.wa.work[0], (C2<T>*)bc.wa.scratch[0], st);
  }
  if (rc != NK_OK) return rc;
  if (any) {
    hipLaunchKernelGGL(k_zero_slots_b, dim3(64, count), dim3(256), 0, st, sa, n);
    rc = nk_check_launch("k_zero_slots_b");
    if (rc != NK_OK) return rc;
  }
  {
    NkProfScope ps(st, 3, fuse[0].pro, fuse[0].epi, count);
    rc = nk_dispatch_final<T>(hp.g.nl, q.pf, bc.final_fuse[0], (const C2<T>*)P->d_tw_f, (const C2<T>*)bc.wa.work[0], st);
  }
  if (rc != NK_OK || !any) return rc;
  hipLaunchKernelGGL(k_fold_slots_a_b, dim3(NK_FOLD_BLOCKS, count), dim3(256), 0, st, sa, n);
  hipLaunchKernelGGL(k_fold_slots_b_b, dim3(1, count), dim3(NK_FOLD_BLOCKS), 0, st, sa, n);
  return nk_check_launch("k_fold_slots_b");
}

extern "C" int nk_hartley_fused_batch(const nk_plan* P, const nk_fuse* fuse, int count, int convention, void* const* workspace,
                                      void* stream) {
  if (!P || !fuse || !workspace || count < 1 || count > NK_MAX_BATCH)
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused_batch: bad argument (1 <= count <= NK_MAX_BATCH)");
  for (int m = 0; m < count; ++m) {
    const int rc = nk_fused_check(P, fuse + m, convention);
    if (rc != NK_OK) return rc;
    if (!workspace[m]) return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused_batch: every member needs a workspace");
    for (int k = 0; k < m; ++k)
      if (workspace[k] == workspace[m]) return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused_batch: members share a workspace");
    if (!nk_same_class(fuse[0], fuse[m]))
      return nk_set_error(NK_ERR_INVALID, "nk_hartley_fused_batch: the members select different kernel classes");
  }
  hipStream_t st = (hipStream_t)stream;
  if (count == 1 || !nk_plan_batch_ok(P) || !nk_batch_class_ok(P->hp.g, fuse[0])) {  // one launch set per member: same results
    for (int m = 0; m < count; ++m) {
      const int rc = P->hp.dtype == NK_F32 ? nk_run_hartley<float>(P, fuse[m], convention, workspace[m], st)
                                           : nk_run_hartley<double>(P, fuse[m], convention, workspace[m], st);
      if (rc != NK_OK) return rc;
    }
    return NK_OK;
  }
  if (P->hp.dtype == NK_F32) return nk_run_hartley_batch<float>(P, fuse, count, convention, workspace, st);
  return nk_run_hartley_batch<double>(P, fuse, count, convention, workspace, st);
}

// sandwich: out = EPI( scale * H( scale_first * mul_scalar * mul . H( PRO(in) ) ) ), five passes (nk_fft3.h)
static bool nk_strided_tile_divides(const nk_plan* P, int n, int64_t inner) {
  return P->hp.dtype == NK_F32 ? nk_fast_strided_ok<float>(n, inner) : nk_fast_strided_ok<double>(n, inner);
}
extern "C" int nk_plan_sandwich(const nk_plan* P) {
  if (!P || !nk_plan_uses_pipeline2(P) || !nk_fast_contig_ok(P->hp.g.h)) return 0;
  const NkGeom& g = P->hp.g;
  const int64_t rs = g.h + (P->hp.dtype == NK_F32 ? nk_pipe3_colpad<float>() : nk_pipe3_colpad<double>());
  if (g.ndim == 3 && !nk_strided_tile_divides(P, g.nm, rs)) return 0;
  return nk_strided_tile_divides(P, g.na, g.ndim == 3 ? (int64_t)g.nm * rs : rs) ? 1 : 0;
}

template <typename T>
static bool nk_contig3_quad_ok(int h) {
  switch (h) {
#define NK_CASE(NN) \
  case NN:          \
    return Contig3Tile<T, NN>::QUAD_OK;
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  }
  return false;
}
// 1 if nk_hartley_sandwich on this plan accepts nk_fuse.pipe_chunks == chunks (with an octant amplitude prologue)
extern "C" int nk_plan_pipe_ok(const nk_plan* P, int chunks) {
  if (!nk_plan_sandwich(P) || chunks < 2 || (chunks & 1)) return 0;
  const NkGeom& g = P->hp.g;
  if (g.ndim != 3 || g.batch != 1 || g.na % chunks != 0) return 0;
  if (!nk_env_int("NK_CONTIG_QUAD", 1) || (nk_env_int("NK_XMAP", NK_XMAP_DEFAULT) & 4)) return 0;
  return (P->hp.dtype == NK_F32 ? nk_contig3_quad_ok<float>(g.h) : nk_contig3_quad_ok<double>(g.h)) ? 1 : 0;
}

template <typename T>
static int nk_run_sandwich(const nk_plan* P, const NkFuse& f, double scale_first, int convention, void* workspace, hipStream_t st,
                           bool with_final = true, NkPipe3* q_out = nullptr) {
  const NkHostPlan& hp = P->hp;
  const int sign = convention == NK_HARTLEY_CANONICAL ? -1 : 1;
  static const int work_pad = nk_env_int("NK_WORK_PAD", 2080);
  const NkPipe3 q = nk_pipe3_setup<T>(hp, sign, work_pad, scale_first * (f.mul_scalar != 0.0 ? f.mul_scalar : 1.0));
  if (q_out) *q_out = q;
  if (nk_pipe3_work_elems(hp.g, nk_pipe3_colpad<T>(), work_pad) * sizeof(C2<T>) > hp.work_bytes)
    return nk_set_error(NK_ERR_RUNTIME, "nk_hartley_sandwich: plan workspace too small");
  if (128 * (q.ss > q.pm.s.inner ? q.ss : q.pm.s.inner) * (int64_t)sizeof(C2<T>) >= ((int64_t)1 << 32))
    return nk_set_error(NK_ERR_UNSUPPORTED, "transform too large for the 32-bit thread offsets of the strided passes");
  C2<T>* work = (C2<T>*)workspace;
  int rc;
  // slab pipelining (nk_fuse.pipe_chunks): C/2 stages of the first and of the final pass, see include/niftyk.h
  const int C = f.pipe_chunks >= 2 ? f.pipe_chunks : 0;
  if (C) {
    if (hp.g.ndim != 3 || hp.g.batch != 1 || (C & 1) || hp.g.na % C != 0 || !f.field_octant ||
        (f.pro != NK_PRO_AMP && f.pro != NK_PRO_AMP_JVP))
      return nk_set_error(NK_ERR_UNSUPPORTED, "nk_fuse.pipe_chunks: needs a 3-D plan with batch 1, an even chunk count that divides "
                                               "the first axis, and an octant amplitude prologue");
  }
  const int wc = C ? hp.g.na / C : 0;  // slabs per chunk
  if (C) {
    const int Mh = hp.g.nm / 2 + 1;
    for (int j = 0; j < C / 2; ++j) {
      // rows a8 = j wc .. (j+1) wc - 1 (the last stage also a8 = na / 2) and their mirrors: chunks <= j and >= C-1-j of `in`
      if (f.pipe_wait) {
        hipError_t e = hipStreamWaitEvent(st, (hipEvent_t)f.pipe_wait[j], 0);
        if (e != hipSuccess) return nk_set_hip_error(e, "hipStreamWaitEvent(pipe_wait)");
      }
      NkPass3 p1 = q.p1;
      const int a_lo = j * wc, a_hi = (j == C / 2 - 1) ? hp.g.na / 2 + 1 : (j + 1) * wc;
      p1.blk0 = (int64_t)a_lo * Mh;
      p1.nblk = (int64_t)(a_hi - a_lo) * Mh;
      ProfScope ps(st, 5, f.cg_r ? 4 : f.pro, f.epi);  // (prologue key 4: AMP_JVP with the CG direction update riding along)
      rc = nk_dispatch_contig3<T>(hp.g.h, p1, f, (const C2<T>*)P->d_tw_a, (const C2<T>*)P->d_twr_a, work, st);
      if (rc != NK_OK) return rc;
    }
  } else {
    ProfScope ps(st, 5, f.cg_r ? 4 : f.pro, f.epi);
    rc = nk_dispatch_contig3<T>(hp.g.h, q.p1, f, (const C2<T>*)P->d_tw_a, (const C2<T>*)P->d_twr_a, work, st);
    if (rc != NK_OK) return rc;
  }
  if (hp.g.ndim == 3) {
    ProfScope ps(st, 6, f.pro, f.epi);
    rc = nk_dispatch_strided<T, 0>(hp.g.nm, q.s2, f, (const C2<T>*)P->d_tw_b, work, (C2<T>*)nullptr, st);
    if (rc != NK_OK) return rc;
  }
  {
    ProfScope ps(st, 7, f.pro, f.epi);
    rc = nk_dispatch_mid<T>(hp.g.na, q.pm, f, (const C2<T>*)P->d_tw_c, work, st);
  }
  if (rc != NK_OK) return rc;
  if (hp.g.ndim == 3) {
    ProfScope ps(st, 6, f.pro, f.epi);
    rc = nk_dispatch_strided<T, 0>(hp.g.nm, q.s2, f, (const C2<T>*)P->d_tw_b, work, (C2<T>*)nullptr, st);
    if (rc != NK_OK) return rc;
  }
  if (!with_final) return NK_OK;  // nk_hartley_sandwich_pair launches the final passes of two samples together
  ProfScope ps(st, 3, f.pro, f.epi);
  return nk_final_with_slots(hp, workspace, f, st, [&](const NkFuse& f2) {
    if (!C) return nk_dispatch_final3<T>(hp.g.nl, q.pf, f2, (const C2<T>*)P->d_tw_f, (const C2<T>*)work, st);
    for (int j = 0; j < C / 2; ++j) {
      // pair indices a = j wc + 1 .. (j+1) wc (stage 0 also a = 0): afterwards the chunks j and C-1-j of `out` are final
      NkPassF pf = q.pf;
      pf.a0 = j == 0 ? 0 : j * wc + 1;
      pf.a_cnt = (j + 1) * wc + 1 - pf.a0;
      int rcj = nk_dispatch_final3<T>(hp.g.nl, pf, f2, (const C2<T>*)P->d_tw_f, (const C2<T>*)work, st);
      if (rcj != NK_OK) return rcj;
      if (f.pipe_record) {
        hipError_t e = hipEventRecord((hipEvent_t)f.pipe_record[j], st);
        if (e != hipSuccess) return nk_set_hip_error(e, "hipEventRecord(pipe_record)");
      }
    }
    return (int)NK_OK;
  });
}

static int nk_sandwich_check(const nk_plan* P, const nk_fuse* fuse, int convention, void* workspace) {
  if (!P || !fuse) return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: null argument");
  if (!fuse->in || !fuse->out || !workspace) return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: in/out/workspace must be set");
  if (convention != NK_HARTLEY_NON_CANONICAL && convention != NK_HARTLEY_CANONICAL)
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: unknown hartley convention");
  if (!nk_plan_sandwich(P)) return nk_set_error(NK_ERR_UNSUPPORTED, "nk_hartley_sandwich: needs nk_plan_sandwich(plan) != 0");
  if (fuse->epi == NK_EPI_MUL)
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: `mul` is the diagonal between the transforms; the MUL epilogue is not available");
  if (fuse->cg_r && !(fuse->field_octant && fuse->pro == NK_PRO_AMP_JVP && fuse->dafield && fuse->cg_scal))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: cg_r needs field_octant, the AMP_JVP prologue with dafield, and cg_scal");
  if ((fuse->pro == NK_PRO_AMP || fuse->pro == NK_PRO_AMP_JVP) && (!fuse->pidx || !fuse->amp))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: AMP prologue needs pidx and amp");
  if (fuse->pro == NK_PRO_AMP_JVP && (!fuse->damp || !fuse->in2))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: AMP_JVP prologue needs damp and in2");
  if (fuse->pro == NK_PRO_MUL && !fuse->in2) return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: MUL prologue needs in2");
  if (fuse->epi == NK_EPI_VJP && (!fuse->pidx || !fuse->amp || !fuse->xi || !fuse->abar))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: VJP epilogue needs pidx, amp, xi and abar");
  if (fuse->epi == NK_EPI_LIKELIHOOD && (!fuse->data || !fuse->value))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: LIKELIHOOD epilogue needs data and value");
  if (fuse->field_octant) {
    const bool amp_pro = fuse->pro == NK_PRO_AMP || fuse->pro == NK_PRO_AMP_JVP;
    if ((amp_pro && !fuse->afield) || (fuse->pro == NK_PRO_AMP_JVP && !fuse->dafield && !(fuse->pidx_octant && fuse->dampT)) ||
        (fuse->epi == NK_EPI_VJP && !fuse->afield))
      return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich: field_octant needs afield (and dafield, or pidx_octant + dampT, for AMP_JVP)");
    const NkGeom& g = P->hp.g;
    if ((int64_t)(g.na / 2 + 1) * (g.nm / 2 + 1) * (g.nl / 2 + 1) >= ((int64_t)1 << 31))
      return nk_set_error(NK_ERR_UNSUPPORTED, "nk_hartley_sandwich: octant field too large (>= 2^31 elements)");
  }
  return NK_OK;
}

extern "C" int nk_hartley_sandwich(const nk_plan* P, const nk_fuse* fuse, double scale_first, int convention,
                                   void* workspace, void* stream) {
  const int rc = nk_sandwich_check(P, fuse, convention, workspace);
  if (rc != NK_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (P->hp.dtype == NK_F32) return nk_run_sandwich<float>(P, *fuse, scale_first, convention, workspace, st);
  return nk_run_sandwich<double>(P, *fuse, scale_first, convention, workspace, st);
}

template <typename T>
static int nk_run_sandwich_pair(const nk_plan* P, const NkFuse& fa, const NkFuse& fb, double scale_first, int convention,
                                void* wsa, void* wsb, hipStream_t st) {
  NkPipe3 q;
  int rc = nk_run_sandwich<T>(P, fa, scale_first, convention, wsa, st, false, &q);
  if (rc != NK_OK) return rc;
  rc = nk_run_sandwich<T>(P, fb, scale_first, convention, wsb, st, false, nullptr);
  if (rc != NK_OK) return rc;
  const NkHostPlan& hp = P->hp;
  ProfScope ps(st, 9, fa.pro, fa.epi);  // its own profile key: one launch, two final passes
  return nk_final_with_slots(hp, wsa, fa, st, [&](const NkFuse& fa2) {
    return nk_final_with_slots(hp, wsb, fb, st, [&](const NkFuse& fb2) {
      return nk_dispatch_final_pair<T>(hp.g.nl, q.pf, fa2, fb2, (const C2<T>*)P->d_tw_f, (const C2<T>*)wsa, (const C2<T>*)wsb, st);
    });
  });
}

extern "C" int nk_hartley_sandwich_pair(const nk_plan* P, const nk_fuse* fa, const nk_fuse* fb, double scale_first,
                                        int convention, void* workspace_a, void* workspace_b, void* stream) {
  int rc = nk_sandwich_check(P, fa, convention, workspace_a);
  if (rc != NK_OK) return rc;
  rc = nk_sandwich_check(P, fb, convention, workspace_b);
  if (rc != NK_OK) return rc;
  const bool vjp = fa->epi == NK_EPI_VJP && fb->epi == NK_EPI_VJP && fa->afield && fb->afield && fa->field_octant && fb->field_octant;
  if (!vjp || P->hp.g.ndim != 3 || fa->pipe_chunks || fb->pipe_chunks || workspace_a == workspace_b)
    return nk_set_error(NK_ERR_UNSUPPORTED, "nk_hartley_sandwich_pair: two VJP epilogues with octant amplitude fields on a 3-D plan, "
                                             "two workspaces, no slab pipelining");
  // B joins A's fresh lines to its sum: as its running sum (one shared `out`) or as its innermost partial sum (carry1 = A's
  // out, B's own `out` holding an older partial sum: the pairwise order over samples)
  const bool joined = (fa->out == fb->out && fb->accumulate) || (fb->carry1 == fa->out && fb->out != fa->out);
  if (!joined || fa->w8 == fb->w8 || (fa->mul_scalar != fb->mul_scalar))
    return nk_set_error(NK_ERR_INVALID, "nk_hartley_sandwich_pair: B accumulates onto A's `out` or takes it as carry1; separate "
                                         "w8 areas, the same scalar diagonal");
  hipStream_t st = (hipStream_t)stream;
  if (P->hp.dtype == NK_F32) return nk_run_sandwich_pair<float>(P, *fa, *fb, scale_first, convention, workspace_a, workspace_b, st);
  return nk_run_sandwich_pair<double>(P, *fa, *fb, scale_first, convention, workspace_a, workspace_b, st);
}

extern "C" int nk_hartley(const nk_plan* P, const void* in, void* out, double scale, int convention, void* workspace,
                          void* stream) {
  nk_fuse f;
  memset(&f, 0, sizeof(f));
  f.pro = NK_PRO_PLAIN;
  f.in = in;
  f.epi = NK_EPI_AFFINE;
  f.out = out;
  f.scale = scale;
  f.offset = 0.0;
  return nk_hartley_fused(P, &f, convention, workspace, stream);
}

template <typename T>
static int nk_run_c2c(const nk_plan* P, const void* in, void* out, int inverse, double scale, hipStream_t st) {
  if (P->cc.lp.n == 0) return nk_set_error(NK_ERR_UNSUPPORTED, "nk_fftn: last axis too long for the c2c kernel");
  NkPassCC cc = P->cc;
  cc.swap = inverse ? 1 : 0;
  cc.scale = scale;
  const int64_t blocks = (cc.nlines + cc.tl.tile - 1) / cc.tl.tile;
  hipLaunchKernelGGL(k_c2c_contig<T>, dim3((unsigned)blocks), dim3(nk_gen_threads<T>(P->threads_cc)), P->lds_cc, st, cc,
                     (const C2<T>*)P->d_tw_cc, (const C2<T>*)in, (C2<T>*)out);
  int rc = nk_check_launch("k_c2c_contig");
  if (rc != NK_OK) return rc;
  if (P->ndim == 3) {
    const NkPassS& ps = P->c2c_mid;
    hipLaunchKernelGGL(k_c2c_strided<T>, dim3((unsigned)(ps.outer * ps.tiles_per_slab)), dim3(nk_gen_threads<T>(P->threads_cm)),
                       P->lds_cm, st, ps, cc.swap, (const C2<T>*)P->d_tw_b, (C2<T>*)out);
    rc = nk_check_launch("k_c2c_strided(mid)");
    if (rc != NK_OK) return rc;
  }
  if (P->ndim >= 2) {
    const NkPassS& ps = P->c2c_first;
    hipLaunchKernelGGL(k_c2c_strided<T>, dim3((unsigned)(ps.outer * ps.tiles_per_slab)), dim3(nk_gen_threads<T>(P->threads_cf)),
                       P->lds_cf, st, ps, cc.swap, (const C2<T>*)P->d_tw_c, (C2<T>*)out);
    rc = nk_check_launch("k_c2c_strided(first)");
  }
  return rc;
}

extern "C" int nk_fftn(const nk_plan* P, const void* in, void* out, int inverse, double scale, void* workspace,
                       void* stream) {
  (void)workspace;
  if (!P || !in || !out) return nk_set_error(NK_ERR_INVALID, "nk_fftn: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (P->hp.dtype == NK_F32) return nk_run_c2c<float>(P, in, out, inverse, scale, st);
  return nk_run_c2c<double>(P, in, out, inverse, scale, st);
}
