// nk_fft_batch.h -- what nk_fft.hip and nk_fft_b.hip share: the batched twins of the strided-first pass kernels live in a
// translation unit of their own (every twin is another instantiation of a heavy template; two units compile side by side).
#pragma once
#include "nk_plan.h"
#include "nk_fft2.h"
#include "nk_util.h"

#ifndef NK_S0_WAVES
#define NK_S0_WAVES 1
#endif
#ifndef NK_PAIR_HAND_LDS_KB
#define NK_PAIR_HAND_LDS_KB 64  // k2_final2 hands A's lines to B through LDS while planes + stash fit this (nk_fft_p.hip)
#endif
#ifndef NK_S1_TWO_WG
#define NK_S1_TWO_WG 0
#endif

// Wavefronts per SIMD a final-pass kernel is compiled for (the second argument of __launch_bounds__ = its register budget:
// 3 -> 168 VGPRs, 2 -> 256).  fp64 kernels on line couples keep 2 x TILE x PITCH x 8 bytes of planes with TILE >= 2: their
// LDS admits two waves per SIMD whatever the register count, so a tighter budget only buys spills -- k2_final2<double,512>
// 168 VGPRs + 184 spilled -> 234 + 0: 1.87 -> 1.54 ms per pair launch at 512^3 (round 5).  The fp32 pair kernel at 1024 is
// the opposite case (six resident workgroups at 168 VGPRs beat four at 175: 51.9 vs 54.0 ms per four-sample application).
template <typename T, bool COUPLES, int EC, int THREADS>
constexpr int nk_final_waves() {
  return THREADS > 256 ? 1 : (sizeof(T) == 8 && (COUPLES || EC == 2)) ? 2 : ((!COUPLES && sizeof(T) == 4 && (EC == 0 || EC == 1)) ? 4 : 3);
}

// fp64 sum of `acc` over the workgroup (only when the epilogue produces an energy).  With slots (set up by the library for
// the final pass of the pipelines): every WAVEFRONT stores its partial sum to its own slot -- no LDS hop, no barrier at
// the end of the kernel (a barrier there kept the workgroup's LDS and wave slots busy until its last wave arrived: 4 % of
// the scatter pass) -- and the slots are folded in a fixed order afterwards: bit-reproducible, no atomics.  Without slots
// (the generic kernels): one atomic per workgroup on *value.
__device__ __forceinline__ void nk_flush_energy(const NkFuse& f, double acc, void* lds_raw) {
  if ((f.epi != NK_EPI_LIKELIHOOD && f.epi != NK_EPI_VJP) || f.value == nullptr) return;
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  if (f.value_slots > 0) {
    const int64_t slot = (int64_t)blockIdx.x * nw + wave;
    if (lane == 0 && slot < f.value_slots) f.value[slot] = acc;
    return;
  }
  __syncthreads();  // LDS tile is dead from here on
  double* red = (double*)lds_raw;
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += red[w];
    atomicAdd(f.value, s);
  }
}

// maximum of |w8| (octant sums of the VJP epilogue) of this wavefront -> its slot; folded by the library afterwards
__device__ __forceinline__ void nk_flush_wmax(const NkFuse& f, float wmax) {
#ifdef NK_WMAX_OFF
  return;
#endif
  if (f.w8max == nullptr || f.w8 == nullptr || f.epi != NK_EPI_VJP || f.value_slots <= 0) return;
  for (int off = 32; off > 0; off >>= 1) wmax = nk_wmax_join(wmax, __shfl_down(wmax, off, 64));  // NaN / inf survive
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t slot = (int64_t)blockIdx.x * ((blockDim.x + 63) >> 6) + wave;
  if (lane == 0 && slot < f.value_slots) f.w8max[slot] = (double)wmax;
}


// ---- batched launches of the strided-first pipeline (include/niftyk.h, "batched launches") ---------------------------------
// The members' fuse records travel in the kernel arguments (8 x 328 bytes), blockIdx.y picks the member; work arrays and
// reduction slots are the members' own.  A member runs exactly the code of the single launch: same tiles, same block
// order within the member, same slots -- the same bits.
struct NkFuseArr {
  NkFuse f[NK_MAX_BATCH];
};
struct NkWorkArr {
  void* work[NK_MAX_BATCH];
  void* scratch[NK_MAX_BATCH];
};
// what nk_hartley_fused_batch hands to the innermost launchers of this host thread (they launch the batched twin of the
// kernel they would launch for member 0)
struct NkBatchCtx {
  int count;
  const NkFuse* fuse;    // [count] as given by the caller (first pass)
  NkFuse final_fuse[NK_MAX_BATCH];  // with the members' reduction slots filled in (final pass)
  NkWorkArr wa;
};
extern thread_local const NkBatchCtx* t_batch;  // defined in nk_fft.hip
// Which kernel classes have a batched twin (every twin is another instantiation of a heavy template: the classes the fused
// engine launches on 2-D grids from 512 points per axis on; any other class runs member by member):
//   first pass: PLAIN (0), MUL (6), octant AMP (4), octant AMP_JVP (5);
//   final pass: MUL (1), likelihood (3), VJP with amplitude field on line couples (2), run-time generic (-1: NONLIN).
template <int N, int PC>
constexpr bool nk_twin_strided() {
  return N >= 512 && (PC == 0 || PC == 4 || PC == 5 || PC == 6);
}
template <int NL, bool COUPLES, int EC, int PAIR>
constexpr bool nk_twin_final() {
  return NL >= 512 && PAIR == 0 && ((COUPLES && EC == 2) || (!COUPLES && (EC == 1 || EC == 3 || EC == -1)) ||
                                    (!COUPLES && EC == 2 && NL >= 2048));  // (the single-pair VJP build of 2-D grids, nk_final_single_2d)
}

static bool nk_batch_class_ok(const NkGeom& g, const nk_fuse& f) {
  static const int generic = nk_env_int("NK_EC_GENERIC", 0);
  if (generic || g.na < 512 || g.nl < 512) return false;
  const bool pro_ok = f.pro == NK_PRO_PLAIN || f.pro == NK_PRO_MUL ||
                      (f.field_octant && !f.io32 && (f.pro == NK_PRO_AMP || f.pro == NK_PRO_AMP_JVP));
  const bool epi_ok = f.epi == NK_EPI_MUL || (f.epi == NK_EPI_LIKELIHOOD && !f.io32) || f.epi == NK_EPI_NONLIN ||
                      (f.epi == NK_EPI_VJP && f.afield);
  return pro_ok && epi_ok;
}


// the twins' launchers: defined and explicitly instantiated in nk_fft_b.hip for exactly the classes nk_twin_* name
template <typename T, int N, int PC>
int nk_twin_launch_strided(const NkPassS& ps, int64_t blocks, const C2<T>* tw, int xmap, hipStream_t st);
template <typename T, int NL, bool COUPLES, int EC>
int nk_twin_launch_final(const NkPassF& pf, int64_t blocks, const C2<T>* tw, int xmap, hipStream_t st);

// the two launches of the two-level first-axis pass (nk_fft2.h: nk_tl_split, nk_strided_body MODE 4 / 5), defined in
// nk_fft_t.hip: sub-lines of n1 points through the prologue class of `f` (run-time class for the rare ones), then n2-point
// sub-lines in place.  Inside nk_hartley_fused_batch (t_batch set) both launches cover all members (grid.y).
template <typename T>
int nk_tl_first_axis(const NkPassS& s1, int n1, int n2, const NkFuse& f, const C2<T>* tw_n1, const C2<T>* tw_n2, const C2<T>* tw_full,
                     C2<T>* work, hipStream_t st);

// nk_hartley_sandwich_pair's final-pass launch (k2_final2): defined and explicitly instantiated in nk_fft_p.hip
template <typename T, int NL>
int nk_launch_final_pair(NkPassF pf, const NkFuse& fa, const NkFuse& fb, const C2<T>* tw, const C2<T>* worka, const C2<T>* workb,
                         hipStream_t st);
