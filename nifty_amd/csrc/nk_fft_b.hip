// nk_fft_b.hip -- the batched twins of the strided-first pass kernels (nk_hartley_fused_batch, include/niftyk.h) and their
// launchers; see nk_fft_batch.h.  Same phase functions as the single kernels of nk_fft.hip (nk_fft2.h).
#include <hip/hip_runtime.h>

#include "nk_fft_batch.h"

template <typename T, int N, int MODE, int PC>
__global__ void __launch_bounds__((StridedTile<T, N, nk_strided_cx<MODE, PC>(), MODE>::THREADS),
                                  (StridedTile<T, N, nk_strided_cx<MODE, PC>(), MODE>::SC::E == 64
                                       ? 2
                                       : (MODE == 3 && NK_S1_TWO_WG && StridedTile<T, N>::LDS_BYTES <= 80 * 1024 &&
                                                  StridedTile<T, N>::THREADS <= 512
                                              ? 2 * StridedTile<T, N>::THREADS / 256
                                              : NK_S0_WAVES)))
    k2_strided_b(NkPassS p, NkFuseArr fa, const C2<T>* __restrict__ tw, NkWorkArr wa, int xmap) {
  extern __shared__ __align__(16) unsigned char smem[];
  using ST = StridedTile<T, N, nk_strided_cx<MODE, PC>(), MODE>;
  DeviceExec<T, ST::SC::E> ex;
  double acc = 0.0;
  const int m = blockIdx.y;
  constexpr bool OCT = MODE == 3 && (PC == 4 || PC == 5 || PC == 9);
  const int64_t blk = (xmap && (!OCT || p.g.ndim != 3)) ? nk_xcd_contig(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  C2<T>* tw_lds = ST::TWLDS ? reinterpret_cast<C2<T>*>(smem + ST::LDS_BYTES) : nullptr;
  nk_strided_body<T, N, ST::TILE, MODE, PC, nk_strided_cx<MODE, PC>()>(ex, p, fa.f[m], blk, (T*)smem, tw, (C2<T>*)wa.work[m],
                                                                       (C2<T>*)wa.scratch[m], &acc, tw_lds);
  (void)acc;
}

template <typename T, int NL, bool COUPLES, int EC, int PAIR>
__global__ void __launch_bounds__((FinalTile<T, NL, EC, COUPLES ? 2 : 1>::THREADS),
                                  (nk_final_waves<T, COUPLES, EC, FinalTile<T, NL, EC, COUPLES ? 2 : 1>::THREADS>()))
    k2_final_b(NkPassF p, NkFuseArr fa, const C2<T>* __restrict__ tw, NkWorkArr wa, int xmap) {
  extern __shared__ __align__(16) unsigned char smem[];
  DeviceExec<T, SchedF<T, NL>::E> ex;
  double acc = 0.0;
  float wmax = 0.0f;
  const int m = blockIdx.y;
  const int64_t blk = xmap ? nk_xcd_contig(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x + p.blk0;
  nk_final_body<T, NL, FinalTile<T, NL, EC, COUPLES ? 2 : 1>::TILE, COUPLES, EC, PAIR>(ex, p, fa.f[m], blk, (T*)smem, tw,
                                                                                      (const C2<T>*)wa.work[m], &acc,
                                                                                      (COUPLES && (EC == 2 || EC == -1)) ? &wmax : nullptr);
  nk_flush_energy(fa.f[m], acc, smem);
}


template <typename T, int N, int PC>
int nk_twin_launch_strided(const NkPassS& ps, int64_t blocks, const C2<T>* tw, int xmap, hipStream_t st) {
  using ST = StridedTile<T, N, nk_strided_cx<3, PC>(), 3>;
  const NkBatchCtx& bc = *t_batch;
  auto kb = k2_strided_b<T, N, 3, PC>;
  static unsigned long long attr_mask = 0;  // per-device attribute
  if (ST::LDS_TOTAL > 64 * 1024 && nk_first_on_device(attr_mask)) {
    hipError_t e = hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, ST::LDS_TOTAL);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(k2_strided_b)");
  }
  NkFuseArr fa;
  for (int m = 0; m < NK_MAX_BATCH; ++m) fa.f[m] = bc.fuse[m < bc.count ? m : 0];
  hipLaunchKernelGGL(kb, dim3((unsigned)blocks, (unsigned)bc.count), dim3(ST::THREADS), ST::LDS_TOTAL, st, ps, fa, tw, bc.wa, xmap);
  return nk_check_launch("k2_strided_b");
}

template <typename T, int NL, bool COUPLES, int EC>
int nk_twin_launch_final(const NkPassF& pf, int64_t blocks, const C2<T>* tw, int xmap, hipStream_t st) {
  using CT = FinalTile<T, NL, EC, COUPLES ? 2 : 1>;
  const NkBatchCtx& bc = *t_batch;
  auto kb = k2_final_b<T, NL, COUPLES, EC, 0>;
  static unsigned long long attr_mask = 0;  // per-device attribute
  if (CT::LDS_BYTES > 64 * 1024 && nk_first_on_device(attr_mask)) {
    hipError_t e = hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, CT::LDS_BYTES);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(k2_final_b)");
  }
  NkFuseArr fa;
  for (int m = 0; m < NK_MAX_BATCH; ++m) fa.f[m] = bc.final_fuse[m < bc.count ? m : 0];
  // every wavefront owns one slot of a member's energy area: never drop a partial silently
  for (int m = 0; m < bc.count; ++m)
    if (fa.f[m].value_slots > 0 && blocks * ((CT::THREADS + 63) / 64) > fa.f[m].value_slots)
      return nk_set_error(NK_ERR_RUNTIME, "final pass: more wavefronts than reduction slots (nk_value_slot_count)");
  hipLaunchKernelGGL(kb, dim3((unsigned)blocks, (unsigned)bc.count), dim3(CT::THREADS), CT::LDS_BYTES, st, pf, fa, tw, bc.wa, xmap);
  return nk_check_launch("k2_final_b");
}

// explicit instantiations: exactly the classes of nk_twin_strided / nk_twin_final (nk_fft_batch.h)
#define NK_TWIN_SIZES(X) X(512) X(1024) X(2048) X(4096)
#define NK_TWINS(NN)                                                                                                             \
  template int nk_twin_launch_strided<float, NN, 0>(const NkPassS&, int64_t, const C2<float>*, int, hipStream_t);                \
  template int nk_twin_launch_strided<float, NN, 4>(const NkPassS&, int64_t, const C2<float>*, int, hipStream_t);                \
  template int nk_twin_launch_strided<float, NN, 5>(const NkPassS&, int64_t, const C2<float>*, int, hipStream_t);                \
  template int nk_twin_launch_strided<float, NN, 6>(const NkPassS&, int64_t, const C2<float>*, int, hipStream_t);                \
  template int nk_twin_launch_strided<double, NN, 0>(const NkPassS&, int64_t, const C2<double>*, int, hipStream_t);              \
  template int nk_twin_launch_strided<double, NN, 4>(const NkPassS&, int64_t, const C2<double>*, int, hipStream_t);              \
  template int nk_twin_launch_strided<double, NN, 5>(const NkPassS&, int64_t, const C2<double>*, int, hipStream_t);              \
  template int nk_twin_launch_strided<double, NN, 6>(const NkPassS&, int64_t, const C2<double>*, int, hipStream_t);              \
  template int nk_twin_launch_final<float, NN, true, 2>(const NkPassF&, int64_t, const C2<float>*, int, hipStream_t);            \
  template int nk_twin_launch_final<float, NN, false, 1>(const NkPassF&, int64_t, const C2<float>*, int, hipStream_t);           \
  template int nk_twin_launch_final<float, NN, false, 3>(const NkPassF&, int64_t, const C2<float>*, int, hipStream_t);           \
  template int nk_twin_launch_final<float, NN, false, -1>(const NkPassF&, int64_t, const C2<float>*, int, hipStream_t);          \
  template int nk_twin_launch_final<double, NN, true, 2>(const NkPassF&, int64_t, const C2<double>*, int, hipStream_t);          \
  template int nk_twin_launch_final<double, NN, false, 1>(const NkPassF&, int64_t, const C2<double>*, int, hipStream_t);         \
  template int nk_twin_launch_final<double, NN, false, 3>(const NkPassF&, int64_t, const C2<double>*, int, hipStream_t);         \
  template int nk_twin_launch_final<double, NN, false, -1>(const NkPassF&, int64_t, const C2<double>*, int, hipStream_t);
NK_TWIN_SIZES(NK_TWINS)
// the single-pair VJP final pass of 2-D grids (nk_final_single_2d)
template int nk_twin_launch_final<double, 4096, false, 2>(const NkPassF&, int64_t, const C2<double>*, int, hipStream_t);
template int nk_twin_launch_final<double, 2048, false, 2>(const NkPassF&, int64_t, const C2<double>*, int, hipStream_t);
template int nk_twin_launch_final<float, 4096, false, 2>(const NkPassF&, int64_t, const C2<float>*, int, hipStream_t);
