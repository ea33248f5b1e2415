// nk_fft_t.hip -- the two-level first-axis pass of 2-D grids (nk_fft2.h: nk_tl_split, nk_strided_body MODE 4 / 5) and its
// launchers, single and batched; a translation unit of its own like the batched twins (nk_fft_b.hip).
#include <hip/hip_runtime.h>

#include "nk_fft_batch.h"

// MODE 4: first launch (fused prologue class PC, inter-level twiddles on the way out); MODE 5: second launch, in place
template <typename T, int N, int MODE, int PC>
__global__ void __launch_bounds__((StridedTile<T, N, false, MODE>::THREADS))
    k2_tl(NkPassS p, NkFuse f, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ tw_full, C2<T>* __restrict__ work) {
  extern __shared__ __align__(16) unsigned char smem[];
  using ST = StridedTile<T, N, false, MODE>;
  DeviceExec<T, ST::SC::E> ex;
  double acc = 0.0;
  nk_strided_body<T, N, ST::TILE, MODE, PC, false>(ex, p, f, nk_xcd_contig(blockIdx.x, gridDim.x), (T*)smem, tw, work, (C2<T>*)nullptr,
                                                   &acc, nullptr, tw_full);
  (void)acc;
}
template <typename T, int N, int MODE, int PC>
__global__ void __launch_bounds__((StridedTile<T, N, false, MODE>::THREADS))
    k2_tl_b(NkPassS p, NkFuseArr fa, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ tw_full, NkWorkArr wa) {
  extern __shared__ __align__(16) unsigned char smem[];
  using ST = StridedTile<T, N, false, MODE>;
  DeviceExec<T, ST::SC::E> ex;
  double acc = 0.0;
  const int m = blockIdx.y;
  nk_strided_body<T, N, ST::TILE, MODE, PC, false>(ex, p, fa.f[m], nk_xcd_contig(blockIdx.x, gridDim.x), (T*)smem, tw, (C2<T>*)wa.work[m],
                                                   (C2<T>*)nullptr, &acc, nullptr, tw_full);
  (void)acc;
}

template <typename T, int N, int MODE, int PC>
static int nk_tl_launch(NkPassS ps, int other, const NkFuse& f, const C2<T>* tw, const C2<T>* tw_full, C2<T>* work, hipStream_t st) {
  using ST = StridedTile<T, N, false, MODE>;
  static_assert(ST::LDS_BYTES <= 64 * 1024, "two-level tiles are small");
  if (ps.inner % ST::TILE != 0) return nk_set_error(NK_ERR_UNSUPPORTED, "two-level first-axis pass: row length not a multiple of the tile");
  ps.tl.tile = ST::TILE;
  ps.tl.dtile = nk_make_div(ST::TILE);
  ps.tiles_per_slab = (int)(ps.inner / ST::TILE);
  ps.sub = other;
  const int64_t blocks = (int64_t)ps.g.batch * other * ps.tiles_per_slab;
  if (blocks > 0x7fffffffLL) return nk_set_error(NK_ERR_UNSUPPORTED, "two-level first-axis pass: too many tiles for one launch");
  if (t_batch != nullptr) {
    const NkBatchCtx& bc = *t_batch;
    NkFuseArr fa;
    for (int m = 0; m < NK_MAX_BATCH; ++m) fa.f[m] = bc.fuse[m < bc.count ? m : 0];
    hipLaunchKernelGGL((k2_tl_b<T, N, MODE, PC>), dim3((unsigned)blocks, (unsigned)bc.count), dim3(ST::THREADS), ST::LDS_BYTES, st, ps, fa, tw,
                       tw_full, bc.wa);
    return nk_check_launch("k2_tl_b");
  }
  hipLaunchKernelGGL((k2_tl<T, N, MODE, PC>), dim3((unsigned)blocks), dim3(ST::THREADS), ST::LDS_BYTES, st, ps, f, tw, tw_full, work);
  return nk_check_launch("k2_tl");
}

// the prologue classes of nk_launch_strided that the fused engine launches on 2-D grids; everything else: run-time class
template <typename T, int N1>
static int nk_tl_first(const NkPassS& ps, int n2, const NkFuse& f, const C2<T>* tw, const C2<T>* tw_full, C2<T>* work, hipStream_t st) {
  if constexpr (sizeof(T) == 8)
    if (f.field_octant && f.pro == NK_PRO_AMP && f.io32) return nk_tl_launch<T, N1, 4, 9>(ps, n2, f, tw, tw_full, work, st);
  if (f.field_octant && f.pro == NK_PRO_AMP) return nk_tl_launch<T, N1, 4, 4>(ps, n2, f, tw, tw_full, work, st);
  if (f.field_octant && f.pro == NK_PRO_AMP_JVP) return nk_tl_launch<T, N1, 4, 5>(ps, n2, f, tw, tw_full, work, st);
  if (f.pro == NK_PRO_PLAIN) return nk_tl_launch<T, N1, 4, 0>(ps, n2, f, tw, tw_full, work, st);
  if (f.pro == NK_PRO_MUL) return nk_tl_launch<T, N1, 4, 6>(ps, n2, f, tw, tw_full, work, st);
  return nk_tl_launch<T, N1, 4, -1>(ps, n2, f, tw, tw_full, work, st);
}

template <typename T>
int nk_tl_first_axis(const NkPassS& s1, int n1, int n2, const NkFuse& f, const C2<T>* tw_n1, const C2<T>* tw_n2, const C2<T>* tw_full,
                     C2<T>* work, hipStream_t st) {
  if (n1 != 64 || (n2 != 64 && n2 != 32) || s1.g.ndim != 2 || s1.ss != 0)
    return nk_set_error(NK_ERR_UNSUPPORTED, "two-level first-axis pass: 2-D grids, 64 x 64 or 64 x 32 points");
  int rc = nk_tl_first<T, 64>(s1, n2, f, tw_n1, tw_full, work, st);
  if (rc != NK_OK) return rc;
  if (n2 == 64) return nk_tl_launch<T, 64, 5, -1>(s1, n1, f, tw_n2, tw_full, work, st);
  return nk_tl_launch<T, 32, 5, -1>(s1, n1, f, tw_n2, tw_full, work, st);
}
template int nk_tl_first_axis<float>(const NkPassS&, int, int, const NkFuse&, const C2<float>*, const C2<float>*, const C2<float>*,
                                     C2<float>*, hipStream_t);
template int nk_tl_first_axis<double>(const NkPassS&, int, int, const NkFuse&, const C2<double>*, const C2<double>*, const C2<double>*,
                                      C2<double>*, hipStream_t);
