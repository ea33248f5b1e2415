// nk_fft_p.hip -- nk_hartley_sandwich_pair's final pass: the final passes of TWO samples in one launch (k2_final2) and its
// launcher; see nk_fft_batch.h.  Same phase functions as the single kernels of nk_fft.hip (nk_fft2.h); a translation unit
// of its own because every instantiation inlines the heavy final-pass body twice.
#include <hip/hip_runtime.h>

#include "nk_fft_batch.h"

// TWO sandwich final passes (scatter class, row-mirror pairing) in ONE launch: every workgroup runs the final pass of its
// line pairs for sample A and then for sample B, whose epilogue joins A's lines to its sum.  Both bodies are the same code on
// the same workgroup index, so the thread that produced an output value of A is the one that needs it for B.
//   HAND = 0  A's lines go through memory: plain stores by A (FinalCt::keep), B's loads meet them in the cache hierarchy --
//             in principle; at 1024^3 the launch moves 39.4 GB where two single launches move 40.8: they do NOT stay.
//   HAND = 2  A's lines wait in an LDS stash behind the transform planes -- [group][slot][image][NL/2 + 1] values, each
//             written and read by the same thread -- and B takes them from there as its running sum (one shared `out`):
//             A's store and B's load of the whole array never happen.
//   HAND = 3  the same with B taking them as its innermost partial sum (carry1 == A's out, B's own `out` holding an older
//             partial sum: the pairwise order over samples).  A's `out` is then not written at all.
// The stashed values are the stored ones, rounded to T: the same bits as through memory (tests/test_engine_gpu.py::
// test_pair_final_pass_is_bit_identical passes with either).  Two inlined bodies, no loop around the phases.
//
// MEASURED (round 5, 1024^3 fp32, profiles/r05c_pair_hand_over.txt): the hand-over removes 8.6 of 39.4 GB per launch and
// LOSES: 7.19 ms against 6.96.  The stash doubles the workgroup's LDS (16.9 -> 33.3 KB: four resident workgroups per CU
// instead of six), and the final pass is more sensitive to its occupancy than to its bytes -- the HAND = 0 kernel launched
// with the same LDS footprint (NK_PAIR_PAD_LDS) takes 8.1 ms, so at EQUAL occupancy the hand-over gains 8 %, and the
// occupancy costs 16 %.  Compiling for two waves per SIMD (no spills) or loading two coefficients ahead changes nothing
// (all within 1 % of each other).  The variants are therefore only built with -DNK_PAIR_HAND_BUILD=1 and selected with
// NK_PAIR_HAND=1; the product runs HAND = 0.
#ifndef NK_PAIR_HAND_BUILD
#define NK_PAIR_HAND_BUILD 0
#endif
template <typename T, int NL>
struct PairTile {
  using CT = FinalTile<T, NL, 2, 2>;
  static constexpr int STASH_BYTES = CT::TILE * 4 * (NL / 2 + 1) * (int)sizeof(T);
  static constexpr int LDS_HAND = CT::LDS_BYTES + STASH_BYTES;
  // the stash doubles the LDS of a workgroup: worth it while enough workgroups stay resident to keep the loads in flight
  static constexpr bool HAND_OK = NK_PAIR_HAND_BUILD != 0 && LDS_HAND <= NK_PAIR_HAND_LDS_KB * 1024;
};

#ifndef NK_PAIR_HAND_WAVES
#define NK_PAIR_HAND_WAVES 3  // wavefronts per SIMD the hand-over variants are compiled for
#endif
template <typename T, int NL, int HAND>
__global__ void __launch_bounds__((FinalTile<T, NL, 2, 2>::THREADS),
                                  (HAND ? (FinalTile<T, NL, 2, 2>::THREADS > 256 ? 1 : NK_PAIR_HAND_WAVES)
                                        : nk_final_waves<T, true, 2, FinalTile<T, NL, 2, 2>::THREADS>()))
    k2_final2(NkPassF p, NkFuse fa, NkFuse fb, const C2<T>* __restrict__ tw, const C2<T>* __restrict__ worka,
              const C2<T>* __restrict__ workb) {
  extern __shared__ __align__(16) unsigned char smem[];
  using CT = FinalTile<T, NL, 2, 2>;
  DeviceExec<T, SchedF<T, NL>::E> ex;
  const int64_t blk = (int64_t)blockIdx.x;
  T* stash = HAND ? (T*)(smem + CT::LDS_BYTES) : nullptr;
  fa.pipe_chunks = -1;  // HAND = 0: A's output lines are read again by B below, keep them in the cache hierarchy (FinalCt::keep)
  {
    double acc = 0.0;
    float wmax = 0.0f;
    nk_final_body<T, NL, CT::TILE, true, 2, 1, HAND ? 1 : 0>(ex, p, fa, blk, (T*)smem, tw, worka, &acc, &wmax, stash);
    nk_flush_energy(fa, acc, smem);
    nk_flush_wmax(fa, wmax);
  }
  __syncthreads();  // A's lines are out (HAND = 0: visible to this workgroup) before B's body reuses the planes
  {
    double acc = 0.0;
    float wmax = 0.0f;
    nk_final_body<T, NL, CT::TILE, true, 2, 1, HAND>(ex, p, fb, blk, (T*)smem, tw, workb, &acc, &wmax, stash);
    nk_flush_energy(fb, acc, smem);
    nk_flush_wmax(fb, wmax);
  }
}

template <typename T, int NL, int HAND>
static int nk_launch_final_pair_h(NkPassF pf, const NkFuse& fa, const NkFuse& fb, const C2<T>* tw, const C2<T>* worka,
                                  const C2<T>* workb, hipStream_t st) {
  using CT = FinalTile<T, NL, 2, 2>;
  static const int pad = nk_env_int("NK_PAIR_PAD_LDS", 0);  // experiment: occupancy of the HAND = 0 launch at the stash's LDS cost
  const int LDS = (HAND ? PairTile<T, NL>::LDS_HAND : CT::LDS_BYTES) + (HAND ? 0 : pad);
  auto kern = k2_final2<T, NL, HAND>;
  static unsigned long long attr_mask = 0;  // per-device attribute
  if (LDS > 64 * 1024 && nk_first_on_device(attr_mask)) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return nk_set_hip_error(e, "hipFuncSetAttribute(k2_final2)");
  }
  pf.tiles_per_a = (pf.A > 1 && CT::TILE >= 2) ? (pf.M / 2 + 1 + CT::TILE / 2 - 1) / (CT::TILE / 2) : (pf.M + CT::TILE - 1) / CT::TILE;
  pf.blk0 = 0;
  const int64_t blocks = (int64_t)pf.g.batch * (pf.A / 2 + 1) * pf.tiles_per_a;
  const int64_t waves = blocks * ((CT::THREADS + 63) / 64);
  if ((fa.value_slots > 0 && waves > fa.value_slots) || (fb.value_slots > 0 && waves > fb.value_slots))
    return nk_set_error(NK_ERR_RUNTIME, "final pass: more wavefronts than reduction slots (nk_value_slot_count)");
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(CT::THREADS), LDS, st, pf, fa, fb, tw, worka, workb);
  return nk_check_launch("k2_final2");
}

template <typename T, int NL>
int nk_launch_final_pair(NkPassF pf, const NkFuse& fa, const NkFuse& fb, const C2<T>* tw, const C2<T>* worka, const C2<T>* workb,
                         hipStream_t st) {
  static const int hand = nk_env_int("NK_PAIR_HAND", 0);
  // 3-D line couples only (the launcher's caller guarantees a 3-D plan; A > 1 is what makes the bodies pair lines as couples)
  if constexpr (PairTile<T, NL>::HAND_OK) {
    if (hand && pf.A > 1) {
      if (fa.out == fb.out && fb.accumulate) return nk_launch_final_pair_h<T, NL, 2>(pf, fa, fb, tw, worka, workb, st);
      if (fb.carry1 == fa.out && fb.out != fa.out) return nk_launch_final_pair_h<T, NL, 3>(pf, fa, fb, tw, worka, workb, st);
    }
  }
  return nk_launch_final_pair_h<T, NL, 0>(pf, fa, fb, tw, worka, workb, st);
}

#define NK_CASE(NN)                                                                                                             \
  template int nk_launch_final_pair<float, NN>(NkPassF, const NkFuse&, const NkFuse&, const C2<float>*, const C2<float>*,        \
                                               const C2<float>*, hipStream_t);                                                  \
  template int nk_launch_final_pair<double, NN>(NkPassF, const NkFuse&, const NkFuse&, const C2<double>*, const C2<double>*,     \
                                                const C2<double>*, hipStream_t);
NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
