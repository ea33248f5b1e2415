// nk_fft2.h -- register-resident, compile-time specialised line FFT passes (the fast path).
//
// Each thread keeps E complex elements of one line in registers for the whole pass; a line of N points is
// transformed by S <= 3 Stockham stages of radix R_s <= E done entirely in registers (recursive radix-2
// decimation in time with compile-time twiddles), with ONE LDS exchange between consecutive stages.  The
// exchange can be split into a real and an imaginary half so the LDS tile is N*TILE*sizeof(T) bytes (half of
// the complex tile): 1024 x 16 fp32 lines fit in 64 KiB -> two workgroups per CU.
//
// Stage s (Ns = R_1..R_{s-1}, R = R_s), butterfly j in [0, N/R):   (Stockham autosort, natural order in and out)
//     in   x[j + r N/R]              r = 0..R-1,   twiddled by W_{Ns R}^{(j mod Ns) r}
//     out  y[(j - j mod Ns) R + (j mod Ns) + r' Ns]
// Thread p of a line (P = N/E threads per line) owns butterflies j = p + q P, q = 0..E/R-1, so stage inputs
// are always rows p, p+P, ...: consecutive lanes touch consecutive rows (coalesced global loads in the first
// stage, conflict-free LDS reads in later ones).
//
// The body is written against an executor: on the GPU a phase is the code between two __syncthreads(); the
// test-only host emulation (tests/emu) runs every phase for all thread ids in turn with the per-thread
// registers kept in an array.
#pragma once
#include <type_traits>

#include "nk_fft_phases.h"

// ---------------------------------------------------------------------------------------------
// in-register DFT of compile-time size R <= 64 (natural order in / out), forward sign
// ---------------------------------------------------------------------------------------------
template <typename T>
struct TwConst {
  // cos(2 pi k / 64), sin(2 pi k / 64), k = 0..31
  static NK_HD T c(int k) {
    constexpr double v[32] = {1.0,
                              0.99518472667219688624,
                              0.98078528040323044913,
                              0.95694033573220886494,
                              0.92387953251128675613,
                              0.88192126434835502971,
                              0.83146961230254523708,
                              0.77301045336273696081,
                              0.70710678118654752440,
                              0.63439328416364549822,
                              0.55557023301960222474,
                              0.47139673682599764856,
                              0.38268343236508977173,
                              0.29028467725446236764,
                              0.19509032201612826785,
                              0.09801714032956060199,
                              0.0,
                              -0.09801714032956060199,
                              -0.19509032201612826785,
                              -0.29028467725446236764,
                              -0.38268343236508977173,
                              -0.47139673682599764856,
                              -0.55557023301960222474,
                              -0.63439328416364549822,
                              -0.70710678118654752440,
                              -0.77301045336273696081,
                              -0.83146961230254523708,
                              -0.88192126434835502971,
                              -0.92387953251128675613,
                              -0.95694033573220886494,
                              -0.98078528040323044913,
                              -0.99518472667219688624};
    return (T)v[k];
  }
  static NK_HD T s(int k) {
    constexpr double v[32] = {0.0,
                              0.09801714032956060199,
                              0.19509032201612826785,
                              0.29028467725446236764,
                              0.38268343236508977173,
                              0.47139673682599764856,
                              0.55557023301960222474,
                              0.63439328416364549822,
                              0.70710678118654752440,
                              0.77301045336273696081,
                              0.83146961230254523708,
                              0.88192126434835502971,
                              0.92387953251128675613,
                              0.95694033573220886494,
                              0.98078528040323044913,
                              0.99518472667219688624,
                              1.0,
                              0.99518472667219688624,
                              0.98078528040323044913,
                              0.95694033573220886494,
                              0.92387953251128675613,
                              0.88192126434835502971,
                              0.83146961230254523708,
                              0.77301045336273696081,
                              0.70710678118654752440,
                              0.63439328416364549822,
                              0.55557023301960222474,
                              0.47139673682599764856,
                              0.38268343236508977173,
                              0.29028467725446236764,
                              0.19509032201612826785,
                              0.09801714032956060199};
    return (T)v[k];
  }
};

// v[0..R-1] at stride STR inside a register array; result replaces the inputs in natural order.
template <typename T, int R>
struct RegDft {
  static NK_HD void run(C2<T>* v) {
    C2<T> ev[R / 2], od[R / 2];
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
      ev[i] = v[2 * i];
      od[i] = v[2 * i + 1];
    }
    RegDft<T, R / 2>::run(ev);
    RegDft<T, R / 2>::run(od);
#pragma unroll
    for (int k = 0; k < R / 2; ++k) {
      // w = exp(-2 pi i k / R) = (cos, -sin) with angle index k*(64/R)
      const int a = k * (64 / R);
      C2<T> t;
      if (a == 0) {
        t = od[k];
      } else if (a == 16) {
        t = C2<T>{od[k].y, -od[k].x};
      } else {
        const T c = TwConst<T>::c(a), s = TwConst<T>::s(a);
        t = C2<T>{od[k].x * c + od[k].y * s, od[k].y * c - od[k].x * s};
      }
      v[k] = C2<T>{ev[k].x + t.x, ev[k].y + t.y};
      v[k + R / 2] = C2<T>{ev[k].x - t.x, ev[k].y - t.y};
    }
  }
};
template <typename T>
struct RegDft<T, 1> {
  static NK_HD void run(C2<T>*) {}
};
template <typename T>
struct RegDft<T, 2> {
  static NK_HD void run(C2<T>* v) {
    const C2<T> a = v[0], b = v[1];
    v[0] = C2<T>{a.x + b.x, a.y + b.y};
    v[1] = C2<T>{a.x - b.x, a.y - b.y};
  }
};

// ---------------------------------------------------------------------------------------------
// compile-time schedules
// ---------------------------------------------------------------------------------------------
template <int N_, int E_, int S_, int R0_, int R1_, int R2_>
struct SchedDef {
  static constexpr int N = N_, E = E_, S = S_, R0 = R0_, R1 = R1_, R2 = R2_;
  static constexpr int P = N_ / E_;  // threads per line
  static_assert(R0_ * R1_ * R2_ == N_, "radices must multiply to N");
  static_assert(E_ % R0_ == 0 && E_ % R1_ == 0 && E_ % R2_ == 0, "radices must divide E");
  static constexpr int radix(int s) { return s == 0 ? R0_ : s == 1 ? R1_ : R2_; }
};
template <typename T, int N>
struct Sched;
#define NK_SCHED(TT, NN, EE, SS, A, B, C) \
  template <>                              \
  struct Sched<TT, NN> : SchedDef<NN, EE, SS, A, B, C> {}
NK_SCHED(float, 32, 8, 2, 8, 4, 1);  // (sub-lines of the two-level first-axis pass only)
NK_SCHED(float, 64, 8, 2, 8, 8, 1);
NK_SCHED(float, 128, 16, 2, 16, 8, 1);
NK_SCHED(float, 256, 16, 2, 16, 16, 1);
NK_SCHED(float, 512, 32, 2, 32, 16, 1);
NK_SCHED(float, 1024, 32, 2, 32, 32, 1);
NK_SCHED(float, 2048, 32, 3, 32, 32, 2);
NK_SCHED(float, 4096, 32, 3, 32, 32, 4);
NK_SCHED(double, 32, 8, 2, 8, 4, 1);
NK_SCHED(double, 64, 8, 2, 8, 8, 1);
NK_SCHED(double, 128, 16, 2, 16, 8, 1);
NK_SCHED(double, 256, 16, 2, 16, 16, 1);
NK_SCHED(double, 512, 16, 3, 16, 16, 2);
NK_SCHED(double, 1024, 16, 3, 16, 16, 4);
NK_SCHED(double, 2048, 16, 3, 16, 16, 8);
NK_SCHED(double, 4096, 16, 3, 16, 16, 16);
#undef NK_SCHED

// schedule of the plain IN-PLACE strided pass (MODE 0): 1024 fp32 lines as 256 threads x 64 elements (radix 64 x 16) --
// 250 VGPRs, 64 KiB of LDS, so TWO independent workgroups share a CU and one computes while the other waits for its
// rows (1024^3 fp32: middle-axis pass 1.80 -> 1.68 ms, first-axis pass 1.95 -> 1.89 ms).  The first pass with its
// prologue operands and the fused middle pass of the sandwich spill with 64 elements per thread and keep Sched.
template <typename T, int N>
struct SchedW : Sched<T, N> {};
#ifndef NK_NO_WIDE
template <>
struct SchedW<float, 1024> : SchedDef<1024, 64, 2, 64, 16, 1> {};
#endif
template <typename T, int N, int MODE>
using StridedSched = std::conditional_t<MODE == 0, SchedW<T, N>, Sched<T, N>>;

// schedule of the FINAL (contiguous) pass: E = 16 everywhere -- twice the threads per line pair, half the LDS
// tile per workgroup -> twice the resident waves for the latency-bound epilogue
template <typename T, int N>
struct SchedF : Sched<T, N> {};
template <>
struct SchedF<float, 512> : SchedDef<512, 16, 3, 16, 16, 2> {};
template <>
struct SchedF<float, 1024> : SchedDef<1024, 16, 3, 16, 16, 4> {};
template <>
struct SchedF<float, 2048> : SchedDef<2048, 16, 3, 16, 16, 8> {};
template <>
struct SchedF<float, 4096> : SchedDef<4096, 16, 3, 16, 16, 16> {};

// ---------------------------------------------------------------------------------------------
// LDS exchange planes (scalar T).
//   strided passes : plane[row][t], lanes run over the TILE columns first.  Rows written by neighbouring
//                    butterflies are R0 apart and would hit the same half of the bank period, so bit 0 of the
//                    row is flipped by bit log2(R0):  row' = row ^ ((row >> LOGR0) & 1)   (a bijection)
//   contiguous pass: plane[t][idx + (idx >> 5)], lanes run over the line elements first; the pad turns the
//                    stride-R0 write pattern into stride R0+1 and the pitch staggers the lines over the banks
// ---------------------------------------------------------------------------------------------
constexpr int nk_ilog2(int v) { return v <= 1 ? 0 : 1 + nk_ilog2(v / 2); }

template <int TILE, int LOGR0>
NK_HD int nk_xrow(int row, int t) {
  return ((row ^ ((row >> LOGR0) & 1)) * TILE) + t;
}

template <int N, int P>
struct ContigLayout {
  static constexpr int NPAD = N + (N >> 5);
  static constexpr int PITCH = NPAD + ((P % 32) - (NPAD % 32) + 32) % 32;
  static NK_HD int addr(int t, int idx) { return t * PITCH + idx + (idx >> 5); }
};

// tile choices of the fast path ------------------------------------------------------------------
// strided passes: rows of 128 B when the thread count (P*TILE <= 1024) and the LDS plane (<= 128 KiB) allow
// CX: the light first-pass classes exchange through ONE complex plane and read their twiddles from LDS (see below)
#ifndef NK_S1_TWO_WG
#define NK_S1_TWO_WG 0
#endif
template <int MODE, int PC>
constexpr bool nk_strided_cx() {
  return !NK_S1_TWO_WG && MODE == 3 && (PC == 0 || PC == 1 || PC == 6);
}
template <typename T, int N, bool CX = false, int MODE = 3>
struct StridedTile {
  using SC = StridedSched<T, N, MODE>;
  static constexpr int P = SC::P;
  static constexpr int want = (N <= 32 ? 256 : 128) / (2 * (int)sizeof(T));  // (32-point sub-lines: rows of 256 B fill a wavefront)
  static constexpr int by_threads = 1024 / P;
  static constexpr int by_lds = (128 * 1024) / (N * (int)sizeof(T));
  static constexpr int m1 = want < by_threads ? want : by_threads;
  static constexpr int TILE = m1 < by_lds ? m1 : by_lds;
  static constexpr int THREADS = P * TILE;
  // CPLX: exchange through ONE complex plane -- a single write + read round per stage boundary with 8-byte LDS
  // accesses (half the LDS instructions, one barrier instead of three, no tmp registers) -- where the workgroup is
  // alone on its CU anyway: the fp32 radix-32 kernels at 512 threads need 130-170 VGPRs.  Otherwise two half rounds
  // through one scalar plane (half the LDS: two workgroups per CU, the fp64 kernels).  Measured at 1024^3 fp32
  // (gpurun_out/r02c_probe.log): plain first pass 1.84 -> 1.76 ms; the in-place pass (1.95 -> 2.00 ms) and the
  // register-heavy JVP prologues (they spill with the complex plane: 3.3 -> 5.7 ms) keep the split exchange -> CX.
  static constexpr bool CPLX = CX && sizeof(T) == 4 && SC::E == 32 && P * TILE == 512 && N * TILE * 8 <= 128 * 1024;
  static constexpr int LDS_BYTES = N * TILE * (int)sizeof(T) * (CPLX ? 2 : 1);
  // TWLDS: the workgroup copies the axis' twiddle table (N complex) behind the exchange plane once and the stages read
  // it with ds_read instead of global loads.  The ~R twiddle loads per thread and stage are vector-memory instructions
  // with four distinct addresses per wave: they queue in the same address / L1 pipeline as the tile's own loads and
  // stores (tile copy with the pass's access pattern, tools/micro/copy_bench.hip: 5.81 TB/s, 5.33 with 31 such loads
  // per thread, 4.93 with the LDS exchange on top -- the measured speed of the pass).
  static constexpr int TW_BYTES = N * 2 * (int)sizeof(T);
  static constexpr bool TWLDS = CPLX && (LDS_BYTES + TW_BYTES) < 160 * 1024 && TW_BYTES % 1024 == 0;
  static constexpr int LDS_TOTAL = LDS_BYTES + (TWLDS ? TW_BYTES : 0);
};
// contiguous pass: ~256 threads, both planes within 48 KiB
template <typename T, int H>
struct ContigTile {
  static constexpr int P = Sched<T, H>::P;
  static constexpr int PITCH = ContigLayout<H, P>::PITCH;
  static constexpr int fit(int tile) {
    return (tile > 1 && (P * tile > 256 || 2 * tile * PITCH * (int)sizeof(T) > 48 * 1024)) ? fit(tile / 2) : tile;
  }
  static constexpr int TILE = fit(16);
  static constexpr int THREADS = P * TILE;
  static constexpr int LDS_BYTES = 2 * TILE * PITCH * (int)sizeof(T);
};

#ifndef NK_FINAL_LDS_KB
#define NK_FINAL_LDS_KB 20
#endif
// final pass of the strided-first pipeline: up to 256 threads and <= NK_FINAL_LDS_KB of planes (20 KiB: two line pairs
// per workgroup at 1024 fp32 -- many small workgroups hide the load/epilogue latency best)
// Every compile-time epilogue class runs best with HALF that tile while the workgroup keeps a full wavefront (one line
// pair per workgroup at 1024 fp32: 2.44 -> 2.17 ms for the multiply epilogue, 3.9 -> 3.6 ms for the VJP).
// MINT: the scatter/VJP kernels (COUPLES) need BOTH lines (b0, M - b0) of a couple in one workgroup -- the octant
// outputs (w8, field_octant) are indexed by the couple -- so their tile never drops below 2, whatever the caps say.
template <typename T, int NL, int EC = -1, int MINT = 1>
struct FinalTile {
  static constexpr int P = SchedF<T, NL>::P;
  static constexpr int PITCH = ContigLayout<NL, P>::PITCH;
  static constexpr int fit(int tile) {
    return (tile > MINT && (P * tile > 256 || 2 * tile * PITCH * (int)sizeof(T) > NK_FINAL_LDS_KB * 1024)) ? fit(tile / 2) : tile;
  }
  static constexpr int T0 = fit(16);
  // halve only while the workgroup keeps a full wavefront
  static constexpr int TILE = (EC >= 0 && T0 / 2 >= MINT && P * (T0 / 2) >= 64) ? T0 / 2 : T0;
  static constexpr int THREADS = P * TILE;
  static constexpr int LDS_BYTES = 2 * TILE * PITCH * (int)sizeof(T);
};

#ifndef NK_FINAL_SINGLE_2D_KB
#define NK_FINAL_SINGLE_2D_KB 60  // fp64 lines of 2048 / 4096 and fp32 lines of 4096 points: couple tiles of 68 / 135 KiB
#endif
// VJP final pass of 2-D grids on single line pairs instead of the couple tile, where the couple tile leaves one or two
// workgroups per CU (measured: 4096^2 fp64 195 -> 150 us per launch, C4 1.59 -> 1.64 it/s; 2048^2 fp64 C2 14.39 -> 14.76 it/s)
template <typename T, int NL>
constexpr bool nk_final_single_2d() {
  return FinalTile<T, NL, 2, 2>::LDS_BYTES > NK_FINAL_SINGLE_2D_KB * 1024;
}

// store of a streamed element (written once, read by a later kernel from HBM anyway): non-temporal stores keep the
// write stream out of the way of the reads -- in one bench step at 1024^3 fp32: first-pass family 2.53 -> 2.49 ms, in-place
// pass 2.03 -> 1.96 ms, final-pass family unchanged within noise, step -0.5 % (non-temporal LOADS of the streamed inputs
// measured slower: not used)
#ifndef NK_NT_STORE
#define NK_NT_STORE 1
#endif
template <typename T>
NK_HD void nk_store_stream(C2<T>* p, C2<T> v) {
#if !defined(NK_HOST_EMU) && NK_NT_STORE
  typedef T __attribute__((ext_vector_type(2))) V2;
  V2 w;
  w.x = v.x, w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<V2*>(p));
#else
  *p = v;
#endif
}

template <typename T>
NK_HD void nk_store_stream_s(T* p, T v) {
#if !defined(NK_HOST_EMU) && NK_NT_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// per-thread register file of one pass
template <typename T, int E>
struct PassRegs {
  C2<T> v[E];
  T tmp[E];
};

#ifndef NK_HOST_EMU
// device executor: a phase is the code between two workgroup barriers, registers live in `regs`
template <typename T, int E>
struct DeviceExec {
  PassRegs<T, E> regs;
  template <typename F>
  __device__ __forceinline__ void phase(F f) {
    f((int)threadIdx.x, regs);
    __syncthreads();
  }
  template <typename F>
  __device__ __forceinline__ void last_phase(F f) {
    f((int)threadIdx.x, regs);
  }
};
#endif

template <typename T, typename SC, int S>
struct StageInfo {
  static constexpr int R = SC::radix(S);
  static constexpr int NS = S == 0 ? 1 : S == 1 ? SC::R0 : SC::R0 * SC::R1;  // product of previous radices
  static constexpr int Q = SC::E / R;                                        // butterflies per thread
};

#ifndef NK_TW_GROUP
#define NK_TW_GROUP 0
#endif
// twiddle + in-register butterflies of stage S on v (inputs ordered v[q*R + r])
// TWC: compose the R - 1 twiddles of a butterfly from 7 + (R/8 - 1) table entries, w^r = w^(8 (r/8)) w^(r%8): one more
// complex multiplication per element but ~40 fewer live registers than R - 1 hoisted table loads (R >= 16)
template <typename T, typename SC, int S, bool TWC = false>
NK_HD void nk_stage_compute(C2<T>* v, int p, const C2<T>* __restrict__ tw) {
  using SI = StageInfo<T, SC, S>;
  constexpr int R = SI::R, NS = SI::NS, Q = SI::Q, N = SC::N, P = SC::P;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    if constexpr (TWC && NS > 1 && R >= 16) {
      const int k = (p + q * P) & (NS - 1);
      constexpr int step = N / (NS * R);
      C2<T> wl[8], wh[R / 8];
#pragma unroll
      for (int r = 1; r < 8; ++r) wl[r] = tw[(k * r * step) & (N - 1)];
#pragma unroll
      for (int h = 1; h < R / 8; ++h) wh[h] = tw[(k * 8 * h * step) & (N - 1)];
#pragma unroll
      for (int r = 1; r < R; ++r) {
        const int h = r / 8, l = r % 8;
        const C2<T> w = h == 0 ? wl[l] : (l == 0 ? wh[h] : cmul(wh[h], wl[l]));
        v[q * R + r] = cmul(v[q * R + r], w);
      }
    } else if (NS > 1) {
      const int k = (p + q * P) & (NS - 1);
      constexpr int step = N / (NS * R);
#pragma unroll
      for (int r = 1; r < R; ++r) {
        const C2<T> w = tw[(k * r * step) & (N - 1)];
        v[q * R + r] = cmul(v[q * R + r], w);
#if !defined(NK_HOST_EMU) && NK_TW_GROUP > 0
        // scheduling fence: at most NK_TW_GROUP twiddles in flight (all R - 1 hoisted loads cost 2 (R - 1) VGPRs)
        if (r % NK_TW_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    RegDft<T, R>::run(v + q * R);
  }
}

// row (= line element index) held in v[q*R + r'] AFTER stage S
template <typename SC, int S>
NK_HD int nk_out_row(int p, int q, int rp) {
  constexpr int R = SC::radix(S);
  constexpr int NS = S == 0 ? 1 : S == 1 ? SC::R0 : SC::R0 * SC::R1;
  const int j = p + q * SC::P;
  const int k = j & (NS - 1);
  return (j - k) * R + k + rp * NS;
}
// row that must be in v[q*R + r] BEFORE stage S
template <typename SC, int S>
NK_HD int nk_in_row(int p, int q, int r) {
  constexpr int R = SC::radix(S);
  return p + q * SC::P + r * (SC::N / R);
}

// exchange helpers for the strided layout: COMP 0 = real parts, 1 = imaginary parts
template <typename T, typename SC, int S, int TILE, int COMP>
NK_HD void nk_xwrite_cols(const C2<T>* v, T* plane, int pp, int t) {
  constexpr int R = SC::radix(S), Q = SC::E / R, L0 = nk_ilog2(SC::R0);
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) plane[nk_xrow<TILE, L0>(nk_out_row<SC, S>(pp, q, r), t)] = COMP ? v[q * R + r].y : v[q * R + r].x;
}
template <typename T, typename SC, int S, int TILE>
NK_HD void nk_xread_cols(T* dst, const T* plane, int pp, int t) {
  constexpr int R = SC::radix(S), Q = SC::E / R, L0 = nk_ilog2(SC::R0);
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) dst[q * R + r] = plane[nk_xrow<TILE, L0>(nk_in_row<SC, S>(pp, q, r), t)];
}

// the same exchange through a complex plane (StridedTile::CPLX): 8-byte accesses, rows of TILE * 8 bytes
template <typename T, typename SC, int S, int TILE>
NK_HD void nk_xwrite_c2(const C2<T>* v, C2<T>* plane, int pp, int t) {
  constexpr int R = SC::radix(S), Q = SC::E / R, L0 = nk_ilog2(SC::R0);
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) plane[nk_xrow<TILE, L0>(nk_out_row<SC, S>(pp, q, r), t)] = v[q * R + r];
}
template <typename T, typename SC, int S, int TILE>
NK_HD void nk_xread_c2(C2<T>* dst, const C2<T>* plane, int pp, int t) {
  constexpr int R = SC::radix(S), Q = SC::E / R, L0 = nk_ilog2(SC::R0);
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int r = 0; r < R; ++r) dst[q * R + r] = plane[nk_xrow<TILE, L0>(nk_in_row<SC, S>(pp, q, r), t)];
}

// XCD-aware block order of the first pass with OCTANT amplitude fields (3-D): the slabs a and A-a and, inside a
// slab, the tiles c and NL-c read the same octant lines.  Workgroups are dealt round-robin to the 8 XCDs
// (blockIdx % 8), each XCD keeps 2 * 32 of them in flight -- so every XCD is handed whole slab PAIRS (a, A-a),
// all their tiles back to back: the octant lines are fetched from HBM once and re-used out of that XCD's L2.
// A bijection of [0, nblocks); placement only matters for speed.
NK_HD int64_t nk_oct_block_remap(int64_t v, const NkPassS& p) {
  const int na = p.g.na, tiles = p.tiles_per_slab;
  if (p.g.ndim == 2 && tiles % 2 == 0) {
    // 2-D (the caller hands in the XCD-contiguous order): column tile j and its last-axis mirror tiles-1-j read the same
    // columns of the octant fields -- run them back to back, 0, T-1, 1, T-2, ...; neighbouring tiles stay two steps apart
    const int64_t bat = v / tiles;
    const int u = (int)(v - bat * tiles);
    return bat * tiles + ((u & 1) ? tiles - 1 - u / 2 : u / 2);
  }
  if (p.g.ndim != 3 || (na / 2) % 8 != 0) return v;
  const int64_t per = (int64_t)na * tiles;
  const int64_t bat = v / per;
  v -= bat * per;
  const int x = (int)(v % 8);
  const int64_t s = v / 8;
  const int64_t g = (s / (2 * tiles)) * 8 + x;  // slab pair handled by XCD x
  const int within = (int)(s % (2 * tiles));
  // inside the pair the four workgroups that read the same octant lines run back to back:
  // (a, j), (a, tiles-1-j), (A-a, j), (A-a, tiles-1-j)   (tiles even: the last-axis mirror of tile j is tile tiles-1-j)
  int sel = within / tiles, tile = within % tiles;
  if (tiles % 2 == 0) {
    const int quad = within / 4, m = within % 4;
    sel = m / 2;
    tile = (m % 2 == 0) ? quad : tiles - 1 - quad;
  }
  const int64_t idx = 2 * g + sel;   // slab sequence 0, A/2, 1, A-1, 2, A-2, ...
  const int64_t slab = idx == 0 ? 0 : idx == 1 ? na / 2 : ((idx & 1) ? na - idx / 2 : idx / 2);
  return bat * per + slab * tiles + tile;
}

// Twiddle table -> LDS without touching a VGPR: global_load_lds_dwordx4 moves 16 bytes per lane straight into LDS
// (wave w of the workgroup deposits 1 KiB per instruction at the wave-uniform LDS address base + offset); the
// workgroup barrier that follows waits for it (vmcnt) and publishes it.  Table sizes are multiples of 1 KiB
// (StridedTile::TWLDS).
template <typename T>
NK_HD void nk_tw_to_lds(const C2<T>* __restrict__ tw_global, C2<T>* tw_lds, int n, int tid, int nthreads) {
#ifndef NK_HOST_EMU
  const int bytes = n * (int)sizeof(C2<T>);
  const int wave = tid >> 6, lane = tid & 63;
  for (int off = wave * 1024; off < bytes; off += (nthreads >> 6) * 1024)
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(tw_global) + off + lane * 16,
                                     (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(tw_lds) + off), 16, 0, 0);
#else
  for (int i = tid; i < n; i += nthreads) tw_lds[i] = tw_global[i];
#endif
}

// XCD-contiguous block order: hardware deals workgroup b to XCD b % 8 (observed, MI355X_MICROARCH.md); with this
// remap the workgroups of one XCD take a CONTIGUOUS range of work items, so the tiles that share a 4 KiB row of the
// array (adjacent 128 B pieces) meet in one XCD's L2 / TLB instead of being spread over all eight.  A pure copy with
// the first pass's access pattern (1024 rows x 128 B at 4 KiB stride, 1024^3 fp32) runs at 4.77 TB/s in natural
// order and at 5.51 TB/s with this order (tools/micro/copy_bench.hip).  A bijection of [0, nb); speed only.
#ifndef NK_XMAP_DEFAULT
#define NK_XMAP_DEFAULT 9  // bit 0: first strided pass, bit 1: every in-place strided pass, bit 2: final pass, bit 3: the
                           // in-place middle-axis pass of the sandwich
#endif
NK_HD int64_t nk_xcd_contig(int64_t blk, int64_t nb) {
  const int64_t q = nb / 8, r = nb % 8;
  const int64_t x = blk % 8, i = blk / 8;
  return x * q + (x < r ? x : r) + i;
}

// ---------------------------------------------------------------------------------------------
#ifndef NK_TL_PLAIN_STORE
#define NK_TL_PLAIN_STORE 1  // bit 0 / 1: plain instead of non-temporal stores in the first / second launch of the two-level pass
#endif
#ifndef NK_STRIDED_TWC
#define NK_STRIDED_TWC 0  // experiment: composed twiddles in the strided passes (see nk_stage_compute)
#endif
// strided pass body (pass B: in place c2c; pass C: c2c + Hartley combine + epilogue)
// thread id -> column t = tid % TILE, line thread pp = tid / TILE;  blockDim = P * TILE
// LDS: ONE scalar plane of N*TILE elements (split real / imaginary exchange)
// ---------------------------------------------------------------------------------------------
// MODE 3: FIRST pass of the strided-first pipeline: the real input (through the fused prologue) is read as
//         complex pairs along the contiguous axis, transformed along this strided axis and written to `work`
// MODE 0: plain in-place c2c
// MODE 4 / 5: the TWO-LEVEL first-axis pass of 2-D grids (long fp64 lines, nk_tl_split): a line of F = N * p.sub points,
//         j = j1 * p.sub + j2, is transformed as N-point sub-lines over j1 (MODE 4: first pass with the fused prologue, rows
//         p.sub apart; the result X1(k1; j2) times the inter-level twiddle w_F^(j2 k1) goes to row j2 * N + k1 of `work`) and
//         then, with the roles of the factors swapped (MODE 5, in place: N = the other factor, sub-line k1 at rows
//         k1 + p.sub * j2), which leaves X(k1 + p.sub * k2) at row k1 + p.sub * k2: natural order for the final pass.  A tile is
//         N <= 64 rows of 128 ... 256 bytes -- 64 threads and 4 KiB of LDS instead of 1024 threads and 128 KiB that hold
//         4096 rows of 64 bytes -- so that many workgroups share a CU and one's arithmetic hides the other's loads.
template <typename T, int N, int TILE, int MODE, int PC, bool CX = false, typename Exec>
NK_HD void nk_strided_body(Exec& ex, const NkPassS& p, const NkFuse& f, int64_t blk, T* plane,
                           const C2<T>* __restrict__ tw_global, C2<T>* __restrict__ work, C2<T>* __restrict__ scratch,
                           double* acc_out, C2<T>* tw_lds = nullptr, const C2<T>* __restrict__ tw_full = nullptr) {
  using SC = StridedSched<T, N, MODE>;
  constexpr int E = SC::E, S = SC::S;
  constexpr bool FIRST = MODE == 3 || MODE == 4, TL = MODE == 4 || MODE == 5;
  constexpr bool CPLX = StridedTile<T, N, CX, MODE>::CPLX && TILE == StridedTile<T, N, CX, MODE>::TILE;
  // twiddles of the later stages: from the workgroup's LDS copy when the caller provides the room (StridedTile::TWLDS)
  const C2<T>* tw = tw_lds ? tw_lds : tw_global;
  [[maybe_unused]] C2<T>* cplane = reinterpret_cast<C2<T>*>(plane);
  if constexpr (FIRST && (PC == 4 || PC == 5 || PC == 9)) blk = nk_oct_block_remap(blk, p);
  int64_t o = blk / p.tiles_per_slab;
  const int64_t c0 = (blk % p.tiles_per_slab) * (int64_t)TILE;
  // two-level passes: o = (batch, sub-line); the sub-lines j2 and p.sub - j2 read the same rows of the octant amplitude fields
  // (their rows are mirror images): they run back to back, 0, sub/2, 1, sub-1, 2, sub-2, ...
  [[maybe_unused]] int tl_sub = 0;
  [[maybe_unused]] int64_t tl_bat = 0;
  if constexpr (TL) {
    tl_bat = o / p.sub;
    const int i = (int)(o - tl_bat * p.sub);
    tl_sub = i == 0 ? 0 : i == 1 ? p.sub / 2 : ((i & 1) ? p.sub - i / 2 : i / 2);
  }
  // MODE 3 reads the user array: line element j of column c0 at (o*N + j)*inner, and writes the work array.
  // MODE 0 runs in place on the work array.  Work layouts of the strided-first pipeline (3-D; p.ss = slab stride):
  //   blo == 0 : [batch][first] slabs of ss >= mid*last/2 elements, natural order inside
  //   blo == 1 : [batch][mid] slabs of ss >= first*last/2 elements, each [first][last/2]
  // (2-D, or ss == 0: plain natural layout)
  // Every address is split into a WAVE-UNIFORM 64-bit part (scalar registers) and a small per-thread 32-bit offset:
  // row(pp, q, r) = row(pp, 0, 0) + row(0, q, r) for both the input and the output order of a stage.
  int64_t in_off = o * N * p.inner + c0;
  int64_t rstride = p.inner;
  if constexpr (TL) {
    in_off = (tl_bat * N * p.sub + tl_sub) * p.inner + c0;
    rstride = (int64_t)p.sub * p.inner;
  }
  C2<T>* base = work + in_off;
  if (MODE == 0 && p.ss > 0) {
    if (p.blo > 0) {  // lines over the first axis, o = batch*mid + b
      base = work + o * p.ss + c0;
    } else {          // o = batch, columns run over (mid, last/2)
      rstride = p.ss;
      base = work + o * N * p.ss + c0;
    }
  }

  ex.phase([&](int tid, PassRegs<T, E>& rg) {
    const int t = tid % TILE, pp = tid / TILE;
    constexpr int R = SC::radix(0), Q = E / R;
    constexpr bool OCT = FIRST && (PC == 4 || PC == 5 || PC == 9);
    // octant amplitude fields [A/2+1][N/2+1][nl/2+1] (this axis is the middle one; A = 1 in 2-D): the two reals of
    // a pair sit at folded last-axis offsets c8a, c8b of the folded row
    [[maybe_unused]] uint32_t o8 = 0, ch = 0, c8 = 0;
    [[maybe_unused]] bool desc = false;
    if constexpr (OCT) {
      const int nl = p.g.nl, c = 2 * (int)(c0 + t);
      ch = nl / 2 + 1;
      desc = 2 * c >= nl;  // (c, c+1) -> (nl-c, nl-c-1): stored descending, lower position nl-c-1
      c8 = desc ? nl - c - 1 : c;
      o8 = p.g.ndim == 3 ? (uint32_t)nk_fold((int)(o % p.g.na), p.g.na) * (N / 2 + 1) : 0u;
    }
    const uint32_t toff = (uint32_t)(pp * rstride + t);  // thread part (< 2^31: checked by the host driver)
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t uoff = (int64_t)nk_in_row<SC, 0>(0, q, r) * rstride;  // uniform part
        if constexpr (OCT) {
          uint32_t j8;
          if constexpr (TL)  // row of the whole line: tl_sub + p.sub * (row of the sub-line)   (2-D: o8 == 0)
            j8 = (uint32_t)nk_fold(tl_sub + p.sub * nk_in_row<SC, 0>(pp, q, r), N * p.sub) * ch + c8;
          else
            j8 = (o8 + (uint32_t)nk_fold(nk_in_row<SC, 0>(pp, q, r), N)) * ch + c8;  // < 2^31 elements
          rg.v[q * R + r] = nk_prologue_oct<T, PC>(f, 2 * (in_off + uoff), 2 * toff, j8, desc);
        } else if (FIRST) {
          rg.v[q * R + r] = nk_prologue_ct<T, PC>(f, 2 * (in_off + uoff), 2 * toff);
        } else {
          rg.v[q * R + r] = (NK_NT_LOAD & 1) ? nk_ld_stream(&nk_at32<C2<T>>(base, uoff, toff)) : nk_at32<C2<T>>(base, uoff, toff);
        }
      }
    if (tw_lds) nk_tw_to_lds<T>(tw_global, tw_lds, N, tid, SC::P * TILE);  // published by this phase's barrier
    nk_stage_compute<T, SC, 0>(rg.v, pp, tw);
    if constexpr (CPLX)
      nk_xwrite_c2<T, SC, 0, TILE>(rg.v, cplane, pp, t);
    else
      nk_xwrite_cols<T, SC, 0, TILE, 0>(rg.v, plane, pp, t);
  });
  auto store_tile = [&](int tid, PassRegs<T, E>& rg) {
    const int t = tid % TILE, pp = tid / TILE;
    constexpr int LS = S - 1;
    constexpr int R = SC::radix(LS), Q = E / R;
    (void)scratch;
    (void)acc_out;
    // store: MODE 0 in place; MODE 3 into the work array -- natural slabs (o = batch*first + a, or 2-D) or, blo == 1,
    // [batch][this axis][first][last/2]: row b -> slab b, position a
    C2<T>* obase = base;
    int64_t ostride = rstride;
    if constexpr (MODE == 4) {
      // sub-line tl_sub -> the N consecutive rows tl_sub * N + k1, each value times w_F^(tl_sub k1), F = N * p.sub
      obase = work + (tl_bat * p.sub + tl_sub) * N * p.inner + c0;
      ostride = p.inner;
      const int fmask = N * p.sub - 1;
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int k1 = nk_out_row<SC, LS>(pp, q, r);
          rg.v[q * R + r] = cmul(rg.v[q * R + r], tw_full[(tl_sub * k1) & fmask]);
        }
    }
    if (MODE == 3 && p.ss > 0) {
      if (p.blo > 0) {
        const int64_t bat = o / p.g.na, a = o % p.g.na;
        obase = work + bat * N * p.ss + a * p.inner + c0;
        ostride = p.ss;
      } else {
        obase = work + o * p.ss + c0;
      }
    }
    const uint32_t toff = (uint32_t)(nk_out_row<SC, LS>(pp, 0, 0) * ostride + t);
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t uo = (int64_t)nk_out_row<SC, LS>(0, q, r) * ostride;
        if constexpr ((MODE == 4 && (NK_TL_PLAIN_STORE & 1)) || (MODE == 5 && (NK_TL_PLAIN_STORE & 2)))  // (the next launch reads these rows: leave them to the cache)
          *nk_ptr32<C2<T>>(obase, uo, toff) = rg.v[q * R + r];
        else if constexpr (S == 2)  // thread part = pp * ostride + t (small): 32-bit byte offset from a scalar base
          nk_store_stream(nk_ptr32<C2<T>>(obase, uo, toff), rg.v[q * R + r]);
        else
          nk_store_stream((obase + uo) + toff, rg.v[q * R + r]);
      }
  };
  if constexpr (CPLX) {
    // one round per stage boundary; the last stage's butterflies and the stores share the final phase (no barrier)
    if constexpr (S == 3) {
      ex.phase([&](int tid, PassRegs<T, E>& rg) {
        nk_xread_c2<T, SC, 1, TILE>(rg.v, cplane, tid / TILE, tid % TILE);
        nk_stage_compute<T, SC, 1, NK_STRIDED_TWC != 0>(rg.v, tid / TILE, tw);
      });
      ex.phase([&](int tid, PassRegs<T, E>& rg) { nk_xwrite_c2<T, SC, 1, TILE>(rg.v, cplane, tid / TILE, tid % TILE); });
    }
    ex.last_phase([&](int tid, PassRegs<T, E>& rg) {
      nk_xread_c2<T, SC, S - 1, TILE>(rg.v, cplane, tid / TILE, tid % TILE);
      nk_stage_compute<T, SC, S - 1, NK_STRIDED_TWC != 0>(rg.v, tid / TILE, tw);
      store_tile(tid, rg);
    });
    return;
  }
  // exchange 0 -> stage 1 (split: real parts, then imaginary parts through the same scalar plane)
  ex.phase([&](int tid, PassRegs<T, E>& rg) { nk_xread_cols<T, SC, 1, TILE>(rg.tmp, plane, tid / TILE, tid % TILE); });
  ex.phase([&](int tid, PassRegs<T, E>& rg) { nk_xwrite_cols<T, SC, 0, TILE, 1>(rg.v, plane, tid / TILE, tid % TILE); });
  if constexpr (S == 2) {
    ex.last_phase([&](int tid, PassRegs<T, E>& rg) {
      const int t = tid % TILE, pp = tid / TILE;
      T im[E];
      nk_xread_cols<T, SC, 1, TILE>(im, plane, pp, t);
#pragma unroll
      for (int e = 0; e < E; ++e) rg.v[e] = C2<T>{rg.tmp[e], im[e]};
      nk_stage_compute<T, SC, 1, NK_STRIDED_TWC != 0>(rg.v, pp, tw);
      store_tile(tid, rg);
    });
  } else {
    ex.phase([&](int tid, PassRegs<T, E>& rg) {
      const int t = tid % TILE, pp = tid / TILE;
      T im[E];
      nk_xread_cols<T, SC, 1, TILE>(im, plane, pp, t);
#pragma unroll
      for (int e = 0; e < E; ++e) rg.v[e] = C2<T>{rg.tmp[e], im[e]};
      nk_stage_compute<T, SC, 1, NK_STRIDED_TWC != 0>(rg.v, pp, tw);
    });
    ex.phase([&](int tid, PassRegs<T, E>& rg) { nk_xwrite_cols<T, SC, 1, TILE, 0>(rg.v, plane, tid / TILE, tid % TILE); });
    ex.phase([&](int tid, PassRegs<T, E>& rg) { nk_xread_cols<T, SC, 2, TILE>(rg.tmp, plane, tid / TILE, tid % TILE); });
    ex.phase([&](int tid, PassRegs<T, E>& rg) { nk_xwrite_cols<T, SC, 1, TILE, 1>(rg.v, plane, tid / TILE, tid % TILE); });
    ex.last_phase([&](int tid, PassRegs<T, E>& rg) {
      const int t = tid % TILE, pp = tid / TILE;
      T im[E];
      nk_xread_cols<T, SC, 2, TILE>(im, plane, pp, t);
#pragma unroll
      for (int e = 0; e < E; ++e) rg.v[e] = C2<T>{rg.tmp[e], im[e]};
      nk_stage_compute<T, SC, 2, NK_STRIDED_TWC != 0>(rg.v, pp, tw);
      store_tile(tid, rg);
    });
  }
}

// ---------------------------------------------------------------------------------------------
// contiguous pass body (pass A: real lines -> packed half spectrum;  IS_1D: full 1-D Hartley)
// thread id -> line thread pp = tid % P, line t = tid / P;  blockDim = P * TILE
// LDS: TWO scalar planes (re, im) of TILE * PITCH elements
// ---------------------------------------------------------------------------------------------
template <typename T, int H, int TILE, bool IS_1D, typename Exec>
NK_HD void nk_contig_body(Exec& ex, const NkPassA& p, const NkFuse& f, int64_t blk, T* planes,
                          const C2<T>* __restrict__ tw, const C2<T>* __restrict__ twr, C2<T>* __restrict__ work,
                          double* acc_out) {
  using SC = Sched<T, H>;
  using LY = ContigLayout<H, SC::P>;
  constexpr int E = SC::E, S = SC::S, P = SC::P;
  T* pre = planes;
  T* pim = planes + TILE * LY::PITCH;
  const int64_t line0 = blk * TILE;
  const int nl = p.g.nl;

  ex.phase([&](int tid, PassRegs<T, E>& rg) {
    const int pp = tid % P, t = tid / P;
    const int64_t line = line0 + t;
    constexpr int R = SC::radix(0), Q = E / R;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        C2<T> z{(T)0, (T)0};
        if (line < p.nlines) {
          z = nk_prologue_pair<T>(f, line * nl + 2 * (int64_t)nk_in_row<SC, 0>(pp, q, r));
        }
        rg.v[q * R + r] = z;
      }
    nk_stage_compute<T, SC, 0>(rg.v, pp, tw);
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int a = LY::addr(t, nk_out_row<SC, 0>(pp, q, r));
        pre[a] = rg.v[q * R + r].x;
        pim[a] = rg.v[q * R + r].y;
      }
  });
  // stage 1
  ex.phase([&](int tid, PassRegs<T, E>& rg) {
    const int pp = tid % P, t = tid / P;
    constexpr int R = SC::radix(1), Q = E / R;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int a = LY::addr(t, nk_in_row<SC, 1>(pp, q, r));
        rg.v[q * R + r] = C2<T>{pre[a], pim[a]};
      }
    nk_stage_compute<T, SC, 1>(rg.v, pp, tw);
  });
  ex.phase([&](int tid, PassRegs<T, E>& rg) {
    const int pp = tid % P, t = tid / P;
    constexpr int R = SC::radix(1), Q = E / R;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int a = LY::addr(t, nk_out_row<SC, 1>(pp, q, r));
        pre[a] = rg.v[q * R + r].x;
        pim[a] = rg.v[q * R + r].y;
      }
  });
  if constexpr (S == 3) {
    ex.phase([&](int tid, PassRegs<T, E>& rg) {
      const int pp = tid % P, t = tid / P;
      constexpr int R = SC::radix(2), Q = E / R;
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int a = LY::addr(t, nk_in_row<SC, 2>(pp, q, r));
          rg.v[q * R + r] = C2<T>{pre[a], pim[a]};
        }
      nk_stage_compute<T, SC, 2>(rg.v, pp, tw);
    });
    ex.phase([&](int tid, PassRegs<T, E>& rg) {
      const int pp = tid % P, t = tid / P;
      constexpr int R = SC::radix(2), Q = E / R;
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int a = LY::addr(t, nk_out_row<SC, 2>(pp, q, r));
          pre[a] = rg.v[q * R + r].x;
          pim[a] = rg.v[q * R + r].y;
        }
    });
  }
  // untangle (+ Hartley combine for 1-D) from the natural-order planes
  ex.last_phase([&](int tid, PassRegs<T, E>& rg) {
    (void)rg;
    constexpr int NK = H / 2 + 1;
    constexpr int NT = P * TILE;
    const T sg = (T)p.g.sign;
    double acc = 0.0;
    for (int idx = tid; idx < TILE * NK; idx += NT) {
      const int k = idx % NK, t = idx / NK;
      const int64_t line = line0 + t;
      if (line >= p.nlines) continue;
      if (k == 0) {
        const T zx = pre[LY::addr(t, 0)], zy = pim[LY::addr(t, 0)];
        if (IS_1D) {
          nk_epilogue<T>(f, line * nl, zx + zy, acc);
          nk_epilogue<T>(f, line * nl + H, zx - zy, acc);
        } else {
          work[line * H] = C2<T>{zx + zy, zx - zy};
        }
        continue;
      }
      const int a1 = LY::addr(t, k), a2 = LY::addr(t, H - k);
      const C2<T> Zk{pre[a1], pim[a1]}, Zm{pre[a2], pim[a2]};
      const C2<T> Ev{(T)0.5 * (Zk.x + Zm.x), (T)0.5 * (Zk.y - Zm.y)};
      const C2<T> Od{(T)0.5 * (Zk.x - Zm.x), (T)0.5 * (Zk.y + Zm.y)};
      const C2<T> G = cmul(twr[k], Od);
      const C2<T> Fk{Ev.x + G.y, Ev.y - G.x}, Fm{Ev.x - G.y, -Ev.y - G.x};
      if (IS_1D) {
        const int64_t ob = line * nl;
        nk_epilogue_pair<T>(f, ob + k, Fk.x + sg * Fk.y, ob + nl - k, Fk.x - sg * Fk.y, acc);
        if (k != H - k) nk_epilogue_pair<T>(f, ob + H - k, Fm.x + sg * Fm.y, ob + H + k, Fm.x - sg * Fm.y, acc);
      } else {
        work[line * H + k] = Fk;
        work[line * H + H - k] = Fm;
      }
    }
    *acc_out += acc;
  });
}

// |x| rounded UP to fp32: the running maximum of the octant sums is kept in one 32-bit register (an upper bound is all the
// fixed-point scatter needs; an fp64 maximum cost 6 % of the scatter pass and pushed the generic kernels over 128 VGPRs)
NK_HD float nk_abs_up(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __double2float_ru(fabs(x));
#else
  return nextafterf((float)fabs(x), INFINITY);
#endif
}
// running maximum of |x| that PROPAGATES NaN: for non-negative floats the order of the bit patterns is the order of the
// values, and every NaN pattern lies above +inf -- an unsigned integer maximum of the bits (one VALU instruction, like
// fmaxf) therefore carries a NaN or an infinity of the octant sums through to the scale of the fixed-point scatter,
// which then answers with NaN instead of rounding them to finite garbage
NK_HD float nk_wmax_join(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned int ua = __float_as_uint(a), ub = __float_as_uint(b);
  return __uint_as_float(ua > ub ? ua : ub);
#else
  return (a != a || b != b) ? NAN : (a > b ? a : b);
#endif
}
// EC 5 = EC 3 (likelihood) with FLOAT data / inverse-covariance / output arrays under a wider T (nk_fuse.io32)
template <int EC>
constexpr bool nk_ec_lh() {
  return EC == 3 || EC == 5;
}
template <typename T, int EC>
struct NkLhIo {
  typedef typename std::conditional<EC == 5, float, T>::type type;
};
// per-group constants of the final epilogue (one slot, or one couple of slots)
template <int NH>
struct FinalGroup {
  int64_t okh[NH], omh[NH];
  int base[NH], mlo[NH];
};

// compile-time specialised epilogue of the final pass (EC 0 affine, 1 multiply, 2 scatter/VJP with materialised
// amplitude field): per-launch constants hoisted into registers, arithmetic in the field type T (the bin sums of
// the scatter stay fp64), and -- the point of the exercise -- no per-output run-time switch
template <typename T>
struct FinalCt {
  T* out;
  const T *mul, *xi, *addend, *af;
  const T *c1, *c2;  // partial sums of other samples that join the running sum (nk_fuse.carry1 / carry2, nk_vjp_chain)
  T sc, off, asc;
  bool accum, dot;  // dot: also accumulate sum addend[o] * out[o] (the CG curvature d.(A d) when addend = d)
  bool keep;        // `out` is read again right away (first sample of a pair launch, k2_final2): plain stores, not streaming ones
  T* stash;         // pair launch with the hand-over in LDS (STASH != 0 below): this group's [NH][4][NL/2 + 1] values
};
template <typename T, int EC>
NK_HD FinalCt<T> nk_final_ct(const NkFuse& f, T* stash = nullptr) {
  FinalCt<T> c;
  c.out = (T*)f.out;
  c.mul = (const T*)f.mul;
  c.xi = (const T*)f.xi;
  c.addend = (const T*)f.addend;
  c.af = (const T*)f.afield;
  c.c1 = (const T*)f.carry1;
  c.c2 = (const T*)f.carry2;
  c.sc = (T)(f.scale * (EC == 1 ? f.mul_scalar : 1.0));
  c.off = (T)f.offset;
  c.asc = (T)f.addend_scale;
  c.accum = f.accumulate != 0;
  c.dot = EC == 2 && f.value != nullptr && f.addend != nullptr;
  c.keep = f.pipe_chunks < 0;  // host-side field otherwise; k2_final2 marks its first sample with -1
  c.stash = stash;
  return c;
}

#ifndef NK_FINAL_UNROLL
#define NK_FINAL_UNROLL 1
#endif
#ifndef NK_LH_UNROLL
#define NK_LH_UNROLL 4  // coefficients per trip of the likelihood epilogue's fast paths
#endif
// EC 2 (scatter / VJP) comes as LOAD (every operand of the four images: xi, the addend, the running sum in `out`) and
// APPLY (arithmetic, stores): nk_final_coeff issues the loads of both slots of a couple before the first store -- `out`
// may alias the addend, so no load could move above a store -- and nothing is consumed inside a conditional block (the ISA
// before the split: one or two loads, s_waitcnt vmcnt(0), branch, per image).  Loads of images that do not exist (self-
// paired line: om == ok, the addresses are valid) are issued regardless and their values dropped.
// MODE >= 0: bit 0 = there is an addend, bit 1 = `out` holds a running sum, bit 2 / 3 = carry1 / carry2 join the sum
// (nk_vjp_chain) -- compile-time, and every slot of the group
// active, so the load block has no branch at all (a load under a run-time condition merges with a constant zero, and the
// compiler moves the first use of the merged value up into the load's block: s_waitcnt per slot again).
// MODE < 0: the flags are read at run time and inactive slots skipped (edge groups, 2-D launches of the couple kernels).
template <typename T>
struct NkVjpOps {
  T x[4], d[4], o[4], p[4], q[4];  // xi, addend, running sum, carry1, carry2
};
// STASH (pair launch k2_final2, both samples run by the same threads on the same coefficients): 1 = sample A, whose output
// values go to the slot's LDS stash st[image * nkp] INSTEAD of memory; 2 / 3 = sample B, which takes them from there as its
// running sum (2: one shared `out`) or as its innermost partial sum carry1 (3) -- A's lines never travel to HBM and back.
// The values are the stored ones (rounded to T): the same bits as through memory.
template <typename T, bool BOTH, int MODE, int STASH = 0>
NK_HD NkVjpOps<T> nk_final_vjp_load(const FinalCt<T>& c, int64_t ok, int64_t om, int k2, int k2m, const T* st = nullptr,
                                    int nkp = 0) {
  NkVjpOps<T> v;
  const bool add = MODE < 0 ? c.addend != nullptr : (MODE & 1) != 0;
  const bool run = MODE < 0 ? c.accum : (MODE & 2) != 0;
  const bool k1 = MODE < 0 ? c.c1 != nullptr : (MODE & 4) != 0;
  const bool k2nd = MODE < 0 ? c.c2 != nullptr : (MODE & 8) != 0;
  // image i lives at row base (ok for i = 0, 2; om for i = 1, 3) + column (k2 for i = 0, 3; k2m for i = 1, 2): the row bases are
  // workgroup-uniform where a whole wave serves one slot (scalar registers), the column a 32-bit lane offset -- spelled as
  // base[column], not as one 64-bit sum per operand, so that the loads can take the scalar-base addressing form
  const int64_t rowb[4] = {ok, om, ok, om};
  const uint32_t colb[4] = {(uint32_t)k2, (uint32_t)k2m, (uint32_t)k2m, (uint32_t)k2};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool on = BOTH || i < 2;
    if constexpr (STASH == 3)
      v.p[i] = on ? st[i * nkp] : (T)0;
    else
      v.p[i] = (on && k1) ? (c.c1 + rowb[i])[colb[i]] : (T)0;
    v.q[i] = (on && k2nd) ? (c.c2 + rowb[i])[colb[i]] : (T)0;
    if constexpr ((NK_NT_LOAD & 8) != 0) {
      v.x[i] = on ? nk_ld_stream((c.xi + rowb[i]) + colb[i]) : (T)0;
      v.d[i] = (on && add) ? nk_ld_stream((c.addend + rowb[i]) + colb[i]) : (T)0;
      if constexpr (STASH == 2)
        v.o[i] = on ? st[i * nkp] : (T)0;
      else
        v.o[i] = (on && run) ? nk_ld_stream((c.out + rowb[i]) + colb[i]) : (T)0;
    } else {
      v.x[i] = on ? (c.xi + rowb[i])[colb[i]] : (T)0;
      v.d[i] = (on && add) ? (c.addend + rowb[i])[colb[i]] : (T)0;
      if constexpr (STASH == 2)
        v.o[i] = on ? st[i * nkp] : (T)0;
      else
        v.o[i] = (on && run) ? (c.out + rowb[i])[colb[i]] : (T)0;
    }
  }
  return v;
}
// returns the fp64 bin-sum contribution sum_images xi * t
template <typename T, bool BOTH, int MODE, int STASH = 0>
NK_HD double nk_final_vjp_apply(const FinalCt<T>& c, const NkVjpOps<T>& v, int64_t ok, int64_t om, bool self, T v0, T v1,
                                T v2, T v3, int k2, int k2m, T a, double& acc, T* st = nullptr, int nkp = 0) {
  const T t[4] = {v0 * c.sc, v1 * c.sc, v2 * c.sc, v3 * c.sc};
  const bool k1 = STASH == 3 || (MODE < 0 ? c.c1 != nullptr : (MODE & 4) != 0);
  const bool k2nd = MODE < 0 ? c.c2 != nullptr : (MODE & 8) != 0;
  T r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // the sample's own contribution g = fma(asc, d, a t) with a t rounded -- spelled out, so that every instantiation of this
    // epilogue (compile-time and run-time flags, single and pair launches) rounds alike; d is zero without an addend (an exact
    // no-op).  Then the partial sums, innermost first (nk_vjp_chain)
    r[i] = nk_settle(a * t[i]);
    r[i] = nk_settle(nk_fma(c.asc, v.d[i], r[i]));
    if (k1) r[i] = v.p[i] + r[i];
    if (k2nd) r[i] = v.q[i] + r[i];
    r[i] = v.o[i] + r[i];    // o is zero without a running sum
  }
  if (c.dot) {
    double e = (double)v.d[0] * (double)r[0];
    if (BOTH) e += (double)v.d[2] * (double)r[2];
    if (!self) e += (double)v.d[1] * (double)r[1] + (BOTH ? (double)v.d[3] * (double)r[3] : 0.0);
    acc += e;
  }
  T* outk = c.out + ok;
  T* outm = c.out + om;
  auto put = [&](T* p, T val, int image) {
    if constexpr (STASH == 1)
      st[image * nkp] = val;
    else if (c.keep)
      *p = val;
    else
      nk_store_stream_s(p, val);
  };
  put(outk + (uint32_t)k2, r[0], 0);
  if (BOTH) put(outk + (uint32_t)k2m, r[2], 2);
  double s = (double)v.x[0] * (double)t[0];
  if (BOTH) s += (double)v.x[2] * (double)t[2];
  if (!self) {
    put(outm + (uint32_t)k2m, r[1], 1);
    s += (double)v.x[1] * (double)t[1];
    if (BOTH) {
      put(outm + (uint32_t)k2, r[3], 3);
      s += (double)v.x[3] * (double)t[3];
    }
  }
  return s;
}

// the (up to) four images of one coefficient of ONE slot: H(k,kl) -> ok+k2, H(-k,-kl) -> om+k2m, and when BOTH
// (k2m != k2) H(k,-kl) -> ok+k2m, H(-k,kl) -> om+k2.  `self`: the line is its own partner (images 1, 3 coincide
// with 2, 0).  Returns the fp64 bin-sum contribution for EC 2.
template <typename T, int EC, bool BOTH, int MODE = -1>
NK_HD double nk_final_slot(const NkFuse& f, const FinalCt<T>& c, int64_t ok, int64_t om, bool self, T sg, T fx, T fy, T gx,
                           T gy, int k2, int k2m, T a, double& acc) {
  const T v0 = fx + sg * fy, v1 = fx - sg * fy, v2 = gx + sg * gy, v3 = gx - sg * gy;
  T* outk = c.out + ok;
  T* outm = c.out + om;
  if constexpr (nk_ec_lh<EC>()) {  // likelihood: the reference arithmetic (fp64) per output, energy terms returned
    double e = 0.0;
    const int64_t o4[4] = {ok + k2, om + k2m, ok + k2m, om + k2};
    const T v4[4] = {v0, v1, v2, v3};
    nk_epi_likelihood4<T, typename NkLhIo<T, EC>::type>(f, o4, v4, (self ? 1 : 3) | (BOTH ? (self ? 4 : 12) : 0), e);
    return e;
  } else if constexpr (EC == 0) {
    nk_store_stream_s(outk + k2, (T)(v0 * c.sc + c.off));
    if (BOTH) nk_store_stream_s(outk + k2m, (T)(v2 * c.sc + c.off));
    if (!self) {
      nk_store_stream_s(outm + k2m, (T)(v1 * c.sc + c.off));
      if (BOTH) nk_store_stream_s(outm + k2, (T)(v3 * c.sc + c.off));
    }
    return 0.0;
  } else if constexpr (EC == 1) {
    if (c.mul) {
      const T* mk = c.mul + ok;
      const T* mm = c.mul + om;
      const T m0 = mk[k2], m2 = BOTH ? mk[k2m] : (T)0;
      T m1 = (T)0, m3 = (T)0;
      if (!self) {
        m1 = mm[k2m];
        if (BOTH) m3 = mm[k2];
      }
      nk_store_stream_s(outk + k2, (T)(v0 * c.sc * m0));
      if (BOTH) nk_store_stream_s(outk + k2m, (T)(v2 * c.sc * m2));
      if (!self) {
        nk_store_stream_s(outm + k2m, (T)(v1 * c.sc * m1));
        if (BOTH) nk_store_stream_s(outm + k2, (T)(v3 * c.sc * m3));
      }
    } else {
      nk_store_stream_s(outk + k2, (T)(v0 * c.sc));
      if (BOTH) nk_store_stream_s(outk + k2m, (T)(v2 * c.sc));
      if (!self) {
        nk_store_stream_s(outm + k2m, (T)(v1 * c.sc));
        if (BOTH) nk_store_stream_s(outm + k2, (T)(v3 * c.sc));
      }
    }
    return 0.0;
  } else {
    static_assert(EC != 2, "the scatter epilogue runs through nk_final_vjp_load / nk_final_vjp_apply");
    return 0.0;
  }
}

// EC 2: all slots of a group for the coefficients k2[0 .. n_on-1] (k2[u >= n_on] repeat a valid one: loaded, not applied) --
// the loads of all U coefficients ahead of the first store
template <typename T, int NL, int NH, bool BOTH, int MODE, int U, int STASH = 0>
NK_HD void nk_final_vjp_coeffs(const NkFuse& f, const FinalCt<T>& c, const FinalGroup<NH>& gp, const T* pre, const T* pim, T sg,
                               const int (&k2s)[U], int n_on, int hv, const T* afline, double* w8line, double& acc,
                               float& wmax) {
  constexpr int NKP = NL / 2 + 1;  // stash: [NH][4 images][NKP] per group
  NkVjpOps<T> ops[U][NH];
  T a[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k2 = k2s[u], k2m = (NL - k2) & (NL - 1);
    a[u] = afline[k2];
#pragma unroll
    for (int h = 0; h < NH; ++h)
      if (MODE >= 0 || gp.mlo[h]) {
        if constexpr (STASH >= 2)
          ops[u][h] = nk_final_vjp_load<T, BOTH, MODE, STASH>(c, gp.okh[h], gp.omh[h], k2, k2m, c.stash + h * 4 * NKP + k2, NKP);
        else
          ops[u][h] = nk_final_vjp_load<T, BOTH, MODE>(c, gp.okh[h], gp.omh[h], k2, k2m);
      }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (u >= n_on) break;
    const int k2 = k2s[u], k2m = (NL - k2) & (NL - 1);
    const int d1 = k2 + (k2 >> 5), d2 = k2m + (k2m >> 5);
    double ssum = 0.0;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (MODE < 0 && !gp.mlo[h]) continue;
      const T fx = pre[gp.base[h] + d1], fy = pim[gp.base[h] + d1];
      const T gx = pre[gp.base[h] + d2], gy = pim[gp.base[h] + d2];
      if constexpr (STASH != 0)
        ssum += nk_final_vjp_apply<T, BOTH, MODE, STASH>(c, ops[u][h], gp.okh[h], gp.omh[h], gp.mlo[h] == 1, fx + sg * fy,
                                                         fx - sg * fy, gx + sg * gy, gx - sg * gy, k2, k2m, a[u], acc,
                                                         c.stash + h * 4 * NKP + k2, NKP);
      else
        ssum += nk_final_vjp_apply<T, BOTH, MODE>(c, ops[u][h], gp.okh[h], gp.omh[h], gp.mlo[h] == 1, fx + sg * fy, fx - sg * fy,
                                                  gx + sg * gy, gx - sg * gy, k2, k2m, a[u], acc);
    }
    if (w8line) {
      w8line[k2] = ssum;
      wmax = nk_wmax_join(wmax, nk_abs_up(ssum));
    } else {
      NK_VJP_SCATTER(f, f.pidx[gp.okh[hv] + k2], ssum);
    }
  }
}

// EC 3 with MODE = 1 (Gaussian, scalar N^-1) / 2 (Poissonian) and every slot a regular line pair: all slots of a group for
// the coefficients k2[0 .. n_on-1] -- the data loads of all U coefficients ahead of the first store (nk_lh4_load / _apply)
template <typename T, int NL, int NH, bool BOTH, int MODE, int U, typename TD = T>
NK_HD void nk_final_lh_coeffs(const NkFuse& f, const FinalGroup<NH>& gp, const T* pre, const T* pim, T sg, const int (&k2s)[U],
                              int n_on, double& acc) {
  constexpr int MASK = BOTH ? 15 : 3;
  typename NkLhData<TD, MODE>::type d[U][NH][4];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k2 = k2s[u], k2m = (NL - k2) & (NL - 1);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int64_t o4[4] = {gp.okh[h] + k2, gp.omh[h] + k2m, gp.okh[h] + k2m, gp.omh[h] + k2};
      nk_lh4_load<TD, MODE, MASK>(f, o4, d[u][h]);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (u >= n_on) break;
    const int k2 = k2s[u], k2m = (NL - k2) & (NL - 1);
    const int d1 = k2 + (k2 >> 5), d2 = k2m + (k2m >> 5);
    double ssum = 0.0;  // summed like the run-time path: images of a slot, slots of the group, then the thread's total
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const T fx = pre[gp.base[h] + d1], fy = pim[gp.base[h] + d1];
      const T gx = pre[gp.base[h] + d2], gy = pim[gp.base[h] + d2];
      const int64_t o4[4] = {gp.okh[h] + k2, gp.omh[h] + k2m, gp.okh[h] + k2m, gp.omh[h] + k2};
      const T v4[4] = {fx + sg * fy, fx - sg * fy, gx + sg * gy, gx - sg * gy};
      double e = 0.0;
      nk_lh4_apply<T, MODE, MASK, TD>(f, o4, v4, d[u][h], e);
      ssum += e;
    }
    acc += ssum;
  }
}

// all slots of a group for coefficient k2
template <typename T, int NL, int NH, int EC, bool BOTH, int MODE = -1, int STASH = 0>
NK_HD void nk_final_coeff(const NkFuse& f, const FinalCt<T>& c, const FinalGroup<NH>& gp, const T* pre, const T* pim, T sg,
                          int k2, int hv, const T* afline, double* w8line, double& acc, float& wmax) {
  if constexpr (EC == 2) {
    const int one[1] = {k2};
    nk_final_vjp_coeffs<T, NL, NH, BOTH, MODE, 1, STASH>(f, c, gp, pre, pim, sg, one, 1, hv, afline, w8line, acc, wmax);
  } else {
    const int k2m = (NL - k2) & (NL - 1);
    const int d1 = k2 + (k2 >> 5), d2 = k2m + (k2m >> 5);
    double ssum = 0.0;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      if (MODE <= 0 && !gp.mlo[h]) continue;
      const T fx = pre[gp.base[h] + d1], fy = pim[gp.base[h] + d1];
      const T gx = pre[gp.base[h] + d2], gy = pim[gp.base[h] + d2];
      ssum += nk_final_slot<T, EC, BOTH, MODE>(f, c, gp.okh[h], gp.omh[h], gp.mlo[h] == 1, sg, fx, fy, gx, gy, k2, k2m, (T)0,
                                               acc);
    }
    if constexpr (nk_ec_lh<EC>()) acc += ssum;
  }
}

// ---------------------------------------------------------------------------------------------
// FINAL pass of the strided-first pipeline (contiguous axis).
// After the strided passes  Z_j(k) = A_j(k) + i B_j(k)  with A_j = FFT_strided(x[.., 2j]), B_j = FFT_strided(x[.., 2j+1])
// (both Hermitian in the strided wave vector k).  For the line pair (k, -k):
//     G(k, 2j)   = A_j(k) = (Z_j(k) + conj Z_j(-k)) / 2
//     G(k, 2j+1) = B_j(k) = (Z_j(k) - conj Z_j(-k)) / (2i)
// ONE complex FFT of length nl over n_last gives F(k, k_last); the partner line follows from
// F(-k, k_last) = conj F(k, -k_last), so  H(k, kl) = Re F + s Im F  and  H(-k, -kl) = Re F - s Im F  are both
// written as whole contiguous rows (no scattered mirror segments, no packed column).
// Lines are (a, b): a on the first axis (A values, 1 for 2-D), b on the middle axis (M values).
// thread id -> line thread pp = tid % P, pair slot t = tid / P;  LDS: two planes of TILE * PITCH
// ---------------------------------------------------------------------------------------------
struct NkPassF {
  NkGeom g;
  int A, M;          // line index space (first, middle)
  int tiles_per_a;   // M / TILE
  int blo;           // work array layout, see nk_strided_body
  int64_t ss;        // slab stride (0: plain natural layout)
  int64_t rs;        // row stride of the work array in complex elements (0: nl / 2)
  int a0, a_cnt;     // one pipeline stage of the sandwich: first-axis pair indices a0 .. a0 + a_cnt - 1 (a_cnt == 0: all)
  int64_t blk0;      // first workgroup of the stage (= a0 * tiles_per_a, set by the launcher)
};

// EC: compile-time epilogue class (0 affine, 1 multiply, 2 scatter/VJP with materialised amplitude field,
// 3 likelihood, 5 likelihood with float arrays under T = double (nk_fuse.io32), -1 generic)
// PAIR: how the two work rows (k, -k) encode the line X(k, .) of length NL that is transformed here
//   0  even/odd columns (strided-first pipeline, above):  Z_j(k) = X(k, 2j) + i X(k, 2j+1)
//   1  row-mirror pairing of the sandwich pipeline (nk_fft3.h): rows of NL/2 + 1 columns,
//        G_c(k) = X(k, c) + i X(-k, NL - c),  X(-k, c) = conj X(k, c)   (x real)
//      =>  2 X(k, c)      = G_c(k) + conj G_c(-k)                  c = 0 .. NL/2
//          2 X(k, NL - c) = conj( (G_c(k) - conj G_c(-k)) / i )    c = 1 .. NL/2 - 1
// STASH / stash: see nk_final_vjp_load (k2_final2 only; TILE * 4 * (NL/2 + 1) values behind the planes)
template <typename T, int NL, int TILE, bool COUPLES, int EC, int PAIR = 0, int STASH = 0, typename Exec>
NK_HD void nk_final_body(Exec& ex, const NkPassF& p, const NkFuse& f_in, int64_t blk, T* planes,
                         const C2<T>* __restrict__ tw, const C2<T>* __restrict__ work, double* acc_out,
                         float* wmax_out = nullptr, T* stash = nullptr) {
  // the line FFT runs on 2A / 2B (see the load phase): fold the 1/2 into the output scale every epilogue applies first
  NkFuse f = f_in;
  f.scale = 0.5 * f_in.scale;
  using SC = SchedF<T, NL>;
  using LY = ContigLayout<NL, SC::P>;
  constexpr int E = SC::E, S = SC::S, P = SC::P, H = NL / 2;
  T* pre = planes;
  T* pim = planes + TILE * LY::PITCH;
  const int A = p.A, M = p.M;
  const int bt0 = (int)(blk % p.tiles_per_a);
  const int64_t r0 = blk / p.tiles_per_a;
  const int a = (int)(r0 % (A / 2 + 1));
  const int64_t bat = r0 / (A / 2 + 1);
  const int am = a ? A - a : 0;
  const bool a_self = (a == am);

  // slot -> line.  3-D (A > 1): slots come in couples (b0, M - b0) so that one epilogue work item owns all EIGHT
  // sign-flip images of a coefficient (one scatter atomic per eight outputs); 2-D: consecutive lines.
  const bool couples = COUPLES && (A > 1) && (TILE >= 2);
  auto line_of = [&](int t, int& b, int& bm, bool& active, bool& self) {
    if (couples) {
      const int b0 = bt0 * (TILE / 2) + (t >> 1);
      const int flip = t & 1;
      const int b0m = b0 ? M - b0 : 0;
      b = flip ? b0m : b0;
      bm = flip ? b0 : b0m;
      active = b0 <= M / 2 && !(flip && (b0m == b0 || a_self));
      self = a_self && b == bm;
    } else {
      b = bt0 * TILE + t;
      bm = b ? M - b : 0;
      active = b < M && !(a_self && b > bm);
      self = a_self && b == bm;
    }
  };

  ex.phase([&](int tid, PassRegs<T, E>& rg) {
    const int pp = tid % P;
    int t = tid / P;
    if constexpr (P % 64 == 0) t = nk_uniform(t);
    int b, bm;
    bool active, self;
    line_of(t, b, bm, active, self);
    const int64_t rs = p.rs > 0 ? p.rs : (int64_t)H;
    auto line_at = [&](int aa, int bb) {
      if (p.ss > 0 && p.blo > 0) return work + (bat * M + bb) * p.ss + (int64_t)aa * rs;
      if (p.ss > 0) return work + (bat * A + aa) * p.ss + (int64_t)bb * rs;
      return work + ((bat * A + aa) * M + bb) * rs;
    };
    const C2<T>* lk = line_at(a, b);
    const C2<T>* lm = line_at(am, bm);
    constexpr int R = SC::radix(0), Q = E / R;
    // stage-0 rows of a thread are pp + multiples of P (P even): the parity of the element index is a per-thread
    // constant.  even: 2A = Zk + conj Zm, odd: 2B = -i (Zk - conj Zm); the factor 1/2 is folded into the output scale
    const bool odd = pp & 1;
    if (active) {
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int n2 = nk_in_row<SC, 0>(pp, q, r);
          if constexpr (PAIR == 1) {
            const bool up = n2 > H;
            const int c = up ? NL - n2 : n2;
            const C2<T> Zk = (NK_NT_LOAD & 4) ? nk_ld_stream(lk + c) : lk[c], Zm = (NK_NT_LOAD & 4) ? nk_ld_stream(lm + c) : lm[c];
            const T c0 = up ? Zk.y : Zk.x, c1 = up ? Zm.y : Zm.x, c2 = up ? Zk.x : Zk.y, c3 = up ? Zm.x : Zm.y;
            rg.v[q * R + r] = C2<T>{c0 + c1, c2 - c3};
          } else {
            const C2<T> Zk = (NK_NT_LOAD & 4) ? nk_ld_stream(lk + (n2 >> 1)) : lk[n2 >> 1],
                        Zm = (NK_NT_LOAD & 4) ? nk_ld_stream(lm + (n2 >> 1)) : lm[n2 >> 1];
            const T c0 = odd ? Zk.y : Zk.x, c1 = odd ? Zm.y : Zm.x, c2 = odd ? Zm.x : Zk.y, c3 = odd ? Zk.x : Zm.y;
            rg.v[q * R + r] = C2<T>{c0 + c1, c2 - c3};
          }
        }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) rg.v[e] = C2<T>{(T)0, (T)0};
    }
    nk_stage_compute<T, SC, 0>(rg.v, pp, tw);
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ad = LY::addr(t, nk_out_row<SC, 0>(pp, q, r));
        pre[ad] = rg.v[q * R + r].x;
        pim[ad] = rg.v[q * R + r].y;
      }
  });
  ex.phase([&](int tid, PassRegs<T, E>& rg) {
    const int pp = tid % P, t = tid / P;
    constexpr int R = SC::radix(1), Q = E / R;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ad = LY::addr(t, nk_in_row<SC, 1>(pp, q, r));
        rg.v[q * R + r] = C2<T>{pre[ad], pim[ad]};
      }
    nk_stage_compute<T, SC, 1>(rg.v, pp, tw);
  });
  if constexpr (S == 3) {
    ex.phase([&](int tid, PassRegs<T, E>& rg) {
      const int pp = tid % P, t = tid / P;
      constexpr int R = SC::radix(1), Q = E / R;
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ad = LY::addr(t, nk_out_row<SC, 1>(pp, q, r));
          pre[ad] = rg.v[q * R + r].x;
          pim[ad] = rg.v[q * R + r].y;
        }
    });
    ex.phase([&](int tid, PassRegs<T, E>& rg) {
      const int pp = tid % P, t = tid / P;
      constexpr int R = SC::radix(2), Q = E / R;
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int ad = LY::addr(t, nk_in_row<SC, 2>(pp, q, r));
          rg.v[q * R + r] = C2<T>{pre[ad], pim[ad]};
        }
      nk_stage_compute<T, SC, 2>(rg.v, pp, tw);
    });
  }
  // natural-order spectrum back to the planes, then an LDS-driven epilogue loop: each work item owns the four
  // mirror images (k,kl), (k,-kl), (-k,-kl), (-k,kl) of one coefficient pair -- they share their power bin, so the
  // VJP scatter needs ONE fp64 atomic per four outputs; rows are written contiguously (ascending / descending)
  ex.phase([&](int tid, PassRegs<T, E>& rg) {
    const int pp = tid % P, t = tid / P;
    constexpr int LS = S - 1;
    constexpr int R = SC::radix(LS), Q = E / R;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ad = LY::addr(t, nk_out_row<SC, LS>(pp, q, r));
        pre[ad] = rg.v[q * R + r].x;
        pim[ad] = rg.v[q * R + r].y;
      }
  });
  ex.last_phase([&](int tid, PassRegs<T, E>& rg) {
    (void)rg;
    constexpr int NK = NL / 2 + 1;
    constexpr int NT = P * TILE;
    const T sg = (T)p.g.sign;
    double acc = 0.0;
    float wmax = 0.0f;
    // a fixed group of threads serves one slot (2-D) / one couple of slots (3-D): all line bookkeeping is hoisted
    // out of the k_last loop, which then only advances by the group width
    const int nslot = couples ? TILE / 2 : TILE;
    const int tps = NT / nslot;
    int u = tid / tps;  // tps is P or 2P: whole waves when P is a multiple of 64
    if constexpr (P % 64 == 0) u = nk_uniform(u);
    const int lane = tid - u * tps;
    constexpr int NH = COUPLES ? 2 : 1;
    FinalGroup<NH> gp;  // mlo per slot: 0 inactive, 1 self-paired line, 3 regular pair
    int any = 0;
    // octant slot of this group for the optional w8 output (VJP): [batch][a][b0][k_last]
    double* w8line = nullptr;
    const int b0g = couples ? bt0 * (TILE / 2) + u : bt0 * TILE + u;
    if (f.w8 && f.epi == NK_EPI_VJP)
      w8line = f.w8 + ((bat * (A / 2 + 1) + a) * (int64_t)(M / 2 + 1) + b0g) * (NL / 2 + 1);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int t = couples ? 2 * u + h : u;
      int b, bm;
      bool active, self;
      line_of(t, b, bm, active, self);
      if (h == 1 && !couples) active = false;
      gp.okh[h] = (((bat * A + a) * M + b)) * (int64_t)NL;
      gp.omh[h] = (((bat * A + am) * M + bm)) * (int64_t)NL;
      gp.base[h] = LY::addr(t, 0);
      gp.mlo[h] = active ? (self ? 1 : 3) : 0;
      any |= gp.mlo[h];
    }
    if constexpr (EC >= 0) {
      if (any) {
        const FinalCt<T> c = nk_final_ct<T, EC>(f, STASH != 0 ? stash + u * (NH * 4 * (NL / 2 + 1)) : nullptr);
        int hv = 0;
#pragma unroll
        for (int h = NH - 1; h >= 0; --h)
          if (gp.mlo[h]) hv = h;
        // amplitude value of the group's coefficients: a full field line, or (field_octant) the octant line (a, b0)
        const T* afline = nullptr;
        if constexpr (EC == 2)
          afline = f.field_octant ? c.af + ((int64_t)a * (M / 2 + 1) + b0g) * (NL / 2 + 1) : c.af + gp.okh[hv];
        // k_last = 0 and NL/2 are their own mirrors; everything in between has four distinct images per slot
        auto coefficients = [&](auto mode_c) {
          constexpr int MODE = decltype(mode_c)::value;
          if (lane < 2)
            nk_final_coeff<T, NL, NH, EC, false, MODE, STASH>(f, c, gp, pre, pim, sg, lane ? NL / 2 : 0, hv, afline, w8line, acc, wmax);
          if constexpr (nk_ec_lh<EC>() && MODE > 0) {
            constexpr int U = sizeof(T) == 8 ? (NK_LH_UNROLL + 1) / 2 : NK_LH_UNROLL;  // fp64: 168 VGPRs + 116 B / lane of spills with four
            for (int k2 = 1 + lane; k2 < NL / 2; k2 += U * tps) {
              int ks[U], n_on = 0;
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int k = k2 + u * tps;
                const bool on = k < NL / 2;
                ks[u] = on ? k : k2;
                n_on += on;
              }
              nk_final_lh_coeffs<T, NL, NH, true, MODE, U, typename NkLhIo<T, EC>::type>(f, gp, pre, pim, sg, ks, n_on, acc);
            }
          } else if constexpr (EC == 2 && MODE >= 0 && NK_FINAL_UNROLL > 1) {
            constexpr int U = NK_FINAL_UNROLL;
            for (int k2 = 1 + lane; k2 < NL / 2; k2 += U * tps) {
              int ks[U], n_on = 0;
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int k = k2 + u * tps;
                const bool on = k < NL / 2;
                ks[u] = on ? k : k2;
                n_on += on;
              }
              nk_final_vjp_coeffs<T, NL, NH, true, MODE, U, STASH>(f, c, gp, pre, pim, sg, ks, n_on, hv, afline, w8line, acc, wmax);
            }
          } else {
            for (int k2 = 1 + lane; k2 < NL / 2; k2 += tps)
              nk_final_coeff<T, NL, NH, EC, true, MODE, STASH>(f, c, gp, pre, pim, sg, k2, hv, afline, w8line, acc, wmax);
          }
        };
        if constexpr (EC == 2) {  // see nk_final_vjp_load
          bool full = true;
#pragma unroll
          for (int h = 0; h < NH; ++h) full = full && gp.mlo[h] != 0;
          switch (full ? (c.addend ? 1 : 0) | (c.accum ? 2 : 0) | (c.c1 ? 4 : 0) | (c.c2 ? 8 : 0) : -1) {
            case 0: coefficients(std::integral_constant<int, 0>{}); break;
            case 1: coefficients(std::integral_constant<int, 1>{}); break;
            case 2: coefficients(std::integral_constant<int, 2>{}); break;
            case 3: coefficients(std::integral_constant<int, 3>{}); break;
            case 6: coefficients(std::integral_constant<int, 6>{}); break;    // the sample that closes a subtree of four
            case 14: coefficients(std::integral_constant<int, 14>{}); break;  // ... of eight (pairwise sum over samples)
            default: coefficients(std::integral_constant<int, -1>{}); break;
          }
        } else if constexpr (nk_ec_lh<EC>()) {  // see nk_epi_likelihood4
          bool regular = true;
#pragma unroll
          for (int h = 0; h < NH; ++h) regular = regular && gp.mlo[h] == 3;
          const bool gauss = f.lh_kind == NK_LH_GAUSS;
          switch (regular && !(gauss && f.icov) ? (gauss ? 1 : 2) : -1) {
            case 1: coefficients(std::integral_constant<int, 1>{}); break;
            case 2: coefficients(std::integral_constant<int, 2>{}); break;
            default: coefficients(std::integral_constant<int, -1>{}); break;
          }
        } else {
          coefficients(std::integral_constant<int, -1>{});
        }
      }
    } else if (any) {
      const bool vjp = f.epi == NK_EPI_VJP;
      for (int k2 = lane; k2 < NK; k2 += tps) {
        const int k2m = k2 ? NL - k2 : 0;
        const int d1 = k2 + (k2 >> 5), d2 = k2m + (k2m >> 5);
        if (vjp) {
          // one slot (four images) at a time keeps the register footprint of the scatter epilogue small
          int hv = 0;
#pragma unroll
          for (int h = NH - 1; h >= 0; --h)
            if (gp.mlo[h]) hv = h;
          int32_t bin = 0;
          double a;
          if (f.afield) {
            a = (double)((const T*)f.afield)[gp.okh[hv] + k2];
            if (!w8line) bin = f.pidx[gp.okh[hv] + k2];
          } else {
            bin = f.pidx[gp.okh[hv] + k2];
            a = f.amp[bin];
          }
          double ssum = 0.0;
#pragma unroll
          for (int h = 0; h < NH; ++h) {
            if (!gp.mlo[h]) continue;
            const T fx = pre[gp.base[h] + d1], fy = pim[gp.base[h] + d1];
            const T gx = pre[gp.base[h] + d2], gy = pim[gp.base[h] + d2];
            const int64_t o4[4] = {gp.okh[h] + k2, gp.omh[h] + k2m, gp.okh[h] + k2m, gp.omh[h] + k2};
            const T v4[4] = {fx + sg * fy, fx - sg * fy, gx + sg * gy, gx - sg * gy};
            const int m4 = gp.mlo[h] == 1 ? (k2m != k2 ? 5 : 1) : (k2m != k2 ? 15 : 3);
            ssum += nk_vjp_quad<T>(f, o4, v4, m4, a);
          }
          if (w8line) {
            w8line[k2] = ssum;
            wmax = nk_wmax_join(wmax, nk_abs_up(ssum));
          } else {
            NK_VJP_SCATTER(f, bin, ssum);
          }
          continue;
        }
        int64_t o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        T v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int mask = 0;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const T fx = pre[gp.base[h] + d1], fy = pim[gp.base[h] + d1];
          const T gx = pre[gp.base[h] + d2], gy = pim[gp.base[h] + d2];
          o[4 * h + 0] = gp.okh[h] + k2, v[4 * h + 0] = fx + sg * fy;   // H(k, kl)
          o[4 * h + 1] = gp.omh[h] + k2m, v[4 * h + 1] = fx - sg * fy;  // H(-k, -kl)
          o[4 * h + 2] = gp.okh[h] + k2m, v[4 * h + 2] = gx + sg * gy;  // H(k, -kl)
          o[4 * h + 3] = gp.omh[h] + k2, v[4 * h + 3] = gx - sg * gy;   // H(-k, kl)
          const int m4 = gp.mlo[h] == 0 ? 0 : (gp.mlo[h] == 1 ? (k2m != k2 ? 5 : 1) : (k2m != k2 ? 15 : 3));
          mask |= m4 << (4 * h);
        }
        nk_epilogue_multi<T, 4 * NH>(f, o, v, mask, acc);
      }
    }
    *acc_out += acc;
    if (wmax_out) *wmax_out = nk_wmax_join(*wmax_out, wmax);
  });
}

// ---------------------------------------------------------------------------------------------
// eligibility of the fast path (shared by the HIP driver and the host emulation)
// ---------------------------------------------------------------------------------------------
// (experimental builds may pass a shorter list, e.g. -D'NK_FAST_SIZES(X)=X(1024)': ~6x faster to compile)
#ifndef NK_FAST_SIZES
#define NK_FAST_SIZES(X) X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096)
#endif

static inline bool nk_fast_size(int n) {
#define NK_CASE(NN) \
  if (n == NN) return true;
  NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
  return false;
}
template <typename T>
static inline int nk_fast_strided_tile(int n) {
  switch (n) {
#define NK_CASE(NN) \
  case NN:          \
    return StridedTile<T, NN>::TILE > StridedTile<T, NN, false, 0>::TILE ? StridedTile<T, NN>::TILE : StridedTile<T, NN, false, 0>::TILE;
    NK_FAST_SIZES(NK_CASE)
#undef NK_CASE
    default:
      return 0;
  }
}
template <typename T>
static inline bool nk_fast_strided_ok(int n, int64_t inner) {
  if (!nk_fast_size(n)) return false;
  const int tile = nk_fast_strided_tile<T>(n);
  return tile > 0 && inner % tile == 0;
}
static inline bool nk_fast_contig_ok(int h) { return nk_fast_size(h); }

// ---------------------------------------------------------------------------------------------
// pass parameters of the strided-first pipeline (shared by the HIP driver and the host emulation)
// ---------------------------------------------------------------------------------------------
#include "nk_plan.h"
// Two-level first-axis pass of 2-D grids (nk_strided_body MODE 4 / 5): n = n1 * n2, n1-point sub-lines in the first launch,
// n2-point ones in the second.  The single-kernel pass holds a whole line per workgroup: at 4096 fp64 points that is 1024 threads
// with rows of 64 bytes and ONE workgroup per CU (the register file admits 4 columns), whose loads, arithmetic and stores do
// not overlap: 112 us per plain pass at 4096^2 where its access pattern alone takes 57 (tools/micro/col_tile_bench.hip).
// Measured, round 6 (rocprofv3 on tools/gpu_fused_probe.py 4096,4096 f64; profiles/r06_two_level_*): two launches of 64-row
// tiles take 54 + 48 = 102 us plain (-9 %), 109 + 48 against 177 with the JVP prologue (-11 %), 69 + 48 against 125 with the
// amplitude prologue -- each launch streams at the 5 TB/s of an HBM copy, the second one partly out of the Infinity Cache
// when the first leaves its rows there (plain instead of non-temporal stores: 63 -> 54 us for the first, 52 -> 48 us for the
// second launch; NK_TL_PLAIN_STORE).  At 2048 fp64 points (rows of 128 bytes already, 32 MiB fields that live in the cache)
// the second launch costs more than the overlap gains: 29 against 22 us per member -- single kernel kept.  Default: fp64
// lines of 4096 points; NK_TWO_LEVEL=0: never, =2: also 2048 (fp64) and 4096 (fp32).  The two schedules round differently.
template <typename T>
static inline bool nk_tl_split(const NkGeom& g, int& n1, int& n2) {
#ifdef NK_HOST_EMU
  const int on = nk_env_int("NK_TWO_LEVEL", 1);  // (tests switch it per call)
#else
  static const int on = nk_env_int("NK_TWO_LEVEL", 1);
#endif
  n1 = n2 = 0;
  if (!on || g.ndim != 2) return false;
  if (g.na == 4096 && (sizeof(T) == 8 || on >= 2)) n1 = 64, n2 = 64;
  else if (g.na == 2048 && sizeof(T) == 8 && on >= 2) n1 = 64, n2 = 32;
  return n1 > 0;
}
#define NK_WORK_PAD_MAX 8192  // elements; the plan's workspace reserves this much per slab
#include "nk_fft3.h"
struct NkPipe2 {
  NkPassS s1, s0;  // first pass (fused prologue); second, in-place pass (3-D only)
  NkPassF pf;
};
static inline NkPipe2 nk_pipe2_setup(const NkHostPlan& hp, int sign, int blo, int pad) {
  NkPipe2 q{};
  const NkGeom& g = hp.g;
  if (pad < 0) pad = 0;
  if (pad > NK_WORK_PAD_MAX) pad = NK_WORK_PAD_MAX;
  if (g.ndim == 3) {
    blo = blo > 0 ? 1 : 0;
    q.s1 = hp.pb;
    q.s0 = hp.pc;
    const int64_t ss = blo ? (int64_t)g.na * g.h + pad : (int64_t)g.nm * g.h + pad;
    q.s1.blo = q.s0.blo = q.pf.blo = blo;
    q.s1.ss = q.s0.ss = q.pf.ss = ss;
    if (blo) {  // second pass: one line per (batch, mid) and last-axis tile, rows `h` apart
      q.s0.outer = (int64_t)g.batch * g.nm;
      q.s0.inner = g.h;
    }
  } else {
    q.s1 = hp.pc;
  }
  q.pf.g = g;
  q.pf.g.sign = sign;
  q.pf.A = g.ndim == 3 ? g.na : 1;
  q.pf.M = g.ndim == 3 ? g.nm : g.na;
  return q;
}
