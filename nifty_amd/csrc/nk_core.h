// nk_core.h -- device-side building blocks of libniftyk (gfx950 / MI355X).
//
// Every kernel in this library is written as a sequence of PHASES separated by workgroup
// barriers.  A phase is a plain function  phase(tid, nthreads, block, lds, params)  that loops
// over its work items with stride `nthreads`.  On the GPU the kernel calls the phases with
// __syncthreads() in between; the test-only host emulation (tests/emu, -DNK_HOST_EMU) calls the
// same phase for tid = 0..nthreads-1 sequentially.  That lets the index math, digit reversal,
// twiddles and the fused prologue/epilogue be checked bit-for-bit on a CPU-only box.  The
// emulation is never linked into the product library.
#pragma once
#include <stdint.h>
#include <type_traits>
#include "../../include/niftyk.h"

#ifdef NK_HOST_EMU
#include <cmath>
struct int2 {
  int x, y;
};
#define NK_HD inline
#define NK_PIN(x) (void)(x)
#define NK_ATOMIC_ADD(p, v) (*(p) += (v))
#define NK_ATOMIC_ADD_XCD(p, v) (*(p) += (v))
static inline int nk_xcc_id() { return 0; }
static inline int nk_uniform(int v) { return v; }
#else
#include <hip/hip_runtime.h>
#define NK_HD __host__ __device__ __forceinline__
// the value exists in a register HERE, and no memory access moves across this point (orders register-only arithmetic
// against the loads of a following batch; the optimiser is otherwise free to sink it below them)
#ifdef __HIP_DEVICE_COMPILE__
#define NK_PIN(x) asm volatile("" : "+v"(x)::"memory")
#else
#define NK_PIN(x) (void)(x)
#endif
#define NK_ATOMIC_ADD(p, v) atomicAdd((p), (v))
// accumulator private to one XCD: every contributor sits behind the same L2, so workgroup scope (an L2 atomic
// without the memory-side round trip of agent scope) is sufficient -- see nk_epilogue VJP
#define NK_ATOMIC_ADD_XCD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
__device__ __forceinline__ int nk_xcc_id() {
  return (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7);  // HW_REG_XCC_ID
}
// value known to be identical in all lanes of a wave: move it to a scalar register (scalar address arithmetic,
// scalar branches); identity on the host
__host__ __device__ __forceinline__ int nk_uniform(int v) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}
#endif

// VJP scatter target: one accumulator, or one per XCD selected by the hardware XCC id at run time (placement
// independent: whatever XCD a workgroup lands on, it only ever touches that XCD's private copy)
#define NK_VJP_SCATTER(f, p, val)                                                   \
  do {                                                                              \
    if ((f).abar_copies > 1)                                                        \
      NK_ATOMIC_ADD_XCD((f).abar + (int64_t)nk_xcc_id() * (f).abar_stride + (p), (val)); \
    else                                                                            \
      NK_ATOMIC_ADD((f).abar + (p), (val));                                         \
  } while (0)

// ... or, with nk_fuse.wfull, no accumulation at all: the contribution lands at its own grid point (the partners of a
// work item write 0) and the caller sums the bins in a fixed order
#define NK_VJP_SCATTER_AT(f, o, p, val)     \
  do {                                      \
    if ((f).wfull)                          \
      (f).wfull[(o)] = (val);               \
    else                                    \
      NK_VJP_SCATTER(f, p, val);            \
  } while (0)

#define NK_MAX_STAGES 8

template <typename T>
struct alignas(2 * sizeof(T)) C2 {
  T x, y;
};

template <typename T>
NK_HD C2<T> cmul(C2<T> a, C2<T> b) {
  return C2<T>{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T>
NK_HD C2<T> cadd(C2<T> a, C2<T> b) {
  return C2<T>{a.x + b.x, a.y + b.y};
}
template <typename T>
NK_HD C2<T> csub(C2<T> a, C2<T> b) {
  return C2<T>{a.x - b.x, a.y - b.y};
}
// multiply by -i  (forward-transform rotation)
template <typename T>
NK_HD C2<T> cmul_mi(C2<T> a) {
  return C2<T>{a.y, -a.x};
}

// ---------------------------------------------------------------------------------------------
// line FFT plan: in-place decimation-in-frequency, mixed radix {8,4,2}; result is left in
// digit-reversed order and un-reversed by the store phase.
// ---------------------------------------------------------------------------------------------
// Division of tile-local indices by a run-time constant: floor(x / d) = (x * ceil(2^32 / d)) >> 32, exact while
// x * d < 2^32 (indices inside an LDS tile are < 2^16, divisors < 2^13).  The generic kernels decompose an index per
// element and stage (line / position, butterfly / sub-block, the digits of the reversal): as hardware integer divisions
// (~40 instructions each) that index arithmetic was most of their run time (mixed-radix grids at 1/5 of the fast path's
// bandwidth); a multiply-high is one instruction.
struct NkDiv {
  uint32_t d, inv;  // inv == 0: d == 1
};
static inline NkDiv nk_make_div(int64_t d) {
  NkDiv v;
  v.d = (uint32_t)(d < 1 ? 1 : d);
  v.inv = v.d == 1 ? 0u : (uint32_t)(((uint64_t)1 << 32) / v.d + ((((uint64_t)1 << 32) % v.d) ? 1 : 0));
  return v;
}
NK_HD uint32_t nk_fdiv(uint32_t x, const NkDiv& v) { return v.inv ? (uint32_t)(((uint64_t)x * v.inv) >> 32) : x; }
// x = q * d + r
NK_HD void nk_fdivmod(uint32_t x, const NkDiv& v, int& q, int& r) {
  const uint32_t qq = nk_fdiv(x, v);
  q = (int)qq;
  r = (int)(x - qq * v.d);
}

struct NkLinePlan {
  int n;                       // complex line length (product of radices 8, 4, 2, 3, 5, 7; >= 1)
  int nstage;                  // number of DIF stages
  int radix[NK_MAX_STAGES];    // radices in execution order
  int span[NK_MAX_STAGES];     // n / (radix[0] * ... * radix[s]): sub-block length AFTER stage s (= Lr of stage s)
  NkDiv dradix[NK_MAX_STAGES]; // / radix[s]
  NkDiv dspan[NK_MAX_STAGES];  // / span[s]
  NkDiv dnbf[NK_MAX_STAGES];   // / (n / radix[s]): butterflies per line of stage s
};

NK_HD int nk_digit_reverse(const NkLinePlan& lp, int k) {
  int p = 0;
  for (int s = 0; s < lp.nstage; ++s) {
    int q, r;
    nk_fdivmod((uint32_t)k, lp.dradix[s], q, r);
    p += r * lp.span[s];
    k = q;
  }
  return p;
}

// LDS tile layout.  t_fastest: element (pos, t) at pos*tstride + t        (strided-axis tiles)
//                   else     : element (pos, t) at t*lstride + pos + (pos >> 4)  (contiguous lines,
//                              one pad slot per 16 elements keeps small-stride stages conflict-free)
struct NkTile {
  int tile;        // lines per workgroup
  int t_fastest;   // 1: lanes run over t first
  int tstride;     // t_fastest: row pitch (>= tile)
  int lstride;     // !t_fastest: line pitch (>= n + n/16)
  NkDiv dtile;     // / tile
};

NK_HD int nk_lds_addr(const NkTile& tl, int pos, int t) {
  return tl.t_fastest ? pos * tl.tstride + t : t * tl.lstride + pos + (pos >> 4);
}

template <typename T, int R>
struct Butterfly;

template <typename T>
struct Butterfly<T, 2> {
  static NK_HD void run(C2<T>* v) {
    C2<T> a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  }
};

template <typename T>
struct Butterfly<T, 4> {
  static NK_HD void run(C2<T>* v) {
    C2<T> a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]);
    C2<T> b0 = cadd(v[1], v[3]), b1 = cmul_mi(csub(v[1], v[3]));
    v[0] = cadd(a0, b0);
    v[2] = csub(a0, b0);
    v[1] = cadd(a1, b1);
    v[3] = csub(a1, b1);
  }
};

template <typename T>
struct Butterfly<T, 8> {
  static NK_HD void run(C2<T>* v) {
    const T h = (T)0.70710678118654752440;
    // layer 1 (pairs q, q+4)
    C2<T> a[8];
    for (int q = 0; q < 4; ++q) {
      a[q] = cadd(v[q], v[q + 4]);
      a[q + 4] = csub(v[q], v[q + 4]);
    }
    // twiddle the odd half by w8^q
    a[5] = C2<T>{(a[5].x + a[5].y) * h, (a[5].y - a[5].x) * h};
    a[6] = cmul_mi(a[6]);
    a[7] = C2<T>{(a[7].y - a[7].x) * h, -(a[7].x + a[7].y) * h};
    // two radix-4 butterflies on (a0..a3) -> even outputs, (a4..a7) -> odd outputs
    C2<T> e[4] = {a[0], a[1], a[2], a[3]};
    C2<T> o[4] = {a[4], a[5], a[6], a[7]};
    Butterfly<T, 4>::run(e);
    Butterfly<T, 4>::run(o);
    for (int q = 0; q < 4; ++q) {
      v[2 * q] = e[q];
      v[2 * q + 1] = o[q];
    }
  }
};

// odd radices 3, 5, 7: plain O(R^2) DFT with compile-time roots of unity (natural order in and out)
template <int R>
struct NkRoot;
template <>
struct NkRoot<3> {
  static constexpr double c[3] = {1.0, -0.5, -0.5};
  static constexpr double s[3] = {0.0, 0.86602540378443864676, -0.86602540378443864676};
};
template <>
struct NkRoot<5> {
  static constexpr double c[5] = {1.0, 0.30901699437494742410, -0.80901699437494742410, -0.80901699437494742410,
                                  0.30901699437494742410};
  static constexpr double s[5] = {0.0, 0.95105651629515357212, 0.58778525229247312917, -0.58778525229247312917,
                                  -0.95105651629515357212};
};
template <>
struct NkRoot<7> {
  static constexpr double c[7] = {1.0, 0.62348980185873353053, -0.22252093395631440429, -0.90096886790241912624,
                                  -0.90096886790241912624, -0.22252093395631440429, 0.62348980185873353053};
  static constexpr double s[7] = {0.0, 0.78183148246802980871, 0.97492791218182360702, 0.43388373911755812048,
                                  -0.43388373911755812048, -0.97492791218182360702, -0.78183148246802980871};
};
template <typename T, int R>
struct ButterflyOdd {
  static NK_HD void run(C2<T>* v) {
    C2<T> o[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      C2<T> acc = v[0];
#pragma unroll
      for (int r = 1; r < R; ++r) {
        const int k = (q * r) % R;  // w_R^{qr} = cos(2 pi k/R) - i sin(2 pi k/R)
        const T c = (T)NkRoot<R>::c[k], sn = (T)NkRoot<R>::s[k];
        acc.x += v[r].x * c + v[r].y * sn;
        acc.y += v[r].y * c - v[r].x * sn;
      }
      o[q] = acc;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = o[q];
  }
};
template <typename T>
struct Butterfly<T, 3> : ButterflyOdd<T, 3> {};
template <typename T>
struct Butterfly<T, 5> : ButterflyOdd<T, 5> {};
template <typename T>
struct Butterfly<T, 7> : ButterflyOdd<T, 7> {};

// run-time radix -> compile-time stage
#define NK_STAGE_DISPATCH(RADIX, ...)                       \
  switch (RADIX) {                                          \
    case 8: nk_dif_stage<T, 8>(__VA_ARGS__); break;         \
    case 4: nk_dif_stage<T, 4>(__VA_ARGS__); break;         \
    case 2: nk_dif_stage<T, 2>(__VA_ARGS__); break;         \
    case 3: nk_dif_stage<T, 3>(__VA_ARGS__); break;         \
    case 5: nk_dif_stage<T, 5>(__VA_ARGS__); break;         \
    default: nk_dif_stage<T, 7>(__VA_ARGS__); break;        \
  }

// one in-place DIF stage of radix R on sub-blocks of length L for all `tile` lines in LDS
template <typename T, int R>
NK_HD void nk_dif_stage(C2<T>* lds, int tid, int nthr, const NkLinePlan& lp, const NkTile& tl, int L,
                        const C2<T>* __restrict__ tw, int stage) {
  const int nbf = lp.n / R;  // butterflies per line
  const int Lr = lp.span[stage];  // = L / R
  const int total = nbf * tl.tile;
  const int twstep = lp.n / L;
  const NkDiv dtile = tl.dtile, dnbf = lp.dnbf[stage], dlr = lp.dspan[stage];
  for (int idx = tid; idx < total; idx += nthr) {
    int t, bf;
    if (tl.t_fastest) {
      nk_fdivmod((uint32_t)idx, dtile, bf, t);
    } else {
      nk_fdivmod((uint32_t)idx, dnbf, t, bf);
    }
    int blk, j;
    nk_fdivmod((uint32_t)bf, dlr, blk, j);
    const int base = blk * L + j;
    C2<T> v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = lds[nk_lds_addr(tl, base + r * Lr, t)];
    Butterfly<T, R>::run(v);
    if (Lr > 1) {
      const C2<T> w1 = tw[j * twstep];
      C2<T> w = w1;
#pragma unroll
      for (int r = 1; r < R; ++r) {
        v[r] = cmul(v[r], w);
        if (r + 1 < R) w = cmul(w, w1);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) lds[nk_lds_addr(tl, base + r * Lr, t)] = v[r];
  }
}

// ---------------------------------------------------------------------------------------------
// fused prologue / epilogue descriptors (plain pointers: this struct crosses the C ABI)
// ---------------------------------------------------------------------------------------------
// descriptor + enums live in the public C header (the struct crosses the C ABI)
typedef nk_fuse NkFuse;

template <typename T>
NK_HD T nk_prologue(const NkFuse& f, int64_t i) {
  const T* in = (const T*)f.in;
  switch (f.pro) {
    case NK_PRO_AMP:
      if (f.afield) return ((const T*)f.afield)[i] * in[i];
      return (T)(f.amp[f.pidx[i]] * (double)in[i]);
    case NK_PRO_AMP_JVP: {
      if (f.afield && f.dafield)
        return ((const T*)f.afield)[i] * in[i] + ((const T*)f.dafield)[i] * ((const T*)f.in2)[i];
      const int32_t p = f.pidx[i];
      if (f.afield) {
        const T da = f.dampT ? ((const T*)f.dampT)[p] : (T)f.damp[p];
        return ((const T*)f.afield)[i] * in[i] + da * ((const T*)f.in2)[i];
      }
      return (T)(f.amp[p] * (double)in[i] + f.damp[p] * (double)((const T*)f.in2)[i]);
    }
    case NK_PRO_MUL:
      return in[i] * ((const T*)f.in2)[i];
    default:
      return in[i];
  }
}

// prologue of two adjacent real elements (i even): 2*sizeof(T)-byte vector loads of every operand stream
template <typename T>
NK_HD C2<T> nk_prologue_pair(const NkFuse& f, int64_t i) {
  const C2<T> a = *reinterpret_cast<const C2<T>*>((const T*)f.in + i);
  switch (f.pro) {
    case NK_PRO_AMP: {
      if (f.afield) {
        const C2<T> m = *reinterpret_cast<const C2<T>*>((const T*)f.afield + i);
        return C2<T>{m.x * a.x, m.y * a.y};
      }
      const int32_t p0 = f.pidx[i], p1 = f.pidx[i + 1];
      return C2<T>{(T)(f.amp[p0] * (double)a.x), (T)(f.amp[p1] * (double)a.y)};
    }
    case NK_PRO_AMP_JVP: {
      const C2<T> x = *reinterpret_cast<const C2<T>*>((const T*)f.in2 + i);
      if (f.afield && f.dafield) {  // both amplitude factors as fields: no bin index needed (product spectra)
        const C2<T> m = *reinterpret_cast<const C2<T>*>((const T*)f.afield + i);
        const C2<T> dm = *reinterpret_cast<const C2<T>*>((const T*)f.dafield + i);
        return C2<T>{m.x * a.x + dm.x * x.x, m.y * a.y + dm.y * x.y};
      }
      const int32_t p0 = f.pidx[i], p1 = f.pidx[i + 1];
      if (f.afield) {
        const C2<T> m = *reinterpret_cast<const C2<T>*>((const T*)f.afield + i);
        const T d0 = f.dampT ? ((const T*)f.dampT)[p0] : (T)f.damp[p0];
        const T d1 = f.dampT ? ((const T*)f.dampT)[p1] : (T)f.damp[p1];
        return C2<T>{m.x * a.x + d0 * x.x, m.y * a.y + d1 * x.y};
      }
      return C2<T>{(T)(f.amp[p0] * (double)a.x + f.damp[p0] * (double)x.x),
                   (T)(f.amp[p1] * (double)a.y + f.damp[p1] * (double)x.y)};
    }
    case NK_PRO_MUL: {
      const C2<T> x = *reinterpret_cast<const C2<T>*>((const T*)f.in2 + i);
      return C2<T>{a.x * x.x, a.y * x.y};
    }
    default:
      return a;
  }
}

// compile-time specialised pair prologues of the hot configurations (no run-time switch in the load loop):
//   PC = 0 plain, 1 afield*in, 2 afield*in + dampT[pidx]*in2, 3 afield*in + dafield*in2, 6 in*in2,
//   else generic run-time version
// Element base[iu + it] as a V, with the address split into a wave-uniform 64-bit part (scalar registers) and a
// 32-bit per-thread BYTE offset: compiles to the scalar-base + VGPR-offset form of global_load / global_store (ONE
// address VGPR per access instead of a 64-bit pair -- 32 accesses in flight otherwise cost 64 VGPRs).  The drivers
// guarantee it * sizeof(T) < 2^32.
template <typename V, typename T>
NK_HD const V& nk_at32(const T* base, int64_t iu, uint32_t it) {
  const char* b = reinterpret_cast<const char*>(base + iu);
  return *reinterpret_cast<const V*>(b + (uint32_t)(it * (uint32_t)sizeof(T)));
}
template <typename V, typename T>
NK_HD V* nk_ptr32(T* base, int64_t iu, uint32_t it) {
  char* b = reinterpret_cast<char*>(base + iu);
  return reinterpret_cast<V*>(b + (uint32_t)(it * (uint32_t)sizeof(T)));
}

// NK_NT_LOAD (bit mask): non-temporal LOADS of streams that are read exactly once per launch -- 1: work array of the
// in-place strided passes, 2: operand streams of the octant prologues (in, in2, cg_r), 4: work array in the final pass,
// 8: xi / addend / running sum in the scatter epilogue, 16: work array in the fused first-axis pass.  A plain streaming
// copy gains 6 % from them (6.2 -> 6.6 TB/s, tools/micro/launch_shape.hip); in the passes (1024^3 fp32, tools/nt_load_sweep.sh,
// profiles/r04_nt_load_sweep.log): bit 1 in-place pass 1.90 -> 3.00 ms (the line is written right back), bit 8 scatter
// epilogue 4.29 -> 4.94 ms, bit 4 final pass 2.94 -> 3.06 ms, bit 2 first pass 3.06 -> 3.03-3.05 ms (noise), bit 16 fused
// first-axis pass 2.22 -> 2.14-2.17 ms.  Default: bit 16 only.
#ifndef NK_NT_LOAD
#define NK_NT_LOAD 16
#endif
template <typename V>
NK_HD V nk_ld_stream(const V* p) {
#if !defined(NK_HOST_EMU)
  if constexpr (sizeof(V) == 8) {
    typedef float __attribute__((ext_vector_type(2))) W;
    const W w = __builtin_nontemporal_load(reinterpret_cast<const W*>(p));
    V v;
    __builtin_memcpy(&v, &w, 8);
    return v;
  } else if constexpr (sizeof(V) == 16) {
    typedef float __attribute__((ext_vector_type(4))) W;
    const W w = __builtin_nontemporal_load(reinterpret_cast<const W*>(p));
    V v;
    __builtin_memcpy(&v, &w, 16);
    return v;
  } else if constexpr (sizeof(V) == 4) {
    const float w = __builtin_nontemporal_load(reinterpret_cast<const float*>(p));
    V v;
    __builtin_memcpy(&v, &w, 4);
    return v;
  } else {
    return *p;
  }
#else
  return *p;
#endif
}

// iu: wave-uniform part of the flat index (scalar registers), it: per-thread part (32 bit)
template <typename T, int PC>
NK_HD C2<T> nk_prologue_ct(const NkFuse& f, int64_t iu, uint32_t it) {
  if constexpr (PC == 0) {
    return nk_at32<C2<T>>((const T*)f.in, iu, it);
  } else if constexpr (PC == 1) {
    const C2<T> a = nk_at32<C2<T>>((const T*)f.in, iu, it);
    const C2<T> m = nk_at32<C2<T>>((const T*)f.afield, iu, it);
    return C2<T>{m.x * a.x, m.y * a.y};
  } else if constexpr (PC == 2) {
    const C2<T> a = nk_at32<C2<T>>((const T*)f.in, iu, it);
    const C2<T> m = nk_at32<C2<T>>((const T*)f.afield, iu, it);
    const C2<T> x = nk_at32<C2<T>>((const T*)f.in2, iu, it);
    const int2 p = nk_at32<int2>(f.pidx, iu, it);
    const T* dt = (const T*)f.dampT;
    return C2<T>{m.x * a.x + dt[p.x] * x.x, m.y * a.y + dt[p.y] * x.y};
  } else if constexpr (PC == 3) {
    const C2<T> a = nk_at32<C2<T>>((const T*)f.in, iu, it);
    const C2<T> m = nk_at32<C2<T>>((const T*)f.afield, iu, it);
    const C2<T> x = nk_at32<C2<T>>((const T*)f.in2, iu, it);
    const C2<T> dm = nk_at32<C2<T>>((const T*)f.dafield, iu, it);
    return C2<T>{m.x * a.x + dm.x * x.x, m.y * a.y + dm.y * x.y};
  } else if constexpr (PC == 6) {  // MUL: in * in2
    const C2<T> a = nk_at32<C2<T>>((const T*)f.in, iu, it);
    const C2<T> x = nk_at32<C2<T>>((const T*)f.in2, iu, it);
    return C2<T>{a.x * x.x, a.y * x.y};
  } else {
    return C2<T>{nk_prologue<T>(f, iu + it), nk_prologue<T>(f, iu + it + 1)};
  }
}

// octant-shaped fields (nk_fuse.field_octant): fold a grid coordinate onto k <= n/2
NK_HD int nk_fold(int x, int n) { return 2 * x <= n ? x : n - x; }

// two adjacent elements of an octant row (only element-aligned): one 2*sizeof(T) load
template <typename T>
struct __attribute__((packed, aligned(sizeof(T)))) NkPairU {
  T x, y;
};
template <typename T>
NK_HD C2<T> nk_load_pair_u(const T* p) {
  const NkPairU<T> v = *reinterpret_cast<const NkPairU<T>*>(p);
  return C2<T>{v.x, v.y};
}

// pair prologues with OCTANT amplitude fields: j = octant offset of the lower of the two folded positions of the
// adjacent reals (i, i+1); desc: the pair is stored in descending order (mirrored half of the last axis)
//   PC = 4: afield8 * in,  5: afield8 * in + dafield8 * in2
//   PC = 7: afield8 * in + dampT[pidx_octant] * in2   (da gathered from its table: no expanded da field)
//   PC = 8: as 5, after the pending CG direction update  in <- beta * in + cg_r  (written back; same fp64 arithmetic as
//           nk_cg_direction, so the fused and the separate update agree bit for bit)
//   PC = 9: as 4 with `in` a FLOAT array under a wider T (nk_fuse.io32: fp32 excitations promoted at their product with the
//           fp64 amplitude, library/correlated_fields.py:755-764)
// The prologue comes in two halves, LOAD (every operand of the pair, no arithmetic, no store) and APPLY (arithmetic and
// the write-back of class 8), so that a pass can issue the loads of all its elements before the first use: class 8's
// store to `in` may alias every later load as far as the compiler knows, and a per-element load -> use chain leaves four
// loads in flight per thread (ISA of the contiguous first pass before the split: 16 x "4 loads, s_waitcnt vmcnt(0)").
template <typename T>
struct NkOctOps {
  C2<T> a, r, m, x, dm;
};
template <typename T, int PC>
NK_HD NkOctOps<T> nk_oct_load(const NkFuse& f, int64_t iu, uint32_t it, uint32_t j) {
  NkOctOps<T> o;
  if constexpr (PC == 9) {
    const C2<float> a32 = *reinterpret_cast<const C2<float>*>((const float*)f.in + iu + it);
    o.a = C2<T>{(T)a32.x, (T)a32.y};
  } else {
    o.a = (NK_NT_LOAD & 2) && PC != 8 ? nk_ld_stream(reinterpret_cast<const C2<T>*>((const T*)f.in + iu + it))
                                      : *reinterpret_cast<const C2<T>*>((const T*)f.in + iu + it);
  }
  if constexpr (PC == 8)
    o.r = (NK_NT_LOAD & 2) ? nk_ld_stream(reinterpret_cast<const C2<T>*>((const T*)f.cg_r + iu + it))
                           : *reinterpret_cast<const C2<T>*>((const T*)f.cg_r + iu + it);
  o.m = nk_load_pair_u<T>((const T*)f.afield + j);
  if constexpr (PC != 4 && PC != 9)
    o.x = (NK_NT_LOAD & 2) ? nk_ld_stream(reinterpret_cast<const C2<T>*>((const T*)f.in2 + iu + it))
                           : *reinterpret_cast<const C2<T>*>((const T*)f.in2 + iu + it);
  if constexpr (PC == 7) {
    const NkPairU<int32_t> p = *reinterpret_cast<const NkPairU<int32_t>*>(f.pidx_octant + j);
    const T* dt = (const T*)f.dampT;
    o.dm = C2<T>{dt[p.x], dt[p.y]};
  } else if constexpr (PC != 4 && PC != 9) {
    o.dm = nk_load_pair_u<T>((const T*)f.dafield + j);
  }
  return o;
}
// beta of class 8: the pending direction update's coefficient (same fp64 arithmetic as nk_cg_direction)
NK_HD double nk_oct_beta(const NkFuse& f) {
  const double beta = f.cg_scal[2] / f.cg_scal[0];
  return beta > 0.0 ? beta : 0.0;
}
template <typename T, int PC>
NK_HD C2<T> nk_oct_apply(const NkFuse& f, const NkOctOps<T>& o, int64_t iu, uint32_t it, bool desc, double beta) {
  C2<T> a = o.a;
  if constexpr (PC == 8) {
    a = C2<T>{(T)(beta * (double)a.x + (double)o.r.x), (T)(beta * (double)a.y + (double)o.r.y)};
    *reinterpret_cast<C2<T>*>(const_cast<T*>((const T*)f.in) + iu + it) = a;
  }
  const C2<T> m = desc ? C2<T>{o.m.y, o.m.x} : o.m;
  if constexpr (PC == 4 || PC == 9) {
    return C2<T>{m.x * a.x, m.y * a.y};
  } else {
    const C2<T> dm = desc ? C2<T>{o.dm.y, o.dm.x} : o.dm;
    return C2<T>{m.x * a.x + dm.x * o.x.x, m.y * a.y + dm.y * o.x.y};
  }
}
template <typename T, int PC>
NK_HD C2<T> nk_prologue_oct(const NkFuse& f, int64_t iu, uint32_t it, uint32_t j, bool desc) {
  double beta = 0.0;
  if constexpr (PC == 8) beta = nk_oct_beta(f);
  return nk_oct_apply<T, PC>(f, nk_oct_load<T, PC>(f, iu, it, j), iu, it, desc, beta);
}

NK_HD void nk_nonlin(int kind, double s, double& g, double& gp) {
  if (kind == NK_NL_EXP) {
    g = gp = exp(s);
  } else if (kind == NK_NL_SIGMOID) {  // NIFTy's sigmoid = 0.5 + 0.5 tanh(s)
    const double th = tanh(s);
    g = 0.5 + 0.5 * th;
    gp = 0.5 - 0.5 * th * th;
  } else {
    g = s;
    gp = 1.0;
  }
}

// single-output epilogue; `acc` collects the per-thread energy contribution
// LIKELIHOOD epilogue of one output (energy_operators.py:517-640): energy term -> acc, dE/ds -> out, Fisher weight -> out2
template <typename T>
NK_HD void nk_epi_likelihood(const NkFuse& f, int64_t o, T v, double& acc) {
  const double s = (double)v * f.scale + f.offset;
  double g, gp;
  nk_nonlin(f.nonlin, s, g, gp);
  double gs, w;
  if (f.lh_kind == NK_LH_GAUSS) {
    const double ic = f.icov ? (double)((const T*)f.icov)[o] : f.icov_scalar;
    const double r = g - (double)((const T*)f.data)[o];
    acc += 0.5 * ic * r * r;
    gs = gp * ic * r;
    w = gp * gp * ic;
  } else {
    const double d = (double)((const int64_t*)f.data)[o];
    acc += g - d * log(g);
    gs = gp * (1.0 - d / g);
    w = gp * gp / g;
  }
  ((T*)f.out)[o] = (T)gs;
  if (f.out2) ((T*)f.out2)[o] = (T)w;
}

// FAST (compile-time, chosen per group of lines by the final pass): every image in `MASK` exists and the likelihood is
// 1 = Gaussian with a scalar inverse covariance, 2 = Poissonian.  The data loads (nk_lh4_load: raw values, no arithmetic)
// then stand in straight-line code and the caller issues them for several coefficients ahead of the first store; under
// run-time conditions each load sat in its own block with an s_waitcnt behind it.
// TD: the type of the data, inverse-covariance and output ARRAYS -- T, or float under a double pipeline (nk_fuse.io32)
template <typename TD, int FAST>
struct NkLhData {
  typedef typename std::conditional<FAST == 2, int64_t, TD>::type type;
};
template <typename TD, int FAST, int MASK>
NK_HD void nk_lh4_load(const NkFuse& f, const int64_t (&o)[4], typename NkLhData<TD, FAST>::type (&d)[4]) {
  typedef typename NkLhData<TD, FAST>::type D;
#pragma unroll
  for (int i = 0; i < 4; ++i) d[i] = ((MASK >> i) & 1) ? ((const D*)f.data)[o[i]] : (D)1;
}
// one output of the likelihood epilogue: energy term returned, dE/ds and the Fisher weight through gs, w
NK_HD double nk_lh_term(const NkFuse& f, bool gauss, double v, double d, double ic, double& gs, double& w) {
  const double s = v * f.scale + f.offset;
  double g, gp;
  nk_nonlin(f.nonlin, s, g, gp);
  if (gauss) {
    const double r = g - d;
    gs = gp * ic * r;
    w = gp * gp * ic;
    return 0.5 * ic * r * r;
  }
  gs = gp * (1.0 - d / g);
  w = gp * gp / g;
  return g - d * log(g);
}
template <typename T, int FAST, int MASK, typename TD = T>
NK_HD void nk_lh4_apply(const NkFuse& f, const int64_t (&o)[4], const T (&v)[4],
                        const typename NkLhData<TD, FAST>::type (&d)[4], double& acc) {
  double gs[4], w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double e = nk_lh_term(f, FAST == 1, (double)v[i], (double)d[i], f.icov_scalar, gs[i], w[i]);
    if ((MASK >> i) & 1) acc += e;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!((MASK >> i) & 1)) continue;
    ((TD*)f.out)[o[i]] = (TD)gs[i];
    if (f.out2) ((TD*)f.out2)[o[i]] = (TD)w[i];
  }
}

// the (up to) four images of one coefficient, everything decided at run time: every load is issued before the first store
// -- `out` may alias nothing here, but the compiler cannot know, and four dependent load -> store chains per work item cost
// the final pass 1 ms at 1024^3
template <typename T, typename TD = T>
NK_HD void nk_epi_likelihood4(const NkFuse& f, const int64_t (&o)[4], const T (&v)[4], int mask, double& acc) {
  double dv[4], ic[4];
  const bool gauss = f.lh_kind == NK_LH_GAUSS;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool on = (mask >> i) & 1;
    dv[i] = !on ? 1.0 : gauss ? (double)((const TD*)f.data)[o[i]] : (double)((const int64_t*)f.data)[o[i]];
    ic[i] = (on && gauss && f.icov) ? (double)((const TD*)f.icov)[o[i]] : f.icov_scalar;
  }
  double gs[4], w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double e = nk_lh_term(f, gauss, (double)v[i], dv[i], ic[i], gs[i], w[i]);
    if ((mask >> i) & 1) acc += e;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!((mask >> i) & 1)) continue;
    ((TD*)f.out)[o[i]] = (TD)gs[i];
    if (f.out2) ((TD*)f.out2)[o[i]] = (TD)w[i];
  }
}

// The running sum of a VJP output (nk_fuse.accumulate / carry1 / carry2): the sample's own contribution g is ROUNDED to the
// field type first and then added to the partial sums innermost first, out <- [out +] ([carry2 +] ([carry1 +] g)) -- plain
// additions of stored values, so that a sum over samples gives the same bits whether its partial sums were formed in this
// epilogue or by a separate addition on another rank (the pairwise order of utilities.py:349-414).
// nk_settle: the value as it would be stored -- keeps the compiler from contracting the product inside g with the addition
// that follows (fma(a, t, out) rounds once, a stored a*t plus out rounds twice).
template <typename T>
NK_HD T nk_settle(T x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#endif
  return x;
}
NK_HD float nk_fma(float x, float y, float z) { return __builtin_fmaf(x, y, z); }
NK_HD double nk_fma(double x, double y, double z) { return __builtin_fma(x, y, z); }
// a sample's own VJP contribution in the generic epilogues (fp64 arithmetic, rounded to T once): a t + s addend, the product
// a t rounded before the fused multiply-add -- spelled out for the same reason as in nk_final_vjp_apply
NK_HD double nk_vjp_own(double a, double t, double s, double addend) { return nk_fma(s, addend, nk_settle(a * t)); }
template <typename T>
NK_HD T nk_vjp_chain(const NkFuse& f, T g, T c1, T c2, T ov) {
  T r = nk_settle(g);
  if (f.carry1) r = c1 + r;
  if (f.carry2) r = c2 + r;
  if (f.accumulate) r = ov + r;
  return r;
}
template <typename T>
NK_HD T nk_carry_at(const void* carry, int64_t o) {
  return carry ? ((const T*)carry)[o] : (T)0;
}

template <typename T>
NK_HD void nk_epilogue(const NkFuse& f, int64_t o, T v, double& acc) {
  T* out = (T*)f.out;
  switch (f.epi) {
    case NK_EPI_MUL: {
      double r = (double)v * f.scale * f.mul_scalar;
      if (f.mul) r *= (double)((const T*)f.mul)[o];
      out[o] = (T)r;
    } break;
    case NK_EPI_VJP: {
      const double t = (double)v * f.scale;
      const int32_t p = f.pidx[o];
      const double r = nk_vjp_own(f.afield ? (double)((const T*)f.afield)[o] : f.amp[p], t, f.addend_scale,
                                  f.addend ? (double)((const T*)f.addend)[o] : 0.0);
      out[o] = nk_vjp_chain<T>(f, (T)r, nk_carry_at<T>(f.carry1, o), nk_carry_at<T>(f.carry2, o), f.accumulate ? out[o] : (T)0);
      NK_VJP_SCATTER_AT(f, o, p, (double)((const T*)f.xi)[o] * t);
    } break;
    case NK_EPI_LIKELIHOOD:
      nk_epi_likelihood<T>(f, o, v, acc);
      break;
    case NK_EPI_NONLIN: {
      const double s = (double)v * f.scale + f.offset;
      double g, gp;
      nk_nonlin(f.nonlin, s, g, gp);
      out[o] = (T)g;
      if (f.out2) ((T*)f.out2)[o] = (T)gp;
    } break;
    default:
      out[o] = (T)((double)v * f.scale + f.offset);
  }
}

// paired epilogue for the two mirror outputs (o1 = k, o2 = -k) of one Fourier coefficient; the two
// points share their power bin, so the VJP scatter needs a single atomic.
template <typename T>
NK_HD void nk_epilogue_pair(const NkFuse& f, int64_t o1, T v1, int64_t o2, T v2, double& acc) {
  if (f.epi == NK_EPI_VJP) {
    T* out = (T*)f.out;
    const double t1 = (double)v1 * f.scale, t2 = (double)v2 * f.scale;
    const int32_t p = f.pidx[o1];
    const double a = f.afield ? (double)((const T*)f.afield)[o1] : f.amp[p];
    const double r1 = nk_vjp_own(a, t1, f.addend_scale, f.addend ? (double)((const T*)f.addend)[o1] : 0.0);
    const double r2 = nk_vjp_own(a, t2, f.addend_scale, f.addend ? (double)((const T*)f.addend)[o2] : 0.0);
    const T q1 = nk_vjp_chain<T>(f, (T)r1, nk_carry_at<T>(f.carry1, o1), nk_carry_at<T>(f.carry2, o1), f.accumulate ? out[o1] : (T)0);
    const T q2 = nk_vjp_chain<T>(f, (T)r2, nk_carry_at<T>(f.carry1, o2), nk_carry_at<T>(f.carry2, o2), f.accumulate ? out[o2] : (T)0);
    out[o1] = q1;
    out[o2] = q2;
    const T* xi = (const T*)f.xi;
    NK_VJP_SCATTER_AT(f, o1, p, (double)xi[o1] * t1 + (double)xi[o2] * t2);
    if (f.wfull && o2 != o1) f.wfull[o2] = 0.0;
  } else {
    nk_epilogue<T>(f, o1, v1, acc);
    nk_epilogue<T>(f, o2, v2, acc);
  }
}

// up to eight outputs that share one power bin (the sign-flip images of one coefficient): one atomic for the VJP
// scatter (VJP) epilogue of the four images of one slot: all loads first, then the stores; returns sum xi*t
template <typename T>
NK_HD double nk_vjp_quad(const NkFuse& f, const int64_t (&o)[4], const T (&v)[4], int mask, double a) {
  T* out = (T*)f.out;
  const T* xi = (const T*)f.xi;
  const T* addend = (const T*)f.addend;
  T xv[4], av[4], ov[4], c1[4], c2[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool on = (mask >> i) & 1;
    xv[i] = on ? xi[o[i]] : (T)0;
    av[i] = (on && addend) ? addend[o[i]] : (T)0;
    ov[i] = (on && f.accumulate) ? out[o[i]] : (T)0;
    c1[i] = on ? nk_carry_at<T>(f.carry1, o[i]) : (T)0;
    c2[i] = on ? nk_carry_at<T>(f.carry2, o[i]) : (T)0;
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!((mask >> i) & 1)) continue;
    const double t = (double)v[i] * f.scale;
    out[o[i]] = nk_vjp_chain<T>(f, (T)nk_vjp_own(a, t, f.addend_scale, (double)av[i]), c1[i], c2[i], ov[i]);
    s += (double)xv[i] * t;
  }
  return s;
}

template <typename T, int NOUT>
NK_HD void nk_epilogue_multi(const NkFuse& f, const int64_t (&o)[8], const T (&v)[8], int mask, double& acc,
                             double* w8slot = nullptr) {
  if (f.epi == NK_EPI_VJP) {
    T* out = (T*)f.out;
    const T* xi = (const T*)f.xi;
    const T* addend = (const T*)f.addend;
    const T* afield = (const T*)f.afield;
    const int first = (mask & 15) ? 0 : 4;  // slot 0 of an active half is always valid
    // all loads first (nothing below may be reordered across the stores by the compiler: out may alias)
    T xv[NOUT], av[NOUT], ov[NOUT], c1[NOUT], c2[NOUT];
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      const bool on = (mask >> i) & 1;
      xv[i] = on ? xi[o[i]] : (T)0;
      av[i] = (on && addend) ? addend[o[i]] : (T)0;
      ov[i] = (on && f.accumulate) ? out[o[i]] : (T)0;
      c1[i] = on ? nk_carry_at<T>(f.carry1, o[i]) : (T)0;
      c2[i] = on ? nk_carry_at<T>(f.carry2, o[i]) : (T)0;
    }
    int32_t p = 0;
    double a;
    if (afield) {
      a = (double)afield[o[first]];
      if (!w8slot) p = f.pidx[o[first]];
    } else {
      p = f.pidx[o[first]];
      a = f.amp[p];
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      if (!((mask >> i) & 1)) continue;
      const double t = (double)v[i] * f.scale;
      out[o[i]] = nk_vjp_chain<T>(f, (T)nk_vjp_own(a, t, f.addend_scale, (double)av[i]), c1[i], c2[i], ov[i]);
      s += (double)xv[i] * t;
    }
    if (w8slot) {
      *w8slot = s;  // octant array: every slot is written exactly once, reduced later by nk_octant_scatter
    } else if (f.wfull) {
#pragma unroll
      for (int i = 0; i < NOUT; ++i)
        if ((mask >> i) & 1) f.wfull[o[i]] = i == first ? s : 0.0;
    } else {
      NK_VJP_SCATTER(f, p, s);
    }
  } else {
#pragma unroll
    for (int i = 0; i < NOUT; ++i)
      if (mask & (1 << i)) nk_epilogue<T>(f, o[i], v[i], acc);
  }
}

// ---------------------------------------------------------------------------------------------
// transform geometry shared by the pass kernels
// ---------------------------------------------------------------------------------------------
struct NkGeom {
  int ndim;          // 1, 2 or 3 transformed axes
  int batch;         // leading batch count
  int na;            // length of the FIRST transformed axis  (pass C; 1 if ndim == 1)
  int nm;            // length of the MIDDLE transformed axis (pass B; 1 if ndim < 3)
  int nl;            // length of the LAST (contiguous) transformed axis (pass A), nl = 2*h
  int h;             // complex length of the packed half spectrum along the last axis
  int sign;          // +1: Re F + Im F (non-canonical, NIFTy default)   -1: Re F - Im F
};
