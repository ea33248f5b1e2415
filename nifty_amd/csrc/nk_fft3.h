// nk_fft3.h -- the SANDWICH pipeline:  out = epilogue( H( m . H( prologue(in) ) ) )  in five passes instead of six.
//
// A metric application J^T M J d (energy_operators.py:146-152 through harmonic_operators.py:144-161) is two genuine
// Hartley transforms with a diagonal in between.  Run back to back through the strided-first pipeline they cost six
// passes over the array: the last pass of the first transform writes the position-space field, the first pass of the
// second one reads it again.  Here the first transform runs CONTIGUOUS AXIS FIRST and the second one contiguous axis
// last, so that the last pass of the former and the first pass of the latter meet on the same (first) axis and become
// ONE kernel with the whole line in registers -- the position-space field never exists in memory.
//
//   F1  contiguous axis, real lines through the fused prologue -> half spectrum W(a, b, c), c = 0 .. nl/2, rows of
//       RS = nl/2 + 16 (fp32) / + 8 (fp64) complex columns (unpacked: no packed Nyquist column; the pad columns are 0)
//   F2  middle axis (3-D), in place, plain c2c                                      (k2_strided, MODE 0)
//   FM  first axis, in place: c2c -> F(a, b, c) = p + i q, the full spectrum of the real input.  The Hartley field at
//       the two points this coefficient feeds is  s(a, b, c) = p + sg q  and  s(-a, -b, -c) = p - sg q.  With the
//       diagonal m and ROW-MIRROR PAIRING
//           y_c(a) = m(a, b, c) s(a, b, c)  +  i m(-a, -b, -c) s(-a, -b, -c)
//       is an element-wise function of F(a, b, c) alone (no partner element, no second row), and the c2c transform
//       of y along a is  G_c(k_a; b) = X~(k_a; b, c) + i X~(-k_a; M - b, nl - c)  with x = m . s the (real) input of the
//       second transform and X~ its spectrum along a: the first pass of the second transform, done on the spot.
//   A2  middle axis (3-D), in place: G_c(k) = X2(k, c) + i X2(-k, nl - c), k = (k_a, k_b)
//   A3  contiguous axis, line pairs (k, -k): X2(-k, c) = conj X2(k, c) gives the whole line X2(k, 0..nl-1) from the two
//       rows (nk_final_body, PAIR = 1), one complex FFT of length nl, Hartley combine and the usual fused epilogues.
//
// Every access of F2 / FM / A2 is a 128-byte row segment of the ONE work array (all three run in place); only a
// non-constant diagonal m is read in 64-byte pieces (its own and the mirrored row).
#pragma once
#include <type_traits>

// contiguous first pass -------------------------------------------------------------------------------------
struct NkPass3 {
  NkGeom g;
  int64_t nlines;     // batch * na * nm
  int rows_per_slab;  // lines per work slab (3-D: nm, 2-D: na)
  int64_t rs, ss;     // work row stride / slab stride in complex elements
  int64_t blk0, nblk; // QUAD launches of one pipeline stage: workgroups blk0 .. blk0 + nblk - 1 (nblk == 0: all)
  NkDiv dmh;          // QUAD launches: / (nm / 2 + 1) of the workgroup's index inside its batch member (set by the launcher)
};

// thread id -> line thread pp = tid % P, line t = tid / P; LDS: two scalar planes (re, im) of TILE * PITCH elements
// PC: compile-time prologue class as in nk_strided_body (0 plain, 1 afield, 3 afield + dafield, 4 / 5 octant fields,
// 6 multiply, 7 octant a field + da gathered from its table, 8 = 5 with the pending CG direction update, -1 run-time)
// Schedule and tile: like the final pass (SchedF, E = 16: twice the threads per line, small workgroups) -- a row pass is
// latency-bound, many small workgroups hide the loads of the prologue operands best (JVP prologue at 1024^3 fp32:
// 4.96 ms with 8 lines x 16 threads x 32 elements per workgroup)
#ifndef NK_CONTIG3_LDS_KB
#define NK_CONTIG3_LDS_KB 20
#endif
template <typename T, int H>
struct Contig3Tile {
  using SC = SchedF<T, H>;
  static constexpr int P = SC::P;
  static constexpr int PITCH = ContigLayout<H, P>::PITCH;
  static constexpr int fit(int tile) {
    return (tile > 1 && (P * tile > 256 || 2 * tile * PITCH * (int)sizeof(T) > NK_CONTIG3_LDS_KB * 1024)) ? fit(tile / 2) : tile;
  }
  static constexpr int T0 = fit(16);
  static constexpr int TILE = (T0 / 2 >= 1 && P * (T0 / 2) >= 64) ? T0 / 2 : T0;  // halve while a full wavefront remains
  static constexpr int THREADS = P * TILE;
  static constexpr int LDS_BYTES = 2 * TILE * PITCH * (int)sizeof(T);
  // QUAD launches (3-D, octant prologue classes): one workgroup = the four rows (a, b), (a, M-b), (A-a, b), (A-a, M-b)
  static constexpr bool QUAD_OK = P * 4 <= 256;
  static constexpr int QTHREADS = P * 4;
  static constexpr int QLDS_BYTES = 2 * 4 * PITCH * (int)sizeof(T);
};

#ifndef NK_MID_EB
#define NK_MID_EB 8  // elements per load batch of a field diagonal in the fused middle pass
#endif
// load batches of prologue class 8 in the first phase (see there)
#ifndef NK_OCT8_BATCHES
#define NK_OCT8_BATCHES 2
#endif

// QUAD (3-D grids, octant prologue classes; TILE == 4): workgroup blk = (batch, a8, b8) over the OCTANT of the first two
// axes takes the rows (a8, b8), (a8, M - b8), (A - a8, b8), (A - a8, M - b8) -- the four rows of the grid that read the
// SAME lines of the octant amplitude fields a[pidx], da[pidx].  In natural row order these four rows run at unrelated
// times on different XCDs and every octant line was fetched about four times (rocprofv3 FETCH_SIZE of the JVP class:
// 15.2 GB against 9.7 GB of operands, profiles/r02g_pmc_traffic.json); here the first wavefront's request brings the line
// in and the other rows hit the CU's vector cache.  Rows that coincide with their mirror (a8 or b8 equal to 0 or n/2) stay
// idle in their duplicate slots (0.4 % of the slots at 1024^2 rows).
// QUAD: blk = (a8, b8) index of the workgroup inside batch member `bat` (< (na/2 + 1) * (nm/2 + 1): the octant of the first two
// axes), taken apart with a multiply-high (NkPass3::dmh) -- the rows, their work-array addresses and the octant line follow
// from (bat, a8, b8) without a single integer division (round 5: the per-wave index bookkeeping of this straight-line kernel
// -- 64-bit divisions by run-time values, repeated in the store loop -- was a fifth of its vector instructions, and the pass
// keeps its vector pipes 64 % busy).
template <typename T, int H, int TILE, int PC, bool QUAD = false, typename Exec>
NK_HD void nk_contig3_body(Exec& ex, const NkPass3& p, const NkFuse& f, int64_t blk, T* planes,
                           const C2<T>* __restrict__ tw, const C2<T>* __restrict__ twr, C2<T>* __restrict__ work, int bat = 0) {
  using SC = typename Contig3Tile<T, H>::SC;
  using LY = ContigLayout<H, SC::P>;
  constexpr int E = SC::E, S = SC::S, P = SC::P;
  static_assert(!QUAD || TILE == 4, "a QUAD workgroup holds exactly four rows");
  T* pre = planes;
  T* pim = planes + TILE * LY::PITCH;
  const int64_t line0 = blk * TILE;
  constexpr int nl = 2 * H;  // (= p.g.nl: the launcher dispatches on H = nl / 2; a constant lets the mirror selects of the prologue fold)
  [[maybe_unused]] int a8 = 0, b8 = 0;
  if constexpr (QUAD) nk_fdivmod((uint32_t)blk, p.dmh, a8, b8);
  // QUAD: first / middle index of tile slot t (its mirror images of (a8, b8)), and whether the slot exists
  auto quad_ab = [&](int t, int& a, int& b) -> bool {
    const bool ma = (t & 2) != 0, mb = (t & 1) != 0;
    a = ma ? p.g.na - a8 : a8;
    b = mb ? p.g.nm - b8 : b8;
    return !((ma && (a8 == 0 || 2 * a8 == p.g.na)) || (mb && (b8 == 0 || 2 * b8 == p.g.nm)));
  };
  // row of tile slot t, or -1 for an idle slot
  auto line_of = [&](int t) -> int64_t {
    if constexpr (!QUAD) {
      const int64_t l = line0 + t;
      return l < p.nlines ? l : -1;
    } else {
      int a, b;
      if (!quad_ab(t, a, b)) return -1;
      return ((int64_t)bat * p.g.na + a) * p.g.nm + b;
    }
  };

  ex.phase([&](int tid, PassRegs<T, E>& rg) {
    const int pp = tid % P, t = tid / P;
    const int64_t line = line_of(t);
    constexpr int R = SC::radix(0), Q = E / R;
    constexpr bool OCT = PC == 4 || PC == 5 || PC == 7 || PC == 8;
    static_assert(!QUAD || OCT, "QUAD launches exist for the octant prologue classes only");
    [[maybe_unused]] uint32_t o8 = 0;
    if constexpr (OCT) {
      const uint32_t ch = nl / 2 + 1;
      if constexpr (QUAD) {
        o8 = ((uint32_t)a8 * (uint32_t)(p.g.nm / 2 + 1) + (uint32_t)b8) * ch;  // all four rows fold onto (a8, b8)
      } else if (p.g.ndim == 3) {
        const int b = (int)(line % p.g.nm), a = (int)((line / p.g.nm) % p.g.na);
        o8 = ((uint32_t)nk_fold(a, p.g.na) * (p.g.nm / 2 + 1) + (uint32_t)nk_fold(b, p.g.nm)) * ch;
      } else {
        o8 = (uint32_t)nk_fold((int)(line % p.g.na), p.g.na) * ch;
      }
    }
    // flat real index of the row = iu (wave-uniform where the rows of a tile are consecutive) + tl (per thread)
    const int64_t iu = QUAD ? (line < 0 ? 0 : line * nl) : line0 * nl;
    const uint32_t tl = QUAD ? 0u : (uint32_t)(t * nl);  // < 2^31: a tile of lines
    if constexpr (OCT) {
      // ONE branch around the whole row, and all E elements' operand loads ahead of the first use (nk_oct_load / _apply)
      if (line >= 0) {
        // class 8 carries five operand pairs and fp64 temporaries per element: two half batches keep it at 4 waves/SIMD
        constexpr int NB = PC == 8 ? NK_OCT8_BATCHES * (sizeof(T) == 8 ? 2 : 1) : 1, EB = E / NB;  // fp64: 214 VGPRs with two
        double beta = 0.0;
        if constexpr (PC == 8) beta = nk_oct_beta(f);
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += EB) {
          NkOctOps<T> ops[EB];
#pragma unroll
          for (int e = e0; e < e0 + EB; ++e) {
            const int row = nk_in_row<SC, 0>(pp, e / R, e % R);  // complex index j: reals 2j, 2j+1
            const bool desc = 4 * row >= nl;  // (c, c+1) -> (nl-c, nl-c-1): stored descending, lower position nl-c-1
            ops[e - e0] = nk_oct_load<T, PC>(f, iu, tl + 2 * row, o8 + (desc ? nl - 2 * row - 1 : 2 * row));
          }
#pragma unroll
          for (int e = e0; e < e0 + EB; ++e) {
            const int row = nk_in_row<SC, 0>(pp, e / R, e % R);
            rg.v[e] = nk_oct_apply<T, PC>(f, ops[e - e0], iu, tl + 2 * row, 4 * row >= nl, beta);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) rg.v[e] = C2<T>{(T)0, (T)0};
      }
    } else {
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int row = nk_in_row<SC, 0>(pp, q, r);
          C2<T> z{(T)0, (T)0};
          if (line >= 0) {
            if constexpr (PC >= 0) {
              z = nk_prologue_ct<T, PC>(f, iu, tl + 2 * row);
            } else {
              z = nk_prologue_pair<T>(f, iu + tl + 2 * row);
            }
          }
          rg.v[q * R + r] = z;
        }
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
  // untangle the half-length transform of z_j = x_2j + i x_2j+1 into F[0 .. H] (natural-order planes) and write the
  // unpacked row; the pad columns H+1 .. rs-1 are zeroed so that the in-place passes never touch uninitialised data
  ex.last_phase([&](int tid, PassRegs<T, E>& rg) {
    (void)rg;
    constexpr int NK = H / 2 + 1;
    constexpr int NT = P * TILE;
    auto row_of = [&](int64_t line) { return work + (line / p.rows_per_slab) * p.ss + (line % p.rows_per_slab) * p.rs; };
    auto untangle = [&](C2<T>* dst, int t, int k) {
      if (k == 0) {
        const T zx = pre[LY::addr(t, 0)], zy = pim[LY::addr(t, 0)];
        nk_store_stream(dst, C2<T>{zx + zy, (T)0});
        nk_store_stream(dst + H, C2<T>{zx - zy, (T)0});
        return;
      }
      const int a1 = LY::addr(t, k), a2 = LY::addr(t, H - k);
      const C2<T> Zk{pre[a1], pim[a1]}, Zm{pre[a2], pim[a2]};
      const C2<T> Ev{(T)0.5 * (Zk.x + Zm.x), (T)0.5 * (Zk.y - Zm.y)};
      const C2<T> Od{(T)0.5 * (Zk.x - Zm.x), (T)0.5 * (Zk.y + Zm.y)};
      const C2<T> G = cmul(twr[k], Od);
      nk_store_stream(dst + k, C2<T>{Ev.x + G.y, Ev.y - G.x});
      if (k != H - k) nk_store_stream(dst + (H - k), C2<T>{Ev.x - G.y, -Ev.y - G.x});
    };
    const int npad = (int)(p.rs - (H + 1));
    if constexpr (QUAD) {
      // slot by slot: the row's address is workgroup-uniform (3-D work layout: slab (bat, a), row b), the threads run over k
#pragma unroll
      for (int t = 0; t < TILE; ++t) {
        int a, b;
        if (!quad_ab(t, a, b)) continue;
        C2<T>* dst = work + ((int64_t)bat * p.g.na + a) * p.ss + (int64_t)b * p.rs;
        for (int k = tid + 1; k < NK; k += NT) untangle(dst, t, k);
        if (tid == 0) untangle(dst, t, 0);
        for (int c = tid; c < npad; c += NT) dst[H + 1 + c] = C2<T>{(T)0, (T)0};
      }
    } else {
      for (int idx = tid; idx < TILE * NK; idx += NT) {
        const int k = idx % NK, t = idx / NK;
        const int64_t line = line_of(t);
        if (line < 0) continue;
        untangle(row_of(line), t, k);
      }
      for (int idx = tid; idx < TILE * npad; idx += NT) {
        const int64_t line = line_of(idx / npad);
        if (line >= 0) row_of(line)[H + 1 + idx % npad] = C2<T>{(T)0, (T)0};
      }
    }
  });
}

// fused middle pass -----------------------------------------------------------------------------------------
struct NkPassM {
  NkPassS s;         // strided geometry over the first axis (MODE 0 addressing of nk_strided_body)
  int64_t rs;        // work row stride in complex elements
  int M;             // rows per slab (3-D: nm, 2-D: 1)
  double mid_scale;  // (scale of the first transform) * mul_scalar
};

// registers of the fused middle pass: the pass registers plus the prefetched first-stage inputs of the NEXT tile
template <typename T, int E, bool PF = true>
struct MidRegs : PassRegs<T, E> {
  C2<T> nxt[PF ? E : 1];
};
#ifndef NK_HOST_EMU
template <typename Regs>
struct DeviceExecR {
  Regs regs;
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

// thread id -> column t = tid % TILE, line thread pp = tid / TILE; blockDim = P * TILE
// The workgroup walks the tiles v = v0, v0 + vstep, ... < nblk (a persistent launch: vstep = gridDim; one tile per
// workgroup: vstep >= nblk).  One workgroup fills a CU (registers, LDS) and the two line transforms cost about as much
// VALU time as the tile's HBM traffic takes (1024 x 16 fp32: ~11 us each), so with PF the loads of the NEXT tile are
// issued into spare registers before the current tile's first butterfly and land while it is computed.
// XM: exchange mode.  0: two half rounds (real, imaginary parts) through a scalar plane of N*TILE*sizeof(T) bytes;
//     1: one round through a complex plane (twice the LDS);  2: a complex plane of HALF the columns, used twice (columns
//     t < TILE/2, then the rest) -- the LDS of mode 0 without its E temporaries per thread, which is what lets the
//     wide schedule (64 elements per thread, two workgroups per CU) run without spills
// MF: the diagonal is a field (fuse.mul), else only the scalar mid_scale;  TWC: composed twiddles (nk_stage_compute)
template <typename T, int N, int TILE, int XM, bool MF, bool PF, typename SC, bool TWC = false, typename Exec>
NK_HD void nk_mid_body(Exec& ex, const NkPassM& pm, const NkFuse& f, int64_t v0, int64_t vstep, int64_t nblk, int xmap,
                       T* plane, const C2<T>* __restrict__ tw_global, C2<T>* __restrict__ work, C2<T>* tw_lds = nullptr) {
  using RG = MidRegs<T, SC::E, PF>;
  constexpr bool CX = XM == 1;
  constexpr int E = SC::E, S = SC::S, LS = S - 1;
  const NkPassS& p = pm.s;
  const C2<T>* tw = tw_lds ? tw_lds : tw_global;
  [[maybe_unused]] C2<T>* cplane = reinterpret_cast<C2<T>*>(plane);
  const int64_t rstride = p.ss > 0 ? p.ss : p.inner;
  auto tile_of = [&](int64_t v, int64_t& o, int64_t& c0) {
    const int64_t blk = xmap ? nk_xcd_contig(v, nblk) : v;
    o = blk / p.tiles_per_slab;
    c0 = (blk % p.tiles_per_slab) * (int64_t)TILE;
  };
  auto load_tile = [&](C2<T>* dst, int64_t v, int tid) {
    int64_t o, c0;
    tile_of(v, o, c0);
    const C2<T>* base = work + o * N * rstride + c0;
    const int t = tid % TILE, pp = tid / TILE;
    constexpr int R = SC::radix(0), Q = E / R;
    const uint32_t toff = (uint32_t)(pp * rstride + t);
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < R; ++r)
        dst[q * R + r] = (NK_NT_LOAD & 16) ? nk_ld_stream(&nk_at32<C2<T>>(base, (int64_t)nk_in_row<SC, 0>(0, q, r) * rstride, toff))
                                           : nk_at32<C2<T>>(base, (int64_t)nk_in_row<SC, 0>(0, q, r) * rstride, toff);
  };

  // exchange: the output of stage SA (rows nk_out_row<SA>) becomes the input of stage SB (rows nk_in_row<SB>)
  auto xwrite = [&](auto sa, int half, int tid, RG& rg) {
    constexpr int SA = decltype(sa)::value;
    const int t = tid % TILE, pp = tid / TILE;
    if constexpr (XM == 1) {
      nk_xwrite_c2<T, SC, SA, TILE>(rg.v, cplane, pp, t);
    } else if constexpr (XM == 2) {
      if ((t >= TILE / 2) == (half != 0)) nk_xwrite_c2<T, SC, SA, TILE / 2>(rg.v, cplane, pp, t % (TILE / 2));
    } else {
      if (half == 0)
        nk_xwrite_cols<T, SC, SA, TILE, 0>(rg.v, plane, pp, t);
      else
        nk_xwrite_cols<T, SC, SA, TILE, 1>(rg.v, plane, pp, t);
    }
  };
  auto xread = [&](auto sb, int half, int tid, RG& rg) {
    constexpr int SB = decltype(sb)::value;
    const int t = tid % TILE, pp = tid / TILE;
    if constexpr (XM == 1) {
      nk_xread_c2<T, SC, SB, TILE>(rg.v, cplane, pp, t);
    } else if constexpr (XM == 2) {
      if ((t >= TILE / 2) == (half != 0)) nk_xread_c2<T, SC, SB, TILE / 2>(rg.v, cplane, pp, t % (TILE / 2));
    } else if (half == 0) {
      nk_xread_cols<T, SC, SB, TILE>(rg.tmp, plane, pp, t);
    } else {
      T im[E];
      nk_xread_cols<T, SC, SB, TILE>(im, plane, pp, t);
#pragma unroll
      for (int e = 0; e < E; ++e) rg.v[e] = C2<T>{rg.tmp[e], im[e]};
    }
  };
  // the barrier-separated rounds of one exchange after the producing phase has written (half 0 of) its output
  auto exchange_rest = [&](auto sa, auto sb) {
    ex.phase([&](int tid, RG& rg) { xread(sb, 0, tid, rg); });
    if constexpr (!CX) {
      ex.phase([&](int tid, RG& rg) { xwrite(sa, 1, tid, rg); });
      ex.phase([&](int tid, RG& rg) { xread(sb, 1, tid, rg); });
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using ILS = std::integral_constant<int, LS>;

  if (tw_lds || PF) {
    ex.phase([&](int tid, RG& rg) {
      if (tw_lds) nk_tw_to_lds<T>(tw_global, tw_lds, N, tid, SC::P * TILE);  // published by this phase's barrier
      if constexpr (PF) {
        if (v0 < nblk) load_tile(rg.nxt, v0, tid);
      }
    });
  }
  auto do_tile = [&](int64_t v) {
    int64_t o, c0;
    tile_of(v, o, c0);
    C2<T>* base = work + o * N * rstride + c0;
    // ---- first transform: the tile (prefetched, or loaded here), stages 0 .. LS
    ex.phase([&](int tid, RG& rg) {
      if constexpr (PF) {
#pragma unroll
        for (int e = 0; e < E; ++e) rg.v[e] = rg.nxt[e];
        if (v + vstep < nblk) load_tile(rg.nxt, v + vstep, tid);
      } else {
        load_tile(rg.v, v, tid);
      }
      nk_stage_compute<T, SC, 0, TWC>(rg.v, tid / TILE, tw);
      xwrite(I0{}, 0, tid, rg);
    });
    exchange_rest(I0{}, I1{});
    if constexpr (S == 3) {
      ex.phase([&](int tid, RG& rg) {
        nk_stage_compute<T, SC, 1, TWC>(rg.v, tid / TILE, tw);
        xwrite(I1{}, 0, tid, rg);
      });
      exchange_rest(I1{}, I2{});
    }
    // ---- last stage of the first transform, the diagonal with row-mirror pairing, hand-over to the second transform
    ex.phase([&](int tid, RG& rg) {
      const int t = tid % TILE, pp = tid / TILE;
      nk_stage_compute<T, SC, LS, TWC>(rg.v, pp, tw);
      constexpr int R = SC::radix(LS);
      [[maybe_unused]] constexpr int Q = E / R;
      const T sg = (T)p.g.sign, ms = (T)pm.mid_scale;
      if constexpr (MF) {
        // m(a, b, c) for the real part, m(-a, -b, -c) for the imaginary part: 64-byte pieces of the user's field.
        // Addresses as a wave-uniform 64-bit part (the register slot's row) + a 32-bit per-thread byte offset (nk_at32):
        // 2 E loads with 64-bit per-lane addresses would cost 4 E VGPRs
        const int nl = p.g.nl, M = pm.M;
        const int64_t col = c0 + t;
        const int b = (int)(col / pm.rs), c = (int)(col % pm.rs);
        const bool valid = c <= nl / 2;
        const int64_t rowlen = (int64_t)M * nl;
        const T* mb = (const T*)f.mul + o * N * rowlen;  // this batch entry
        const int rp = nk_out_row<SC, LS>(pp, 0, 0);
        const int RPMAX = nk_out_row<SC, LS>(SC::P - 1, 0, 0);  // rp <= RPMAX (rows of a thread grow with pp)
        const uint32_t off1 = (uint32_t)((int64_t)b * nl + c), off2 = (uint32_t)((int64_t)(b ? M - b : 0) * nl + (c ? nl - c : 0));
        const uint32_t t1 = (uint32_t)(rp * rowlen) + off1;
        const uint32_t t2 = (uint32_t)((RPMAX - rp) * rowlen) + off2;
        // the two diagonal values of EB elements at a time, all their loads ahead of the first use, ONE test of `valid`
        // around the batch (per element -- test, two loads, s_waitcnt, multiply -- the 2 E loads ran one pair at a time)
        constexpr int EB = E < NK_MID_EB ? E : NK_MID_EB;
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += EB) {
          T m1[EB], m2[EB];
          if (valid) {
#pragma unroll
            for (int e = e0; e < e0 + EB; ++e) {
              const int ru = nk_out_row<SC, LS>(0, e / R, e % R);  // a = rp + ru
              m1[e - e0] = nk_at32<T>(mb, (int64_t)ru * rowlen, t1);
              // mirror row (N - a) % N: (N - ru - RPMAX) + (RPMAX - rp), except a = 0 (its own mirror)
              const T* own = mb + off2;
              const T* mir = &nk_at32<T>(mb, (int64_t)(N - ru - RPMAX) * rowlen, t2);
              m2[e - e0] = *((ru == 0 && rp == 0) ? own : mir);
            }
          } else {
#pragma unroll
            for (int e = 0; e < EB; ++e) m1[e] = m2[e] = (T)0;
          }
#pragma unroll
          for (int e = e0; e < e0 + EB; ++e) {
            const C2<T> F = rg.v[e];
            rg.v[e] = C2<T>{ms * m1[e - e0] * (F.x + sg * F.y), ms * m2[e - e0] * (F.x - sg * F.y)};
          }
          // keep the batches apart: left alone hipcc issues all 2 E loads first and sinks the multiplications below them
          // (212 B / lane of spills, 4.9 -> 6.4 ms); pinning the batch's results in front of a compiler fence orders them
#pragma unroll
          for (int e = e0; e < e0 + EB; ++e) {
            NK_PIN(rg.v[e].x);
            NK_PIN(rg.v[e].y);
          }
        }
      } else {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const C2<T> F = rg.v[e];
          rg.v[e] = C2<T>{ms * (F.x + sg * F.y), ms * (F.x - sg * F.y)};
        }
      }
      xwrite(ILS{}, 0, tid, rg);
    });
    exchange_rest(ILS{}, I0{});
    // ---- second transform
    ex.phase([&](int tid, RG& rg) {
      nk_stage_compute<T, SC, 0, TWC>(rg.v, tid / TILE, tw);
      xwrite(I0{}, 0, tid, rg);
    });
    if constexpr (S == 3) {
      exchange_rest(I0{}, I1{});
      ex.phase([&](int tid, RG& rg) {
        nk_stage_compute<T, SC, 1, TWC>(rg.v, tid / TILE, tw);
        xwrite(I1{}, 0, tid, rg);
      });
    }
    // last exchange, then the last stage and the stores; the plane is free again after the exchange's last barrier
    using IPREV = std::integral_constant<int, LS - 1>;
    exchange_rest(IPREV{}, ILS{});
    ex.last_phase([&](int tid, RG& rg) {
      const int t = tid % TILE, pp = tid / TILE;
      nk_stage_compute<T, SC, LS, TWC>(rg.v, pp, tw);
      constexpr int R = SC::radix(LS), Q = E / R;
      const uint32_t toff = (uint32_t)(nk_out_row<SC, LS>(pp, 0, 0) * rstride + t);
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int64_t uo = (int64_t)nk_out_row<SC, LS>(0, q, r) * rstride;
          if constexpr (S == 2)
            nk_store_stream(nk_ptr32<C2<T>>(base, uo, toff), rg.v[q * R + r]);
          else
            nk_store_stream((base + uo) + toff, rg.v[q * R + r]);
        }
    });
  };
  if constexpr (PF) {
    for (int64_t v = v0; v < nblk; v += vstep) do_tile(v);
  } else {
    if (v0 < nblk) do_tile(v0);  // one tile per workgroup (vstep >= nblk): no loop around the phases
  }
}

// pass parameters of the sandwich pipeline (shared by the HIP driver and the host emulation) --------------------
struct NkPipe3 {
  NkPass3 p1;   // contiguous first pass
  NkPassS s2;   // in-place middle-axis pass (3-D), used for F2 and A2
  NkPassM pm;   // fused first-axis pass
  NkPassF pf;   // final pass (PAIR = 1)
  int64_t rs, ss;
};
template <typename T>
static inline int nk_pipe3_colpad() {
  return 128 / (2 * (int)sizeof(T));  // one 128-byte row segment: every strided tile divides it
}
// complex elements of the work array the sandwich needs
static inline size_t nk_pipe3_work_elems(const NkGeom& g, int colpad, int pad) {
  const int64_t rs = g.h + colpad;
  if (g.ndim == 3) return (size_t)g.batch * g.na * ((int64_t)g.nm * rs + pad);
  return (size_t)g.batch * g.na * rs;
}
template <typename T>
static inline NkPipe3 nk_pipe3_setup(const NkHostPlan& hp, int sign, int pad, double mid_scale) {
  NkPipe3 q{};
  const NkGeom& g = hp.g;
  if (pad < 0) pad = 0;
  if (pad > NK_WORK_PAD_MAX) pad = NK_WORK_PAD_MAX;
  const int cp = nk_pipe3_colpad<T>();
  pad = pad / cp * cp;
  q.rs = g.h + cp;
  q.ss = g.ndim == 3 ? (int64_t)g.nm * q.rs + pad : 0;
  q.p1.g = g;
  q.p1.g.sign = sign;
  q.p1.nlines = (int64_t)g.batch * g.na * g.nm;
  q.p1.rows_per_slab = g.ndim == 3 ? g.nm : g.na;
  q.p1.rs = q.rs;
  q.p1.ss = g.ndim == 3 ? q.ss : (int64_t)g.na * q.rs;
  if (g.ndim == 3) {
    q.s2 = hp.pb;  // lines over the middle axis inside slab o = batch * na + a: base = work + o * ss, rows rs apart
    q.s2.outer = (int64_t)g.batch * g.na;
    q.s2.inner = q.rs;
    q.s2.blo = 1;
    q.s2.ss = q.ss;
  }
  q.pm.s = hp.pc;
  q.pm.s.g.sign = sign;
  q.pm.s.outer = g.batch;
  q.pm.s.inner = g.ndim == 3 ? (int64_t)g.nm * q.rs : q.rs;
  q.pm.s.blo = 0;
  q.pm.s.ss = q.ss;
  q.pm.rs = q.rs;
  q.pm.M = g.ndim == 3 ? g.nm : 1;
  q.pm.mid_scale = mid_scale;
  q.pf.g = g;
  q.pf.g.sign = sign;
  q.pf.A = g.ndim == 3 ? g.na : 1;
  q.pf.M = g.ndim == 3 ? g.nm : g.na;
  q.pf.blo = 0;
  q.pf.ss = q.ss;
  q.pf.rs = q.rs;
  return q;
}
