// nk_fft_phases.h -- load / store phases of the genuine N-D Hartley transform (real -> real).
//
// Algorithm (replaces ducc0.fft.genuine_hartley / scipy.fft.fftn + Re+-Im, reference
// nifty/cl/ducc_dispatch.py:88-100,135-142):
//   pass A  last axis, contiguous real lines of length nl=2h:  z[j] = x[2j] + i x[2j+1], complex FFT of
//           length h in LDS, untangle to the half spectrum F[0..h], PACK  P[0] = F[0] + i F[h],
//           P[k] = F[k] (0<k<h)  -> half-complex array C[...][h]   (exactly as many bytes as the input)
//   pass B  middle axis (3-D only): in-place strided c2c FFT on C, tiles of `tile` adjacent columns
//   pass C  first axis: strided c2c FFT, then Hartley combine  H[k] = Re F[k] + s Im F[k],
//           H[-k] = Re F[k] - s Im F[k]; both outputs go through the fused epilogue.  The packed column
//           k_last = 0 is written to a small scratch S instead ...
//   pass D  ... and untangled there: A = (Z(k) + conj Z(-k))/2 is the spectrum at k_last = 0,
//           B = (Z(k) - conj Z(-k))/(2i) the one at k_last = h.
// 1-D transforms are a single kernel (pass A load + FFT + untangle + combine).
// HBM traffic per transform: (read + write) of the array once per transformed axis.
#pragma once
#include "nk_core.h"

struct NkPassA {
  NkGeom g;
  NkLinePlan lp;   // n = h
  NkTile tl;       // contiguous layout
  int64_t nlines;  // batch * na * nm
  NkDiv dh, dnk;   // / h, / (h / 2 + 1): index decomposition of the load / store phases
};

// load `tile` real lines, apply the prologue, store as complex pairs (natural order) in LDS
template <typename T>
NK_HD void nk_passA_load(const NkPassA& p, const NkFuse& f, int64_t blk, int tid, int nthr, C2<T>* lds) {
  const int h = p.g.h;
  const int total = p.tl.tile * h;
  const int64_t line0 = blk * p.tl.tile;
  for (int idx = tid; idx < total; idx += nthr) {
    int j, t;
    nk_fdivmod((uint32_t)idx, p.dh, t, j);
    const int64_t line = line0 + t;
    C2<T> z{(T)0, (T)0};
    if (line < p.nlines) {
      const int64_t i = line * p.g.nl + 2 * j;
      z.x = nk_prologue<T>(f, i);
      z.y = nk_prologue<T>(f, i + 1);
    }
    lds[nk_lds_addr(p.tl, j, t)] = z;
  }
}

// F[k] and F[h-k] of the real line from the half-length complex FFT Z (digit-reversed in LDS)
template <typename T>
NK_HD void nk_untangle(const NkLinePlan& lp, const NkTile& tl, const C2<T>* lds, int t, int k, int h,
                       const C2<T>* __restrict__ twr, C2<T>& Fk, C2<T>& Fm) {
  const C2<T> Zk = lds[nk_lds_addr(tl, nk_digit_reverse(lp, k), t)];
  const C2<T> Zm = lds[nk_lds_addr(tl, nk_digit_reverse(lp, h - k), t)];
  const C2<T> E{(T)0.5 * (Zk.x + Zm.x), (T)0.5 * (Zk.y - Zm.y)};
  const C2<T> O{(T)0.5 * (Zk.x - Zm.x), (T)0.5 * (Zk.y + Zm.y)};
  const C2<T> G = cmul(twr[k], O);  // twr[k] = exp(-2 pi i k / (2h))
  Fk = C2<T>{E.x + G.y, E.y - G.x};
  Fm = C2<T>{E.x - G.y, -E.y - G.x};
}

// pass A store: packed half spectrum to the half-complex work array
template <typename T>
NK_HD void nk_passA_store(const NkPassA& p, int64_t blk, int tid, int nthr, const C2<T>* lds,
                          const C2<T>* __restrict__ twr, C2<T>* __restrict__ work) {
  const int h = p.g.h;
  const int nk = h / 2 + 1;
  const int total = p.tl.tile * nk;
  const int64_t line0 = blk * p.tl.tile;
  for (int idx = tid; idx < total; idx += nthr) {
    int k, t;
    nk_fdivmod((uint32_t)idx, p.dnk, t, k);
    const int64_t line = line0 + t;
    if (line >= p.nlines) continue;
    C2<T>* dst = work + line * h;
    if (k == 0) {
      const C2<T> Z0 = lds[nk_lds_addr(p.tl, 0, t)];
      dst[0] = C2<T>{Z0.x + Z0.y, Z0.x - Z0.y};
    } else {
      C2<T> Fk, Fm;
      nk_untangle<T>(p.lp, p.tl, lds, t, k, h, twr, Fk, Fm);
      dst[k] = Fk;
      dst[h - k] = Fm;
    }
  }
}

// 1-D: untangle + Hartley combine + epilogue straight to the output
template <typename T>
NK_HD void nk_pass1d_store(const NkPassA& p, const NkFuse& f, int64_t blk, int tid, int nthr, const C2<T>* lds,
                           const C2<T>* __restrict__ twr, double& acc) {
  const int h = p.g.h, nl = p.g.nl;
  const T sg = (T)p.g.sign;
  const int nk = h / 2 + 1;
  const int total = p.tl.tile * nk;
  const int64_t line0 = blk * p.tl.tile;
  for (int idx = tid; idx < total; idx += nthr) {
    int k, t;
    nk_fdivmod((uint32_t)idx, p.dnk, t, k);
    const int64_t line = line0 + t;
    if (line >= p.nlines) continue;
    const int64_t o = line * nl;
    if (k == 0) {
      const C2<T> Z0 = lds[nk_lds_addr(p.tl, 0, t)];
      nk_epilogue<T>(f, o, Z0.x + Z0.y, acc);
      nk_epilogue<T>(f, o + h, Z0.x - Z0.y, acc);
    } else {
      C2<T> Fk, Fm;
      nk_untangle<T>(p.lp, p.tl, lds, t, k, h, twr, Fk, Fm);
      nk_epilogue_pair<T>(f, o + k, Fk.x + sg * Fk.y, o + nl - k, Fk.x - sg * Fk.y, acc);
      if (k != h - k) nk_epilogue_pair<T>(f, o + h - k, Fm.x + sg * Fm.y, o + h + k, Fm.x - sg * Fm.y, acc);
    }
  }
}

// strided c2c passes on the half-complex work array viewed as [outer][n][inner]
struct NkPassS {
  NkGeom g;
  NkLinePlan lp;   // n = axis length
  NkTile tl;       // t_fastest layout
  int64_t outer;   // number of outer slabs
  int64_t inner;   // complex elements between consecutive line elements
  int tiles_per_slab;  // inner / tile
  int blo;             // strided-first pipeline, 3-D: work array blocked as [batch][mid/blo][first][blo][last/2] (0: natural)
  int64_t ss;          // ... and its slab stride in elements (>= slab size: padding de-aliases the power-of-two strides)
  int sub;             // two-level first-axis pass (nk_strided_body MODE 4 / 5): the OTHER factor of the line length, else 0
};

// columns of the tile that starts at column c0 of its slab which exist (tiles need not divide the slab width)
NK_HD int nk_tile_columns(const NkPassS& p, int64_t c0) {
  const int64_t left = p.inner - c0;
  return left < p.tl.tile ? (int)left : p.tl.tile;
}

template <typename T>
NK_HD void nk_passS_load(const NkPassS& p, int64_t blk, int tid, int nthr, C2<T>* lds,
                         const C2<T>* __restrict__ work) {
  const int n = p.lp.n, tile = p.tl.tile;
  const int64_t o = blk / p.tiles_per_slab;
  const int64_t c0 = (blk % p.tiles_per_slab) * (int64_t)tile;
  const C2<T>* src = work + o * n * p.inner + c0;
  const int total = n * tile;
  const int valid = nk_tile_columns(p, c0);  // (the last tile of a slab may hang over its end: those columns are zeros)
  for (int idx = tid; idx < total; idx += nthr) {
    int t, j;
    nk_fdivmod((uint32_t)idx, p.tl.dtile, j, t);
    lds[nk_lds_addr(p.tl, j, t)] = t < valid ? src[(int64_t)j * p.inner + t] : C2<T>{(T)0, (T)0};
  }
}

template <typename T>
NK_HD void nk_passB_store(const NkPassS& p, int64_t blk, int tid, int nthr, const C2<T>* lds,
                          C2<T>* __restrict__ work) {
  const int n = p.lp.n, tile = p.tl.tile;
  const int64_t o = blk / p.tiles_per_slab;
  const int64_t c0 = (blk % p.tiles_per_slab) * (int64_t)tile;
  C2<T>* dst = work + o * n * p.inner + c0;
  const int total = n * tile;
  const int valid = nk_tile_columns(p, c0);
  for (int idx = tid; idx < total; idx += nthr) {
    int t, k;
    nk_fdivmod((uint32_t)idx, p.tl.dtile, k, t);
    if (t < valid) dst[(int64_t)k * p.inner + t] = lds[nk_lds_addr(p.tl, nk_digit_reverse(p.lp, k), t)];
  }
}

// pass C store: Hartley combine + epilogue; the packed column goes to scratch S[batch][nm][na]
template <typename T>
NK_HD void nk_passC_store(const NkPassS& p, const NkFuse& f, int64_t blk, int tid, int nthr, const C2<T>* lds,
                          C2<T>* __restrict__ scratch, double& acc) {
  const int na = p.lp.n, tile = p.tl.tile, h = p.g.h, nm = p.g.nm, nl = p.g.nl;
  const T sg = (T)p.g.sign;
  const int64_t b = blk / p.tiles_per_slab;
  const int64_t c0 = (blk % p.tiles_per_slab) * (int64_t)tile;
  const int total = na * tile;
  const int m0 = (int)(c0 / h), kl0 = (int)(c0 % h);  // (once per workgroup; the columns of the tile follow by counting)
  const int valid = nk_tile_columns(p, c0);
  for (int idx = tid; idx < total; idx += nthr) {
    int t, k0;
    nk_fdivmod((uint32_t)idx, p.tl.dtile, k0, t);
    if (t >= valid) continue;
    int m = m0, kl = kl0 + t;
    while (kl >= h) {
      kl -= h;
      ++m;
    }
    const C2<T> F = lds[nk_lds_addr(p.tl, nk_digit_reverse(p.lp, k0), t)];
    if (kl == 0) {
      scratch[(b * nm + m) * na + k0] = F;
    } else {
      const int k0m = k0 ? na - k0 : 0, mm = m ? nm - m : 0;
      const int64_t o1 = ((b * na + k0) * nm + m) * nl + kl;
      const int64_t o2 = ((b * na + k0m) * nm + mm) * nl + (nl - kl);
      nk_epilogue_pair<T>(f, o1, F.x + sg * F.y, o2, F.x - sg * F.y, acc);
    }
  }
}

// pass D: untangle the packed column (k_last = 0 and k_last = h planes)
template <typename T>
NK_HD void nk_passD(const NkGeom& g, const NkFuse& f, int64_t gid, const C2<T>* __restrict__ scratch, double& acc) {
  const int na = g.na, nm = g.nm, nl = g.nl, h = g.h;
  const T sg = (T)g.sign;
  const int k0 = (int)(gid % na);
  const int64_t r = gid / na;
  const int m = (int)(r % nm);
  const int64_t b = r / nm;
  const int k0m = k0 ? na - k0 : 0, mm = m ? nm - m : 0;
  const C2<T> Z1 = scratch[(b * nm + m) * na + k0];
  const C2<T> Z2 = scratch[(b * nm + mm) * na + k0m];
  const C2<T> A{(T)0.5 * (Z1.x + Z2.x), (T)0.5 * (Z1.y - Z2.y)};
  const C2<T> D{(T)0.5 * (Z1.x - Z2.x), (T)0.5 * (Z1.y + Z2.y)};
  const C2<T> B{D.y, -D.x};
  const int64_t o = ((b * na + k0) * nm + m) * nl;
  nk_epilogue<T>(f, o, A.x + sg * A.y, acc);
  nk_epilogue<T>(f, o + h, B.x + sg * B.y, acc);
}
