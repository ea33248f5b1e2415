// nk_plan.h -- host-side planning shared by the HIP library and the test-only host emulation:
// radix decomposition, LDS tile selection for the three pass kinds, twiddle tables.
#pragma once
#include <cmath>
#include <cstdlib>
#include <vector>

#include "nk_fft_phases.h"

// radices 8/4/2 first, then the odd ones; nstage = -1 when n has a prime factor > 7 or needs too many stages.
// (Round 4 tried composite radices 6, 9, 10, 12, 15 with a planner that minimises the number of stages -- 768 = 12*8*8 instead
// of 8*8*4*3: one LDS round trip and barrier less.  The direct O(R^2) butterflies cost more than that saves (960^3 fp32
// 17.4 -> 18.0 ms, 3000^2 fp64 0.25 -> 0.40 ms), and their mere presence in the stage switch cost every length 6 % through the
// kernel's register count; removed again, numbers in profiles/r04_generic_sweep.log.)
static inline NkLinePlan nk_make_line_plan(int n) {
  NkLinePlan lp{};
  lp.n = n;
  lp.nstage = 0;
  int rem = n;
  while (rem > 1) {
    int R;
    if (rem % 8 == 0) R = (rem == 16) ? 4 : 8;  // 8*2 -> 4*4
    else if (rem % 4 == 0) R = 4;
    else if (rem % 2 == 0) R = 2;
    else if (rem % 3 == 0) R = 3;
    else if (rem % 5 == 0) R = 5;
    else if (rem % 7 == 0) R = 7;
    else {
      lp.nstage = -1;
      return lp;
    }
    if (lp.nstage == NK_MAX_STAGES) {
      lp.nstage = -1;
      return lp;
    }
    lp.radix[lp.nstage++] = R;
    rem /= R;
  }
  int span = n;
  for (int s = 0; s < lp.nstage; ++s) {  // what the kernels divide by, as multiply-high constants (NkDiv)
    span /= lp.radix[s];
    lp.span[s] = span;
    lp.dradix[s] = nk_make_div(lp.radix[s]);
    lp.dspan[s] = nk_make_div(span);
    lp.dnbf[s] = nk_make_div(n / lp.radix[s]);
  }
  return lp;
}
static inline bool nk_factorable(int64_t n) { return n >= 1 && n < ((int64_t)1 << 30) && nk_make_line_plan((int)n).nstage >= 0; }

static inline int nk_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

struct NkHostPlan {
  NkGeom g{};
  int dtype = 1;
  size_t csize = 16;  // bytes per complex element
  // pass A (last axis)
  NkPassA pa{};
  int threads_a = 256;
  size_t lds_a = 0;
  // pass B (middle axis, 3-D only) and pass C (first axis, ndim >= 2)
  NkPassS pb{}, pc{};
  int threads_b = 256, threads_c = 256;
  size_t lds_b = 0, lds_c = 0;
  // twiddles (as double pairs; converted to the plan dtype when uploaded)
  std::vector<double> tw_a, twr_a, tw_b, tw_c, tw_f;  // tw_f: full-length table of the last axis (final pass)
  std::vector<double> tw_t64, tw_t32;                 // sub-line tables of the two-level first-axis pass (2-D)
  size_t work_bytes = 0, scratch_bytes = 0;
};

static inline void nk_fill_twiddle(std::vector<double>& tw, int n, int denom) {
  // tw[k] = exp(-2 pi i k / denom), k = 0..n-1
  tw.resize(2 * (size_t)(n > 0 ? n : 1));
  const long double w = -2.0L * 3.14159265358979323846264338327950288L / (long double)denom;
  for (int k = 0; k < n; ++k) {
    tw[2 * k] = (double)cosl(w * k);
    tw[2 * k + 1] = (double)sinl(w * k);
  }
  if (n <= 0) tw[0] = 1.0, tw[1] = 0.0;
}

static inline int nk_round_threads(int64_t work) {
  int64_t t = (work + 63) / 64 * 64;
  if (t < 64) t = 64;
  if (t > 1024) t = 1024;
  return (int)t;
}

// strided tile: largest power-of-two T <= inner with n*T*csize <= budget and T*csize <= 256 B.  The tile need not divide
// the slab width (the last tile of a slab is cut short, nk_tile_columns): a width like 500 = 4 * 125 used to end up with
// rows of four elements (32 B per request).  NK_TILE_DIVIDES=1 restores the old rule.
static inline int nk_pick_strided_tile(int n, int64_t inner, size_t csize, const char* env, int64_t outer = 1 << 20) {
  int forced = nk_env_int(env, 0);
  // 128 KiB of LDS per workgroup (one workgroup per CU): rows of 16 columns for lines of 768 ... 1000 fp32 elements beat two
  // resident workgroups with rows of 8 (768^3: -12 %, profiles/r04_generic_sweep.log)
  size_t budget = 128 * 1024;
  int T = 1;
  static const int divides = nk_env_int("NK_TILE_DIVIDES", 0);
  auto fits = [&](int t) { return (int64_t)t <= inner && (!divides || inner % t == 0); };
  while (fits(T * 2) && (size_t)n * (T * 2) * csize <= budget && (size_t)(T * 2) * csize <= 256) T *= 2;
  if ((size_t)T * csize < 64) {  // rows shorter than 64 B: all the LDS a workgroup can have
    budget = 152 * 1024;
    while (fits(T * 2) && (size_t)n * (T * 2) * csize <= budget && (size_t)(T * 2) * csize <= 256) T *= 2;
  }
  // small problems: rather narrower tiles than fewer workgroups than CUs (1000^2 fp64: 63 workgroups with rows of 8)
  while (T > 4 && outer * ((inner + T - 1) / T) < 256) T /= 2;
  if (forced > 0) {
    T = 1;
    while (T * 2 <= forced && fits(T * 2) && (size_t)n * (T * 2) * csize <= 152 * 1024) T *= 2;
  }
  return T;
}

// returns 0 on success, negative nk_status otherwise; msg receives a static description
static inline int nk_host_plan_init(NkHostPlan& P, int ndim, const int64_t* shape, int dtype, int64_t batch,
                                    const char** msg) {
  *msg = "";
  if (ndim < 1 || ndim > 3) {
    *msg = "only 1, 2 or 3 transformed axes are supported";
    return NK_ERR_UNSUPPORTED;
  }
  if (dtype != NK_F32 && dtype != NK_F64) {
    *msg = "dtype must be NK_F32 or NK_F64";
    return NK_ERR_INVALID;
  }
  if (batch < 1) {
    *msg = "batch must be >= 1";
    return NK_ERR_INVALID;
  }
  for (int d = 0; d < ndim; ++d) {
    if (shape[d] < 1) {
      *msg = "axis lengths must be positive";
      return NK_ERR_INVALID;
    }
    if (!nk_factorable(shape[d])) {
      *msg = "axis lengths must factor into 2, 3, 5 and 7";
      return NK_ERR_UNSUPPORTED;
    }
  }
  if (shape[ndim - 1] < 2 || shape[ndim - 1] % 2 != 0) {
    *msg = "the last axis must have an even length >= 2 (real-to-complex packing)";
    return NK_ERR_UNSUPPORTED;
  }
  P.dtype = dtype;
  P.csize = dtype == NK_F32 ? 8 : 16;
  NkGeom& g = P.g;
  g.ndim = ndim;
  g.batch = (int)batch;
  g.nl = (int)shape[ndim - 1];
  g.h = g.nl / 2;
  g.na = ndim >= 2 ? (int)shape[0] : 1;
  g.nm = ndim == 3 ? (int)shape[1] : 1;
  g.sign = 1;
  const size_t max_line_bytes = 144 * 1024;
  if ((size_t)(g.h + g.h / 16 + 1) * P.csize > max_line_bytes || (size_t)g.na * P.csize > max_line_bytes ||
      (size_t)g.nm * P.csize > max_line_bytes) {
    *msg = "axis too long for a single-LDS line transform";
    return NK_ERR_UNSUPPORTED;
  }
  const int64_t n_total = batch * g.na * g.nm * g.nl;
  if (n_total >= ((int64_t)1 << 40)) {
    *msg = "array too large";
    return NK_ERR_UNSUPPORTED;
  }
  // ---- pass A
  P.pa.g = g;
  P.pa.lp = nk_make_line_plan(g.h);
  P.pa.nlines = batch * g.na * g.nm;
  {
    const int lstride = g.h + g.h / 16 + 1;
    const size_t line_bytes = (size_t)lstride * P.csize;
    int64_t tile = (int64_t)(32 * 1024 / line_bytes);
    const int64_t want = (3072 + g.h - 1) / g.h;  // ~3072 complex elements per workgroup, 12 per thread (sweep: r04_generic_sweep.log)
    if (tile > want) tile = want;
    if (tile < 1) tile = 1;
    if (tile > P.pa.nlines) tile = P.pa.nlines;
    int forced = nk_env_int("NK_TILE_A", 0);
    if (forced > 0 && (size_t)forced * line_bytes <= 152 * 1024) tile = forced;
    P.pa.tl.tile = (int)tile;
    P.pa.tl.t_fastest = 0;
    P.pa.tl.tstride = 0;
    P.pa.tl.lstride = lstride;
    P.pa.tl.dtile = nk_make_div(tile);
    P.pa.dh = nk_make_div(g.h);
    P.pa.dnk = nk_make_div(g.h / 2 + 1);
    P.lds_a = (size_t)tile * line_bytes;
    P.threads_a = nk_round_threads(tile * g.h / 12);
    if (P.threads_a > 256) P.threads_a = 256;
    int ft = nk_env_int("NK_THREADS_A", 0);
    if (ft >= 64) P.threads_a = ft;
  }
  nk_fill_twiddle(P.tw_a, g.h, g.h);
  nk_fill_twiddle(P.twr_a, g.h / 2 + 1, g.nl);
  nk_fill_twiddle(P.tw_f, g.nl, g.nl);
  if (ndim == 2) nk_fill_twiddle(P.tw_t64, 64, 64), nk_fill_twiddle(P.tw_t32, 32, 32);
  // ---- pass B / C
  auto setup_strided = [&](NkPassS& ps, int n, int64_t outer, int64_t inner, const char* env, int& threads,
                           size_t& lds, std::vector<double>& tw) {
    ps.g = g;
    ps.lp = nk_make_line_plan(n);
    ps.outer = outer;
    ps.inner = inner;
    const int T = nk_pick_strided_tile(n, inner, P.csize, env, outer);
    ps.tl.tile = T;
    ps.tl.t_fastest = 1;
    ps.tl.tstride = T;
    ps.tl.lstride = 0;
    ps.tl.dtile = nk_make_div(T);
    ps.tiles_per_slab = (int)((inner + T - 1) / T);
    lds = (size_t)n * T * P.csize;
    threads = nk_round_threads((int64_t)n * T / 4);
    if (threads > 512) threads = 512;  // (1024-thread workgroups: +10 % on the strided passes of 768^3)
    int ft = nk_env_int("NK_THREADS_S", 0);
    if (ft >= 64) threads = ft;
    nk_fill_twiddle(tw, n, n);
  };
  if (ndim == 3) setup_strided(P.pb, g.nm, batch * g.na, g.h, "NK_TILE_B", P.threads_b, P.lds_b, P.tw_b);
  if (ndim >= 2)
    setup_strided(P.pc, g.na, batch, (int64_t)g.nm * g.h, "NK_TILE_C", P.threads_c, P.lds_c, P.tw_c);
  // + room for the padded slab strides of the strided-first pipeline (nk_pipe2_setup: <= NK_WORK_PAD_MAX per slab)
  // ... and for the rows of nl/2 + 16 columns of the sandwich pipeline (nk_fft3.h)
  P.work_bytes = ((size_t)(n_total / 2) + (size_t)batch * g.na * g.nm * 16 +
                  (size_t)batch * (g.na > g.nm ? g.na : g.nm) * 8192) * P.csize;
  P.scratch_bytes = ndim >= 2 ? (size_t)batch * g.nm * g.na * P.csize : 0;
  return NK_OK;
}
