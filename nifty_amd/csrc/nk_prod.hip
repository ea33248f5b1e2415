// nk_prod.hip -- amplitude fields of PRODUCT spectra: a correlated field on a product of harmonic sub-spaces has the
// amplitude  a(k) = s * prod_i t_i[pidx_i(k_i)]  -- one table per sub-space over that sub-space's power bins, one overall
// factor s (the zero-mode amplitude) -- (reference library/correlated_fields.py:713-764: the distributed amplitudes are
// multiplied on the full harmonic domain; :809-858 for the normalisation).  The reference materialises every factor on the
// full grid (ContractionOperator.adjoint @ PowerDistributor) and multiplies the fields; here the product is formed once,
// directly from the tables, and its adjoint -- the gradient with respect to every table -- is a weighted marginal sum.
//
// The grid is seen as [S0][S1][S2] (C order, up to three sub-spaces, missing ones have size 1); a sub-space may itself be
// multi-dimensional, S_i is its number of points and pidx_i its flattened bin index.  The same code serves the full grid
// and the OCTANT arrays of the register-resident transform pipeline (nk_fuse.field_octant / w8): the caller passes the
// octant sizes and the bin index of the sub-spaces' octant points.
#include <hip/hip_runtime.h>

#include "../../include/niftyk.h"
#include "nk_util.h"

namespace {

constexpr int PROD_THREADS = 256;

struct ProdArgs {
  int nsub;
  int64_t size[3];
  const int32_t* pidx[3];
  const double* tab[3];
  const double* dtab[3];
  const double* scale;
  const double* dscale;
};

__device__ __forceinline__ void prod_split(const ProdArgs& p, int64_t k, int64_t (&ks)[3]) {
  ks[2] = k % p.size[2];
  k /= p.size[2];
  ks[1] = k % p.size[1];
  ks[0] = k / p.size[1];
}

// value: out[k] = s prod_i t_i;   tangent (TAN): out[k] = ds prod_i t_i + s sum_i dt_i prod_{j != i} t_j
template <typename T, bool TAN>
__global__ void __launch_bounds__(PROD_THREADS) k_product_field(ProdArgs p, int64_t n, T* __restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * PROD_THREADS + threadIdx.x;
  if (k >= n) return;
  int64_t ks[3];
  prod_split(p, k, ks);
  double t[3] = {1.0, 1.0, 1.0}, dt[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i >= p.nsub) break;
    const int32_t b = p.pidx[i][ks[i]];
    t[i] = p.tab[i][b];
    if (TAN && p.dtab[i]) dt[i] = p.dtab[i][b];
  }
  const double s = *p.scale;
  if (!TAN) {
    out[k] = (T)(s * t[0] * t[1] * t[2]);
  } else {
    const double ds = p.dscale ? *p.dscale : 0.0;
    out[k] = (T)(ds * t[0] * t[1] * t[2] + s * (dt[0] * t[1] * t[2] + t[0] * dt[1] * t[2] + t[0] * t[1] * dt[2]));
  }
}

// ---- adjoint: marg[b] = sum over the points k with k_which = b of  w[k] * s * prod_{j != which} t_j[pidx_j(k_j)]
// The grid is [A][B][C] around the sub-space `which` (A = points of the sub-spaces before it, C = after it).
// rows: tmp[a][b] = before(a) * sum_c w[a][b][c] after(c) -- one wavefront per row (lanes stride c, then a shuffle tree);
// cols: partial[chunk][b] = sum of tmp over a chunk of rows a, in order; fold: marg[b] = sum of the partials, in order.
// Every sum has a fixed order: bit-reproducible.
__device__ __forceinline__ double prod_other(const ProdArgs& p, int which, bool before, int64_t idx) {
  // weight of the sub-spaces before (after) `which` at their combined point index idx
  double w = 1.0;
  if (before) {
    if (which == 2) {
      w = p.tab[0][p.pidx[0][idx / p.size[1]]] * p.tab[1][p.pidx[1][idx % p.size[1]]];
    } else if (which == 1) {
      w = p.tab[0][p.pidx[0][idx]];
    }
  } else {
    if (which == 0) {
      if (p.nsub > 1) w = p.tab[1][p.pidx[1][idx / p.size[2]]];
      if (p.nsub > 2) w *= p.tab[2][p.pidx[2][idx % p.size[2]]];
    } else if (which == 1) {
      if (p.nsub > 2) w = p.tab[2][p.pidx[2][idx]];
    }
  }
  return w;
}

__global__ void __launch_bounds__(PROD_THREADS) k_marg_rows(ProdArgs p, int which, int64_t A, int64_t Bn, int64_t C,
                                                            const double* __restrict__ w, double* __restrict__ tmp) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (PROD_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= A * Bn) return;
  const int64_t a = row / Bn;
  double s = 0.0;
  for (int64_t c = lane; c < C; c += 64) s += w[row * C + c] * prod_other(p, which, false, c);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) tmp[row] = s * prod_other(p, which, true, a) * *p.scale;
}

constexpr int MARG_ROWS = 128;  // rows per partial sum

__global__ void __launch_bounds__(PROD_THREADS) k_marg_cols(int64_t A, int64_t Bn, const double* __restrict__ tmp,
                                                            double* __restrict__ partial) {
  const int64_t b = (int64_t)blockIdx.x * PROD_THREADS + threadIdx.x;
  if (b >= Bn) return;
  const int64_t a0 = (int64_t)blockIdx.y * MARG_ROWS, a1 = a0 + MARG_ROWS < A ? a0 + MARG_ROWS : A;
  double s = 0.0;
  for (int64_t a = a0; a < a1; ++a) s += tmp[a * Bn + b];
  partial[(int64_t)blockIdx.y * Bn + b] = s;
}

__global__ void __launch_bounds__(PROD_THREADS) k_marg_fold(int64_t chunks, int64_t Bn, const double* __restrict__ partial,
                                                            double* __restrict__ marg) {
  const int64_t b = (int64_t)blockIdx.x * PROD_THREADS + threadIdx.x;
  if (b >= Bn) return;
  double s = 0.0;
  for (int64_t c = 0; c < chunks; ++c) s += partial[c * Bn + b];
  marg[b] = s;
}

// ---- separable Hartley transforms from the genuine one.  The harmonic transform of a PRODUCT domain is one Hartley
// transform per sub-space (library/correlated_fields.py:726-730: a chain of HarmonicTransformOperators with `space=`), i.e.
// the kernel prod_i cas(k_i.x_i), not cas(sum_i k_i.x_i).  cas(a) cas(b) = 1/2 [cas(a+b) + cas(a-b) + cas(-a+b) - cas(-a-b)]
// (and its three-factor analogue), so the separable transform is a fixed combination of the genuine transform at the
// points whose sub-space wave vectors are mirrored: out[k] = sum_s coef[s] in[flip_s(k)], s over the 2^nsub sign patterns.
// The combination commutes with the transform and is symmetric: it is applied to the OUTPUT of a forward evaluation and to
// the INPUT of an adjoint one, so that prologue and epilogue fusion of the single N-D transform stay what they are.
struct MirrorArgs {
  int ndim, nsub;
  int64_t n[3];
  int group[3];
  double coef[8];
};
template <typename T>
__global__ void __launch_bounds__(PROD_THREADS) k_mirror_combine(MirrorArgs m, int64_t total, const T* __restrict__ in,
                                                                  T* __restrict__ out, double scale, double offset) {
  const int64_t k = (int64_t)blockIdx.x * PROD_THREADS + threadIdx.x;
  if (k >= total) return;
  int64_t idx[3], r = k;
  for (int ax = m.ndim - 1; ax >= 0; --ax) {
    idx[ax] = r % m.n[ax];
    r /= m.n[ax];
  }
  double acc = 0.0;
  for (int s = 0; s < (1 << m.nsub); ++s) {
    int64_t flat = 0;
    for (int ax = 0; ax < m.ndim; ++ax) {
      const bool neg = (s >> m.group[ax]) & 1;
      const int64_t i = neg && idx[ax] ? m.n[ax] - idx[ax] : idx[ax];
      flat = flat * m.n[ax] + i;
    }
    acc += m.coef[s] * (double)in[flat];
  }
  out[k] = (T)(scale * acc + offset);
}

int prod_args(const nk_product* q, ProdArgs* p, int64_t* n) {
  if (!q || q->nsub < 1 || q->nsub > 3 || !q->scale) return nk_set_error(NK_ERR_INVALID, "nk_product: 1..3 sub-spaces and a scale");
  p->nsub = q->nsub;
  *n = 1;
  for (int i = 0; i < 3; ++i) {
    const bool on = i < q->nsub;
    if (on && (q->size[i] < 1 || !q->pidx[i] || !q->tab[i])) return nk_set_error(NK_ERR_INVALID, "nk_product: bad sub-space");
    p->size[i] = on ? q->size[i] : 1;
    p->pidx[i] = on ? q->pidx[i] : nullptr;
    p->tab[i] = on ? q->tab[i] : nullptr;
    p->dtab[i] = on ? q->dtab[i] : nullptr;
    *n *= p->size[i];
  }
  p->scale = q->scale;
  p->dscale = q->dscale;
  return NK_OK;
}

}  // namespace

extern "C" int nk_product_field(const nk_product* q, int tangent, void* out, int dtype, void* stream) {
  ProdArgs p;
  int64_t n;
  const int rc = prod_args(q, &p, &n);
  if (rc != NK_OK) return rc;
  if (!out || (dtype != NK_F32 && dtype != NK_F64)) return nk_set_error(NK_ERR_INVALID, "nk_product_field: bad argument");
  const int64_t blocks = (n + PROD_THREADS - 1) / PROD_THREADS;
  if (blocks > 0x7fffffffLL) return nk_set_error(NK_ERR_UNSUPPORTED, "nk_product_field: grid too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)blocks), block(PROD_THREADS);
  if (dtype == NK_F32) {
    if (tangent)
      hipLaunchKernelGGL((k_product_field<float, true>), grid, block, 0, st, p, n, (float*)out);
    else
      hipLaunchKernelGGL((k_product_field<float, false>), grid, block, 0, st, p, n, (float*)out);
  } else {
    if (tangent)
      hipLaunchKernelGGL((k_product_field<double, true>), grid, block, 0, st, p, n, (double*)out);
    else
      hipLaunchKernelGGL((k_product_field<double, false>), grid, block, 0, st, p, n, (double*)out);
  }
  return nk_check_launch("nk_product_field");
}

extern "C" size_t nk_product_marginal_scratch(const nk_product* q, int which) {
  ProdArgs p;
  int64_t n;
  if (prod_args(q, &p, &n) != NK_OK || which < 0 || which >= p.nsub) return 0;
  int64_t A = 1;
  for (int i = 0; i < which; ++i) A *= p.size[i];
  const int64_t Bn = p.size[which];
  return sizeof(double) * (size_t)(A * Bn + ((A + MARG_ROWS - 1) / MARG_ROWS) * Bn);
}

extern "C" int nk_product_marginal(const nk_product* q, int which, const double* w, double* scratch, double* marg,
                                   void* stream) {
  ProdArgs p;
  int64_t n;
  const int rc = prod_args(q, &p, &n);
  if (rc != NK_OK) return rc;
  if (which < 0 || which >= p.nsub || !w || !scratch || !marg)
    return nk_set_error(NK_ERR_INVALID, "nk_product_marginal: bad argument");
  int64_t A = 1, C = 1;
  for (int i = 0; i < which; ++i) A *= p.size[i];
  for (int i = which + 1; i < 3; ++i) C *= p.size[i];
  const int64_t Bn = p.size[which];
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = A * Bn, row_blocks = (rows + PROD_THREADS / 64 - 1) / (PROD_THREADS / 64);
  const int64_t chunks = (A + MARG_ROWS - 1) / MARG_ROWS, col_blocks = (Bn + PROD_THREADS - 1) / PROD_THREADS;
  if (row_blocks > 0x7fffffffLL || chunks > 65535 || col_blocks > 0x7fffffffLL)
    return nk_set_error(NK_ERR_UNSUPPORTED, "nk_product_marginal: grid too large");
  double* tmp = scratch;
  double* partial = scratch + rows;
  hipLaunchKernelGGL(k_marg_rows, dim3((unsigned)row_blocks), dim3(PROD_THREADS), 0, st, p, which, A, Bn, C, w, tmp);
  hipLaunchKernelGGL(k_marg_cols, dim3((unsigned)col_blocks, (unsigned)chunks), dim3(PROD_THREADS), 0, st, A, Bn, tmp, partial);
  hipLaunchKernelGGL(k_marg_fold, dim3((unsigned)col_blocks), dim3(PROD_THREADS), 0, st, chunks, Bn, partial, marg);
  return nk_check_launch("nk_product_marginal");
}

extern "C" int nk_mirror_combine(int ndim, const int64_t* shape, const int* group, int nsub, const double* coef, const void* in,
                                 void* out, double scale, double offset, int dtype, void* stream) {
  if (ndim < 1 || ndim > 3 || nsub < 1 || nsub > 3 || !shape || !group || !coef || !in || !out || in == out)
    return nk_set_error(NK_ERR_INVALID, "nk_mirror_combine: bad argument (1..3 axes and sub-spaces, out of place)");
  MirrorArgs m;
  m.ndim = ndim, m.nsub = nsub;
  int64_t total = 1;
  for (int ax = 0; ax < 3; ++ax) {
    m.n[ax] = ax < ndim ? shape[ax] : 1;
    m.group[ax] = ax < ndim ? group[ax] : 0;
    if (m.n[ax] < 1 || m.group[ax] < 0 || m.group[ax] >= nsub) return nk_set_error(NK_ERR_INVALID, "nk_mirror_combine: bad axis");
    total *= m.n[ax];
  }
  for (int s = 0; s < 8; ++s) m.coef[s] = s < (1 << nsub) ? coef[s] : 0.0;
  const int64_t blocks = (total + PROD_THREADS - 1) / PROD_THREADS;
  if (blocks > 0x7fffffffLL) return nk_set_error(NK_ERR_UNSUPPORTED, "nk_mirror_combine: grid too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == NK_F32)
    hipLaunchKernelGGL(k_mirror_combine<float>, dim3((unsigned)blocks), dim3(PROD_THREADS), 0, st, m, total, (const float*)in,
                       (float*)out, scale, offset);
  else if (dtype == NK_F64)
    hipLaunchKernelGGL(k_mirror_combine<double>, dim3((unsigned)blocks), dim3(PROD_THREADS), 0, st, m, total, (const double*)in,
                       (double*)out, scale, offset);
  else
    return nk_set_error(NK_ERR_INVALID, "dtype must be NK_F32 or NK_F64");
  return nk_check_launch("nk_mirror_combine");
}
