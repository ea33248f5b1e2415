// nk_amp.hip -- the CorrelatedField amplitude model on the nb power bins, forward / JVP / VJP.
//
// Restates (not copies) nifty/cl/library/correlated_fields.py: _TwoLogIntegrations (:119-162),
// _SlopeRemover (:89-116), _Normalization (:165-208), _Amplitude (:277-386) and the zero-mode handling
// of CorrelatedFieldMaker.finalize / get_normalized_amplitudes (:713-764, :809-859), for the single
// amplitude / total_N == 0 case, in closed form:
//   sig0 = flex sqrt(D) sqrt(D^2/12 + asp),  sig1 = flex sqrt(D)           (D = log-k bin widths)
//   x0 = sig0 xs0, x1 = sig1 xs1
//   c = cumsum(x1);  smooth_{j+2} = cumsum( (c_j + c_{j-1})/2 D_j + x0_j )
//   p = slope rel + smooth - smooth_last sc;  spec = exp(p);  S = sum mult spec
//   a_0 = V zm;  a_b = V fluct sqrt(spec_b / S)
// The double cumulative sum is ONE scan with the associative affine operator
//   (c, s) -> (c + A, s + c*Dk + B),  compose(l, r) = (Al+Ar, Dl+Dr, Bl+Br+Al*Dr)
// so each of forward / JVP / VJP is a single launch.  Round-1 implementation: one 1024-thread
// workgroup walks the bins in coalesced tiles (nb <= ~1.2e6 -> a few hundred tiles); a multi-workgroup
// look-back version is the planned follow-up.
#include <hip/hip_runtime.h>

#include "nk_util.h"

namespace {

constexpr int AMP_THREADS = 1024;
constexpr int AMP_WAVES = AMP_THREADS / 64;

struct Seg {
  double A, D, B;
};
__device__ __forceinline__ Seg seg_combine(const Seg& l, const Seg& r) {  // l first, then r
  return Seg{l.A + r.A, l.D + r.D, l.B + r.B + l.A * r.D};
}
__device__ __forceinline__ Seg seg_shfl_up(const Seg& s, int off) {
  return Seg{__shfl_up(s.A, off, 64), __shfl_up(s.D, off, 64), __shfl_up(s.B, off, 64)};
}

// exclusive prefix (composite of all segments of lower thread ids) and block total
__device__ __forceinline__ void block_scan_seg(const Seg& mine, Seg& excl, Seg& total, Seg* sh /*[AMP_WAVES]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  Seg incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    Seg o = seg_shfl_up(incl, off);
    if (lane >= off) incl = seg_combine(o, incl);
  }
  __syncthreads();
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  Seg wprefix{0.0, 0.0, 0.0};
  for (int w = 0; w < wave; ++w) wprefix = seg_combine(wprefix, sh[w]);
  Seg up = seg_shfl_up(incl, 1);
  Seg lane_excl = lane == 0 ? Seg{0.0, 0.0, 0.0} : up;
  excl = seg_combine(wprefix, lane_excl);
  Seg t{0.0, 0.0, 0.0};
  for (int w = 0; w < AMP_WAVES; ++w) t = seg_combine(t, sh[w]);
  total = t;
}

__device__ __forceinline__ double block_sum(double v, double* sh /*[AMP_WAVES]*/) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < AMP_WAVES; ++w) s += sh[w];
  return s;
}

// geo layout: rel[nb] | sc[nb] | mult[nb] | delta[nb]
// hyp layout: lm_fluct, ls_fluct, lm_flex, ls_flex, lm_asp, ls_asp, lm_zm, ls_zm, slope_mean, slope_sigma, V
// lat layout: xi_asp, xi_flex, xi_fluct, xi_slope, xi_zm, spectrum[2][nb-2]
// state layout: [0] flex [1] asp [2] fluct [3] zm [4] slope [5] S [6] last ; 16: spec[nb] | ahat[nb] | tmp[nb] | tmp2[nb]
struct Hyper {
  double flex, asp, fluct, zm, slope;
};
__device__ __forceinline__ Hyper hyper_from_lat(const double* hyp, const double* lat) {
  Hyper h;
  h.asp = exp(hyp[4] + hyp[5] * lat[0]);
  h.flex = exp(hyp[2] + hyp[3] * lat[1]);
  h.fluct = exp(hyp[0] + hyp[1] * lat[2]);
  h.slope = hyp[8] + hyp[9] * lat[3];
  h.zm = exp(hyp[6] + hyp[7] * lat[4]);
  return h;
}

constexpr int EPT = 4;  // elements per thread per tile

// runs the affine scan over m = nb-2 elements; elem(j) returns the segment of element j (in scan order),
// emit(j, c, s) receives the inclusive state after element j.  `reverse` walks j = m-1 .. 0.
template <typename ElemF, typename EmitF>
__device__ __forceinline__ Seg affine_scan(int m, bool reverse, ElemF elem, EmitF emit, Seg* sh) {
  Seg carry{0.0, 0.0, 0.0};
  const int tile = AMP_THREADS * EPT;
  for (int base = 0; base < m; base += tile) {
    Seg local[EPT];
    Seg agg{0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int q = base + threadIdx.x * EPT + e;  // position in scan order
      if (q < m) {
        local[e] = elem(reverse ? m - 1 - q : q);
      } else {
        local[e] = Seg{0.0, 0.0, 0.0};
      }
      agg = seg_combine(agg, local[e]);
    }
    Seg excl, total;
    block_scan_seg(agg, excl, total, sh);
    Seg run = seg_combine(carry, excl);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int q = base + threadIdx.x * EPT + e;
      run = seg_combine(run, local[e]);
      if (q < m) emit(reverse ? m - 1 - q : q, run.A, run.B);
    }
    carry = seg_combine(carry, total);
    __syncthreads();
  }
  return carry;
}

__global__ void __launch_bounds__(AMP_THREADS) k_amp_forward(int nb, const double* __restrict__ geo,
                                                             const double* __restrict__ hyp,
                                                             const double* __restrict__ lat, double* __restrict__ state,
                                                             double* __restrict__ amp) {
  __shared__ Seg sh_seg[AMP_WAVES];
  __shared__ double sh_d[AMP_WAVES];
  const double* rel = geo;
  const double* sc = geo + nb;
  const double* mult = geo + 2 * (size_t)nb;
  const double* delta = geo + 3 * (size_t)nb;
  const int m = nb - 2;
  const Hyper h = hyper_from_lat(hyp, lat);
  const double* xs0 = lat + 5;
  const double* xs1 = lat + 5 + m;
  double* spec = state + 16;
  double* ahat = spec + nb;
  double* smooth = ahat + nb;  // tmp
  const double V = hyp[10];
  if (threadIdx.x == 0) {
    state[0] = h.flex, state[1] = h.asp, state[2] = h.fluct, state[3] = h.zm, state[4] = h.slope;
    smooth[0] = 0.0;
    smooth[1] = 0.0;
  }
  Seg tot = affine_scan(
      m, false,
      [&](int j) {
        const double d = delta[j];
        const double sq = sqrt(d);
        const double x0 = h.flex * sq * sqrt(d * d / 12.0 + h.asp) * xs0[j];
        const double x1 = h.flex * sq * xs1[j];
        return Seg{x1, d, 0.5 * x1 * d + x0};
      },
      [&](int j, double, double s) { smooth[j + 2] = s; }, sh_seg);
  const double last = tot.B;  // smooth[nb-1]
  __syncthreads();
  double part = 0.0;
  for (int b = threadIdx.x; b < nb; b += AMP_THREADS) {
    const double p = h.slope * rel[b] + smooth[b] - last * sc[b];
    const double s = exp(p);
    spec[b] = s;
    part += mult[b] * s;
  }
  const double S = block_sum(part, sh_d);
  if (threadIdx.x == 0) state[5] = S, state[6] = last;
  for (int b = threadIdx.x; b < nb; b += AMP_THREADS) {
    const double ah = sqrt(spec[b] / S);
    ahat[b] = ah;
    amp[b] = b == 0 ? V * h.zm : V * h.fluct * ah;
  }
}

__global__ void __launch_bounds__(AMP_THREADS) k_amp_jvp(int nb, const double* __restrict__ geo,
                                                         const double* __restrict__ hyp,
                                                         const double* __restrict__ lat, double* __restrict__ state,
                                                         const double* __restrict__ dlat, double* __restrict__ damp) {
  __shared__ Seg sh_seg[AMP_WAVES];
  __shared__ double sh_d[AMP_WAVES];
  const double* rel = geo;
  const double* sc = geo + nb;
  const double* mult = geo + 2 * (size_t)nb;
  const double* delta = geo + 3 * (size_t)nb;
  const int m = nb - 2;
  const double flex = state[0], asp = state[1], fluct = state[2], zm = state[3], S = state[5];
  const double* spec = state + 16;
  const double* ahat = spec + nb;
  double* dsm = state + 16 + 2 * (size_t)nb;  // tmp
  const double V = hyp[10];
  const double dasp = asp * hyp[5] * dlat[0];
  const double dflex = flex * hyp[3] * dlat[1];
  const double dfluct = fluct * hyp[1] * dlat[2];
  const double dslope = hyp[9] * dlat[3];
  const double dzm = zm * hyp[7] * dlat[4];
  const double* xs0 = lat + 5;
  const double* xs1 = lat + 5 + m;
  const double* dxs0 = dlat + 5;
  const double* dxs1 = dlat + 5 + m;
  if (threadIdx.x == 0) dsm[0] = 0.0, dsm[1] = 0.0;
  Seg tot = affine_scan(
      m, false,
      [&](int j) {
        const double d = delta[j];
        const double sq = sqrt(d);
        const double w0 = sqrt(d * d / 12.0 + asp);
        const double sig0 = flex * sq * w0, sig1 = flex * sq;
        const double dsig0 = dflex * sq * w0 + flex * sq * (0.5 / w0) * dasp;
        const double dsig1 = dflex * sq;
        const double dx0 = dsig0 * xs0[j] + sig0 * dxs0[j];
        const double dx1 = dsig1 * xs1[j] + sig1 * dxs1[j];
        return Seg{dx1, d, 0.5 * dx1 * d + dx0};
      },
      [&](int j, double, double s) { dsm[j + 2] = s; }, sh_seg);
  const double last = tot.B;
  __syncthreads();
  double part = 0.0;
  for (int b = threadIdx.x; b < nb; b += AMP_THREADS) {
    const double dp = dslope * rel[b] + dsm[b] - last * sc[b];
    dsm[b] = dp;
    part += mult[b] * spec[b] * dp;
  }
  const double dS = block_sum(part, sh_d);
  for (int b = threadIdx.x; b < nb; b += AMP_THREADS) {
    const double dah = 0.5 * ahat[b] * (dsm[b] - dS / S);
    damp[b] = b == 0 ? V * dzm : V * (dfluct * ahat[b] + fluct * dah);
  }
}

__global__ void __launch_bounds__(AMP_THREADS) k_amp_vjp(int nb, const double* __restrict__ geo,
                                                         const double* __restrict__ hyp,
                                                         const double* __restrict__ lat, double* __restrict__ state,
                                                         const double* __restrict__ abar, double* __restrict__ latbar) {
  __shared__ Seg sh_seg[AMP_WAVES];
  __shared__ double sh_d[AMP_WAVES];
  const double* rel = geo;
  const double* sc = geo + nb;
  const double* mult = geo + 2 * (size_t)nb;
  const double* delta = geo + 3 * (size_t)nb;
  const int m = nb - 2;
  const double flex = state[0], asp = state[1], fluct = state[2], zm = state[3], S = state[5];
  const double* spec = state + 16;
  const double* ahat = spec + nb;
  double* pbar = state + 16 + 2 * (size_t)nb;  // tmp
  const double V = hyp[10];
  const double* xs0 = lat + 5;
  const double* xs1 = lat + 5 + m;
  // reductions: fl_bar, Q = sum q
  double p_fl = 0.0, p_q = 0.0;
  for (int b = threadIdx.x + 1; b < nb; b += AMP_THREADS) {
    p_fl += abar[b] * ahat[b];
    p_q += 0.5 * ahat[b] * (V * fluct * abar[b]);
  }
  const double fl_bar = V * block_sum(p_fl, sh_d);
  const double Q = block_sum(p_q, sh_d);
  double p_sl = 0.0, p_sc = 0.0;
  for (int b = threadIdx.x; b < nb; b += AMP_THREADS) {
    const double q = b == 0 ? 0.0 : 0.5 * ahat[b] * (V * fluct * abar[b]);
    const double pb = q - (Q / S) * mult[b] * spec[b];
    pbar[b] = pb;
    p_sl += pb * rel[b];
    p_sc += pb * sc[b];
  }
  const double slope_bar = block_sum(p_sl, sh_d);
  const double sc_dot = block_sum(p_sc, sh_d);
  __syncthreads();
  // y = pbar with y[nb-1] -= sc_dot   (slope remover adjoint); reverse affine scan
  double* sbar0 = latbar + 5;
  double* sbar1 = latbar + 5 + m;
  double p_flex = 0.0, p_asp = 0.0;
  affine_scan(
      m, true,
      [&](int j) {
        double y = pbar[j + 2];
        if (j + 2 == nb - 1) y -= sc_dot;
        const double dj = delta[j];
        const double dn = j + 1 < m ? delta[j + 1] : 0.0;
        return Seg{y, 0.5 * (dj + dn), 0.5 * y * dj};
      },
      [&](int j, double t, double g1) {
        const double d = delta[j];
        const double sq = sqrt(d);
        const double w0 = sqrt(d * d / 12.0 + asp);
        const double sig0 = flex * sq * w0, sig1 = flex * sq;
        sbar0[j] = t * sig0;
        sbar1[j] = g1 * sig1;
        const double s0b = t * xs0[j], s1b = g1 * xs1[j];
        p_flex += s0b * sq * w0 + s1b * sq;
        p_asp += s0b * flex * sq * 0.5 / w0;
      },
      sh_seg);
  const double flex_bar = block_sum(p_flex, sh_d);
  const double asp_bar = block_sum(p_asp, sh_d);
  if (threadIdx.x == 0) {
    latbar[0] = asp_bar * asp * hyp[5];
    latbar[1] = flex_bar * flex * hyp[3];
    latbar[2] = fl_bar * fluct * hyp[1];
    latbar[3] = slope_bar * hyp[9];
    latbar[4] = V * abar[0] * zm * hyp[7];
  }
}

}  // namespace

extern "C" int nk_amp_forward(int nb, const double* geo, const double* hyp, const double* lat, double* state,
                              double* amp, void* stream) {
  if (nb < 3 || !geo || !hyp || !lat || !state || !amp) return nk_set_error(NK_ERR_INVALID, "nk_amp_forward: bad argument");
  hipLaunchKernelGGL(k_amp_forward, dim3(1), dim3(AMP_THREADS), 0, (hipStream_t)stream, nb, geo, hyp, lat, state, amp);
  return nk_check_launch("k_amp_forward");
}

extern "C" int nk_amp_jvp(int nb, const double* geo, const double* hyp, const double* lat, double* state,
                          const double* dlat, double* damp, void* stream) {
  if (nb < 3 || !geo || !hyp || !lat || !state || !dlat || !damp)
    return nk_set_error(NK_ERR_INVALID, "nk_amp_jvp: bad argument");
  hipLaunchKernelGGL(k_amp_jvp, dim3(1), dim3(AMP_THREADS), 0, (hipStream_t)stream, nb, geo, hyp, lat, state, dlat, damp);
  return nk_check_launch("k_amp_jvp");
}

extern "C" int nk_amp_vjp(int nb, const double* geo, const double* hyp, const double* lat, double* state,
                          const double* abar, double* latbar, void* stream) {
  if (nb < 3 || !geo || !hyp || !lat || !state || !abar || !latbar)
    return nk_set_error(NK_ERR_INVALID, "nk_amp_vjp: bad argument");
  hipLaunchKernelGGL(k_amp_vjp, dim3(1), dim3(AMP_THREADS), 0, (hipStream_t)stream, nb, geo, hyp, lat, state, abar, latbar);
  return nk_check_launch("k_amp_vjp");
}
