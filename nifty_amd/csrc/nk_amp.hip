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
// executed as a two-launch multi-workgroup scan (per-workgroup aggregates, then carry-in + rescan), followed by
// the element-wise / reduction stages: forward and JVP are 3 launches, VJP 5 over <= 256 workgroups (every reduction
// STORES its result, amp_store_sums: nothing needs zeroing in between).
// Every kernel runs a BATCH of latent points (include/niftyk.h, "batched launches"): blockIdx.y = member, the member's
// latent vector, state and in / out arrays come from a pointer table; the single entry points are batches of one.
#include <hip/hip_runtime.h>

#include "nk_util.h"

namespace {

constexpr int AMP_THREADS = 256;
constexpr int AMP_WAVES = AMP_THREADS / 64;

struct Seg {
  double A, D, B;
};
// per-member arrays of a batched launch: lat, state and one more input (dlat / abar) and output (amp / damp / latbar)
struct AmpBatch {
  const double* lat[NK_MAX_BATCH];
  double* state[NK_MAX_BATCH];
  const double* in[NK_MAX_BATCH];
  double* out[NK_MAX_BATCH];
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

// DETERMINISTIC cross-workgroup sums of the amplitude scalars (S, dS, the VJP reductions): every workgroup stores its
// partial(s); the workgroup that takes the last ticket adds them in block order with the fixed tree of block_sum and
// STORES the result -- the same bits on every run (until round 2: one fp64 atomic per workgroup, order-dependent in the
// last bit, and every later kernel of the step inherits S).  The ticket lives in the caller's STATE buffer (state[14],
// zeroed by the first kernel of nk_amp_forward, left at zero by every user), next to the partials: launches that work
// on different states -- other streams, other host threads -- never share scratch.
__device__ __forceinline__ unsigned int* amp_ticket(double* state) { return reinterpret_cast<unsigned int*>(state + 14); }
template <int NV>
__device__ __forceinline__ void amp_store_sums(const double (&v)[NV] /* valid in thread 0 */, double* part,
                                               double* const (&dst)[NV], double* sh /*[AMP_WAVES]*/, unsigned int* ticket) {
  __shared__ bool is_last;
  if (threadIdx.x == 0) {
    // partials -> coherence point, then the ticket; the last workgroup reads them from there (nk_util.h, nk_publish_partial)
#pragma unroll
    for (int k = 0; k < NV; ++k) nk_publish_partial(&part[k * gridDim.x + blockIdx.x], v[k]);
    is_last = nk_take_last_ticket(ticket, gridDim.x);
  }
  __syncthreads();
  if (!is_last) return;
  nk_acquire_partials();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double x = 0.0;
    for (int g = threadIdx.x; g < (int)gridDim.x; g += AMP_THREADS) x += nk_read_partial(&part[k * gridDim.x + g]);
    const double t = block_sum(x, sh);
    if (threadIdx.x == 0) *dst[k] = t;
  }
  if (threadIdx.x == 0) nk_reset_ticket(ticket);
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
constexpr int MAXG = 256;  // workgroups per launch

// Multi-workgroup affine scan in two launches:
//   aggregate: workgroup g reduces its chunk of the scan order to one segment  -> segs[g]
//   apply    : workgroup g combines segs[0..g) (its carry-in) and all segs (the total), rescans its chunk
//              and emits the inclusive state after every element.
// elem(j) returns the segment of element j; `reverse` walks j = m-1 .. 0.
struct ScanGeom {
  int m;       // elements
  int chunk;   // elements per workgroup (multiple of the tile)
  int ngroups;
};

static inline ScanGeom make_scan_geom(int m) {
  const int tile = AMP_THREADS * EPT;
  int ng = (m + tile - 1) / tile;
  if (ng > MAXG) ng = MAXG;
  if (ng < 1) ng = 1;
  int chunk = (m + ng - 1) / ng;
  chunk = (chunk + tile - 1) / tile * tile;
  ng = (m + chunk - 1) / chunk;
  if (ng < 1) ng = 1;
  return ScanGeom{m, chunk, ng};
}

template <typename ElemF>
__device__ __forceinline__ Seg chunk_aggregate(const ScanGeom& sg, bool reverse, ElemF elem, Seg* sh) {
  const int q0 = blockIdx.x * sg.chunk;
  const int q1 = min(q0 + sg.chunk, sg.m);
  Seg carry{0.0, 0.0, 0.0};
  const int tile = AMP_THREADS * EPT;
  for (int base = q0; base < q1; base += tile) {
    Seg agg{0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int q = base + threadIdx.x * EPT + e;
      if (q < q1) agg = seg_combine(agg, elem(reverse ? sg.m - 1 - q : q));
    }
    Seg excl, total;
    block_scan_seg(agg, excl, total, sh);
    carry = seg_combine(carry, total);
    __syncthreads();
  }
  return carry;
}

template <typename ElemF, typename EmitF>
__device__ __forceinline__ void chunk_apply(const ScanGeom& sg, bool reverse, Seg carry, ElemF elem, EmitF emit, Seg* sh) {
  const int q0 = blockIdx.x * sg.chunk;
  const int q1 = min(q0 + sg.chunk, sg.m);
  const int tile = AMP_THREADS * EPT;
  for (int base = q0; base < q1; base += tile) {
    Seg local[EPT];
    Seg agg{0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int q = base + threadIdx.x * EPT + e;
      local[e] = q < q1 ? elem(reverse ? sg.m - 1 - q : q) : Seg{0.0, 0.0, 0.0};
      agg = seg_combine(agg, local[e]);
    }
    Seg excl, total;
    block_scan_seg(agg, excl, total, sh);
    Seg run = seg_combine(carry, excl);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int q = base + threadIdx.x * EPT + e;
      run = seg_combine(run, local[e]);
      if (q < q1) emit(reverse ? sg.m - 1 - q : q, run.A, run.B);
    }
    carry = seg_combine(carry, total);
    __syncthreads();
  }
}

// carry-in of this workgroup and the grand total from the per-workgroup aggregates: thread t takes aggregate t and
// the workgroup scans them (ngroups <= MAXG = AMP_THREADS) -- the serial walk over up to 256 aggregates that every
// thread did before was half of the 33 us these launches took at nb = 3e5
static_assert(MAXG <= AMP_THREADS, "one aggregate per thread");
__device__ __forceinline__ void seg_prefix_total(const double* __restrict__ segs, int ngroups, Seg& prefix, Seg& total,
                                                 Seg* sh /*[AMP_WAVES + 1]*/) {
  const int g = threadIdx.x;
  const Seg mine = g < ngroups ? Seg{segs[3 * g], segs[3 * g + 1], segs[3 * g + 2]} : Seg{0.0, 0.0, 0.0};
  Seg excl;
  block_scan_seg(mine, excl, total, sh);
  if (g == (int)blockIdx.x) sh[AMP_WAVES] = excl;
  __syncthreads();
  prefix = sh[AMP_WAVES];
  __syncthreads();
}

struct AmpPtrs {
  const double *rel, *sc, *mult, *delta;
  double *spec, *ahat, *tmp, *segs, *part;
};
__device__ __forceinline__ AmpPtrs amp_ptrs(int nb, const double* geo, double* state) {
  AmpPtrs a;
  a.rel = geo;
  a.sc = geo + nb;
  a.mult = geo + 2 * (size_t)nb;
  a.delta = geo + 3 * (size_t)nb;
  a.spec = state + 16;
  a.ahat = a.spec + nb;
  a.tmp = a.ahat + nb;
  a.segs = state + 16 + 4 * (size_t)nb;  // 3 doubles per scan workgroup (<= (nb - 2) / 1024 + 1, <= MAXG)
  int ng = (nb + 1023) / 1024 + 1;
  a.part = a.segs + 3 * (ng > MAXG ? MAXG : ng);  // 2 doubles per workgroup of the reducing launch (<= nb / 256 + 1, <= MAXG)
  return a;
}

// state scalars: [0] flex [1] asp [2] fluct [3] zm [4] slope [5] S [6] last [7] dS
//                [8] fl_bar [9] Q [10] slope_bar [11] sc_dot [12] flex_bar [13] asp_bar
__device__ __forceinline__ Seg fwd_elem(const AmpPtrs& a, const Hyper& h, const double* xs0, const double* xs1, int j) {
  const double d = a.delta[j];
  const double sq = sqrt(d);
  const double x0 = h.flex * sq * sqrt(d * d / 12.0 + h.asp) * xs0[j];
  const double x1 = h.flex * sq * xs1[j];
  return Seg{x1, d, 0.5 * x1 * d + x0};
}

__global__ void __launch_bounds__(AMP_THREADS) k_fwd_agg(int nb, ScanGeom sg, const double* __restrict__ geo,
                                                         const double* __restrict__ hyp, AmpBatch mb) {
  __shared__ Seg sh_seg[AMP_WAVES + 1];
  const double* __restrict__ lat = mb.lat[blockIdx.y];
  double* __restrict__ state = mb.state[blockIdx.y];
  const AmpPtrs a = amp_ptrs(nb, geo, state);
  const Hyper h = hyper_from_lat(hyp, lat);
  const double *xs0 = lat + 5, *xs1 = lat + 5 + sg.m;
  const Seg c = chunk_aggregate(sg, false, [&](int j) { return fwd_elem(a, h, xs0, xs1, j); }, sh_seg);
  if (threadIdx.x == 0) {
    a.segs[3 * blockIdx.x] = c.A, a.segs[3 * blockIdx.x + 1] = c.D, a.segs[3 * blockIdx.x + 2] = c.B;
    if (blockIdx.x == 0) {
      state[0] = h.flex, state[1] = h.asp, state[2] = h.fluct, state[3] = h.zm, state[4] = h.slope;
      state[5] = 0.0;
      state[14] = 0.0;  // the reduction ticket (amp_ticket)
    }
  }
}

__global__ void __launch_bounds__(AMP_THREADS) k_fwd_apply(int nb, ScanGeom sg, const double* __restrict__ geo,
                                                           const double* __restrict__ hyp, AmpBatch mb) {
  __shared__ Seg sh_seg[AMP_WAVES + 1];
  __shared__ double sh_d[AMP_WAVES];
  const double* __restrict__ lat = mb.lat[blockIdx.y];
  double* __restrict__ state = mb.state[blockIdx.y];
  const AmpPtrs a = amp_ptrs(nb, geo, state);
  const Hyper h = hyper_from_lat(hyp, lat);
  const double *xs0 = lat + 5, *xs1 = lat + 5 + sg.m;
  Seg prefix, total;
  seg_prefix_total(a.segs, sg.ngroups, prefix, total, sh_seg);
  const double last = total.B;
  double part = 0.0;
  auto bin = [&](int b, double smooth) {
    const double s = exp(h.slope * a.rel[b] + smooth - last * a.sc[b]);
    a.spec[b] = s;
    part += a.mult[b] * s;
  };
  if (blockIdx.x == 0 && threadIdx.x < 2) bin(threadIdx.x, 0.0);
  chunk_apply(sg, false, prefix, [&](int j) { return fwd_elem(a, h, xs0, xs1, j); },
              [&](int j, double, double s) { bin(j + 2, s); }, sh_seg);
  const double S = block_sum(part, sh_d);
  if (threadIdx.x == 0 && blockIdx.x == 0) state[6] = last;
  __syncthreads();
  const double v[1] = {S};
  double* const dst[1] = {state + 5};
  amp_store_sums<1>(v, a.part, dst, sh_d, amp_ticket(state));
}

__global__ void k_fwd_final(int nb, const double* __restrict__ hyp, AmpBatch mb) {
  double* __restrict__ state = mb.state[blockIdx.y];
  double* __restrict__ amp = mb.out[blockIdx.y];
  const double S = state[5], V = hyp[10], zm = state[3], fluct = state[2];
  const double* spec = state + 16;
  double* ahat = state + 16 + nb;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
    const double ah = sqrt(spec[b] / S);
    ahat[b] = ah;
    amp[b] = b == 0 ? V * zm : V * fluct * ah;
  }
}

// ---- JVP --------------------------------------------------------------------------------------------------------
struct JvpScal {
  double flex, asp, dflex, dasp;
};
__device__ __forceinline__ Seg jvp_elem(const AmpPtrs& a, const JvpScal& q, const double* xs0, const double* xs1,
                                        const double* dxs0, const double* dxs1, int j) {
  const double d = a.delta[j];
  const double sq = sqrt(d);
  const double w0 = sqrt(d * d / 12.0 + q.asp);
  const double sig0 = q.flex * sq * w0, sig1 = q.flex * sq;
  const double dsig0 = q.dflex * sq * w0 + q.flex * sq * (0.5 / w0) * q.dasp;
  const double dsig1 = q.dflex * sq;
  const double dx0 = dsig0 * xs0[j] + sig0 * dxs0[j];
  const double dx1 = dsig1 * xs1[j] + sig1 * dxs1[j];
  return Seg{dx1, d, 0.5 * dx1 * d + dx0};
}

__global__ void __launch_bounds__(AMP_THREADS) k_jvp_agg(int nb, ScanGeom sg, const double* __restrict__ geo,
                                                         const double* __restrict__ hyp, AmpBatch mb) {
  __shared__ Seg sh_seg[AMP_WAVES + 1];
  const double* __restrict__ lat = mb.lat[blockIdx.y];
  double* __restrict__ state = mb.state[blockIdx.y];
  const double* __restrict__ dlat = mb.in[blockIdx.y];
  const AmpPtrs a = amp_ptrs(nb, geo, state);
  const JvpScal q{state[0], state[1], state[0] * hyp[3] * dlat[1], state[1] * hyp[5] * dlat[0]};
  const Seg c = chunk_aggregate(
      sg, false, [&](int j) { return jvp_elem(a, q, lat + 5, lat + 5 + sg.m, dlat + 5, dlat + 5 + sg.m, j); }, sh_seg);
  if (threadIdx.x == 0) {
    a.segs[3 * blockIdx.x] = c.A, a.segs[3 * blockIdx.x + 1] = c.D, a.segs[3 * blockIdx.x + 2] = c.B;
    if (blockIdx.x == 0) state[7] = 0.0;
  }
}

__global__ void __launch_bounds__(AMP_THREADS) k_jvp_apply(int nb, ScanGeom sg, const double* __restrict__ geo,
                                                           const double* __restrict__ hyp, AmpBatch mb) {
  __shared__ Seg sh_seg[AMP_WAVES + 1];
  __shared__ double sh_d[AMP_WAVES];
  const double* __restrict__ lat = mb.lat[blockIdx.y];
  double* __restrict__ state = mb.state[blockIdx.y];
  const double* __restrict__ dlat = mb.in[blockIdx.y];
  const AmpPtrs a = amp_ptrs(nb, geo, state);
  const JvpScal q{state[0], state[1], state[0] * hyp[3] * dlat[1], state[1] * hyp[5] * dlat[0]};
  const double dslope = hyp[9] * dlat[3];
  Seg prefix, total;
  seg_prefix_total(a.segs, sg.ngroups, prefix, total, sh_seg);
  const double last = total.B;
  double part = 0.0;
  auto bin = [&](int b, double dsm) {
    const double dp = dslope * a.rel[b] + dsm - last * a.sc[b];
    a.tmp[b] = dp;
    part += a.mult[b] * a.spec[b] * dp;
  };
  if (blockIdx.x == 0 && threadIdx.x < 2) bin(threadIdx.x, 0.0);
  chunk_apply(sg, false, prefix,
              [&](int j) { return jvp_elem(a, q, lat + 5, lat + 5 + sg.m, dlat + 5, dlat + 5 + sg.m, j); },
              [&](int j, double, double s) { bin(j + 2, s); }, sh_seg);
  const double dS = block_sum(part, sh_d);
  __syncthreads();
  const double v[1] = {dS};
  double* const dst[1] = {state + 7};
  amp_store_sums<1>(v, a.part, dst, sh_d, amp_ticket(state));
}

__global__ void k_jvp_final(int nb, const double* __restrict__ hyp, AmpBatch mb) {
  const double* __restrict__ state = mb.state[blockIdx.y];
  const double* __restrict__ dlat = mb.in[blockIdx.y];
  double* __restrict__ damp = mb.out[blockIdx.y];
  const double S = state[5], dS = state[7], V = hyp[10], fluct = state[2], zm = state[3];
  const double dfluct = fluct * hyp[1] * dlat[2], dzm = zm * hyp[7] * dlat[4];
  const double* ahat = state + 16 + nb;
  const double* dp = state + 16 + 2 * (size_t)nb;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
    const double dah = 0.5 * ahat[b] * (dp[b] - dS / S);
    damp[b] = b == 0 ? V * dzm : V * (dfluct * ahat[b] + fluct * dah);
  }
}

// ---- VJP --------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(AMP_THREADS) k_vjp_red1(int nb, const double* __restrict__ hyp, AmpBatch mb) {
  __shared__ double sh_d[AMP_WAVES];
  double* __restrict__ state = mb.state[blockIdx.y];
  const double* __restrict__ abar = mb.in[blockIdx.y];
  const double V = hyp[10], fluct = state[2];
  const double* ahat = state + 16 + nb;
  double p_fl = 0.0, p_q = 0.0;
  for (int b = 1 + blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
    p_fl += abar[b] * ahat[b];
    p_q += 0.5 * ahat[b] * (V * fluct * abar[b]);
  }
  const double s1 = block_sum(p_fl, sh_d);
  const double s2 = block_sum(p_q, sh_d);
  __syncthreads();
  const double v[2] = {V * s1, s2};
  double* const dst[2] = {state + 8, state + 9};
  amp_store_sums<2>(v, amp_ptrs(nb, nullptr, state).part, dst, sh_d, amp_ticket(state));
}

__global__ void __launch_bounds__(AMP_THREADS) k_vjp_red2(int nb, const double* __restrict__ geo, const double* __restrict__ hyp,
                                                          AmpBatch mb) {
  __shared__ double sh_d[AMP_WAVES];
  double* __restrict__ state = mb.state[blockIdx.y];
  const double* __restrict__ abar = mb.in[blockIdx.y];
  const AmpPtrs a = amp_ptrs(nb, geo, state);
  const double V = hyp[10], fluct = state[2], S = state[5], Q = state[9];
  double p_sl = 0.0, p_sc = 0.0;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
    const double q = b == 0 ? 0.0 : 0.5 * a.ahat[b] * (V * fluct * abar[b]);
    const double pb = q - (Q / S) * a.mult[b] * a.spec[b];
    a.tmp[b] = pb;
    p_sl += pb * a.rel[b];
    p_sc += pb * a.sc[b];
  }
  const double s1 = block_sum(p_sl, sh_d);
  const double s2 = block_sum(p_sc, sh_d);
  __syncthreads();
  const double v[2] = {s1, s2};
  double* const dst[2] = {state + 10, state + 11};
  amp_store_sums<2>(v, a.part, dst, sh_d, amp_ticket(state));
}

__device__ __forceinline__ Seg vjp_elem(const AmpPtrs& a, int nb, int m, double sc_dot, int j) {
  double y = a.tmp[j + 2];
  if (j + 2 == nb - 1) y -= sc_dot;
  const double dj = a.delta[j];
  const double dn = j + 1 < m ? a.delta[j + 1] : 0.0;
  return Seg{y, 0.5 * (dj + dn), 0.5 * y * dj};
}

__global__ void __launch_bounds__(AMP_THREADS) k_vjp_agg(int nb, ScanGeom sg, const double* __restrict__ geo, AmpBatch mb) {
  __shared__ Seg sh_seg[AMP_WAVES + 1];
  double* __restrict__ state = mb.state[blockIdx.y];
  const AmpPtrs a = amp_ptrs(nb, geo, state);
  const double sc_dot = state[11];
  const Seg c = chunk_aggregate(sg, true, [&](int j) { return vjp_elem(a, nb, sg.m, sc_dot, j); }, sh_seg);
  if (threadIdx.x == 0) a.segs[3 * blockIdx.x] = c.A, a.segs[3 * blockIdx.x + 1] = c.D, a.segs[3 * blockIdx.x + 2] = c.B;
}

__global__ void __launch_bounds__(AMP_THREADS) k_vjp_apply(int nb, ScanGeom sg, const double* __restrict__ geo, AmpBatch mb) {
  __shared__ Seg sh_seg[AMP_WAVES + 1];
  __shared__ double sh_d[AMP_WAVES];
  const double* __restrict__ lat = mb.lat[blockIdx.y];
  double* __restrict__ state = mb.state[blockIdx.y];
  double* __restrict__ latbar = mb.out[blockIdx.y];
  const AmpPtrs a = amp_ptrs(nb, geo, state);
  const double flex = state[0], asp = state[1], sc_dot = state[11];
  const double *xs0 = lat + 5, *xs1 = lat + 5 + sg.m;
  double *sbar0 = latbar + 5, *sbar1 = latbar + 5 + sg.m;
  Seg prefix, total;
  seg_prefix_total(a.segs, sg.ngroups, prefix, total, sh_seg);
  double p_flex = 0.0, p_asp = 0.0;
  chunk_apply(sg, true, prefix, [&](int j) { return vjp_elem(a, nb, sg.m, sc_dot, j); },
              [&](int j, double t, double g1) {
                const double d = a.delta[j];
                const double sq = sqrt(d);
                const double w0 = sqrt(d * d / 12.0 + asp);
                sbar0[j] = t * flex * sq * w0;
                sbar1[j] = g1 * flex * sq;
                const double s0b = t * xs0[j], s1b = g1 * xs1[j];
                p_flex += s0b * sq * w0 + s1b * sq;
                p_asp += s0b * flex * sq * 0.5 / w0;
              },
              sh_seg);
  const double s1 = block_sum(p_flex, sh_d);
  const double s2 = block_sum(p_asp, sh_d);
  __syncthreads();
  const double v[2] = {s1, s2};
  double* const dst[2] = {state + 12, state + 13};
  amp_store_sums<2>(v, a.part, dst, sh_d, amp_ticket(state));
}

__global__ void k_vjp_final(const double* __restrict__ hyp, AmpBatch mb) {
  const double* __restrict__ state = mb.state[blockIdx.y];
  const double* __restrict__ abar = mb.in[blockIdx.y];
  double* __restrict__ latbar = mb.out[blockIdx.y];
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double flex = state[0], asp = state[1], fluct = state[2], zm = state[3], V = hyp[10];
    latbar[0] = state[13] * asp * hyp[5];
    latbar[1] = state[12] * flex * hyp[3];
    latbar[2] = state[8] * fluct * hyp[1];
    latbar[3] = state[10] * hyp[9];
    latbar[4] = V * abar[0] * zm * hyp[7];
  }
}

}  // namespace

static inline int amp_grid(int nb) {
  int g = (nb + AMP_THREADS - 1) / AMP_THREADS;
  return g < 1 ? 1 : (g > MAXG ? MAXG : g);
}

static int amp_batch(AmpBatch& mb, int count, const double* const* lat, double* const* state, const double* const* in,
                     double* const* out, const char* who) {
  if (count < 1 || count > NK_MAX_BATCH || !lat || !state || !out) return nk_set_error(NK_ERR_INVALID, who);
  for (int m = 0; m < NK_MAX_BATCH; ++m) {
    const int k = m < count ? m : 0;
    if (!lat[k] || !state[k] || !out[k] || (in && !in[k])) return nk_set_error(NK_ERR_INVALID, who);
    mb.lat[m] = lat[k], mb.state[m] = state[k], mb.in[m] = in ? in[k] : nullptr, mb.out[m] = out[k];
  }
  for (int a = 0; a < count; ++a)  // every member scribbles on its state and output: they must not be shared
    for (int b = a + 1; b < count; ++b)
      if (state[a] == state[b] || out[a] == out[b]) return nk_set_error(NK_ERR_INVALID, who);
  return NK_OK;
}

extern "C" int nk_amp_forward_batch(int nb, const double* geo, const double* hyp, int count, const double* const* lat,
                                    double* const* state, double* const* amp, void* stream) {
  AmpBatch mb;
  if (nb < 3 || !geo || !hyp) return nk_set_error(NK_ERR_INVALID, "nk_amp_forward: bad argument");
  int rc = amp_batch(mb, count, lat, state, nullptr, amp, "nk_amp_forward: bad argument");
  if (rc != NK_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const ScanGeom sg = make_scan_geom(nb - 2);
  hipLaunchKernelGGL(k_fwd_agg, dim3(sg.ngroups, count), dim3(AMP_THREADS), 0, st, nb, sg, geo, hyp, mb);
  hipLaunchKernelGGL(k_fwd_apply, dim3(sg.ngroups, count), dim3(AMP_THREADS), 0, st, nb, sg, geo, hyp, mb);
  hipLaunchKernelGGL(k_fwd_final, dim3(amp_grid(nb), count), dim3(AMP_THREADS), 0, st, nb, hyp, mb);
  return nk_check_launch("nk_amp_forward");
}

extern "C" int nk_amp_jvp_batch(int nb, const double* geo, const double* hyp, int count, const double* const* lat,
                                double* const* state, const double* const* dlat, double* const* damp, void* stream) {
  AmpBatch mb;
  if (nb < 3 || !geo || !hyp || !dlat) return nk_set_error(NK_ERR_INVALID, "nk_amp_jvp: bad argument");
  int rc = amp_batch(mb, count, lat, state, dlat, damp, "nk_amp_jvp: bad argument");
  if (rc != NK_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const ScanGeom sg = make_scan_geom(nb - 2);
  hipLaunchKernelGGL(k_jvp_agg, dim3(sg.ngroups, count), dim3(AMP_THREADS), 0, st, nb, sg, geo, hyp, mb);
  hipLaunchKernelGGL(k_jvp_apply, dim3(sg.ngroups, count), dim3(AMP_THREADS), 0, st, nb, sg, geo, hyp, mb);
  hipLaunchKernelGGL(k_jvp_final, dim3(amp_grid(nb), count), dim3(AMP_THREADS), 0, st, nb, hyp, mb);
  return nk_check_launch("nk_amp_jvp");
}

extern "C" int nk_amp_vjp_batch(int nb, const double* geo, const double* hyp, int count, const double* const* lat,
                                double* const* state, const double* const* abar, double* const* latbar, void* stream) {
  AmpBatch mb;
  if (nb < 3 || !geo || !hyp || !abar) return nk_set_error(NK_ERR_INVALID, "nk_amp_vjp: bad argument");
  int rc = amp_batch(mb, count, lat, state, abar, latbar, "nk_amp_vjp: bad argument");
  if (rc != NK_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  const ScanGeom sg = make_scan_geom(nb - 2);
  hipLaunchKernelGGL(k_vjp_red1, dim3(amp_grid(nb), count), dim3(AMP_THREADS), 0, st, nb, hyp, mb);
  hipLaunchKernelGGL(k_vjp_red2, dim3(amp_grid(nb), count), dim3(AMP_THREADS), 0, st, nb, geo, hyp, mb);
  hipLaunchKernelGGL(k_vjp_agg, dim3(sg.ngroups, count), dim3(AMP_THREADS), 0, st, nb, sg, geo, mb);
  hipLaunchKernelGGL(k_vjp_apply, dim3(sg.ngroups, count), dim3(AMP_THREADS), 0, st, nb, sg, geo, mb);
  hipLaunchKernelGGL(k_vjp_final, dim3(1, count), dim3(64), 0, st, hyp, mb);
  return nk_check_launch("nk_amp_vjp");
}

extern "C" int nk_amp_forward(int nb, const double* geo, const double* hyp, const double* lat, double* state, double* amp,
                              void* stream) {
  if (!lat || !state || !amp) return nk_set_error(NK_ERR_INVALID, "nk_amp_forward: bad argument");
  return nk_amp_forward_batch(nb, geo, hyp, 1, &lat, &state, &amp, stream);
}

extern "C" int nk_amp_jvp(int nb, const double* geo, const double* hyp, const double* lat, double* state, const double* dlat,
                          double* damp, void* stream) {
  if (!lat || !state || !dlat || !damp) return nk_set_error(NK_ERR_INVALID, "nk_amp_jvp: bad argument");
  return nk_amp_jvp_batch(nb, geo, hyp, 1, &lat, &state, &dlat, &damp, stream);
}

extern "C" int nk_amp_vjp(int nb, const double* geo, const double* hyp, const double* lat, double* state, const double* abar,
                          double* latbar, void* stream) {
  if (!lat || !state || !abar || !latbar) return nk_set_error(NK_ERR_INVALID, "nk_amp_vjp: bad argument");
  return nk_amp_vjp_batch(nb, geo, hyp, 1, &lat, &state, &abar, &latbar, stream);
}
