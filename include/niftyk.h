/* niftyk.h -- C ABI of libniftyk, the MI355X (gfx950) kernel library behind the nifty.cl MGVI/geoVI
 * hot path.  This is the drop-in boundary: every entry point replaces one native-dispatch call site
 * of the reference (file:line relative to the NIFTy source tree) and takes only plain pointers and
 * sizes.  All pointers are BORROWED device pointers (hipMalloc'd / torch tensor .data_ptr()),
 * contiguous C order; the caller owns memory and lifetime.  `stream` is a hipStream_t passed as
 * void* (0 = default stream).  Every function returns 0 on success or a negative nk_status code and
 * never throws; nk_last_error() returns a thread-local message for the last failure.
 * Plans are immutable after creation; execution is ordered by the stream.  The transforms allocate
 * nothing after plan creation (workspace is passed in; size from nk_plan_workspace_bytes).  The
 * deterministic reductions (nk_vdot, nk_sum, nk_stats, nk_axpby_sqnorm, nk_cg_*, their *_batch forms)
 * keep 4 MiB of block partials per (device, stream) pair, allocated with hipMalloc the FIRST time a
 * reduction runs on that stream and kept for the life of the process: call any of them once on a
 * stream before capturing it into a graph or timing it.
 */
#ifndef NIFTYK_H
#define NIFTYK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NK_OK = 0,
  NK_ERR_INVALID = -1,      /* bad argument (ValueError on the Python side) */
  NK_ERR_UNSUPPORTED = -2,  /* shape / dtype not supported yet (NotImplementedError) */
  NK_ERR_RUNTIME = -3,      /* HIP runtime failure (RuntimeError) */
  NK_ERR_NOMEM = -4
} nk_status;

typedef enum { NK_F32 = 0, NK_F64 = 1 } nk_dtype;

/* hartley conventions, nifty/config.py:42-78 */
#define NK_HARTLEY_NON_CANONICAL 0 /* Re F + Im F : NIFTy default ("non_canonical_hartley") */
#define NK_HARTLEY_CANONICAL 1     /* Re F - Im F */

/* fused prologue (input side of a transform) */
enum { NK_PRO_PLAIN = 0, NK_PRO_AMP = 1, NK_PRO_AMP_JVP = 2, NK_PRO_MUL = 3 };
/* fused epilogue (output side of a transform) */
enum { NK_EPI_AFFINE = 0, NK_EPI_MUL = 1, NK_EPI_VJP = 2, NK_EPI_LIKELIHOOD = 3, NK_EPI_NONLIN = 4 };
enum { NK_LH_GAUSS = 0, NK_LH_POISSON = 1 };
enum { NK_NL_ID = 0, NK_NL_EXP = 1, NK_NL_SIGMOID = 2 };

/* Fusion descriptor for nk_hartley_fused.  T = the plan's real dtype.
 *   prologue  x(i) fed into the transform:
 *     PLAIN    in[i]
 *     AMP      amp[pidx[i]] * in[i]                              (PowerDistributor gather x xi,
 *                                                                  distributors.py:114-119 + diagonal multiply)
 *     AMP_JVP  amp[pidx[i]] * in[i] + damp[pidx[i]] * in2[i]     (product rule, operator.py:579-582)
 *              (with `afield` set the amp gather becomes a stream: afield[i] * in[i]; `dampT` = da table in T)
 *     MUL      in[i] * in2[i]
 *   epilogue  for transform output v at index o, t = v * scale:
 *     AFFINE      out[o] = t + offset                            (scalar_dvol factor + Adder, adder.py:47-52)
 *     MUL         out[o] = t * mul_scalar * mul[o]               (DiagonalOperator / ScalingOperator)
 *     VJP         out[o] = amp[pidx[o]] * t (+ addend_scale * addend[o]) (+ out[o] if accumulate);
 *                 abar[pidx[o]] += xi[o] * t                     (PowerDistributor adjoint, distributors.py:106-112;
 *                                                                 accumulate = sum over samples, sample_list.py:212-270)
 *                 with `value` and `addend` set and nk_plan_octant_vjp(plan) != 0 (+ afield):
 *                 *value += sum_o addend[o] * out[o]             (the CG curvature d.(A d) of conjugate_gradient.py:88
 *                                                                 for addend = d, taken while out is written)
 *     LIKELIHOOD  s = t + offset, g = nonlin(s); Gaussian / Poisson energy -> *value (atomic),
 *                 out[o] = dE/ds, out2[o] = g'(s)^2 * M_d (Fisher metric weight in s-space)
 *                                                                (energy_operators.py:517-640)
 *     NONLIN      out[o] = nonlin(t + offset), out2[o] = nonlin'(.)   (pointwise.py:134-159)
 */
typedef struct nk_fuse {
  int pro;
  const void* in;
  const void* in2;
  const int32_t* pidx;
  const double* amp;
  const double* damp;
  int epi;
  void* out;
  double scale;
  double offset;
  const void* mul;
  double mul_scalar;
  const void* xi;
  const void* addend;
  double addend_scale;
  int accumulate;
  double* abar;
  int lh_kind, nonlin;
  const void* data;
  const void* icov;
  double icov_scalar;
  void* out2;
  double* value;
  const void* afield;   /* optional T* field a[pidx[i]] (materialised once per linearisation): replaces the amp gather */
  const void* dampT;    /* optional da table in the field dtype T (AMP_JVP) instead of the double table `damp` */
  int abar_copies;      /* VJP: 0/1 = one device-scope accumulator; 8 = one private accumulator per XCD */
  int64_t abar_stride;  /* elements between the private accumulators (fold them with nk_fold_copies) */
  const void* dafield;  /* optional T* field da[pidx[i]] (nk_octant_expand of the da table): AMP_JVP without any gather */
  double* w8;           /* VJP, optional: instead of atomics store the per-octant-point sums xi*t (all sign-flip images
                           merged) to w8[batch][A/2+1][M/2+1][NL/2+1]; reduce with nk_octant_scatter.  Only honoured
                           when nk_plan_octant_vjp(plan) != 0 */
  int field_octant;     /* != 0: afield / dafield are OCTANT arrays [A/2+1][M/2+1][NL/2+1] (nk_octant_expand with
                           compact != 0), broadcast over the batch: the sign-flip images of a coefficient share the
                           value, so the prologue of the first pass and the VJP epilogue read 1/8 of the bytes.  Only
                           valid when nk_plan_octant_vjp(plan) != 0 and with prologue PLAIN / MUL / AMP(afield) /
                           AMP_JVP(afield + dafield) and, for the VJP epilogue, afield set (NK_ERR_INVALID otherwise) */
  int value_slots;      /* leave 0.  Set by the library itself when it gives every workgroup of the final pass its own
                           accumulator in the workspace (folded into *value in a fixed order after the pass) */
  const int32_t* pidx_octant; /* optional, nk_hartley_sandwich with field_octant and the AMP_JVP prologue: the bin index of
                           the OCTANT points [A/2+1][M/2+1][NL/2+1]; with it (and dampT) the prologue gathers da from the
                           table itself and `dafield` is not needed -- no per-application expansion of da[pidx] */
  const void* cg_r;     /* optional, nk_hartley_sandwich with field_octant and the AMP_JVP prologue (+ dafield, cg_scal):
                           the pending search-direction update of a conjugate-gradient iteration
                           (conjugate_gradient.py:124), done on the way: in[i] <- max(0, cg_scal[2] / cg_scal[0]) * in[i]
                           + cg_r[i] is WRITTEN BACK to `in` (not const here) and the new value enters the transform */
  const double* cg_scal; /* the device scalars of nk_cg_update; roll them afterwards with
                           nk_cg_direction(0, NULL, NULL, dtype, scal, 1, stream) */
  double* w8max;        /* optional, with w8 on a 3-D plan: device scalar <- an upper bound of max |w8[x]| of this launch, tight to
                           fp32 rounding (fixed-order maximum over the wavefronts; NK_ERR_UNSUPPORTED on 1-D / 2-D plans).  nk_octant_scatter_k2 takes it as the scale of its fixed-point accumulation, which
                           makes the bin sums independent of the order of the additions (bit-reproducible) */
  /* ---- nk_hartley_sandwich only: slab pipelining against an exchange on another stream ("chunked to overlap
   *      with the last adjoint-FFT pass").  pipe_chunks = C (even, divides the first axis; 0 / 1 = off) cuts the first
   *      axis of `in` / `out` into C equal slab chunks.  The contiguous FIRST pass runs in C/2 stages: stage j touches the
   *      rows of `in` in the chunks <= j and >= C-1-j only and first waits for pipe_wait[j] (a hipEvent_t recorded by the
   *      caller once those chunks of `in` have landed -- an all-gather in chunk order j, C-1-j).  The FINAL pass runs in C/2
   *      stages as well: after stage j the chunks j and C-1-j of `out` are final and pipe_record[j] (an existing
   *      hipEvent_t) is recorded on `stream`, so that a reduce-scatter of those chunks can start while the later stages
   *      still compute.  Either array may be NULL.  Results are bit-identical to the unstaged call.  3-D plans with the
   *      octant prologue classes and batch 1 only (NK_ERR_UNSUPPORTED otherwise). */
  int pipe_chunks;
  void* const* pipe_wait;
  void* const* pipe_record;
  double* wfull;        /* optional, VJP epilogue on plans WITHOUT the octant pipeline (nk_plan_octant_vjp == 0: mixed-radix
                           grids, short axes): full-grid fp64 array [batch][shape...] that receives xi[o] * t[o] per point
                           (mirror partners of one kernel work item: their sum at the first, 0 at the others) INSTEAD of
                           the atomics into abar -- the caller reduces it in a fixed order (nk_csr_rowsum over the
                           bin-sorted permutation): bit-reproducible spectrum gradients on every grid */
  int io32;             /* != 0, fp64 plans with nk_plan_octant_vjp(plan) != 0 only, nk_hartley_fused with the AMP prologue
                           (field_octant, afield = fp64 octant field) and the LIKELIHOOD epilogue: `in`, `data` (Gaussian),
                           `icov`, `out` and `out2` are FLOAT arrays while the transform -- amplitude product, all passes, the
                           work array, residual and energy -- runs in fp64.  This is the value / gradient forward transform of
                           a model with fp32 fields: the reference promotes fp32 excitations to fp64 at their product with the
                           fp64 amplitude and transforms in fp64 (library/correlated_fields.py:755-764), and an fp32 forward
                           transform leaves a coherent 6e-8 gain error on the signal that the residual N^-1 (s - d) amplifies
                           (DESIGN 6).  NK_ERR_INVALID with any other prologue / epilogue or on an fp32 plan */
  const void* carry1;
  const void* carry2;   /* VJP epilogue, optional: T* partial sums of OTHER samples that join this one's output.  The sample's
                           own contribution g[o] = amp * t (+ addend_scale * addend[o]) is rounded to T and then added innermost
                           first: out[o] = [out[o] +] ([carry2[o] +] ([carry1[o] +] g[o])) (the bracketed terms when
                           accumulate / the pointers are set) -- plain additions of stored values, so a sum over samples built
                           in the order of the reference's allreduce_sum (nifty/cl/utilities.py:349-414: neighbours first,
                           then pairs of pairs, ...) has the same bits whether its partial sums were formed in these epilogues
                           on one GPU or added across ranks.  nk_hartley_sandwich_pair: B may take A's `out` as carry1 */
} nk_fuse;

typedef struct nk_plan nk_plan;

const char* nk_last_error(void);
int nk_version(void);

/* ---- transforms: replace ducc_dispatch.hartley / fftn / ifftn (nifty/cl/ducc_dispatch.py:116-142),
 *      called from HartleyOperator._apply_cartesian and FFTOperator.apply
 *      (nifty/cl/operators/harmonic_operators.py:77-94,144-161).  All `ndim` trailing axes of a
 *      [batch, shape...] array are transformed; every axis length must factor into {2, 3, 5, 7} (the real transforms
 *      need an even last axis), at most 3 transformed axes, a line must fit one LDS tile -- NK_ERR_UNSUPPORTED otherwise
 *      (the Python array seam serves such shapes through the chirp-z composition of nifty_amd/backend.py). */
int nk_plan_create(nk_plan** plan, int ndim, const int64_t* shape, int dtype, int64_t batch);
int nk_plan_destroy(nk_plan* plan);
size_t nk_plan_workspace_bytes(const nk_plan* plan);
/* out = scale * Hartley(in); in == out is allowed */
int nk_hartley(const nk_plan* plan, const void* in, void* out, double scale, int convention, void* workspace,
               void* stream);
int nk_hartley_fused(const nk_plan* plan, const nk_fuse* fuse, int convention, void* workspace, void* stream);
/* 1 if nk_hartley_sandwich on this plan accepts nk_fuse.pipe_chunks == chunks (slab pipelining, see nk_fuse) */
int nk_plan_pipe_ok(const nk_plan* plan, int chunks);
/* Hartley SANDWICH  H D H  of a metric application  J^T M J  (LikelihoodEnergyOperator.get_metric_at,
 * operators/energy_operators.py:146-152: SandwichOperator.make(J, M) with J ending / J^T starting in HartleyOperator,
 * harmonic_operators.py:144-161), as ONE call:
 *     t1 = scale_first * Hartley(PRO(in));   x[o] = mul_scalar * (mul ? mul[o] : 1) * t1[o];
 *     t2 = fuse->scale * Hartley(x);         EPI(t2)
 * with the prologues / epilogues of nk_fuse (`mul` / `mul_scalar` are the diagonal between the transforms, so the MUL
 * epilogue is not available).  Five passes over the array instead of six and no position-space intermediate: the last
 * pass of the first transform and the first pass of the second one are one kernel (nifty_amd/csrc/nk_fft3.h).
 * Available when nk_plan_sandwich(plan) != 0 (>= 2 axes, every axis length a power of two in 64 .. 4096). */
int nk_plan_sandwich(const nk_plan* plan);
int nk_hartley_sandwich(const nk_plan* plan, const nk_fuse* fuse, double scale_first, int convention, void* workspace,
                        void* stream);
/* Two sandwiches that ACCUMULATE INTO THE SAME `out` (two samples of a KL metric, SampledKLEnergyClass.apply_metric sums
 * them, kl_energies.py:344-350) in one call: the first four passes of A and of B run one after the other on their own
 * workspaces, the two final passes share ONE launch in which every workgroup finishes its lines for A and then for B --
 * B's read-modify-write of `out` meets A's lines in L2 instead of HBM.  Same arithmetic, same bits as
 * nk_hartley_sandwich(A) followed by nk_hartley_sandwich(B).  Requirements: 3-D plan, both epilogues VJP with octant
 * amplitude fields (field_octant, afield), fuse_b->accumulate != 0, fuse_a->out == fuse_b->out, separate w8 (and w8max, value)
 * areas, the same mul_scalar, no slab pipelining, workspace_a != workspace_b (each nk_plan_workspace_bytes).
 * (Or fuse_b->carry1 == fuse_a->out with a different fuse_b->out: B joins A's fresh lines as its innermost partial sum.  A
 * build with -DNK_PAIR_HAND_BUILD=1 run with NK_PAIR_HAND=1 hands A's lines to B inside the workgroup and then leaves A's
 * `out` unwritten in this form; the product build always stores it.) */
int nk_hartley_sandwich_pair(const nk_plan* plan, const nk_fuse* fuse_a, const nk_fuse* fuse_b, double scale_first,
                             int convention, void* workspace_a, void* workspace_b, void* stream);
/* complex-to-complex: in/out interleaved (re,im) of the plan dtype; inverse != 0 uses exp(+i..);
 * result is multiplied by `scale` (pass 1/N for numpy-style ifftn).  in == out allowed. */
int nk_fftn(const nk_plan* plan, const void* in, void* out, int inverse, double scale, void* workspace,
            void* stream);

/* live profiling for bench.py: when enabled every transform pass kernel launch is bracketed by HIP events on
 * its launch stream; nk_profile_collect synchronises and returns summed milliseconds and launch counts in
 * ms[250] / count[250], index = kernel*25 + prologue*5 + epilogue (kernel: 0 pass1d, 1 passA, 2 passB,
 * 3 passC, 4 passD; sandwich: 5 contiguous first pass, 6 in-place middle-axis pass, 7 fused first-axis pass; its
 * final pass counts as 3; kernel 8 = nk_csr_rowsum with "prologue" = lanes class 0..3 (1 / 4 / 16 / 64 lanes per row) and
 * "epilogue" = 0 weighted / 1 unweighted) and resets the record. */
int nk_profile_enable(int on);
int nk_profile_collect(double* ms, int64_t* count);

/* ---- reductions: replace ducc_dispatch.vdot / AnyArray.vdot / norm / sum
 *      (nifty/cl/ducc_dispatch.py:145-150, any_array.py:544-552).  fp64 accumulation for both dtypes;
 *      result is written to a DEVICE double (no host sync).  `result` must be zeroed by the caller
 *      unless accumulate == 0 (then the kernel chain zeroes it first). */
int nk_vdot(int64_t n, const void* a, const void* b, int dtype, double* result, int accumulate, void* stream);

/* ---- rank-count-independent reductions of SHARDED vectors (no reference counterpart as a function: the reference keeps
 *      every vector whole on every task; its sums over samples are task-count independent, nifty/cl/utilities.py:349-414,
 *      and a drop-in that shards the CG vectors has to keep that property for their dot products).
 *      Every reduction of this library (nk_vdot, nk_sum, nk_stats, nk_axpby_sqnorm, nk_cg_curv, nk_cg_update, ...) cuts an
 *      array whose length qualifies -- nk_red_unit(n, dtype) != 0: 64 units of whole 256-vector rows -- into 64 contiguous
 *      UNITS, reduces each unit with a sub-grid that depends on the unit length only and adds the 64 unit sums in order.
 *      A rank that holds a shard made of whole units announces it with nk_red_layout: unit length, the number of units it
 *      holds, the number of units of the full array (64) and where its units sit -- the shard is a sequence of segments
 *      of seg_units units, segment j starting at unit j * seg_stride + seg_off of the full array (the [chunk][rank][m]
 *      ownership of the sharded CG) -- and a device array units_out[nred][k_global].  Reduction launches of THIS host
 *      thread over exactly k_local * unit_elems elements then write the unit sums (zeros for foreign units) to units_out
 *      INSTEAD of accumulating into their result; after a sum all-reduce of units_out over the ranks (exact: one non-zero
 *      term per unit) nk_red_finish adds the units in order into result[0..nred) -- bit-identical to the single-process
 *      reduction of the whole array.  nk_red_layout(0, ...) clears the announcement. */
int64_t nk_red_unit(int64_t n, int dtype);
int nk_red_layout(int64_t unit_elems, int k_local, int k_global, int seg_units, int seg_stride, int seg_off,
                  double* units_out);
int nk_red_finish(const double* units, int k_global, int nred, double* result, int accumulate, void* stream);
int nk_sum(int64_t n, const void* a, int dtype, double* result, int accumulate, void* stream);
/* result3 = {sum, sum of squares, count} of a over the entries that are neither NaN nor exactly 0 (count = the ignored
 * ones): the per-key statistics of extra.minisanity (extra.py:640-654) */
int nk_stats(int64_t n, const void* a, int dtype, double* result3, void* stream);

/* ---- element-wise vector algebra used by Field/MultiField arithmetic and CG
 *      (field.py:755-763, conjugate_gradient.py:100-124, quadratic_energy.py:31-39) */
enum { NK_OP_ADD = 0, NK_OP_SUB = 1, NK_OP_MUL = 2, NK_OP_DIV = 3 };
/* out = a (op) b ;  b == NULL -> out = a (op) bscalar ; a == NULL -> out = ascalar (op) b */
int nk_binary(int op, int64_t n, const void* a, double ascalar, const void* b, double bscalar, void* out,
              int dtype, void* stream);
/* complex element-wise products / quotients on interleaved (re, im) arrays of `dtype` (n COMPLEX elements):
 *   out = a * op(b)  or  a / op(b),  op = conj when conj_b != 0;  operand kind 0 = complex array, 1 = REAL array of n
 *   elements, 2 = the complex scalar (sr, si) (its pointer is ignored).  DiagonalOperator with a complex diagonal in all
 *   four modes (diagonal_operator.py:194-214: TIMES d, ADJOINT conj d, INVERSE 1/d, ADJOINT_INVERSE 1/conj d), complex
 *   weights of Gaussian residuals (energy_operators.py:517-595).
 * nk_cplx_pointwise: fn 0 exp, 1 log, 2 sqrt (principal branch), 3 reciprocal, 4 conjugate -- out complex; 5 |z| -- out a
 *   REAL array of n elements (pointwise.py:134-159 on complex fields). */
int nk_cplx_muldiv(int64_t n, const void* a, int akind, double asr, double asi, const void* b, int bkind, double bsr,
                   double bsi, int conj_b, int divide, void* out, int dtype, void* stream);
int nk_cplx_pointwise(int fn, int64_t n, const void* x, void* out, int dtype, void* stream);
/* out = alpha * x + beta * y   (y may be NULL) */
int nk_axpby(int64_t n, double alpha, const void* x, double beta, const void* y, void* out, int dtype,
             void* stream);
/* out = alpha * x + beta * y and, in the same pass, *result (+)= sum(out[i]^2) of the stored values (fp64 accumulation,
 * fixed order): a KL sample position p +- r together with its prior term 1/2 |x|^2 (kl_energies.py:318-321,
 * energy_operators.py:890-931) */
int nk_axpby_sqnorm(int64_t n, double alpha, const void* x, double beta, const void* y, void* out, int dtype, double* result,
                    int accumulate, void* stream);
/* pointwise nonlinearity with optional derivative output (pointwise.py:134-159):
 * fn: 0 exp, 1 log, 2 sqrt, 3 tanh, 4 sigmoid(0.5+0.5tanh), 5 reciprocal, 6 power(p), 7 abs, 8 log1p, 9 expm1,
 *     10 arctan, 11 sin, 12 cos, 13 tan, 14 sinc (sin(pi x)/(pi x)), 15 log10, 16 sinh, 17 cosh, 18 sign (derivative 0, NaN at
 *     0), 19 softplus (identity above 33, zero below -33), 20 exponentiate (param ** x), 21 unitstep (1 for x >= 0) */
int nk_pointwise(int fn, double param, int64_t n, const void* x, void* fx, void* dfx, int dtype, void* stream);
/* fx = min(max(x, lo), hi), dfx = 0 where fx sits on a bound, else 1 (pointwise.py:76-88); -INFINITY / INFINITY: no bound */
int nk_clip(double lo, double hi, int64_t n, const void* x, void* fx, void* dfx, int dtype, void* stream);
/* gather / scatter-add by bin index (DOFDistributor, distributors.py:106-127): table/in/out are of `dtype`;
 * the scatter accumulates into DOUBLE bins (np.bincount semantics, utilities.py:222-246), caller zeroes them.
 * nk_scatter_add uses fp64 atomics: with colliding indices the sums depend on the order of the atomics in the last bit.
 * For a STATIC index map use nk_csr_rowsum over the bin-sorted permutation instead (fixed order, no atomics): that is
 * what nifty_amd's DOFDistributor / PowerDistributor / ContractionOperator do. */
int nk_gather(int64_t n, const void* table, const int32_t* pidx, void* out, int dtype, void* stream);
int nk_scatter_add(int64_t n, const void* in, const int32_t* pidx, int64_t nbins, void* bins, int dtype,
                   void* stream);

/* ---- sparse line-of-sight response: replaces scipy.sparse matvec / rmatvec of LOSResponse.apply
 *      (library/los_response.py:244-253).  CSR: rowptr[nrows+1] (int64), col (int32 pixel index), wgt (float32, as the
 *      reference stores them, :196); x / y in the field dtype, fp64 accumulation.
 * nk_csr_rowsum: y[i] = sum_j (wgt ? wgt[j] : 1) * x[col[j]] over rowptr[i] <= j < rowptr[i+1], `lanes` (1, 4, 16 or 64)
 *            lanes per row: lane l adds the entries l, l + lanes, ... in ascending order, a fixed shuffle tree joins the
 *            lanes -- the summation order depends on the matrix only (bit-reproducible, no atomics).  Serves TIMES
 *            (lanes 64: thousands of pixels per line), ADJOINT_TIMES through the TRANSPOSED matrix the caller builds once
 *            (lanes 1: a few lines per pixel), and every scatter-add of a static index map (rows = bins, col = the
 *            bin-sorted permutation of the source points, wgt = NULL; np.bincount / _special_add_at, utilities.py:222-246).
 * nk_spmv  : nk_csr_rowsum with lanes = 64
 * nk_spmv_t: x64[col[j]] += wgt[j] * y[i] on a caller-zeroed fp64 array (cast to the field dtype afterwards) by fp64
 *            atomics -- for callers that do not hold the transpose; order-dependent in the last bit */
int nk_csr_rowsum(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, const void* x, void* y,
                  int dtype, int lanes, void* stream);
int nk_spmv(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, const void* x, void* y, int dtype,
            void* stream);
int nk_spmv_t(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, const void* y, double* x64,
              int dtype, void* stream);

/* nk_tiled_rowsum: y = R x for a response whose COLUMNS are the points of a grid [outer][ny][nx] and whose rows are long
 * (TIMES of LOSResponse: thousands of pixels per line of sight), with the matrix re-ordered once at set-up by TILES of
 * th x tw grid points (nifty_amd.los_response.tiled_plan builds these arrays on the host):
 *   item     one non-empty tile; item_tile = (outer * nty + ty) * ntx + tx, item_blk[n_items + 1] = its blocks
 *   block    8 entries of ONE row inside the tile (a row's entries there are padded with zero weights to whole blocks, and
 *            cut after 128): loc[8 n_blocks] = position inside the tile (ly * tw + lx, uint16), wgt[8 n_blocks] = float32
 *            weight (the reference's storage type), both 16-byte aligned; blk_slot[n_blocks] = where the partial sum of
 *            the block's PIECE goes -- a piece = the consecutive blocks of one row inside one 64-block step of the tile
 *            (at most 16) -- with the slots of a row consecutive: row_slot[n_rows + 1]
 * Launch 1: a workgroup loads its tile of x into LDS with whole-line reads; a lane multiplies and adds one block in order,
 * the lanes of a piece are joined by shuffle steps 1, 2, 4, 8, one partial sum per piece into `scratch` (count * n_slots
 * doubles, caller-owned); launch 2: every row adds its partial sums in slot order (16 lanes + tree).  fp64 accumulation,
 * fixed order, no atomics: bit-reproducible.  `count` members (1 .. NK_MAX_BATCH) share the matrix: x / y are host arrays
 * of device pointers. */
typedef struct nk_tiled_csr {
  int64_t n_rows, n_slots;
  int32_t n_items, ny, nx, th, tw;
  const int32_t* item_tile;
  const int64_t* item_blk;
  const int32_t* blk_slot;
  const int64_t* row_slot;
  const uint16_t* loc;
  const float* wgt;
} nk_tiled_csr;
int nk_tiled_rowsum(const nk_tiled_csr* m, int count, const void* const* x, void* const* y, double* scratch, int dtype,
                    void* stream);

/* nk_roll: out[(i_0 + shift_0) mod n_0, ..., (i_{d-1} + shift_{d-1}) mod n_{d-1}] = in[i_0, ..., i_{d-1}] for a contiguous
 * C-ordered array of `ndim` <= 6 axes (any integer shifts); elements are opaque units of elem_bytes = 4, 8 or 16.  Replaces
 * np.fft.fftshift / ifftshift of FFTShiftOperator.apply (operators/harmonic_operators.py:420-423).  in != out. */
int nk_roll(int ndim, const int64_t* shape, const int64_t* shift, int elem_bytes, const void* in, void* out, void* stream);

/* nk_bluestein_rows: out[r][k] = scale * sum_j in[r][j] exp(-+ 2 pi i j k / n) for `rows` contiguous rows of ANY length n in
 * one launch (chirp-z; what ducc0's c2c does for the prime lengths of test_fft_operator.py:58-103): w[n] = the chirp
 * exp(-+ i pi j^2 / n) (its sign is the transform's direction), bhat_br[m] = the length-m FFT of the filter
 * b[j] = conj(w)[|j|] (cyclic) in BIT-REVERSED order, tw[m/2] = exp(-2 pi i k / m); m = a power of two >= max(4, 2 n - 1)
 * with m * sizeof(complex) <= 64 KiB.  dtype NK_F32 / NK_F64 = complex64 / complex128.  in_real != 0: the input rows are
 * real (promotion of the first axis of a real transform); out_hartley = +-1: the output rows are real, Re X + out_hartley Im X
 * (the Hartley combination behind the last axis; ducc_dispatch.py:88-100), else complex.  in may equal out for complex ends. */
int nk_bluestein_rows(int64_t rows, int n, int m, const void* in, const void* w, const void* bhat_br, const void* tw, void* out,
                      double scale, int in_real, int out_hartley, int dtype, void* stream);

/* inclusive prefix sums out[i] = sum_{j<=i} in[j] (reverse != 0: suffix sums), fp64 accumulation: the cumulative sums of
 * _TwoLogIntegrations in the GENERIC amplitude graph (library/correlated_fields.py:147-161); in and out may not overlap
 * partially (in == out is fine) */
int nk_cumsum(int64_t n, const void* in, void* out, int reverse, int dtype, void* stream);

/* row-wise complex helper of the any-length (chirp-z) composition in nifty_amd/backend.py -- the transforms of lengths
 * the planner rejects (ducc0 takes every length, ducc_dispatch.py:116-132) are three power-of-two nk_fftn calls per axis
 * glued by this kernel.  a is rows x in_cols, out is rows x out_cols (both C order), w (may be NULL = 1) has at least
 * min(in_cols, out_cols) complex entries:
 *   out[r][c] = scale * op(a[r][c]) * w[c]   for c < min(in_cols, out_cols),   0 for the padded columns c >= in_cols
 *   mode 0: a complex, out complex;  mode 1: a real, out complex;  mode 2: a complex, out real = Re + sgn * Im (w unused) */
int nk_cplx_rows(int64_t rows, int64_t in_cols, int64_t out_cols, const void* a, const void* w, void* out, int mode,
                 double scale, int sgn, int dtype, void* stream);

/* dst[0..n) = sum over c < copies of src[c*stride + (0..n)]  (folds the per-XCD VJP accumulators) */
int nk_fold_copies(int64_t n, int copies, int64_t stride, const double* src, double* dst, void* stream);

/* ---- octant helpers: |k| binning is invariant under the sign flip of every axis, so bin-indexed tables are
 *      expanded / reduced on the octant k_i <= n_i/2 only (1/8 of the gathers / atomics in 3-D).
 *      shape = the `ndim` (1..3) grid axes; fields are [shape...] C order.
 * nk_octant_expand : field[x] = table[pidx[x]] for every grid point, one table gather per octant point
 *                    (PowerDistributor TIMES, distributors.py:114-119); compact != 0: only the octant array
 *                    field8[A/2+1][M/2+1][NL/2+1] is written (see nk_fuse.field_octant)
 * nk_octant_scatter: abar[pidx[x]] += w8[x] over the octant array w8[A/2+1][M/2+1][NL/2+1] written by the VJP
 *                    epilogue; merge_swapped_lines != 0 (only valid when the first two axes have equal length AND
 *                    equal harmonic distances) folds the lines (a,b) and (b,a) before the atomics
 *                    (PowerDistributor ADJOINT_TIMES, distributors.py:106-112) */
int nk_octant_expand(int ndim, const int64_t* shape, const void* table, const int32_t* pidx, void* field, int dtype,
                     int compact, void* stream);
int nk_octant_scatter(int ndim, const int64_t* shape, const double* w8, const int32_t* pidx, double* abar,
                      int merge_swapped_lines, void* stream);
/* nk_octant_expand (compact) for NATURAL binning on a grid with equal harmonic distances, without the index stream:
 * field8[a][b][c] = table[bin of k^2 = a^2 + b^2 + c^2].  `table`: nb doubles (as nk_amp_forward / nk_amp_jvp produce them),
 * bin_k2[nb]: k^2 of every bin (ascending), `dense`: scratch of (max k^2 + 1) elements of `dtype` (the table spread over
 * k^2, rewritten by every call), field8: the octant array in `dtype`.  line_order (optional, may be NULL): the
 * (n_first/2+1) * (n_middle/2+1) octant line indices a * (n_middle/2+1) + b sorted by a^2 + b^2 -- lines that read the same
 * stretch of the table then share a workgroup and its vector cache (same result, 3-D grids: ~2x faster).
 * PowerDistributor TIMES, distributors.py:114-119. */
int nk_octant_expand_k2(int ndim, const int64_t* shape, const double* table, const int32_t* bin_k2, int64_t nb, void* dense,
                        void* field8, int dtype, const int32_t* line_order, void* stream);
/* nk_octant_scatter for NATURAL binning on a grid with equal harmonic distances on all axes (bins = the distinct integer
 * k^2 = a^2+b^2+c^2 in ascending order; bin_k2[nb] = k^2 of every bin): abar[.] = sum over the octant array, OVERWRITING
 * abar.  Shell-binned: blocks of consecutive bins are spherical shells whose cut with every octant line is a c-range
 * known from two integer square roots, accumulated in LDS -- no global atomics.  scratch: >= 64*(nb+32) doubles.
 * w8max (device scalar, optional): max |w8| as produced by nk_fuse.w8max.  With it the contributions are accumulated in
 * 64-bit FIXED POINT (quantum 2^(e-44), 2^e >= *w8max): integer additions commute, so the bin sums are the same bits on
 * every run at the speed of the atomics; rounding <= 2.8e-14 * max |w8| per point (used while a workgroup's share of the
 * octant stays far below the 2^18-point overflow bound: up to ~1100^3; larger grids fall back to the next case).  NULL: floating-point LDS atomics (sums
 * differ in the last bit from run to run).  NK_SCATTER_FP_ATOMICS=1 (environment) forces the latter. */
int nk_octant_scatter_k2(int ndim, const int64_t* shape, const double* w8, const int32_t* pidx, const int32_t* bin_k2,
                         int64_t nb, double* scratch, double* abar, const double* w8max, void* stream);
/* nk_segment_sum: dst[s] (+)= sum_{rowptr[s] <= i < rowptr[s+1]} src[perm[i]] -- the same scatter-add for a static index map
 * given as a bin-sorted permutation of the source points (rowptr: int32[nseg+1], perm: int32[rowptr[nseg]]): one thread per
 * bin, fixed summation order, no atomics.  The engine uses it for the quadrant sums of 2-D grids. */
int nk_segment_sum(int64_t nseg, const int32_t* rowptr, const int32_t* perm, const double* src, double* dst, int accumulate,
                   void* stream);
/* 1 if nk_hartley_fused on this plan honours nk_fuse.w8 (the register-resident pipeline is active) */
int nk_plan_octant_vjp(const nk_plan* plan);

/* power-bin index of every grid point from integer k^2 (equal harmonic distances): pidx[i] = k2table[k^2(i)],
 * rho[bin] += 1 (rho may be NULL, else zeroed by the caller).  Replaces the int64 full-grid searchsorted of
 * PowerSpace.__init__ (domains/power_space.py:172-180) for natural binning. */
int nk_pindex_from_k2(int ndim, const int64_t* shape, const int32_t* k2table, int32_t* pidx, int64_t* rho,
                      void* stream);

/* ---- fused CG updates with device-resident scalars (conjugate_gradient.py:85-126).
 *      scal = device double[8]: [0] gamma_prev  [1] curv  [2] gamma  [3] x.r  [4] x.b  [5] alpha  [6] beta
 *      nk_cg_curv : scal[1] = d.q
 *      nk_cg_update: alpha = scal[0]/scal[1]; x -= alpha d; r -= alpha q; scal[2] = r.r; scal[3] = x.r;
 *                    scal[4] = x.b  (b may be NULL)
 *      nk_cg_direction: beta = max(0, scal[2]/scal[0]); d = beta d + r
 *      accumulate != 0: do not zero the reduction slots first (second segment of a split vector) */
int nk_cg_curv(int64_t n, const void* d, const void* q, int dtype, double* scal, int accumulate, void* stream);
int nk_cg_update(int64_t n, void* x, void* r, const void* d, const void* q, const void* b, int dtype,
                 double* scal, int accumulate, void* stream);
/* nk_cg_update without the linear term: alpha = scal[0]/scal[1]; scal[3] = d.r (OLD residual); x -= alpha d; r -= alpha q;
 * scal[2] = r.r.  The quadratic energy of the iterate (quadratic_energy.py:31-39, which the reference re-evaluates from
 * x, r, b every iteration: conjugate_gradient.py:100-101) advances by dE = -alpha d.r + alpha^2/2 d.q */
int nk_cg_update_dr(int64_t n, void* x, void* r, const void* d, const void* q, int dtype, double* scal, int accumulate,
                    void* stream);
/* roll != 0: after the update also do scal[5]=alpha, scal[6]=beta, scal[0]=scal[2], scal[2..4]=0 (call once per iteration,
 * on the last segment of a multi-segment vector; n == 0 with d = r = NULL only rolls).  The slots of the vector update are
 * then clean: the next nk_cg_update[_dr] may pass accumulate != 0 for every segment.  roll == 2 also clears scal[1], for a
 * caller whose next d.q is ACCUMULATED into the slot after this call (an epilogue deposit followed by nk_cg_curv with
 * accumulate != 0); roll == 1 leaves it alone, for the fused direction update that runs after the deposit */
int nk_cg_direction(int64_t n, void* d, const void* r, int dtype, double* scal, int roll, void* stream);

/* ---- amplitude fields of PRODUCT spectra (library/correlated_fields.py:713-764: CorrelatedFieldMaker.finalize multiplies
 *      the amplitudes of the sub-spaces, each distributed over the full harmonic domain by ContractionOperator.adjoint @
 *      PowerDistributor, and the zero-mode amplitude; :809-858 get_normalized_amplitudes).  The grid is [S0][S1][S2] in C
 *      order -- up to three sub-spaces, S_i = number of points of sub-space i (a sub-space may be multi-dimensional:
 *      its points flattened), the full grid or the OCTANT arrays of nk_fuse.field_octant / w8 alike.
 *        pidx[i]  device int32[S_i]   power bin of every point of sub-space i
 *        tab[i]   device double[nb_i] the sub-space's amplitude table;  dtab[i]: a tangent of it, or NULL
 *        scale    device double       overall factor (the zero-mode amplitude);  dscale: its tangent, or NULL
 *      nk_product_field     out[k] = scale * prod_i tab_i[pidx_i(k_i)]                     (tangent == 0)
 *                           out[k] = dscale * prod_i tab_i + scale * sum_i dtab_i prod_(j != i) tab_j   (tangent != 0)
 *                           -- the field `afield` / `dafield` of the AMP / AMP_JVP prologues and the VJP epilogue
 *      nk_product_marginal  marg[b] = sum over the points k with k_which = b of w[k] * scale * prod_(j != which) tab_j:
 *                           the adjoint of the distribution for sub-space `which` (reduce it over the bins with
 *                           nk_csr_rowsum); w = the per-point sums xi * t of the VJP epilogue (nk_fuse.wfull / w8).  Every
 *                           sum runs in a fixed order (bit-reproducible).  scratch: nk_product_marginal_scratch bytes */
typedef struct nk_product {
  int nsub;
  int64_t size[3];
  const int32_t* pidx[3];
  const double* tab[3];
  const double* dtab[3];
  const double* scale;
  const double* dscale;
} nk_product;
int nk_product_field(const nk_product* p, int tangent, void* out, int dtype, void* stream);
/* nk_mirror_combine: the SEPARABLE Hartley transform of a product domain -- one Hartley transform per sub-space, which is
 * what a chain of HarmonicTransformOperators with `space=` computes (library/correlated_fields.py:726-730,
 * operators/harmonic_operators.py:97-161) -- from the genuine N-D one: cas(a) cas(b) = 1/2 [cas(a+b) + cas(a-b) + cas(-a+b)
 * - cas(-a-b)], so  out[k] = scale * sum_s coef[s] in[flip_s(k)] + offset  over the 2^nsub sign patterns s (bit i of s set:
 * the axes of sub-space i are mirrored, k -> (n - k) mod n).  group[ax] = sub-space of grid axis ax.  The combination is
 * symmetric and commutes with the transform: apply it to the output of nk_hartley_fused in a forward evaluation, to the
 * input in an adjoint one.  Out of place. */
int nk_mirror_combine(int ndim, const int64_t* shape, const int* group, int nsub, const double* coef, const void* in,
                      void* out, double scale, double offset, int dtype, void* stream);
size_t nk_product_marginal_scratch(const nk_product* p, int which);
int nk_product_marginal(const nk_product* p, int which, const double* w, double* scratch, double* marg, void* stream);

/* ---- amplitude model on the nb power bins (library/correlated_fields.py:89-208,277-386).
 *      geo  = double[4*nb]: rel[nb], sc[nb], mult[nb], delta[nb] (delta uses the first nb-2 slots)
 *      hyp  = double[11]: (logmean, logsigma) of fluctuations, flexibility, asperity, zeromode;
 *                         (mean, sigma) of loglogavgslope; total volume V
 *      lat  = device double[5 + 2*(nb-2)]: xi_asperity, xi_flexibility, xi_fluctuations, xi_slope,
 *             xi_zeromode, then spectrum[2][nb-2]
 *      state= device double[8*nb + 16] scratch kept between forward and jvp/vjp at the same point */
int nk_amp_forward(int nb, const double* geo, const double* hyp, const double* lat, double* state, double* amp,
                   void* stream);
int nk_amp_jvp(int nb, const double* geo, const double* hyp, const double* lat, double* state, const double* dlat,
               double* damp, void* stream);
int nk_amp_vjp(int nb, const double* geo, const double* hyp, const double* lat, double* state, const double* abar,
               double* latbar, void* stream);

/* ---- random fields: numpy's `Generator(PCG64).normal(mean, std, n)` stream reproduced on the device, draw for draw
 *      (replaces the host draws of nifty/cl/random.py:219-237 `Random.normal`, reached from field.py:128-156
 *      `Field.from_random`, multi_field.py:109-153 and kl_energies.py:91-159; numpy is the reference's third-party RNG:
 *      PCG XSL-RR 128/64 + 256-strip ziggurat, restated in nifty_amd/csrc/nk_rng.h).
 *      state, inc: HOST pointers, the 128-bit generator state and increment as {high, low} 64-bit words
 *                  (`bit_generator.state["state"]`); inc is odd
 *      out       : device, n values of dtype (fp32 = the fp64 draw rounded once, like `.astype(float32)`)
 *      scratch   : device, nk_pcg64_normal_scratch_bytes(n, attempt) bytes
 *      status    : device uint64[2]: [0] <- number of raw 64-bit draws the n normals consumed (advance the host generator
 *                  by it to stay in lockstep), [1] <- error bits: 2 = the scratch sizing was too tight, call again with
 *                  attempt + 1 (never seen for attempt 0 beyond n ~ 10^3); 1 = chain merge failed (probability < 1e-60)
 *      Asynchronous on `stream`; values are bit-identical to numpy's except in the |x| > 3.654 tail (2.7e-4 of the
 *      draws), where log1p of the device math library may differ from the host libm in the last bit. */
int64_t nk_pcg64_normal_scratch_bytes(int64_t n, int attempt);
int nk_pcg64_normal(const uint64_t* state, const uint64_t* inc, int64_t n, double mean, double std, void* out, int dtype,
                    void* scratch, int64_t scratch_bytes, int attempt, uint64_t* status, void* stream);
/* Draws that consume a FIXED number of raw PCG64 values per output (plain jump-ahead parallelism), for fields on a GPU:
 * nk_pcg64_uniform: out[i] = low + (high - low) * next_double -- `Generator.uniform(low, high, n)` behind Random.uniform
 *                   (nifty/cl/random.py:249-258); consumes n raw draws.
 * nk_pcg64_pm1    : the reference's Random.pm1 (random.py:239-247): +-1 from `integers(0, 2, n)` (complex_units == 0) or
 *                   one of 1, i, -1, -i from `integers(0, 4, n)` (complex_units != 0; out = interleaved re, im of `dtype`).
 *                   numpy's buffered 32-bit bounded path yields two outputs per raw draw (low half first) and never rejects
 *                   for these ranges; consumes ceil(n / 2) raw draws, the caller keeps numpy's has_uint32 / uinteger
 *                   buffer consistent (nifty_amd.backend.pcg64_pm1).
 * state / inc: the generator BEFORE the first draw, two uint64 (hi, lo) each; scratch: nk_pcg64_fixed_scratch_bytes(). */
int64_t nk_pcg64_fixed_scratch_bytes(void);
int nk_pcg64_uniform(const uint64_t* state, const uint64_t* inc, int64_t n, double low, double high, void* out, int dtype,
                     void* scratch, void* stream);
int nk_pcg64_pm1(const uint64_t* state, const uint64_t* inc, int64_t n, void* out, int dtype, int complex_units, void* scratch,
                 void* stream);

/* nk_pcg64_integers: numpy's Generator.integers(low, low + rng + 1, size = n) on a PCG64 stream (Random.uniform of integer
 * fields, reference random.py:252-256; numpy's random_bounded_uint64_fill without masking: Lemire's method on 32-bit words
 * for rng < 2^32, on 64-bit draws beyond, rejected words skipped) as int64 values.  state / inc as for nk_pcg64_normal (the
 * caller has consumed a buffered 32-bit half itself); `nthreads` threads of 32 raw draws each are walked -- the caller sizes
 * them for the expected number of rejections and retries with more when status[1] reports NK_RNG_ERR_SHORT (bit 1).
 * status[0] = words consumed through output n - 1 (32-bit words for rng < 2^32 - 1... see nk_rng.h; 64-bit draws beyond).
 * scratch: nk_pcg64_integers_scratch_bytes(nthreads) bytes. */
int64_t nk_pcg64_integers_scratch_bytes(int64_t nthreads);
int nk_pcg64_integers(const uint64_t* state, const uint64_t* inc, int64_t n, int64_t low, uint64_t rng, int64_t nthreads, int64_t* out,
                      void* scratch, uint64_t* status, void* stream);

/* ---- batched launches: ONE launch per kernel for up to NK_MAX_BATCH independent members -------------------------------
 *      On small grids (2048^2: 128 workgroups per transform pass on 256 CUs, amplitude kernels of 5-20 us) one member's
 *      kernel chain leaves most of the chip idle and the launches themselves set the time.  The members of a batch are
 *      what the reference loops over: the samples of a KL evaluation / metric application (SampledKLEnergyClass,
 *      minimization/kl_energies.py:306-350: `for s in samples`), and the independent linear solves that draw the samples
 *      of one iteration (draw_samples, kl_energies.py:132-158; SamplingEnabler.special_draw_sample,
 *      operators/sampling_enabler.py:64-86).  Every `*_batch` entry point below is its single-member counterpart for
 *      `count` members with the second grid dimension running over the members: per-member arguments are HOST arrays of
 *      `count` device pointers (or an array of `count` nk_fuse records), everything else is shared.  The arithmetic of a
 *      member -- grid, summation orders, reduction slots -- is exactly that of the single call, so the results are
 *      bit-identical to `count` single calls.  1 <= count <= NK_MAX_BATCH. */
#define NK_MAX_BATCH 8
/* 1 if nk_hartley_fused_batch runs the members of this plan in shared launches (2-D plans of the register-resident
 * pipeline, nk_plan_octant_vjp != 0); 0: it falls back to one nk_hartley_fused per member (same results) */
int nk_plan_batch_ok(const nk_plan* plan);
/* nk_hartley_fused for fuse[0..count) on workspace[0..count) (each nk_plan_workspace_bytes, pairwise distinct).  All
 * members must select the same kernel classes: same prologue / epilogue, the same of {afield, dafield, field_octant,
 * io32} set or unset (NK_ERR_INVALID otherwise); pointers, scalars and the optional addend / accumulate / carry / value
 * may differ per member. */
int nk_hartley_fused_batch(const nk_plan* plan, const nk_fuse* fuse, int count, int convention, void* const* workspace,
                           void* stream);
/* amplitude model (nk_amp_forward / nk_amp_jvp / nk_amp_vjp) for `count` latent points: lat / state / amp / dlat / damp /
 * abar / latbar are arrays of `count` device pointers (state and the outputs pairwise distinct; lat or dlat may repeat) */
int nk_amp_forward_batch(int nb, const double* geo, const double* hyp, int count, const double* const* lat,
                         double* const* state, double* const* amp, void* stream);
int nk_amp_jvp_batch(int nb, const double* geo, const double* hyp, int count, const double* const* lat, double* const* state,
                     const double* const* dlat, double* const* damp, void* stream);
int nk_amp_vjp_batch(int nb, const double* geo, const double* hyp, int count, const double* const* lat, double* const* state,
                     const double* const* abar, double* const* latbar, void* stream);
/* nk_gather with one table and one output per member over a shared index; nk_csr_rowsum with one x / y per member over
 * a shared matrix */
int nk_gather_batch(int64_t n, int count, const void* const* table, const int32_t* pidx, void* const* out, int dtype,
                    void* stream);
int nk_csr_rowsum_batch(int64_t nrows, const int64_t* rowptr, const int32_t* col, const float* wgt, int count,
                        const void* const* x, void* const* y, int dtype, int lanes, void* stream);
/* out[m] = alpha[m] * x[m] + beta[m] * y[m] (y[m] may be NULL); the _sqnorm form also result[m] (+)= sum(out[m]^2) */
int nk_axpby_batch(int64_t n, int count, const double* alpha, const void* const* x, const double* beta,
                   const void* const* y, void* const* out, int dtype, void* stream);
int nk_axpby_sqnorm_batch(int64_t n, int count, const double* alpha, const void* const* x, const double* beta,
                          const void* const* y, void* const* out, int dtype, double* const* result, int accumulate,
                          void* stream);
/* out[m] = a[m] (op) b[m] with the scalar conventions of nk_binary per member (a[m] == NULL: ascalar[m], ...) */
int nk_binary_batch(int op, int64_t n, int count, const void* const* a, const double* ascalar, const void* const* b,
                    const double* bscalar, void* const* out, int dtype, void* stream);
int nk_vdot_batch(int64_t n, int count, const void* const* a, const void* const* b, int dtype, double* const* result,
                  int accumulate, void* stream);
/* out = the sum of term[0..count) element by element in the order of the reference's task-count-independent sum
 * (nifty/cl/utilities.py:349-414: neighbours first, then pairs of pairs, ...; every partial sum rounded to `dtype` like a
 * stored vector) -- the sum over the samples of a batch; out may be term[0] */
int nk_sum_tree(int64_t n, int count, const void* const* term, void* out, int dtype, void* stream);
/* the fused CG updates for `count` independent solves, scal[m] = that solve's device double[8] */
int nk_cg_curv_batch(int64_t n, int count, const void* const* d, const void* const* q, int dtype, double* const* scal,
                     int accumulate, void* stream);
int nk_cg_update_batch(int64_t n, int count, void* const* x, void* const* r, const void* const* d, const void* const* q,
                       const void* const* b, int dtype, double* const* scal, int accumulate, void* stream);
int nk_cg_update_dr_batch(int64_t n, int count, void* const* x, void* const* r, const void* const* d,
                          const void* const* q, int dtype, double* const* scal, int accumulate, void* stream);
int nk_cg_direction_batch(int64_t n, int count, void* const* d, const void* const* r, int dtype, double* const* scal,
                          int roll, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NIFTYK_H */
