"""ONE launch set for all members of a batch on small grids (the `*_batch` entry points of include/niftyk.h).

On a 2048^2 grid one sample's kernel chain -- ~17 launches per metric application: amplitude scans of 5-20 us, transform
passes with 128 workgroups on 256 CUs -- leaves most of the chip idle and is bound by the launches themselves (round 4:
6.6 k launches and 116 ms per MGVI iteration of BASELINE config 2, 0.17 of the HBM roofline; four stream lanes only
overlapped the chains).  The members of a batch are exactly what the reference loops over:

  * the samples of a KL evaluation / metric application (SampledKLEnergyClass, minimization/kl_energies.py:306-350:
    `for s in self._sample_list`), one linearisation point each, one common direction;
  * the independent linear solves that draw the samples of an iteration (draw_samples, kl_energies.py:132-158;
    SamplingEnabler.special_draw_sample, operators/sampling_enabler.py:64-86): one linearisation point, one right-hand side
    and CG state each (`BatchCG`).

Every kernel of the chain is launched once with the second grid dimension running over the members; a member's
arithmetic -- grids, summation orders, reduction slots -- is that of the single launch, and the sum over the samples is
the reference's pairwise tree (nk_sum_tree), so the results are bit-identical to the unbatched paths (NK_BATCH=0;
tests/test_batched_gpu.py).
"""
import copy
import ctypes
import os

import torch

from . import _lib as L
from . import backend as B
from . import parallel

MAX = L.MAX_BATCH


def ready(model):
    """True when the chains of this model can run batched: the strided-first pipeline on a 2-D grid with the quadrant
    pipeline (FusedModel.octant_vjp) and the fixed-order bin sums, a grid small enough that one chain leaves the chip idle
    (NK_LANE_MAX_POINTS, the limit of the stream lanes this replaces), NK_BATCH != 0."""
    cached = model.__dict__.get("_batch_ready")
    if cached is None:
        small = model.N <= int(os.environ.get("NK_LANE_MAX_POINTS", str(1 << 25)))
        cached = bool(os.environ.get("NK_BATCH", "1") != "0" and small and model.octant_vjp and not model.sandwich
                      and model.pidx8 is not None and model.k2_dense is None and model.seg_plan is not None
                      and model.full_plan is None and not (model.wide or model.wide_response)
                      and L.load().nk_plan_batch_ok(model.plan.handle))
        model.__dict__["_batch_ready"] = cached
    return cached


def scalars(count, device):
    """`count` zeroed device doubles as one-element tensors, 16 bytes apart: every member of a batched launch must be
    16-byte aligned to share a launch (nk_launch_map_b), and one allocation + one fill serves them all."""
    return [row[:1] for row in torch.zeros(count, 2, dtype=torch.float64, device=device).unbind(0)]


def _ptrs(tensors):
    return L.ptr_array(list(tensors))


def _check(rc, what):
    L.check(rc, what)


class Point:
    """What a member needs of its linearisation point: the latent point (xi, small), amplitude table / field / state and
    the s-space (or data-space) weights.  `of(lp)` shares everything with an engine.LinPoint; `clone_state()` gives a
    member that shares the read-only parts and owns the amplitude scratch `state` (several members at ONE point: the
    sampling solves)."""

    __slots__ = ("lp", "state")

    def __init__(self, lp, state=None):
        self.lp, self.state = lp, lp.state if state is None else state

    @staticmethod
    def of(lp):
        return Point(lp)

    def clone_state(self):
        return Point(self.lp, self.lp.state.clone())


# ---- batched primitives --------------------------------------------------------------------------------------------------
def axpby(alphas, xs, betas=None, ys=None, outs=None):
    """outs[m] = alphas[m] xs[m] + betas[m] ys[m] (new tensors by default)."""
    n = len(xs)
    outs = [torch.empty_like(x) for x in xs] if outs is None else outs
    betas = [0.0] * n if betas is None else betas
    ys = [None] * n if ys is None else ys
    _check(L.load().nk_axpby_batch(xs[0].numel(), n, L.double_array(alphas), _ptrs(xs), L.double_array(betas), _ptrs(ys),
                                   _ptrs(outs), B.dtype_code(xs[0]), B._stream()), "nk_axpby_batch")
    return outs


def axpby_sqnorm(alphas, xs, betas, ys, results, accumulate=False):
    outs = [torch.empty_like(x) for x in xs]
    _check(L.load().nk_axpby_sqnorm_batch(xs[0].numel(), len(xs), L.double_array(alphas), _ptrs(xs), L.double_array(betas),
                                          _ptrs(ys), _ptrs(outs), B.dtype_code(xs[0]), _ptrs(results), 1 if accumulate else 0,
                                          B._stream()), "nk_axpby_sqnorm_batch")
    return outs


def binary(op, a_list, b_list, a_scalars=None, b_scalars=None, outs=None):
    n = len(a_list)
    like = [a if a is not None else b for a, b in zip(a_list, b_list)]
    outs = [torch.empty_like(t) for t in like] if outs is None else outs
    _check(L.load().nk_binary_batch(op, like[0].numel(), n, _ptrs(a_list), L.double_array(a_scalars or [0.0] * n),
                                    _ptrs(b_list), L.double_array(b_scalars or [0.0] * n), _ptrs(outs), B.dtype_code(like[0]),
                                    B._stream()), "nk_binary_batch")
    return outs


def vdot(a_list, b_list, results, accumulate=False):
    _check(L.load().nk_vdot_batch(a_list[0].numel(), len(a_list), _ptrs(a_list), _ptrs(b_list), B.dtype_code(a_list[0]),
                                  _ptrs(results), 1 if accumulate else 0, B._stream()), "nk_vdot_batch")


def sum_tree(terms, out=None):
    """The sum of the tensors in the order of parallel.pair_tree, in one pass (out defaults to terms[0])."""
    out = terms[0] if out is None else out
    if len(terms) > MAX:
        raise ValueError(f"sum_tree: at most {MAX} terms")
    _check(L.load().nk_sum_tree(out.numel(), len(terms), _ptrs(terms), out.data_ptr(), B.dtype_code(out), B._stream()),
           "nk_sum_tree")
    return out


def rowsum(plan_or_csr, xs, ys, weighted=None, lanes=None, nrows=None, dtype_code=None):
    """nk_csr_rowsum over a shared matrix: plan_or_csr = (rowptr, col, lanes) of backend.bin_plan, or (rowptr, col, wgt)
    with `lanes` and `nrows` given."""
    if weighted is None:
        rowptr, col, lanes = plan_or_csr
        wgt = None
    else:
        rowptr, col, wgt = plan_or_csr
    nrows = rowptr.numel() - 1 if nrows is None else nrows
    _check(L.load().nk_csr_rowsum_batch(nrows, rowptr.data_ptr(), col.data_ptr(), B.ptr(wgt), len(xs), _ptrs(xs), _ptrs(ys),
                                        B.dtype_code(xs[0]) if dtype_code is None else dtype_code, lanes, B._stream()),
           "nk_csr_rowsum_batch")
    return ys


class Scratch:
    """`count` private scratch sets of a model (amplitude tangents, quadrant sums, workspaces ...): the lanes of
    FusedModel.lanes, used as plain buffer holders -- every launch goes to the caller's stream."""

    def __init__(self, model, count):
        self.sets = model.lanes(count)

    def __getitem__(self, name):
        return [getattr(s, name) for s in self.sets]

    def workspaces(self):
        return [s.plan.workspace for s in self.sets]


def _transform(model, fuses, scr):
    count = len(fuses)
    arr = (L.Fuse * count)(*fuses)
    ws = _ptrs(scr.workspaces()[:count])
    if model.plan.device.index != torch.cuda.current_device():
        B._wrong_device(model.plan.device.index)
    _check(L.load().nk_hartley_fused_batch(model.plan.handle, arr, count, B._convention(), ws, B._stream()),
           "nk_hartley_fused_batch")
    model._count("transforms", count)


def _expand(model, tables, outs):
    """table[pindex] of the quadrant points, one table per member (FusedModel._amp_field on the batch path)."""
    if outs[0].dtype != torch.float64:
        tables = [t.to(outs[0].dtype) for t in tables]
    _check(L.load().nk_gather_batch(outs[0].numel(), len(outs), _ptrs(tables), model.pidx8.data_ptr(), _ptrs(outs),
                                    B.dtype_code(outs[0]), B._stream()), "nk_gather_batch")
    return outs


def _amp_jvp(model, points, dsmalls, damps):
    _check(L.load().nk_amp_jvp_batch(model.nb, model.geo.data_ptr(), model.hyp.data_ptr(), len(points),
                                     _ptrs([p.lp.x.small for p in points]), _ptrs([p.state for p in points]), _ptrs(dsmalls),
                                     _ptrs(damps), B._stream()), "nk_amp_jvp_batch")


def _amp_vjp(model, points, abars, latbars):
    _check(L.load().nk_amp_vjp_batch(model.nb, model.geo.data_ptr(), model.hyp.data_ptr(), len(points),
                                     _ptrs([p.lp.x.small for p in points]), _ptrs([p.state for p in points]), _ptrs(abars),
                                     _ptrs(latbars), B._stream()), "nk_amp_vjp_batch")


def _vjp(model, scr, points, ws, w2s, scale, outs_xi, addends, dot_outs=None):
    """engine.FusedModel._vjp for the members: outs_xi[m] = a t (+ fac_m * vec_m.xi), quadrant sums -> bins -> latbar[m];
    t = scale * HT(ws[m] (* w2s[m]))."""
    count = len(points)
    fuses = []
    w8s, abars, latbars = scr["w8"][:count], scr["abar"][:count], scr["latbar"][:count]
    for m, pt in enumerate(points):
        lp = pt.lp
        f = model._fuse()
        f.pro, f.in_ = L.PRO_PLAIN, ws[m].data_ptr()
        if w2s is not None:
            f.pro, f.in2 = L.PRO_MUL, w2s[m].data_ptr()
        f.epi, f.out, f.scale = L.EPI_VJP, outs_xi[m].data_ptr(), model.h_dvol * scale
        f.pidx, f.amp, f.xi = model.pidx.data_ptr(), lp.amp.data_ptr(), lp.x.xi.data_ptr()
        f.afield = B.ptr(lp.afield)
        vec, fac = addends[m] if addends[m] is not None else (None, 0.0)
        f.addend, f.addend_scale, f.accumulate = (B.ptr(vec.xi) if (vec is not None and fac) else None), fac, 0
        if dot_outs is not None and dot_outs[m] is not None:
            f.value = dot_outs[m].data_ptr()
        f.abar, f.w8 = abars[m].data_ptr(), w8s[m].data_ptr()
        fuses.append(f)
    _transform(model, fuses, scr)
    rowsum(model.seg_plan, w8s, abars, dtype_code=L.NK_F64)
    _amp_vjp(model, points, abars, latbars)
    return latbars


def metric(model, scr, points, ds, outs, scale, addends, dot_outs=None):
    """outs[m] <- scale * J^T M J ds[m] + fac_m * vec_m at points[m] (engine.FusedModel.lh_metric_accumulate with
    first=True for every member): outs[m] is a LatentVec whose xi is written and whose small part is created.
    addends[m] = (vector, factor) or None.  dot_outs[m]: device double += ds[m].xi . outs[m].xi (needs addend = ds[m])."""
    count = len(points)
    if count > MAX:
        raise ValueError("batch too large")
    damps, dafields, tmps = scr["damp"][:count], scr["dafield"][:count], scr["tmp"][:count]
    _amp_jvp(model, points, [d.small for d in ds], damps)
    _expand(model, damps, dafields)
    fuses = []
    for m, pt in enumerate(points):
        lp, d = pt.lp, ds[m]
        f = model._fuse()
        f.pro, f.in_, f.in2 = L.PRO_AMP_JVP, d.xi.data_ptr(), lp.x.xi.data_ptr()
        f.pidx, f.amp, f.damp = model.pidx.data_ptr(), lp.amp.data_ptr(), damps[m].data_ptr()
        f.afield, f.dafield = B.ptr(lp.afield), dafields[m].data_ptr()
        if model.response is not None:
            f.epi, f.out, f.mul, f.mul_scalar = L.EPI_MUL, tmps[m].data_ptr(), lp.gp.data_ptr(), 1.0
        else:
            f.epi, f.out, f.mul, f.mul_scalar = L.EPI_MUL, tmps[m].data_ptr(), B.ptr(lp.mid), lp.mid_scalar
        fuses.append(f)
    _transform(model, fuses, scr)
    if model.response is not None:
        ws = _through_response(model, points, tmps)
        w2s = [pt.lp.gp for pt in points]
    else:
        ws, w2s = tmps, None
    latbars = _vjp(model, scr, points, ws, w2s, scale, [o.xi for o in outs], addends, dot_outs)
    facs = [a[1] if a is not None else 0.0 for a in addends]
    vecs = [a[0].small if (a is not None and a[1]) else None for a in addends]
    smalls = axpby([1.0] * count, latbars, facs, vecs)
    for o, s in zip(outs, smalls):
        o.small = s
    model._count("metric", count)
    return outs


def _through_response(model, points, grids):
    """R^T M_d R applied to the members' grid fields (engine: response.adjoint(_weigh_data(response.times(.))))."""
    resp, count = model.response, len(points)
    a = resp._arrays(grids[0].device)
    nnz = resp.host_matrix.nnz
    flat = [g.reshape(-1) for g in grids]
    us = [torch.empty(resp.n_data, dtype=g.dtype, device=g.device) for g in grids]
    tm = resp.tiled(grids[0].device)
    if tm is not None:
        tm.rowsum(flat, us)
    else:
        rowsum((a[0], a[1], a[2]), flat, us, weighted=True, lanes=B.lanes_for(nnz, resp.n_data), nrows=resp.n_data)
    if model.const_wd:
        irs = axpby([model.icov_scalar] * count, us)
    else:
        weights = [model.icov_field if model.lh_kind == L.LH_GAUSS else pt.lp.wd for pt in points]
        irs = binary(L.OP_MUL, us, weights)
    outs = [torch.empty(model.shape, dtype=g.dtype, device=g.device) for g in grids]
    rowsum((a[3], a[4], a[5]), irs, [o.reshape(-1) for o in outs], weighted=True, lanes=B.lanes_for(nnz, resp.n_pix),
           nrows=resp.n_pix)
    return outs


# ---- sampled KL: value / gradient and metric applications over the local samples -------------------------------------------
def kl_apply_metric(kl, d):
    """FusedKL._apply_metric_on_lanes as ONE launch set: every sample's contribution in a vector of its own (the first one
    also adds the prior term), summed in pair_tree order by one pass."""
    from .engine import LatentVec

    model, lins = kl.model, kl.lins
    n, w = len(lins), 1.0 / kl.n_total
    prior = (1.0 if kl._holds_first else 0.0) if kl._tree else n * w
    outs = [LatentVec(torch.empty_like(d.xi), None) for _ in range(n)]
    for lo in range(0, n, MAX):
        part = range(lo, min(n, lo + MAX))
        scr = Scratch(model, len(part))
        metric(model, scr, [Point.of(lins[i]) for i in part], [d] * len(part), [outs[i] for i in part], w,
               [(d, prior) if i == 0 and prior else None for i in part])
    return _fold(kl, outs)


def _fold(kl, vecs):
    """pair_tree sum of the members' vectors (what FusedKL._fold_vectors does with one addition per merge)"""
    from .engine import LatentVec

    if len(vecs) == 1:
        return vecs[0]
    if len(vecs) > MAX:
        return kl._fold_vectors(vecs)
    return LatentVec(sum_tree([v.xi for v in vecs]), sum_tree([v.small for v in vecs]))


def kl_linearize(kl, position):
    """FusedKL._linearize_on_lanes as one launch set per stage: value and gradient of every local sample, the gradients
    summed in pair_tree order.  Models with a response evaluate the data-space part member by member (a handful of small
    launches each); the transforms, amplitude kernels and bin sums are shared."""
    from .engine import LatentVec, LinPoint

    model = kl.model
    n, w = len(kl.residuals), 1.0 / kl.n_total
    dev = model.device
    sq = scalars(n, dev)
    signs = [-1.0 if neg else 1.0 for neg in kl.negs]
    xs_xi, xs_small = [], []
    for lo in range(0, n, MAX):
        hi = min(n, lo + MAX)
        xs_xi += axpby_sqnorm([1.0] * (hi - lo), [position.xi] * (hi - lo), signs[lo:hi], [r.xi for r in kl.residuals[lo:hi]],
                              sq[lo:hi])
        xs_small += axpby_sqnorm([1.0] * (hi - lo), [position.small] * (hi - lo), signs[lo:hi],
                                 [r.small for r in kl.residuals[lo:hi]], sq[lo:hi], accumulate=True)
    xs = [LatentVec(a, b, s) for a, b, s in zip(xs_xi, xs_small, sq)]
    lins, grads, values = [], [], []
    for lo in range(0, n, MAX):
        part = list(range(lo, min(n, lo + MAX)))
        scr = Scratch(model, len(part))
        lps, gs, vs = _linearize(model, scr, [xs[i] for i in part], w)
        lins += lps
        grads += gs
        values += vs
    kl.lins = lins
    value = sum_tree(values, out=scalars(1, dev)[0]) if len(values) <= MAX else parallel.tree_fold(values)
    return value, _fold(kl, grads)


def _linearize(model, scr, xs, w):
    """engine.FusedModel.linearize for the members: returns (linearisation points, weighted gradients, weighted values)."""
    from .engine import LatentVec, LinPoint

    count = len(xs)
    dev = model.device
    lps = []
    for x in xs:
        lp = LinPoint()
        lp.x = x
        lp.amp = torch.empty(model.nb, dtype=torch.float64, device=dev)
        lp.state = torch.empty(8 * model.nb + 16, dtype=torch.float64, device=dev)
        lps.append(lp)
    _check(L.load().nk_amp_forward_batch(model.nb, model.geo.data_ptr(), model.hyp.data_ptr(), count,
                                         _ptrs([x.small for x in xs]), _ptrs([lp.state for lp in lps]),
                                         _ptrs([lp.amp for lp in lps]), B._stream()), "nk_amp_forward_batch")
    afields = _expand(model, [lp.amp for lp in lps],
                      [torch.empty(model.field_shape, dtype=model.tdtype, device=dev) for _ in range(count)])
    for lp, af in zip(lps, afields):
        lp.afield = af
    zeros = scalars(2 * count, dev)
    values, lhvals = zeros[:count], zeros[count:]
    tmps = scr["tmp"][:count]
    points = [Point.of(lp) for lp in lps]
    if model.response is not None:
        gs, w2s = [], []
        for lp, x, tmp, lhval in zip(lps, xs, tmps, lhvals):  # data-space part member by member (see kl_linearize)
            gs.append(_response_residual(model, lp, x, tmp, lhval))
            w2s.append(lp.gp)
    else:
        fuses = []
        for m, (lp, x) in enumerate(zip(lps, xs)):
            f = model._fuse()
            f.pro, f.in_, f.pidx, f.amp = L.PRO_AMP, x.xi.data_ptr(), model.pidx.data_ptr(), lp.amp.data_ptr()
            f.afield = lp.afield.data_ptr()
            if not model.const_mid:
                lp.mid = torch.empty(model.shape, dtype=model.tdtype, device=dev)
            else:
                lp.mid_scalar = model.icov_scalar
            f.epi, f.out, f.out2 = L.EPI_LIKELIHOOD, tmps[m].data_ptr(), B.ptr(lp.mid)
            f.offset, f.lh_kind, f.nonlin = model.offset_mean, model.lh_kind, model.nonlin
            f.data, f.icov, f.icov_scalar = model.data.data_ptr(), B.ptr(model.icov_field), model.icov_scalar
            f.value = lhvals[m].data_ptr()
            fuses.append(f)
        if model.wide:
            raise RuntimeError("batched evaluation: fp32 models with the wide forward transform run unbatched")
        _transform(model, fuses, scr)
        gs, w2s = tmps, None
    grads = [LatentVec(torch.empty_like(x.xi), None) for x in xs]
    latbars = _vjp(model, scr, points, gs, w2s, w, [g.xi for g in grads], [(x, w) for x in xs])
    smalls = axpby([1.0] * count, latbars, [w] * count, [x.small for x in xs])
    for g, s in zip(grads, smalls):
        g.small = s
    # value: lh + 1/2 x.x, weighted (the same two additions per member as engine.FusedModel._finish_linearize)
    axpby([w] * count, lhvals, [1.0] * count, values, outs=values)
    axpby([0.5 * w] * count, [x.sqnorm for x in xs], [1.0] * count, values, outs=values)
    for lp, v in zip(lps, values):
        lp.value, lp.grad = v, None
    model._count("value_grad", count)
    return lps, grads, values


def _response_residual(model, lp, x, tmp, lhval):
    """engine.FusedModel._linearize_response up to the signal-space residual: fills lp.gp (and lp.wd), the likelihood value
    -> lhval, returns R^T (dE/dmu) on the grid."""
    model._forward_nonlin(lp, x, tmp)
    mu = model.response.times(tmp)
    if model.lh_kind == L.LH_GAUSS:
        r = B.axpby(1.0, mu, -1.0, model.data)
        ir = model._weigh_data(r, lp)
        B.vdot(r, ir, result=lhval, accumulate=False)
        B.axpby(0.5, lhval, out=lhval)
    else:
        dat = model.data.to(model.tdtype)
        B.vsum(mu, result=lhval, accumulate=False)
        ld = B.vdot(B.pointwise("log", mu), dat)
        B.axpby(1.0, lhval, -1.0, ld, out=lhval)
        lp.wd = B.pointwise("reciprocal", mu)
        ir = B.binary(L.OP_SUB, 1.0, B.binary(L.OP_MUL, dat, lp.wd))
    return model.response.adjoint(ir, model.shape)


# ---- the linear sampling solves of an iteration, advanced together ----------------------------------------------------------
class BatchCgScalars:
    """The device scalars of `count` conjugate-gradient solves (engine.CgWorkspace for a batch): row m = that solve's
    double[8] of nk_cg_*."""

    def __init__(self, device, count):
        self.scal = torch.zeros(count, 8, dtype=torch.float64, device=device)
        self._host = torch.empty(count, 8, dtype=torch.float64, pin_memory=True)
        self.rows = [self.scal[m] for m in range(count)]

    def fetch_begin(self):
        self._host.copy_(self.scal, non_blocking=True)
        if getattr(self, "_landed", None) is None:
            self._landed = torch.cuda.Event()
        self._landed.record(torch.cuda.current_stream(self.scal.device))

    def fetch_end(self):
        self._landed.synchronize()
        return self._host.numpy()

    def fetch(self):
        self.fetch_begin()
        return self.fetch_end()


def solve_together(model, lp, jobs, controller_factory, nreset=None):
    """engine._solve_on_lanes as batched launches: the solves (J^T M J + 1) y_j = b_j = s_j + nj_j at ONE linearisation point
    (kl_energies.py:132-158), started at the prior draws s_j, advance together -- one launch set per CG iteration for all
    unfinished solves, one host synchronisation per iteration for all of them.  Per solve the arithmetic, the stopping
    rule and therefore the iterate are those of ConjugateGradient._inplace_steps on its own (same bits)."""
    import numpy as np

    from . import minimization as M
    from .engine import LatentVec

    pairs = []
    for w0 in range(0, len(jobs), MAX):
        wave = jobs[w0:w0 + MAX]
        count = len(wave)
        scr = Scratch(model, count)
        points = [Point.of(lp)] + [Point.of(lp).clone_state() for _ in range(count - 1)]
        ss, njs = [s for s, _ in wave], [nj for _, nj in wave]
        bs = [LatentVec(a, b) for a, b in zip(axpby([1.0] * count, [s.xi for s in ss], [1.0] * count, [nj.xi for nj in njs]),
                                              axpby([1.0] * count, [s.small for s in ss], [1.0] * count,
                                                    [nj.small for nj in njs]))]
        # r0 = J^T M J s - nj (+ 0 * s): the gradient of the quadratic energy at the start (QuadraticEnergy(s, A, b, _grad=g0))
        rs = [LatentVec(torch.empty_like(s.xi), None) for s in ss]
        metric(model, scr, points, ss, rs, 1.0, [(nj, -1.0) for nj in njs])
        xs = ss  # the prior draws belong to the solves: updated in place
        ds = [LatentVec(r.xi.clone(), r.small.clone()) for r in rs]
        ws = BatchCgScalars(model.device, count)
        # gamma = r.r
        vdot([r.xi.reshape(-1) for r in rs], [r.xi.reshape(-1) for r in rs], [row[0:1] for row in ws.rows])
        vdot([r.small for r in rs], [r.small for r in rs], [row[0:1] for row in ws.rows], accumulate=True)
        controllers = []
        for _ in range(count):
            c = controller_factory()
            if any(c is other for other in controllers):
                c = copy.deepcopy(c)
            controllers.append(c)
        gam = ws.fetch()[:, 0].copy()
        # the energy of the start exactly as QuadraticEnergy._compute_value forms it: A s = r0 + b, E = 1/2 s.(A s) - b.s
        # (the controllers read it at start(), whether or not the recurrence tracks it afterwards)
        track = M._config_track_energy()
        e0 = scalars(2 * count, model.device)
        e_ax, e_b = e0[:count], e0[count:]
        a_s = [LatentVec(a, b) for a, b in zip(axpby([1.0] * count, [r.xi for r in rs], [1.0] * count, [b.xi for b in bs]),
                                               axpby([1.0] * count, [r.small for r in rs], [1.0] * count,
                                                     [b.small for b in bs]))]
        vdot([x.xi.reshape(-1) for x in xs], [v.xi.reshape(-1) for v in a_s], e_ax)
        vdot([x.small for x in xs], [v.small for v in a_s], e_ax, accumulate=True)
        vdot([b.xi.reshape(-1) for b in bs], [x.xi.reshape(-1) for x in xs], e_b)
        vdot([b.small for b in bs], [x.small for x in xs], e_b, accumulate=True)
        del a_s
        host = torch.stack(e_ax + e_b).cpu().numpy().reshape(2, count)
        values = [float(0.5 * host[0, m] - host[1, m]) for m in range(count)]
        status = [None] * count
        active = []
        for m in range(count):
            st = controllers[m].start(M._ScalarEnergyView(values[m], float(np.sqrt(max(gam[m], 0.0)))))
            if st != M.CONTINUE:
                status[m] = st
            elif np.isnan(gam[m]):
                status[m] = M.ERROR
            elif gam[m] == 0:
                status[m] = M.CONVERGED
            else:
                active.append(m)
        since_reset, iteration = 0, 0
        while active:
            iteration += 1
            k = len(active)
            pts = [points[m] for m in active]
            d_a, r_a, x_a, b_a = ([v[m] for m in active] for v in (ds, rs, xs, bs))
            rows = [ws.rows[m] for m in active]
            lib, st_ = L.load(), B._stream()
            # (d <- beta d + r of this iteration was enqueued at the end of the previous one, ahead of the host's wait)
            # q = (J^T M J + 1) d with d.q (xi part) taken in the epilogue, like the single solve's fused dot
            qs = [LatentVec(torch.empty_like(v.xi), None) for v in d_a]
            slots = [row[1:2] for row in rows]  # (zero: fresh at first, then cleared by the roll -- roll = 2 below)
            metric(model, Scratch(model, k), pts, d_a, qs, 1.0, [(v, 1.0) for v in d_a], dot_outs=slots)
            _check(lib.nk_cg_curv_batch(d_a[0].small.numel(), k, _ptrs([v.small for v in d_a]), _ptrs([v.small for v in qs]),
                                        L.NK_F64, _ptrs(rows), 1, st_), "nk_cg_curv_batch")
            # (the slots of the update's reductions are zero here: fresh scalars at first, then left so by the roll that
            # follows every update below -- accumulate = 1 for both segments, no zeroing launch)
            for seg in ("xi", "small"):
                xx, rr, dd, qq = ([getattr(v, seg).reshape(-1) for v in vs] for vs in (x_a, r_a, d_a, qs))
                if track:
                    _check(lib.nk_cg_update_dr_batch(xx[0].numel(), k, _ptrs(xx), _ptrs(rr), _ptrs(dd), _ptrs(qq),
                                                     B.dtype_code(xx[0]), _ptrs(rows), 1, st_), "nk_cg_update_dr_batch")
                else:
                    bb = [getattr(v, seg).reshape(-1) for v in b_a]
                    _check(lib.nk_cg_update_batch(xx[0].numel(), k, _ptrs(xx), _ptrs(rr), _ptrs(dd), _ptrs(qq), _ptrs(bb),
                                                  B.dtype_code(xx[0]), _ptrs(rows), 1, st_), "nk_cg_update_batch")
            M.counters.add("cg_iterations", k)
            since_reset += 1
            refreshed = False
            # (ConjugateGradient's nreset) r = A x - b -- unless every solve stops at its iteration limit now anyway
            # (minimization._forced_stop: the residual of a final iterate is never read)
            if since_reset >= (M.CG_NRESET if nreset is None else nreset) and not all(M._forced_stop(controllers[m]) for m in active):
                axs = [LatentVec(torch.empty_like(v.xi), None) for v in x_a]
                metric(model, Scratch(model, k), pts, x_a, axs, 1.0, [(v, 1.0) for v in x_a])
                for r, ax, b in zip(r_a, axs, b_a):
                    B.axpby(1.0, ax.xi, -1.0, b.xi, out=r.xi)
                    B.axpby(1.0, ax.small, -1.0, b.small, out=r.small)
                for m_, (x, r, b) in zip(active, zip(x_a, r_a, b_a)):
                    for slot, (u, v) in ((2, (r, r)), (3, (x, r)), (4, (x, b))):
                        res = ws.rows[m_][slot:slot + 1]
                        B.vdot(u.xi.reshape(-1), v.xi.reshape(-1), result=res, accumulate=False)
                        B.vdot(u.small, v.small, result=res, accumulate=True)
                since_reset, refreshed = 0, True
            ws.fetch_begin()
            # the next iteration's d <- beta d + r (xi, then the small part with the roll of the scalars), enqueued before the
            # host waits for this iteration's scalars: it steers nothing, and a solve that stops now never reads its d again
            for seg, roll in (("xi", 0), ("small", 2)):
                dd = [getattr(v, seg).reshape(-1) for v in d_a]
                rr = [getattr(v, seg).reshape(-1) for v in r_a]
                _check(lib.nk_cg_direction_batch(dd[0].numel(), k, _ptrs(dd), _ptrs(rr), B.dtype_code(dd[0]), _ptrs(rows),
                                                 roll, st_), "nk_cg_direction_batch")
            sc = ws.fetch_end()
            for m in list(active):
                g_prev, curv, gamma = sc[m, 0], sc[m, 1], sc[m, 2]
                alpha = g_prev / curv if curv != 0 else float("nan")
                if np.isnan(curv) or curv == 0.0 or np.isnan(alpha) or alpha < 0:
                    M.logger.error("Error: ConjugateGradient: bad curvature / step")
                    status[m] = M.ERROR
                elif np.isnan(gamma) or gamma < 0:
                    status[m] = M.ERROR
                elif gamma == 0:
                    status[m] = M.CONVERGED
                else:
                    if track and not refreshed:
                        values[m] = values[m] - alpha * sc[m, 3] + 0.5 * alpha * alpha * curv
                    else:
                        values[m] = 0.5 * sc[m, 3] - 0.5 * sc[m, 4]
                    st = controllers[m].check(M._ScalarEnergyView(values[m], float(np.sqrt(gamma))))
                    if st != M.CONTINUE:
                        status[m] = st
                if status[m] is not None:
                    active.remove(m)
        pairs += [(b, x) for b, x in zip(bs, xs)]
    return pairs
