"""CorrelatedFieldMaker / SimpleCorrelatedField: the amplitude model and the correlated-field operator.

Counterpart of reference nifty/cl/library/correlated_fields.py (_SlopeRemover :89-116,
_TwoLogIntegrations :119-162, _Normalization :165-208, _Amplitude :277-386, CorrelatedFieldMaker
:389-859), library/correlated_fields_simple.py:36-133 and operators/normal_operators.py:28-72, for the
single-amplitude, total_N == 0, non-Matern case the BASELINE configs use, and for product spectra.

Two realisations share one interface:
* the GENERIC operator graph built from the small linear operators (host Fields, config 1's
  "nifty.cl numpy CPU" plumbing case and the structural cross-check of the fused path), and
* ``CorrelatedFieldOperator``: ONE fused Operator node whose forward pass, Jacobian and adjoint
  Jacobian run in the HIP kernels of libniftyk (amplitude kernels + fused Hartley transform) for
  device Fields.  ``finalize()`` returns this node; on host Fields it evaluates the generic graph.
* several spectra (product spaces) and / or ``total_N > 0``: ``ProductCorrelatedFieldOperator`` -- the
  amplitude graphs (standard or Matern, shared by ``dofdex``) feed ``_ProductFieldNode``, which does the
  N-sized half in nk_product_field / nk_hartley_fused / nk_mirror_combine / nk_product_marginal.
"""
import ctypes
import functools
import os
import threading

import numpy as np
import torch

from . import _lib as L
from . import backend as B
from .domains import DomainTuple, MultiDomain, PowerSpace, RGSpace, UnstructuredDomain, makeDomain
from .engine import SMALL_KEYS, lognormal_moments
from .field import Field, MultiField, full, makeField
from .minimization import logger
from .operators import (ContractionOperator, DiagonalOperator, EndomorphicOperator, HarmonicTransformOperator, Linearization,
                        LinearOperator, Operator, PowerDistributor, Variable, VdotOperator, ducktape, is_linearization, makeOp)


# ------------------------------------------------------------------------------------------------
# hyper-parameter transforms (normal_operators.py:28-72)
# ------------------------------------------------------------------------------------------------
def _per_copy(param, n):
    """A scalar hyper-parameter, or one value per copy (normal_operators.py / utilities.value_reshaper)."""
    arr = np.asarray(param, dtype=np.float64)
    if arr.shape in ((), (1,)):
        return np.full(n, float(arr.reshape(-1)[0]))
    if arr.shape != (n,):
        raise TypeError(f"hyper-parameter of shape {arr.shape} does not match the {n} copies")
    return arr


def NormalTransform(mean, sigma, key, N_copies=0):
    """mean + sigma * xi_key; N_copies >= 1: one value per copy on an UnstructuredDomain(N_copies)."""
    if N_copies == 0:
        domain = DomainTuple.scalar_domain()
        return float(sigma) * ducktape(domain, None, key) + float(mean)
    domain = DomainTuple.make(UnstructuredDomain(N_copies))
    mean_f, sigma_f = (makeField(domain, _per_copy(p, N_copies)) for p in (mean, sigma))
    return (makeOp(sigma_f) @ ducktape(domain, None, key)) + mean_f


def LognormalTransform(mean, sigma, key, N_copies=0):
    if N_copies == 0:
        logmean, logsigma = lognormal_moments(mean, sigma)
    else:
        pairs = [lognormal_moments(m, sg) for m, sg in zip(_per_copy(mean, N_copies), _per_copy(sigma, N_copies))]
        logmean, logsigma = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    return NormalTransform(logmean, logsigma, key, N_copies).ptw("exp")


class _Distributor(LinearOperator):
    """Copies of a few amplitude / zero-mode models onto the total_N fields: out[i] = in[dofdex[i]] along the leading
    axis (reference correlated_fields.py:211-231)."""

    def __init__(self, dofdex, domain, target):
        self._dofdex = torch.as_tensor(np.asarray(dofdex), dtype=torch.int64)
        self._domain, self._target = DomainTuple.make(domain), DomainTuple.make(target)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        v = x.val
        idx = self._dofdex.tolist()
        if mode == self.TIMES:
            return Field(self._target, torch.stack([v[i] for i in idx]).contiguous())
        rows = [None] * self._domain.shape[0]
        for j, i in enumerate(idx):
            rows[i] = v[j] if rows[i] is None else rows[i] + v[j] if not v.is_cuda else B.axpby(1.0, rows[i], 1.0, v[j].contiguous())
        zero = torch.zeros_like(v[0])
        return Field(self._domain, torch.stack([zero if r is None else r for r in rows]).contiguous())


# ------------------------------------------------------------------------------------------------
# generic amplitude building blocks (host)
# ------------------------------------------------------------------------------------------------
def _log_k_lengths(pspace):
    return np.log(pspace.k_lengths[1:])


def _relative_log_k_lengths(pspace):
    logkl = _log_k_lengths(pspace)
    return np.insert(logkl - logkl[0], 0, 0.0)


def _log_vol(pspace):
    logkl = _log_k_lengths(pspace)
    return logkl[1:] - logkl[:-1]


def _dev(t, like):
    """Static operator array on the device / dtype of the field it meets (migrated lazily like the reference's
    `_device_preparation`)."""
    return t.to(device=like.device, dtype=like.dtype)


def _batched(domain, space, cls_name):
    """(leading copies, PowerSpace) domains: `space` must be the last space; returns the number of leading rows (0: none)."""
    domain = makeDomain(domain)
    if space != len(domain) - 1 or len(domain) > 2 or not isinstance(domain[space], PowerSpace):
        raise NotImplementedError(f"{cls_name}: a PowerSpace as the last of at most two spaces is supported")
    return domain, (domain.shape[0] if len(domain) == 2 else 0)


def _rowwise(v, nrows, fn):
    """fn on every leading row of v (nrows == 0: on v itself); slices and stacking are copies."""
    if nrows == 0:
        return fn(v)
    return torch.stack([fn(v[i].contiguous()) for i in range(nrows)]).contiguous()


class _SlopeRemover(EndomorphicOperator):
    def __init__(self, domain, space=0):
        self._domain, self._nrows = _batched(domain, space, "_SlopeRemover")
        logkl = _relative_log_k_lengths(self._domain[space])
        self._sc = torch.from_numpy(logkl / float(logkl[-1]))
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        return Field(self._tgt(mode), _rowwise(x.val, self._nrows, lambda v: self._apply1(v, mode)))

    def _apply1(self, v, mode):
        return self._apply_row(Field(DomainTuple.make(self._domain[-1]), v), mode).val

    def _apply_row(self, x, mode):
        v = x.val
        if v.is_cuda:  # device Fields: libniftyk element-wise kernels, the two scalars cross the host
            sc = _dev(self._sc, v)
            v = v.contiguous()
            if mode == self.TIMES:
                return Field(x.domain, B.axpby(1.0, v, -float(v[-1].item()), sc))
            res = v.clone()
            last = res[-1:]
            B.axpby(1.0, last, -float(B.vdot(v, sc).item()), torch.ones_like(last), out=last)
            return Field(x.domain, res)
        if mode == self.TIMES:
            return Field(x.domain, v - v[-1] * self._sc)
        res = v.clone()
        res[-1] = res[-1] - (v * self._sc).sum()
        return Field(x.domain, res)


class _TwoLogIntegrations(LinearOperator):
    def __init__(self, target, space=0):
        self._target, self._nrows = _batched(target, space, "_TwoLogIntegrations")
        bins = self._target[space]
        excitations = UnstructuredDomain((2, bins.shape[0] - 2))  # two rows of white noise per log-k interval
        self._capability = self.TIMES | self.ADJOINT_TIMES
        self._log_vol = torch.from_numpy(_log_vol(bins))
        self._row_target, self._row_domain = DomainTuple.make(bins), DomainTuple.make(excitations)
        self._domain = makeDomain([excitations if i == space else sub for i, sub in enumerate(self._target)])

    def apply(self, x, mode):
        self._check_input(x, mode)
        rdom = self._row_domain if mode == self.TIMES else self._row_target
        return Field(self._tgt(mode), _rowwise(x.val, self._nrows, lambda v: self._apply_row(Field(rdom, v), mode).val))

    def _apply_row(self, x, mode):
        lv = self._log_vol
        if x.val.is_cuda:
            return self._apply_device(x, mode)
        if mode == self.TIMES:
            v = x.val
            res = torch.zeros(self._row_target.shape, dtype=v.dtype)
            c = torch.cumsum(v[1], 0)
            cprev = torch.cat([torch.zeros(1, dtype=v.dtype), c[:-1]])
            res[2:] = torch.cumsum((c + cprev) / 2 * lv + v[0], 0)
            return Field(self._row_target, res)
        y = x.val
        res = torch.zeros(self._row_domain.shape, dtype=y.dtype)
        t = torch.flip(torch.cumsum(torch.flip(y[2:], [0]), 0), [0])
        res[0] = t
        u = t * (lv / 2.0)
        gc = u.clone()
        gc[:-1] += u[1:]
        res[1] = torch.flip(torch.cumsum(torch.flip(gc, [0]), 0), [0])
        return Field(self._row_domain, res)


    def _apply_device(self, x, mode):
        """Same arithmetic with nk_cumsum and the element-wise kernels; slicing / concatenation are copies."""
        v = x.val.contiguous()
        half_lv = B.axpby(0.5, _dev(self._log_vol, v).contiguous())
        zero1 = torch.zeros(1, dtype=v.dtype, device=v.device)
        if mode == self.TIMES:
            res = torch.zeros(self._row_target.shape, dtype=v.dtype, device=v.device)
            c = B.cumsum(v[1])
            both = B.binary(L.OP_ADD, c, torch.cat([zero1, c[:-1]]).contiguous())  # c_j + c_{j-1}
            inc = B.binary(L.OP_ADD, B.binary(L.OP_MUL, both, half_lv), v[0].contiguous())
            B.cumsum(inc, out=res[2:])
            return Field(self._row_target, res)
        res = torch.zeros(self._row_domain.shape, dtype=v.dtype, device=v.device)
        t = B.cumsum(v[2:], reverse=True, out=res[0])
        u = B.binary(L.OP_MUL, t, half_lv)
        gc = B.binary(L.OP_ADD, u, torch.cat([u[1:], zero1]).contiguous())
        B.cumsum(gc, reverse=True, out=res[1])
        return Field(self._row_domain, res)


def _volume_slots(nbins, zero_mode, other_modes):
    """(factor per bin with 0 in the zero-mode slot, the zero-mode slot alone): how the volume enters an amplitude"""
    factors, slot = np.full(nbins, float(other_modes)), np.zeros(nbins)
    factors[0], slot[0] = 0.0, zero_mode
    return factors, slot


def _on_space(values, domain, space):
    """Diagonal operator on `domain` whose diagonal `values` lives on the sub-space `space` only."""
    return DiagonalOperator(makeField(domain[space], values), domain, space)


class _Normalization(Operator):
    """log-spectrum p -> sqrt( e^p / sum_{k != 0} mult(k) e^p ): unit total power outside the zero mode, per copy
    (reference correlated_fields.py:165-208)."""

    def __init__(self, domain, space=0):
        self._domain = self._target = DomainTuple.make(domain)
        _batched(self._domain, space, "_Normalization")
        multiplicity = self._domain[space].rho.astype(np.float64)
        multiplicity[0] = 0.0  # the zero mode does not count
        self._weigh = _on_space(multiplicity, self._domain, space)
        self._total = ContractionOperator(self._domain, space)

    def apply(self, x):
        self._check_input(x)
        power = x.exp()
        norm = self._total.adjoint(self._total(self._weigh(power)))  # weighted sum over the bins, broadcast back
        return (power * norm.reciprocal()).sqrt()


class _SpectrumLayout:
    """Where the spectra of an amplitude model live: `target` = ([copies,] PowerSpace) with the bins at `space`; when
    several fields share fewer spectra (dofdex) `distribute` copies the spectra onto the fields."""

    def __init__(self, power_space, dofdex):
        self.dofdex = [int(i) for i in dofdex]
        self.distribute = self.field_copies = None
        if not self.dofdex:
            self.target, self.space = makeDomain(power_space), 0
        else:
            spectra = max(self.dofdex) + 1
            self.target, self.space = makeDomain((UnstructuredDomain(spectra), power_space)), 1
            if spectra != len(self.dofdex):
                self.field_copies = UnstructuredDomain(len(self.dofdex))
                self.distribute = _Distributor(self.dofdex, self.target, makeDomain((self.field_copies, power_space)))
        self.bins = self.target[self.space]
        if not isinstance(self.bins, PowerSpace):
            raise TypeError("PowerSpace required")

    def onto_bins(self, per_copy_operator):
        """a per-copy (or scalar) hyper-parameter broadcast over the bins"""
        return ContractionOperator(self.target, self.space).adjoint @ per_copy_operator


class _Amplitude(Operator):
    """a(k) = V zeromode-slot + V fluct * normalised( slope * rel_logk + SlopeRemover(TwoLogIntegrations(sigma xi_s)) )
    (reference correlated_fields.py:277-386).  The log-spectrum is assembled from its two parts -- the power law and
    the integrated-Wiener-process deviation, which may be switched off -- then normalised, scaled by the fluctuation
    amplitude and the volume, and copied onto the fields when spectra are shared (dofdex, total_N > 0)."""

    def __init__(self, target, fluctuations, flexibility, asperity, loglogavgslope, totvol, key, dofdex=()):
        if flexibility is None and asperity is not None:
            raise ValueError("flexibility may not be disabled on its own")
        lay = _SpectrumLayout(target, dofdex)
        log_spectrum = _on_space(_relative_log_k_lengths(lay.bins), lay.target, lay.space) @ lay.onto_bins(loglogavgslope)
        if flexibility is not None:
            log_spectrum = log_spectrum + self._wiener_deviation(lay, flexibility, asperity, key)
        shape = _Normalization(lay.target, lay.space) @ log_spectrum
        # volume factors: the zero-mode slot carries V, every other mode V * fluctuations
        nonzero, zero = _volume_slots(lay.bins.shape, totvol, totvol)
        strength = _on_space(nonzero, lay.target, lay.space) @ lay.onto_bins(fluctuations)
        offset = _on_space(zero, lay.target, lay.space)(full(lay.target, 1.0))
        if lay.distribute is None:
            self._op, self._fluc = strength * shape + offset, fluctuations
        else:
            self._op = (lay.distribute @ strength) * (lay.distribute @ shape) + lay.distribute(offset)
            self._fluc = _Distributor(lay.dofdex, fluctuations.target, lay.field_copies) @ fluctuations
        self._domain, self._target, self._space = self._op.domain, self._op.target, lay.space

    @staticmethod
    def _wiener_deviation(lay, flexibility, asperity, key):
        """SlopeRemover(TwoLogIntegrations(sigma * xi_s)): the deviation of the log-spectrum from the power law.  Per
        log-k interval of length l the two excitation rows get sigma = flex sqrt(l) (sqrt(asp + l^2/12), 1); without
        asperity the first factor is sqrt(l^2/12) (reference correlated_fields.py:315-326, 351-363)."""
        integrate = _TwoLogIntegrations(lay.target, lay.space)
        dom = integrate.domain
        rows = dom[lay.space].shape
        interval = _log_vol(lay.bins)
        spread = ContractionOperator(dom, lay.space).adjoint  # hyper-parameter -> both excitation rows
        root_l, first_row, drift = np.zeros(rows), np.zeros(rows), np.ones(rows)
        root_l[:] = np.sqrt(interval)
        first_row[0], drift[0] = 1.0, interval ** 2 / 12.0
        sigma = _on_space(root_l, dom, lay.space) @ spread @ flexibility
        drift = _on_space(drift, dom, lay.space)(full(dom, 1.0))
        if asperity is None:
            sigma = DiagonalOperator(drift.sqrt()) @ sigma
        else:
            sigma = sigma * ((_on_space(first_row, dom, lay.space) @ spread @ asperity) + drift).sqrt()
        return _SlopeRemover(lay.target, lay.space) @ integrate @ (sigma * Variable(dom, key))

    def apply(self, x):
        self._check_input(x)
        return self._op(x)

    @property
    def fluctuation_amplitude(self):
        return self._fluc


class _AmplitudeMatern(Operator):
    """Parametric amplitude a(k) = scale * (1 + (k/cutoff)^2)^(loglogslope/4), with the position-space volume folded
    in (zero mode ~ V, other modes ~ sqrt(V)) (reference library/correlated_fields.py:231-275)."""

    def __init__(self, pow_spc, scale, cutoff, loglogslope, totvol):
        onto_bins = ContractionOperator(pow_spc, None).adjoint
        k2 = makeField(makeDomain(pow_spc), pow_spc.k_lengths ** 2)
        # log a = log scale + slope/4 * log(1 + k^2 / cutoff^2), evaluated in log space and exponentiated
        ratio = (VdotOperator(k2).adjoint @ cutoff.ptw("power", -2.0)) + 1.0
        log_a = (onto_bins.scale(0.25) @ loglogslope) * ratio.ptw("log") + (onto_bins @ scale.ptw("log"))
        volume, zero_slot = (makeField(log_a.target, v) for v in _volume_slots(pow_spc.shape, totvol, totvol ** 0.5))
        self._op = DiagonalOperator(volume)(log_a.ptw("exp")) + zero_slot
        self._domain, self._target = self._op.domain, self._op.target

    def apply(self, x):
        self._check_input(x)
        return self._op(x)


# ------------------------------------------------------------------------------------------------
# fused device operator
# ------------------------------------------------------------------------------------------------
class _CFJacobian(LinearOperator):
    """Jacobian of the fused correlated field at one latent point (TIMES = JVP, ADJOINT_TIMES = VJP)."""

    def __init__(self, parent, x_small, xi, amp, state):
        self._p = parent
        self._domain, self._target = parent.domain, parent.target
        self._small, self._xi, self._amp, self._state = x_small, xi, amp, state
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        p = self._p
        dev = p._devdata(self._xi.device)
        lib = L.load()
        if mode == self.TIMES:
            dsmall = p._pack_small(x)
            damp = torch.empty(p._nb, dtype=torch.float64, device=dsmall.device)
            L.check(lib.nk_amp_jvp(p._nb, dev["geo"].data_ptr(), dev["hyp"].data_ptr(), self._small.data_ptr(),
                                   self._state.data_ptr(), dsmall.data_ptr(), damp.data_ptr(), B._stream()), "nk_amp_jvp")
            dxi = x[p._prefix + "xi"].val.contiguous()
            out = torch.empty_like(dxi)
            f = p._fuse()
            f.pro, f.in_, f.in2 = L.PRO_AMP_JVP, dxi.data_ptr(), self._xi.data_ptr()
            f.pidx, f.amp, f.damp = dev["pidx"].data_ptr(), self._amp.data_ptr(), damp.data_ptr()
            f.epi, f.out, f.offset = L.EPI_AFFINE, out.data_ptr(), 0.0
            plan = p._plan(dxi)
            fields = p._octant_fields(plan, dxi, dev, self._amp, damp)  # (kept alive until the launch is enqueued)
            if fields is not None:
                f.afield, f.dafield, f.field_octant = fields[0].data_ptr(), fields[1].data_ptr(), 1
            B.hartley_fused(plan, f)
            return Field(self._target, out)
        w = x.val.contiguous()
        out = torch.empty_like(w)
        abar = torch.zeros(p._nb, dtype=torch.float64, device=w.device)
        f = p._fuse()
        f.pro, f.in_ = L.PRO_PLAIN, w.data_ptr()
        f.epi, f.out = L.EPI_VJP, out.data_ptr()
        f.pidx, f.amp, f.xi, f.abar = dev["pidx"].data_ptr(), self._amp.data_ptr(), self._xi.data_ptr(), abar.data_ptr()
        plan = p._plan(w)
        fields = p._octant_fields(plan, w, dev, self._amp)
        if fields is not None:
            # the final pass multiplies by the octant field and stores ONE merged sum xi . t per octant point (all sign-flip
            # images of a coefficient share their bin); nk_octant_scatter reduces those into the bins -- instead of one fp64
            # atomic per grid point through the bin index
            shape = tuple(int(n) for n in w.shape)
            w8 = torch.empty(int(np.prod([n // 2 + 1 for n in shape])), dtype=torch.float64, device=w.device)
            f.afield, f.field_octant, f.w8 = fields[0].data_ptr(), 1, w8.data_ptr()
            B.hartley_fused(plan, f)
            shp = (ctypes.c_int64 * len(shape))(*shape)
            L.check(lib.nk_octant_scatter(len(shape), shp, w8.data_ptr(), dev["pidx"].data_ptr(), abar.data_ptr(), 0,
                                          B._stream()), "nk_octant_scatter")
        else:
            B.hartley_fused(plan, f)
        latbar = torch.empty(p._nsmall, dtype=torch.float64, device=w.device)
        L.check(lib.nk_amp_vjp(p._nb, dev["geo"].data_ptr(), dev["hyp"].data_ptr(), self._small.data_ptr(),
                               self._state.data_ptr(), abar.data_ptr(), latbar.data_ptr(), B._stream()), "nk_amp_vjp")
        return p._unpack_small(latbar, out, w.dtype)


class CorrelatedFieldOperator(Operator):
    """offset + HT( a[pindex] * xi ): one fused node (reference graph: correlated_fields.py:713-764)."""

    def __init__(self, target_space, generic_op, prefix, offset_mean, hyper):
        self._generic = generic_op
        self._domain, self._target = generic_op.domain, generic_op.target
        self._prefix = prefix
        self._pos = target_space
        self._hsp = target_space.get_default_codomain()
        self._ps = PowerSpace(self._hsp)
        self._nb = self._ps.shape[0]
        self._nsmall = 5 + 2 * (self._nb - 2)
        self._offset = float(offset_mean)
        self._hyper = hyper
        self._dev = {}

    # -- parameters of the closed form, in the layout nk_amp_* expect --------------------------------
    def _devdata(self, device):
        key = str(device)
        if key not in self._dev:
            ps = self._ps
            logk = np.log(ps.k_lengths[1:])
            rel = np.insert(logk - logk[0], 0, 0.0)
            delta = np.concatenate([logk[1:] - logk[:-1], [0.0, 0.0]])
            mult = ps.rho.astype(np.float64).copy()
            mult[0] = 0.0
            geo = np.concatenate([rel, rel / rel[-1], mult, delta])
            h = self._hyper
            hyp = np.array([*lognormal_moments(*h["fluctuations"]), *lognormal_moments(*h["flexibility"]),
                            *lognormal_moments(*h["asperity"]), *lognormal_moments(*h["offset_std"]),
                            float(h["loglogavgslope"][0]), float(h["loglogavgslope"][1]), self._pos.total_volume])
            self._dev[key] = dict(geo=torch.from_numpy(geo).to(device), hyp=torch.from_numpy(hyp).to(device),
                                  pidx=ps.device_pindex(device))
        return self._dev[key]

    def _plan(self, tensor):
        return B.get_plan(self._pos.shape, tensor.dtype, 1, tensor.device)

    def _fuse(self):
        f = L.Fuse()
        f.scale, f.mul_scalar, f.addend_scale = self._hsp.scalar_dvol, 1.0, 1.0
        return f

    def _octant_fields(self, plan, like, dev, *tables):
        """The bin tables (a, da, ...) as OCTANT fields in the field type of `like`, streamed by the first pass (its compile-time
        prologue classes; the arithmetic of the fused engine's transforms) instead of table gathers per grid point through the
        4-byte bin index (the run-time class: 9.1 ms instead of 2.3 ms for the first pass at 1024^3 fp32 -- `minisanity`
        evaluates the model once per sample and iteration through this node).  None where the plan has no octant pipeline
        (mixed radix, short axes) or with NK_CF_OCTANT_FORWARD=0."""
        if not L.load().nk_plan_octant_vjp(plan.handle) or os.environ.get("NK_CF_OCTANT_FORWARD", "1") == "0":
            return None
        shape = tuple(int(n) for n in like.shape)
        shp = (ctypes.c_int64 * len(shape))(*shape)
        out = []
        for table in tables:
            t = table if like.dtype == torch.float64 else table.to(like.dtype)
            field = torch.empty(int(np.prod([n // 2 + 1 for n in shape])), dtype=like.dtype, device=like.device)
            L.check(L.load().nk_octant_expand(len(shape), shp, t.data_ptr(), dev["pidx"].data_ptr(), field.data_ptr(),
                                              B.dtype_code(field), 1, B._stream()), "nk_octant_expand")
            out.append(field)
        return out

    def _pack_small(self, x):
        pre = self._prefix
        parts = [x[pre + k].val.reshape(1).to(torch.float64) for k in SMALL_KEYS]
        parts.append(x[pre + "spectrum"].val.reshape(-1).to(torch.float64))
        return torch.cat(parts).contiguous()  # concatenation = data movement only

    def _unpack_small(self, latbar, xi_bar, dtype):
        pre = self._prefix
        dom = self._domain
        vals = {}
        for i, k in enumerate(SMALL_KEYS):
            vals[pre + k] = Field(dom[pre + k], latbar[i].reshape(()).to(dtype))
        vals[pre + "spectrum"] = Field(dom[pre + "spectrum"], latbar[5:].reshape(2, -1).to(dtype))
        vals[pre + "xi"] = Field(dom[pre + "xi"], xi_bar)
        return MultiField.from_dict(vals, dom)

    def apply(self, x):
        self._check_input(x)
        lin = is_linearization(x)
        v = x.val if lin else x
        if v.device_id < 0:
            return self._generic(x)
        xi = v[self._prefix + "xi"].val.contiguous()
        if not B.plan_supported(self._pos.shape, xi.dtype, 1, xi.device):
            return self._generic(x)  # grid the native planner rejects (e.g. a prime axis): generic graph, chirp-z seam
        dev = self._devdata(xi.device)
        small = self._pack_small(v)
        amp = torch.empty(self._nb, dtype=torch.float64, device=xi.device)
        state = torch.empty(8 * self._nb + 16, dtype=torch.float64, device=xi.device)
        L.check(L.load().nk_amp_forward(self._nb, dev["geo"].data_ptr(), dev["hyp"].data_ptr(), small.data_ptr(),
                                        state.data_ptr(), amp.data_ptr(), B._stream()), "nk_amp_forward")
        out = torch.empty_like(xi)
        f = self._fuse()
        f.pro, f.in_, f.pidx, f.amp = L.PRO_AMP, xi.data_ptr(), dev["pidx"].data_ptr(), amp.data_ptr()
        f.epi, f.out, f.offset = L.EPI_AFFINE, out.data_ptr(), self._offset
        plan = self._plan(xi)
        fields = self._octant_fields(plan, xi, dev, amp)  # (kept alive until the launch is enqueued)
        if fields is not None:
            f.afield, f.field_octant = fields[0].data_ptr(), 1
        B.hartley_fused(plan, f)
        val = Field(self._target, out)
        if not lin:
            return val
        return x.new(val, _CFJacobian(self, small, xi, amp, state))

    @property
    def fused_parameters(self):
        """What the fusion pass of optimize_kl needs to build a FusedModel for this operator."""
        return dict(shape=self._pos.shape, distances=self._pos.distances, offset_mean=self._offset, prefix=self._prefix,
                    **self._hyper)

    def __repr__(self):
        return f"CorrelatedFieldOperator(shape={self._pos.shape}, nb={self._nb})"


class _ProductFieldNode(Operator):
    """{xi, the normalised amplitude tables of the sub-spaces, the zero-mode amplitude} -> offset + HT(azm prod_i
    a_i[pindex_i] xi): the big-field half of CorrelatedFieldMaker.finalize for SEVERAL spectra and / or total_N > 0
    (reference correlated_fields.py:713-764) as one fused node.  The amplitude models themselves -- a few numbers per power
    bin, standard or Matern, shared between fields by dofdex -- stay operator graphs on their small domains and feed this
    node their tables; everything that touches the grid happens here, per field copy:
      value     nk_product_field builds azm prod_i a_i[pindex_i] straight from the tables, the transform multiplies it in
                its prologue (AMP with `afield`) and adds the offset in its epilogue;
      TIMES     the first-order variation of the amplitude field (nk_product_field, tangent) enters the AMP_JVP prologue;
      ADJOINT   the VJP epilogue returns a . t and the per-point sums xi . t (nk_fuse.wfull, or the octant sums w8 on plans
                with the register-resident pipeline); nk_product_marginal contracts them with the other sub-spaces'
                amplitudes and nk_csr_rowsum adds them bin by bin, all in a fixed order.
    The harmonic transform of the product domain is one Hartley transform PER SUB-SPACE (a chain of
    HarmonicTransformOperators with `space=`, reference :726-730) -- the kernel prod_i cas(k_i x_i), not the genuine N-D
    Hartley kernel cas(sum_i k_i x_i).  cas(a) cas(b) = 1/2 [cas(a+b) + cas(a-b) + cas(-a+b) - cas(-a-b)] (three factors:
    1/2 [cas(-a+b+c) + cas(a-b+c) + cas(a+b-c) - cas(-a-b-c)]) turns it into ONE fused N-D transform plus a fixed
    combination of mirrored points (nk_mirror_combine) -- behind the transform in value / TIMES, in front of it in ADJOINT --
    instead of per-sub-space transforms with permutation copies around them.
    The reference distributes every amplitude over the full grid and multiplies fields (one PowerDistributor +
    ContractionOperator.adjoint per sub-space, six N-sized products per evaluation for two spectra)."""

    MIRROR = {2: (0.5, 0.5, 0.5, -0.5), 3: (0.0, 0.5, 0.5, 0.0, 0.5, 0.0, 0.0, -0.5)}  # index: bit i = sub-space i mirrored

    def __init__(self, hspace, target, amp_targets, azm_target, total_N, offset):
        self._hspace, self._target = makeDomain(hspace), makeDomain(target)
        self._copies = int(total_N)
        self._lead = 1 if self._copies > 0 else 0
        dom = {"xi": self._hspace, "azm": makeDomain(azm_target)}
        dom.update({f"a{i}": makeDomain(t) for i, t in enumerate(amp_targets)})
        self._domain = MultiDomain.make(dom)
        self._nsub = len(amp_targets)
        self._subspaces = [self._hspace[self._lead + i] for i in range(self._nsub)]
        self._bins = [makeDomain(t)[self._lead] for t in amp_targets]
        self._grid = tuple(n for sp in self._subspaces for n in sp.shape)
        self._offset = float(offset)
        self._scale = float(np.prod([sp.scalar_dvol for sp in self._subspaces]))
        self._dev = {}
        self.calls = {"value": 0, "times": 0, "adjoint": 0}  # (tests check that this node, not the generic graph, ran)

    # -- what the kernels need, per device and dtype ------------------------------------------------------------------------
    def supported(self, dtype, device):
        return 1 <= self._nsub <= 3 and len(self._grid) <= 3 and B.plan_supported(self._grid, dtype, 1, device)

    def _setup(self, dtype, device):
        key = (str(device), dtype)
        if key in self._dev:
            return self._dev[key]
        plan = B.get_plan(self._grid, dtype, 1, device)
        octant = bool(L.load().nk_plan_octant_vjp(plan.handle))
        pidx, sizes, plans = [], [], []
        for sp, bins in zip(self._subspaces, self._bins):
            full = bins.device_pindex(device).reshape(sp.shape)
            part = full[tuple(slice(0, n // 2 + 1) for n in sp.shape)].contiguous() if octant else full
            pidx.append(part.reshape(-1))
            sizes.append(part.numel())
            plans.append(B.bin_plan(pidx[-1], bins.shape[0]))
        npts = int(np.prod(sizes))
        st = dict(plan=plan, octant=octant, pidx=pidx, sizes=sizes, bin_plans=plans, npts=npts,
                  # operands the C ABI insists on although the amplitude arrives as a field
                  no_pidx=torch.zeros(int(np.prod(self._grid)), dtype=torch.int32, device=device),
                  one=torch.ones(1, dtype=torch.float64, device=device), sink=torch.zeros(1, dtype=torch.float64, device=device))
        self._dev[key] = st
        return st

    def _product(self, st, tabs, azm, dtabs=None, dazm=None):
        q = L.Product()
        q.nsub = self._nsub
        for i in range(self._nsub):
            q.size[i], q.pidx[i], q.tab[i] = st["sizes"][i], st["pidx"][i].data_ptr(), tabs[i].data_ptr()
            q.dtab[i] = None if dtabs is None else dtabs[i].data_ptr()
        q.scale = azm.data_ptr()
        q.dscale = None if dazm is None else dazm.data_ptr()
        return q

    def _field(self, st, dtype, q, tangent):
        out = torch.empty(st["npts"], dtype=dtype, device=st["no_pidx"].device)
        L.check(L.load().nk_product_field(ctypes.byref(q), 1 if tangent else 0, out.data_ptr(), B.dtype_code(out), B._stream()),
                "nk_product_field")
        return out

    def _fuse(self, st, field):
        f = L.Fuse()
        f.scale, f.mul_scalar, f.addend_scale = self._scale, 1.0, 1.0
        f.pidx, f.amp, f.damp = st["no_pidx"].data_ptr(), st["one"].data_ptr(), st["one"].data_ptr()
        f.afield, f.field_octant = field.data_ptr(), 1 if st["octant"] else 0
        return f

    def _separable(self, src, dst, offset=0.0):
        """dst <- the per-sub-space Hartley combination of src (both one field copy, flat) + offset (two or three
        sub-spaces; with one the genuine transform is the answer)."""
        shape = (ctypes.c_int64 * len(self._grid))(*self._grid)
        group = (ctypes.c_int * len(self._grid))(*[i for i, sp in enumerate(self._subspaces) for _ in sp.shape])
        coef = (ctypes.c_double * (1 << self._nsub))(*self.MIRROR[self._nsub])
        L.check(L.load().nk_mirror_combine(len(self._grid), shape, group, self._nsub, coef, src.data_ptr(), dst.data_ptr(), 1.0,
                                           float(offset), B.dtype_code(src), B._stream()), "nk_mirror_combine")
        return dst

    def _rows(self, field):
        """[copies][...] view of a Field's values (one row when the maker has no leading copy domain)."""
        return field.val.contiguous().reshape(max(self._copies, 1), -1)

    def _inputs(self, v):
        xi = self._rows(v["xi"])
        tabs = [self._rows(v[f"a{i}"]).to(torch.float64) for i in range(self._nsub)]
        azm = self._rows(v["azm"]).to(torch.float64)
        return xi, tabs, azm

    def apply(self, x):
        self._check_input(x)
        lin = is_linearization(x)
        v = x.val if lin else x
        xi, tabs, azm = self._inputs(v)
        st = self._setup(xi.dtype, xi.device)
        out = torch.empty_like(xi)
        single = self._nsub == 1
        tmp = None if single else torch.empty_like(xi[0])
        fields = []
        for c in range(xi.shape[0]):
            q = self._product(st, [t[c] for t in tabs], azm[c])
            fields.append(self._field(st, xi.dtype, q, False))
            f = self._fuse(st, fields[-1])
            f.pro, f.in_ = L.PRO_AMP, xi[c].data_ptr()
            f.epi, f.out, f.offset = L.EPI_AFFINE, (out[c] if single else tmp).data_ptr(), self._offset if single else 0.0
            B.hartley_fused(st["plan"], f)
            if not single:
                self._separable(tmp, out[c], self._offset)
        self.calls["value"] += 1
        val = Field(self._target, out.reshape(self._target.shape))
        return x.new(val, _ProductFieldJacobian(self, st, xi, tabs, azm, fields)) if lin else val

    def __repr__(self):
        return f"_ProductFieldNode(grid={self._grid}, copies={self._copies}, spectra={self._nsub})"


class _ProductFieldJacobian(LinearOperator):
    """Jacobian of `_ProductFieldNode` at one point (TIMES = JVP, ADJOINT_TIMES = VJP)."""

    def __init__(self, node, st, xi, tabs, azm, fields):
        self._n, self._st = node, st
        self._domain, self._target = node.domain, node.target
        self._xi, self._tabs, self._azm, self._fields = xi, tabs, azm, fields
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        return self._times(x) if mode == self.TIMES else self._adjoint(x)

    def _times(self, x):
        n, st = self._n, self._st
        dxi, dtabs, dazm = n._inputs(x)
        dxi = dxi.to(self._xi.dtype)
        out = torch.empty_like(self._xi)
        single = n._nsub == 1
        tmp = None if single else torch.empty_like(self._xi[0])
        for c in range(self._xi.shape[0]):
            q = n._product(st, [t[c] for t in self._tabs], self._azm[c], [t[c] for t in dtabs], dazm[c])
            dfield = n._field(st, self._xi.dtype, q, True)
            f = n._fuse(st, self._fields[c])
            f.pro, f.in_, f.in2, f.dafield = L.PRO_AMP_JVP, dxi[c].data_ptr(), self._xi[c].data_ptr(), dfield.data_ptr()
            f.epi, f.out, f.offset = L.EPI_AFFINE, (out[c] if single else tmp).data_ptr(), 0.0
            B.hartley_fused(st["plan"], f)
            if not single:
                n._separable(tmp, out[c])
        n.calls["times"] += 1
        return Field(self._target, out.reshape(self._target.shape))

    def _adjoint(self, x):
        n, st, lib = self._n, self._st, L.load()
        w = n._rows(x).to(self._xi.dtype)
        xi_bar = torch.empty_like(self._xi)
        tab_bar = [torch.empty_like(t) for t in self._tabs]
        azm_bar = torch.empty_like(self._azm)
        sums = torch.empty(st["npts"], dtype=torch.float64, device=w.device)  # per-point xi . t (full grid or octant)
        tmp = None if n._nsub == 1 else torch.empty_like(w[0])
        for c in range(w.shape[0]):
            f = n._fuse(st, self._fields[c])
            f.pro, f.in_ = L.PRO_PLAIN, (w[c] if tmp is None else n._separable(w[c], tmp)).data_ptr()
            f.epi, f.out, f.xi, f.abar = L.EPI_VJP, xi_bar[c].data_ptr(), self._xi[c].data_ptr(), st["sink"].data_ptr()
            if st["octant"]:
                f.w8 = sums.data_ptr()
            else:
                f.wfull = sums.data_ptr()
            B.hartley_fused(st["plan"], f)
            q = n._product(st, [t[c] for t in self._tabs], self._azm[c])
            for i in range(n._nsub):
                scratch = torch.empty(max(1, lib.nk_product_marginal_scratch(ctypes.byref(q), i) // 8), dtype=torch.float64,
                                      device=w.device)
                marg = torch.empty(st["sizes"][i], dtype=torch.float64, device=w.device)
                L.check(lib.nk_product_marginal(ctypes.byref(q), i, sums.data_ptr(), scratch.data_ptr(), marg.data_ptr(),
                                                B._stream()), "nk_product_marginal")
                tab_bar[i][c] = B.bin_sum(marg, st["bin_plans"][i])
            # d/d azm of azm * prod_i a_i: the same contraction with every table in place = sum_b a_0[b] abar_0[b] / azm
            azm_bar[c] = B.vdot(self._tabs[0][c].contiguous(), tab_bar[0][c].contiguous()) / self._azm[c]
        n.calls["adjoint"] += 1
        dom, dt = self._domain, self._xi.dtype
        vals = {"xi": Field(dom["xi"], xi_bar.reshape(dom["xi"].shape)),
                "azm": Field(dom["azm"], azm_bar.reshape(dom["azm"].shape).to(dt))}
        for i, t in enumerate(tab_bar):
            vals[f"a{i}"] = Field(dom[f"a{i}"], t.reshape(dom[f"a{i}"].shape).to(dt))
        return MultiField.from_dict(vals, dom)


def _packed_copy(tensors, device):
    """The tensors (all of one dtype) on `device`, moved as ONE buffer: one concatenation and one transfer instead of a
    transfer -- and, from a GPU, a synchronisation -- per tensor."""
    if not tensors:
        return []
    flat = torch.cat([t.reshape(-1) for t in tensors]).to(device)
    sizes = [t.numel() for t in tensors]
    return [part.reshape(t.shape) for part, t in zip(torch.split(flat, sizes), tensors)]


class _one_host_thread:
    """torch's CPU kernels on ONE thread while small host graphs run: with a pool of many threads some element-wise kernels take
    milliseconds on a few hundred numbers (torch.sqrt of 127 doubles: 4.5 ms with 8 threads, 1.6 us with one).  The setting is
    process-wide: nested / concurrent users are counted, the first one in saves the pool size and the last one out restores it."""

    _lock, _users, _saved = threading.Lock(), 0, None

    def __enter__(self):
        cls = _one_host_thread
        with cls._lock:
            if cls._users == 0:
                cls._saved = torch.get_num_threads()
                torch.set_num_threads(1)
            cls._users += 1

    def __exit__(self, *exc):
        cls = _one_host_thread
        with cls._lock:
            cls._users -= 1
            if cls._users == 0:
                torch.set_num_threads(cls._saved)
        return False


class _HostedAmplitudes(Operator):
    """{xi, hyper-parameters and spectrum excitations} -> {xi, azm, a_0, a_1, ...} for device fields: the excitations pass through,
    the amplitude MODELS -- operator graphs over a few hundred power bins each -- are evaluated on the HOST.  On the device every
    one of their ~100 element-wise steps is a kernel launch on a few hundred numbers (2-3 us of work behind ~25 us of launch
    path: a metric application of a 256 x 128 x 64 model spent 3.7 ms of which 0.3 ms in the transforms, the GPU idle 81 % of
    an optimize_kl iteration); here the small inputs come to the host in one packed copy, the graphs run in torch's CPU
    kernels, and the tables go back in one packed copy.  Same arithmetic, fp64, summation order of the host."""

    def __init__(self, excitation_key, models, domain):
        self._key, self._domain = excitation_key, domain
        self._models = models  # host graph: small keys -> {azm, a_0, ...}
        self._small = models.domain
        self._target = MultiDomain.make({"xi": domain[excitation_key], **{k: models.target[k] for k in models.target.keys()}})
        self._dtype = None

    def _to_host(self, mf, dom):
        keys = list(dom.keys())
        parts = _packed_copy([mf[k].val.to(torch.float64) for k in keys], "cpu")
        return MultiField.from_dict({k: Field(dom[k], p) for k, p in zip(keys, parts)}, dom)

    @staticmethod
    def _to_device(mf, dom, like):
        keys = list(dom.keys())
        parts = _packed_copy([mf[k].val.to(torch.float64) for k in keys], like.device)
        return {k: Field(dom[k], p) for k, p in zip(keys, parts)}

    def apply(self, x):
        self._check_input(x)
        lin = is_linearization(x)
        v = x.val if lin else x
        xi = v[self._key]
        small = self._to_host(v, self._small)
        with _one_host_thread():
            out = self._models(Linearization.make_var(small) if lin else small)
        tables = self._to_device(out.val if lin else out, self._models.target, xi.val)
        value = MultiField.from_dict({"xi": Field(self._target["xi"], xi.val), **tables}, self._target)
        if not lin:
            return value
        return x.new(value, _HostedAmplitudesJacobian(self, out.jac, xi.val))


class _HostedAmplitudesJacobian(LinearOperator):
    """Jacobian of `_HostedAmplitudes`: identity on the excitations, the host Jacobian of the amplitude models on the rest"""

    def __init__(self, op, host_jac, like):
        self._op, self._jac, self._like = op, host_jac, like
        self._domain, self._target = op.domain, op.target
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        op = self._op
        if mode == self.TIMES:
            small = op._to_host(x, op._small)
            with _one_host_thread():
                moved = self._jac(small)
            tables = op._to_device(moved, op._models.target, self._like)
            return MultiField.from_dict({"xi": Field(self._target["xi"], x[op._key].val), **tables}, self._target)
        bars = op._to_host(x, op._models.target)
        with _one_host_thread():
            small_bar = self._jac.adjoint_times(bars)
        parts = op._to_device(small_bar, op._small, self._like)
        parts = {k: f.astype(x["xi"].dtype) if f.dtype != x["xi"].dtype else f for k, f in parts.items()}
        return MultiField.from_dict({op._key: Field(self._domain[op._key], x["xi"].val), **parts}, self._domain)


class ProductCorrelatedFieldOperator(Operator):
    """CorrelatedFieldMaker.finalize for several spectra / total_N > 0: the generic operator graph for host fields and for
    grids the transform planner rejects, `_ProductFieldNode` fed by the amplitude graphs for device fields."""

    def __init__(self, generic_op, node, feeder, hosted=None):
        self._generic, self._node, self._fused = generic_op, node, node @ feeder
        self._domain, self._target = generic_op.domain, generic_op.target
        if self._fused.domain is not self._domain or self._fused.target is not self._target:
            raise RuntimeError("fused and generic correlated field disagree about their domains")
        # NK_HOSTED_AMPLITUDES=0: the amplitude graphs on the device as well (rounds 4-5, for A/B)
        self._feeder, self._hosted = feeder, hosted
        if hosted is not None and os.environ.get("NK_HOSTED_AMPLITUDES", "1") != "0":
            self._fused = node @ hosted

    def apply(self, x):
        self._check_input(x)
        v = x.val if is_linearization(x) else x
        any_field = next(iter(v.values()))
        on_device = v.device_id >= 0
        if on_device and self._node.supported(any_field.val.dtype, any_field.val.device):
            return self._fused(x)
        return self._generic(x)

    @property
    def fused_node(self):
        return self._node

    def __repr__(self):
        return f"ProductCorrelatedFieldOperator({self._node!r})"


# ------------------------------------------------------------------------------------------------
# maker
# ------------------------------------------------------------------------------------------------
class _ProductLayout:
    """Where the sub-spaces of a (product) correlated field sit: the harmonic product domain -- with a leading
    UnstructuredDomain(total_N) when several fields share the model -- the axis of every sub-space in it, and the operators
    that move between a sub-space's power bins, the product domain and position space."""

    def __init__(self, amplitudes, position_subdomains, total_N):
        lead = [UnstructuredDomain(total_N)] if total_N > 0 else []
        self.first = len(lead)  # index of the modelled space inside an amplitude's own target (behind the field copies)
        self.bins = [a.target[self.first] for a in amplitudes]
        self.harmonic = makeDomain(lead + [b.harmonic_partner for b in self.bins])
        self.axes = tuple(range(self.first, self.first + len(amplitudes)))
        self.positions = [sub[self.first] for sub in position_subdomains]
        self.broadcast = ContractionOperator(self.harmonic, self.axes).adjoint  # field copies -> product domain

    def spread(self, which, amplitude):
        """the amplitude of sub-space `which` (on its power bins) as a field on the whole product domain: distributed over
        its own harmonic space, constant along the others"""
        others = tuple(ax for ax in self.axes if ax != self.axes[which])
        along_others = ContractionOperator(self.harmonic, others)
        distribute = PowerDistributor(along_others.target, self.bins[which], self.first)
        return along_others.adjoint @ distribute @ amplitude

    def to_position(self, op):
        for axis, position in zip(self.axes, self.positions):
            op = HarmonicTransformOperator(op.target, position, space=axis) @ op
        return op


class CorrelatedFieldMaker:
    """Construction helper (reference correlated_fields.py:389-859): one or several amplitude spectra (power-law +
    integrated Wiener process, reduced variants, Matern); the single full amplitude is fused into one device operator."""

    def __init__(self, prefix, total_N=0):
        self._total_N = int(total_N)  # > 0: total_N fields at once on a leading UnstructuredDomain(total_N)
        if self._total_N < 0:
            raise ValueError("total_N must be >= 0")
        self._prefix, self._hyper = prefix, {}
        self._a, self._target_subdomains = [], []
        self._azm = self._offset_mean = None

    # -- argument plumbing shared by the add_* methods --------------------------------------------------
    @staticmethod
    def _partner_of(subdomain, harmonic_partner):
        """the harmonic space the spectrum lives on: the default codomain unless a compatible one is given"""
        if harmonic_partner is None:
            return subdomain.get_default_codomain()
        for a, b in ((subdomain, harmonic_partner), (harmonic_partner, subdomain)):
            a.check_codomain(b)
        return harmonic_partner

    @staticmethod
    def _mean_std(name, value, positive=False):
        """a (mean, std) hyper-prior; `positive`: both strictly positive (log-normal parameters)"""
        if len(value) != 2:
            raise TypeError(f"{name}: expected a (mean, std) pair")
        if positive and not (value[0] > 0.0 and value[1] > 0.0):
            raise ValueError(f"{name} must be strictly positive (or None)")
        return tuple(value)

    def _check_dofdex(self, dofdex):
        if dofdex is None:
            return [0] * self._total_N
        dofdex = [int(i) for i in dofdex]
        if len(dofdex) != self._total_N:
            raise ValueError("length of dofdex needs to match total_N")
        return dofdex

    def _copies(self, dofdex):
        """(checked dofdex, number of independent models it addresses; 0 without total_N)"""
        dofdex = self._check_dofdex(dofdex)
        return dofdex, (max(dofdex) + 1 if self._total_N > 0 else 0)

    def _register(self, amplitude, subdomain, index=None, fused_prefix=None):
        where = len(self._a) if index is None else index
        self._a.insert(where, amplitude)
        self._target_subdomains.insert(where, subdomain)
        self._amp_prefix = fused_prefix  # None: the model runs on the generic operator graph, never the fused node

    def add_fluctuations(self, target_subdomain, fluctuations, flexibility, asperity, loglogavgslope, prefix="",
                         index=None, dofdex=None, harmonic_partner=None):
        dofdex, copies = self._copies(dofdex)
        partner = self._partner_of(target_subdomain, harmonic_partner)
        priors = dict(fluctuations=self._mean_std("fluctuations", fluctuations),
                      loglogavgslope=self._mean_std("loglogavgslope", loglogavgslope),
                      flexibility=None if flexibility is None else self._mean_std("flexibility", flexibility, positive=True),
                      asperity=None if asperity is None else self._mean_std("asperity", asperity, positive=True))
        if priors["flexibility"] is None and priors["asperity"] is not None:
            raise ValueError("flexibility may not be disabled on its own")
        pre = self._prefix + str(prefix)

        def hyper(name, transform):
            return None if priors[name] is None else transform(*priors[name], pre + name, copies)

        many = self._total_N > 0
        subdomain = makeDomain((UnstructuredDomain(copies), target_subdomain) if many else target_subdomain)
        amplitude = _Amplitude(PowerSpace(partner), hyper("fluctuations", LognormalTransform),
                               hyper("flexibility", LognormalTransform), hyper("asperity", LognormalTransform),
                               hyper("loglogavgslope", NormalTransform), subdomain[-1].total_volume, pre + "spectrum",
                               dofdex if many else ())
        # only the full single-field model has a fused device operator
        fusable = not many and None not in (priors["flexibility"], priors["asperity"])
        self._register(amplitude, subdomain, index, pre if fusable else None)
        if fusable:
            self._hyper.update(priors)

    def add_fluctuations_matern(self, target_subdomain, scale, cutoff, loglogslope, prefix="", adjust_for_volume=True,
                                harmonic_partner=None):
        """Matern-kernel amplitude A(k) = a (1 + (|k|/b)^2)^(c/4) with log-normal scale a and cutoff b and normal
        spectral index c (reference library/correlated_fields.py:577-657); runs on the generic operator graph."""
        if self._total_N > 0:
            raise NotImplementedError("the Matern amplitude only works for total_N == 0 (as in the reference)")
        bins = PowerSpace(self._partner_of(target_subdomain, harmonic_partner))
        subdomain = makeDomain(target_subdomain)
        recipe = (("scale", scale, LognormalTransform), ("cutoff", cutoff, LognormalTransform),
                  ("loglogslope", loglogslope, NormalTransform))
        latent = [transform(*prior, self._prefix + str(prefix) + name) for name, prior, transform in recipe]
        self._register(_AmplitudeMatern(bins, *latent, subdomain[-1].total_volume if adjust_for_volume else 1.0), subdomain)

    def set_amplitude_total_offset(self, offset_mean, offset_std, dofdex=None):
        """Mean and zero-mode amplitude of the field (correlated_fields.py:659-711).  `offset_std`: a (mean, std) pair -- a
        log-normal zero-mode amplitude, the learnable case the fused device operator covers; a number -- a fixed amplitude
        (None or 0: no zero mode at all, the k = 0 coefficient is masked out); an Operator -- a user-built amplitude.  The
        last two run on the generic operator graph."""
        if self._offset_mean is not None and self._azm is not None:
            logger.warning("Overwriting the previous mean offset and zero-mode")
        self._hyper.pop("offset_std", None)
        if offset_std is None or np.isscalar(offset_std) or isinstance(offset_std, Operator):
            self._azm, self._offset_mean = (0.0 if offset_std is None else offset_std), offset_mean
            return
        if len(offset_std) != 2:
            raise TypeError("`offset_std` of invalid type and/or shape; expected a 2D tuple of floats")
        dofdex, copies = self._copies(dofdex)
        zero_mode = LognormalTransform(*offset_std, self._prefix + "zeromode", copies)
        if self._total_N > 0:  # one zero-mode model per entry of dofdex, copied onto the fields
            zero_mode = _Distributor(dofdex, zero_mode.target, UnstructuredDomain(self._total_N)) @ zero_mode
        self._azm, self._offset_mean = zero_mode, offset_mean
        self._hyper["offset_std"] = tuple(offset_std)

    @property
    def amplitude_total_offset(self):
        if self._azm is None:
            raise NotImplementedError("You need to set the `amplitude_total_offset` first")
        return self._azm

    azm = amplitude_total_offset

    @property
    def fluctuations(self):
        return tuple(self._a)

    def _fixed_zero_mode(self):
        """the zero-mode amplitude when it is a plain number, else None"""
        return self.azm if np.isscalar(self.azm) else None

    def get_normalized_amplitudes(self):
        """Amplitudes corrected for the otherwise degenerate zero mode: a_i * (1/azm on k != 0, 1 on k = 0)
        (reference correlated_fields.py:809-858).  A fixed zero mode of 0 removes the k = 0 entry instead (one spectrum
        only), a fixed 1 leaves the amplitudes as they are."""
        space = 1 if self._total_N > 0 else 0
        fixed = self._fixed_zero_mode()
        if fixed is not None and fixed == 0:
            if len(self._a) != 1:
                raise RuntimeError("Zeromode can not be disabled for product spectra")
            keep = np.ones(self._a[0].target.shape)
            keep[(slice(None),) * space + (0,)] = 0.0
            return (makeOp(makeField(self._a[0].target, keep)) @ self._a[0],)
        if fixed is not None and fixed == 1:
            return self.fluctuations
        out = []
        for amp in self._a:
            pspace = amp.target[space]
            mask, unmask = np.zeros(pspace.shape), np.zeros(pspace.shape)
            mask[1:] = unmask[0] = 1.0
            if fixed is not None:  # one constant factor per bin
                factor = np.broadcast_to(mask / fixed + unmask, amp.target.shape)
                out.append(makeOp(makeField(amp.target, np.ascontiguousarray(factor))) @ amp)
                continue
            zm_mask = DiagonalOperator(makeField(pspace, mask), amp.target, space)
            zm_unmask = DiagonalOperator(makeField(pspace, unmask), amp.target, space)(full(amp.target, 1.0))
            na = ContractionOperator(amp.target, space).adjoint @ self.azm.reciprocal()
            out.append(amp * (zm_mask(na) + zm_unmask))
        return tuple(out)

    @property
    def amplitude(self):
        if len(self._a) > 1:  # reference correlated_fields.py:860-866
            raise NotImplementedError("If more than one spectrum is present in the model, no unique set of amplitudes exist "
                                      "because only the relative scale is determined.")
        space = 1 if self._total_N > 0 else 0
        na = self.get_normalized_amplitudes()[0]
        if self._fixed_zero_mode() is not None:
            return na
        return na * (ContractionOperator(na.target, space).adjoint @ self.azm)

    @property
    def power_spectrum(self):
        return self.amplitude ** 2

    # -- fluctuation statistics: operators on latent samples, and the same quantities measured on field realisations
    # (correlated_fields.py:941-1064).  With f_j = (fluctuation amplitude of sub-space j) / (zero-mode amplitude) the variance of
    # the field, in units of the squared zero mode, is prod_j (1 + f_j^2) - 1; a slice along sub-space i keeps f_i^2 instead of
    # 1 + f_i^2. ------------------------------------------------------------------------------------------------------------
    def _relative_fluctuation(self, which):
        return self._a[which].fluctuation_amplitude / self.azm

    def _variance_factor(self, bare=None):
        """prod_j (1 + f_j^2), the factor of sub-space `bare` without its 1"""
        factors = [self._relative_fluctuation(j) ** 2 for j in range(len(self._a))]
        factors = [sq if j == bare else sq + 1 for j, sq in enumerate(factors)]
        return functools.reduce(lambda x, y: x * y, factors)

    def _need_subspace(self, space=0):
        if not self._a:
            raise NotImplementedError
        if space >= len(self._a):
            raise ValueError(f"invalid space specified; got {space!r}")

    def average_fluctuation(self, space):
        self._need_subspace(space)
        return self._a[space if len(self._a) > 1 else 0].fluctuation_amplitude

    def slice_fluctuation(self, space):
        self._need_subspace(space)
        if len(self._a) == 1:
            return self.average_fluctuation(0)
        return self._variance_factor(bare=space).sqrt() * self.azm

    @property
    def total_fluctuation(self):
        self._need_subspace()
        if len(self._a) == 1:
            return self.average_fluctuation(0)
        return (self._variance_factor() - 1).sqrt() * self.azm

    def moment_slice_to_average(self, fluctuations_slice_mean, nsamples=1000):
        """The mean of the `fluctuations` prior of a sub-space still to be added such that its slice fluctuations come out
        as `fluctuations_slice_mean`: that value over the prior mean of sqrt(prod_j (1 + f_j^2)) of the sub-spaces added so
        far, estimated from `nsamples` draws per sub-space (sub-space by sub-space, like the reference draws them)."""
        wanted = float(fluctuations_slice_mean)
        if not wanted > 0:
            raise ValueError(f"fluctuations_slice_mean must be greater zero; got {wanted!r}")
        from .field import from_random

        def prior_draws(op):
            return np.array([op(from_random(op.domain, "normal")).asnumpy() for _ in range(nsamples)])

        inflation = np.prod([1.0 + prior_draws(self._relative_fluctuation(j)) ** 2 for j in range(len(self._a))], axis=0)
        return wanted / np.mean(np.sqrt(inflation))

    def _summary_rows(self):
        """(label, statistic) of everything statistics_summary reports; statistics that are not defined are left out"""
        def maybe(label, getter):
            try:
                return [(label, getter())]
            except NotImplementedError:
                return []

        rows = maybe("Offset amplitude", lambda: self.amplitude_total_offset)
        rows.append(("Total fluctuation amplitude", self.total_fluctuation))
        for i in range(len(self._a) if len(self._a) > 1 else 0):
            rows.append((f"Average fluctuation (space {i})", self.average_fluctuation(i)))
            rows += maybe(f"Slice fluctuation (space {i})", lambda i=i: self.slice_fluctuation(i))
        return [(label, op) for label, op in rows if isinstance(op, Operator)]  # fixed numbers are not sampled

    def statistics_summary(self, prior_info):
        """Logs mean and standard deviation over `prior_info` prior draws of the offset amplitude and the fluctuation
        amplitudes that are sampled (correlated_fields.py:766-803)"""
        from .field import from_random
        from .probing import StatCalculator

        for label, op in self._summary_rows():
            stats = StatCalculator()
            for _ in range(prior_info):
                stats.add(op(from_random(op.domain, "normal")))
            spread = stats.var.ptw("sqrt").asnumpy().ravel()
            for m, sd in zip(stats.mean.asnumpy().ravel(), spread):
                logger.info(f"{label}: {m:.02E} ± {sd:.02E}")

    # realised statistics of signal-space samples: root of the sample average of a per-sample second moment
    @staticmethod
    def _geometry_axes(domain):
        """indices of the sub-domains with geometry: all but a leading UnstructuredDomain (the total_N axis)"""
        return tuple(range(1 if isinstance(domain[0], UnstructuredDomain) else 0, len(domain)))

    @staticmethod
    def _root_of_average(moments):
        """sqrt of the mean of a list of Fields (or scalar Fields), as numbers"""
        total = functools.reduce(lambda x, y: x + y, moments) * (1.0 / len(moments))
        return np.sqrt(total.asnumpy())

    @staticmethod
    def offset_amplitude_realized(samples):
        where = CorrelatedFieldMaker._geometry_axes(samples[0].domain)
        return CorrelatedFieldMaker._root_of_average([smp.mean(where) ** 2 for smp in samples])

    @staticmethod
    def total_fluctuation_realized(samples):
        where = CorrelatedFieldMaker._geometry_axes(samples[0].domain)
        return CorrelatedFieldMaker._root_of_average([smp.var(where) for smp in samples])

    @staticmethod
    def _one_of(samples, space):
        """(geometry axes, the absolute index of geometry sub-space `space`) -- or None when there is only one"""
        where = CorrelatedFieldMaker._geometry_axes(samples[0].domain)
        if space >= len(where):
            raise ValueError(f"invalid space specified; got {space!r}")
        return where, (where[space] if len(where) > 1 else None)

    @staticmethod
    def slice_fluctuation_realized(samples, space):
        """total second moment minus the part that survives averaging over sub-space `space`"""
        where, along = CorrelatedFieldMaker._one_of(samples, space)
        if along is None:
            return CorrelatedFieldMaker.total_fluctuation_realized(samples)

        def moment(smp):
            collapsed = smp.mean(along) ** 2
            return (smp ** 2).mean(where) - collapsed.mean(CorrelatedFieldMaker._geometry_axes(collapsed.domain))

        return CorrelatedFieldMaker._root_of_average([moment(smp) for smp in samples])

    @staticmethod
    def average_fluctuation_realized(samples, space):
        """variance along sub-space `space` of the sample averaged over the other geometry sub-spaces"""
        where, along = CorrelatedFieldMaker._one_of(samples, space)
        if along is None:
            return CorrelatedFieldMaker.total_fluctuation_realized(samples)
        others = tuple(sp for sp in where if sp != along)

        def moment(smp):
            profile = smp.mean(others)
            return profile.var(len(profile.domain) - 1)

        return CorrelatedFieldMaker._root_of_average([moment(smp) for smp in samples])

    def _generic_graph(self):
        """offset + HT( azm * prod_i a_i[pindex_i] * xi ) on the product of the harmonic spaces, with a leading
        UnstructuredDomain(total_N) when several fields are modelled at once (reference correlated_fields.py:713-764):
        every sub-space spreads its amplitude over the whole product domain, the spread amplitudes and the zero mode
        multiply the excitations, one harmonic transform per sub-space takes the product to position space."""
        layout = _ProductLayout(self._a, self._target_subdomains, self._total_N)
        spectrum = None
        for which, amplitude in enumerate(self.get_normalized_amplitudes()):
            spread = layout.spread(which, amplitude)
            spectrum = spread if spectrum is None else spectrum * spread
        excitations = Variable(layout.harmonic, self._prefix + "xi")
        if self._fixed_zero_mode() is None:
            spectrum = (layout.broadcast @ self.azm) * spectrum
        field = layout.to_position(spectrum.real * excitations)
        return field if self._offset_mean is None else field + float(self._offset_mean)

    def _product_operator(self, generic):
        """Several spectra and / or total_N > 0: the amplitude graphs feed one fused node (`_ProductFieldNode`); anything
        the node does not cover (non-regular sub-spaces, more than three sub-spaces) stays generic."""
        space = 1 if self._total_N > 0 else 0
        subs = [sd[space] for sd in self._target_subdomains]
        if len(self._a) > 3 or not all(isinstance(sd, RGSpace) for sd in subs):
            return generic
        lead = [UnstructuredDomain(self._total_N)] if self._total_N > 0 else []
        hspace = makeDomain(lead + [a.target[space].harmonic_partner for a in self._a])
        if hspace is not generic.domain[self._prefix + "xi"]:
            return generic
        amps = self.get_normalized_amplitudes()
        node = _ProductFieldNode(hspace, generic.target, [a.target for a in amps], self.azm.target, self._total_N,
                                 0.0 if self._offset_mean is None else self._offset_mean)
        models = self.azm.ducktape_left("azm")
        for i, a in enumerate(amps):
            models = models + a.ducktape_left(f"a{i}")
        feeder = Variable(hspace, self._prefix + "xi").ducktape_left("xi") + models
        hosted = None
        if isinstance(models.domain, MultiDomain) and self._prefix + "xi" not in models.domain.keys():
            hosted = _HostedAmplitudes(self._prefix + "xi", models, generic.domain)
        return ProductCorrelatedFieldOperator(generic, node, feeder, hosted)

    def finalize(self, prior_info=0):
        if len(self._a) < 1 or self._azm is None:
            raise NotImplementedError("add_fluctuations() and set_amplitude_total_offset() must have been called")
        generic = self._generic_graph()
        if "offset_std" not in self._hyper:  # a fixed or user-built zero mode: the generic graph
            return generic
        pos = self._target_subdomains[0][0]
        if len(self._a) > 1 or self._total_N > 0 or self._amp_prefix != self._prefix:
            return self._product_operator(generic)
        if not isinstance(pos, RGSpace):
            return generic
        return CorrelatedFieldOperator(pos, generic, self._prefix, 0.0 if self._offset_mean is None else self._offset_mean,
                                       dict(self._hyper))


def SimpleCorrelatedField(target, offset_mean, offset_std, fluctuations, flexibility, asperity, loglogavgslope,
                          prefix="", harmonic_partner=None):
    """reference library/correlated_fields_simple.py:36-133 (equal to the maker for one amplitude)."""
    cfm = CorrelatedFieldMaker(prefix)
    cfm.add_fluctuations(target, fluctuations, flexibility, asperity, loglogavgslope, harmonic_partner=harmonic_partner)
    cfm.set_amplitude_total_offset(offset_mean, offset_std)
    op = cfm.finalize()
    op.amplitude, op.power_spectrum = cfm.amplitude, cfm.power_spectrum  # correlated_fields_simple.py:130-131
    return op
