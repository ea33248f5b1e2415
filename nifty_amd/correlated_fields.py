"""CorrelatedFieldMaker / SimpleCorrelatedField: the amplitude model and the correlated-field operator.

Counterpart of reference nifty/cl/library/correlated_fields.py (_SlopeRemover :89-116,
_TwoLogIntegrations :119-162, _Normalization :165-208, _Amplitude :277-386, CorrelatedFieldMaker
:389-859), library/correlated_fields_simple.py:36-133 and operators/normal_operators.py:28-72, for the
single-amplitude, total_N == 0, non-Matern case the BASELINE configs use.

Two realisations share one interface:
* the GENERIC operator graph built from the small linear operators (host Fields, config 1's
  "nifty.cl numpy CPU" plumbing case and the structural cross-check of the fused path), and
* ``CorrelatedFieldOperator``: ONE fused Operator node whose forward pass, Jacobian and adjoint
  Jacobian run in the HIP kernels of libniftyk (amplitude kernels + fused Hartley transform) for
  device Fields.  ``finalize()`` returns this node; on host Fields it evaluates the generic graph.
"""
import numpy as np
import torch

from . import _lib as L
from . import backend as B
from .domains import DomainTuple, MultiDomain, PowerSpace, RGSpace, UnstructuredDomain, makeDomain
from .engine import SMALL_KEYS, lognormal_moments
from .field import Field, MultiField, full, makeField
from .operators import (ContractionOperator, DiagonalOperator, EndomorphicOperator, HarmonicTransformOperator,
                        LinearOperator, Linearization, Operator, PowerDistributor, ScalingOperator, Variable, VdotOperator,
                        ducktape,
                        is_linearization, makeOp)


# ------------------------------------------------------------------------------------------------
# hyper-parameter transforms (normal_operators.py:28-72)
# ------------------------------------------------------------------------------------------------
def NormalTransform(mean, sigma, key, N_copies=0):
    if N_copies != 0:
        raise NotImplementedError("N_copies > 0 (total_N > 0) is out of scope")
    domain = DomainTuple.scalar_domain()
    return float(sigma) * ducktape(domain, None, key) + float(mean)


def LognormalTransform(mean, sigma, key, N_copies=0):
    logmean, logsigma = lognormal_moments(mean, sigma)
    return NormalTransform(logmean, logsigma, key, N_copies).ptw("exp")


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


class _SlopeRemover(EndomorphicOperator):
    def __init__(self, domain, space=0):
        self._domain = makeDomain(domain)
        if len(self._domain) != 1 or not isinstance(self._domain[0], PowerSpace):
            raise NotImplementedError("only a single PowerSpace is supported")
        logkl = _relative_log_k_lengths(self._domain[0])
        self._sc = torch.from_numpy(logkl / float(logkl[-1]))
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        v = x.val
        if v.is_cuda:  # device Fields: libniftyk element-wise kernels, the two scalars cross the host
            sc = _dev(self._sc, v)
            v = v.contiguous()
            if mode == self.TIMES:
                return Field(self._tgt(mode), B.axpby(1.0, v, -float(v[-1].item()), sc))
            res = v.clone()
            last = res[-1:]
            B.axpby(1.0, last, -float(B.vdot(v, sc).item()), torch.ones_like(last), out=last)
            return Field(self._tgt(mode), res)
        if mode == self.TIMES:
            return Field(self._tgt(mode), v - v[-1] * self._sc)
        res = v.clone()
        res[-1] = res[-1] - (v * self._sc).sum()
        return Field(self._tgt(mode), res)


class _TwoLogIntegrations(LinearOperator):
    def __init__(self, target, space=0):
        self._target = makeDomain(target)
        if len(self._target) != 1 or not isinstance(self._target[0], PowerSpace):
            raise NotImplementedError("only a single PowerSpace is supported")
        nb = self._target[0].shape[0]
        self._domain = makeDomain(UnstructuredDomain((2, nb - 2)))
        self._log_vol = torch.from_numpy(_log_vol(self._target[0]))
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        lv = self._log_vol
        if x.val.is_cuda:
            return self._apply_device(x, mode)
        if mode == self.TIMES:
            v = x.val
            res = torch.zeros(self._target.shape, dtype=v.dtype)
            c = torch.cumsum(v[1], 0)
            cprev = torch.cat([torch.zeros(1, dtype=v.dtype), c[:-1]])
            res[2:] = torch.cumsum((c + cprev) / 2 * lv + v[0], 0)
            return Field(self._target, res)
        y = x.val
        res = torch.zeros(self._domain.shape, dtype=y.dtype)
        t = torch.flip(torch.cumsum(torch.flip(y[2:], [0]), 0), [0])
        res[0] = t
        u = t * (lv / 2.0)
        gc = u.clone()
        gc[:-1] += u[1:]
        res[1] = torch.flip(torch.cumsum(torch.flip(gc, [0]), 0), [0])
        return Field(self._domain, res)


    def _apply_device(self, x, mode):
        """Same arithmetic with nk_cumsum and the element-wise kernels; slicing / concatenation are copies."""
        v = x.val.contiguous()
        half_lv = B.axpby(0.5, _dev(self._log_vol, v).contiguous())
        zero1 = torch.zeros(1, dtype=v.dtype, device=v.device)
        if mode == self.TIMES:
            res = torch.zeros(self._target.shape, dtype=v.dtype, device=v.device)
            c = B.cumsum(v[1])
            both = B.binary(L.OP_ADD, c, torch.cat([zero1, c[:-1]]).contiguous())  # c_j + c_{j-1}
            inc = B.binary(L.OP_ADD, B.binary(L.OP_MUL, both, half_lv), v[0].contiguous())
            B.cumsum(inc, out=res[2:])
            return Field(self._target, res)
        res = torch.zeros(self._domain.shape, dtype=v.dtype, device=v.device)
        t = B.cumsum(v[2:], reverse=True, out=res[0])
        u = B.binary(L.OP_MUL, t, half_lv)
        gc = B.binary(L.OP_ADD, u, torch.cat([u[1:], zero1]).contiguous())
        B.cumsum(gc, reverse=True, out=res[1])
        return Field(self._domain, res)


class _Normalization(Operator):
    def __init__(self, domain, space=0):
        self._domain = self._target = DomainTuple.make(domain)
        pspace = self._domain[0]
        mult = pspace.rho.astype(np.float64).copy()
        mult[0] = 0.0
        self._mult = torch.from_numpy(mult)

    def apply(self, x):
        self._check_input(x)
        spec = x.exp()
        # sum over modes with multiplicities, broadcast back
        multop = makeOp(Field(self._domain, self._mult))
        co = ContractionOperator(self._domain, None)
        specsum = co.adjoint(co(multop(spec)))
        return (specsum.reciprocal() * spec).sqrt()


class _Amplitude(Operator):
    """a(k) = vol0 + vol1 fluct * normalised( slope rel_logk + SlopeRemover(TwoLogIntegrations(sigma xi_s)) )."""

    def __init__(self, target, fluctuations, flexibility, asperity, loglogavgslope, totvol, key):
        target = makeDomain(target)
        pspace = target[0]
        if not isinstance(pspace, PowerSpace):
            raise TypeError("PowerSpace required")
        twolog = _TwoLogIntegrations(target)
        dom = twolog.domain
        shp = dom.shape
        expander = ContractionOperator(dom, None).adjoint
        ps_expander = ContractionOperator(target, None).adjoint
        lv = _log_vol(pspace)
        vflex = np.zeros(shp)
        vflex[0] = vflex[1] = np.sqrt(lv)
        vflex = DiagonalOperator(makeField(dom, vflex))
        vasp = np.zeros(shp)
        vasp[0] += 1
        vasp = DiagonalOperator(makeField(dom, vasp))
        shift = np.ones(shp)
        shift[0] = lv ** 2 / 12.0
        shift = makeField(dom, shift)
        vslope = DiagonalOperator(makeField(target, _relative_log_k_lengths(pspace)))
        vol0, vol1 = np.zeros(pspace.shape), np.zeros(pspace.shape)
        vol1[1:] = vol0[0] = totvol
        vol0 = makeField(target, vol0)
        vol1 = DiagonalOperator(makeField(target, vol1))
        slope = vslope @ ps_expander @ loglogavgslope
        sig_flex = vflex @ expander @ flexibility if flexibility is not None else None
        sig_asp = vasp @ expander @ asperity if asperity is not None else None
        sig_fluc = vol1 @ ps_expander @ fluctuations
        xi = Variable(dom, key)
        # flexibility / asperity may be switched off (reference correlated_fields.py:351-363): pure power law, or an
        # integrated Wiener process without the asperity term
        if sig_asp is None and sig_flex is None:
            op = _Normalization(target) @ slope
        elif sig_asp is None:
            sigma = DiagonalOperator(shift.sqrt()) @ sig_flex
            smooth = _SlopeRemover(target) @ twolog @ (sigma * xi)
            op = _Normalization(target) @ (slope + smooth)
        elif sig_flex is None:
            raise ValueError("flexibility may not be disabled on its own")
        else:
            sigma = sig_flex * (sig_asp + shift).sqrt()
            smooth = _SlopeRemover(target) @ twolog @ (sigma * xi)
            op = _Normalization(target) @ (slope + smooth)
        op = (sig_fluc * op) + vol0
        self._op = op
        self._domain, self._target = op.domain, op.target
        self._fluc = fluctuations

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
        expander = ContractionOperator(pow_spc, None).adjoint
        k_squared = makeField(makeDomain(pow_spc), pow_spc.k_lengths ** 2)
        ker = (VdotOperator(k_squared).adjoint @ cutoff.ptw("power", -2.0)) + 1.0
        ker = (expander.scale(0.25) @ loglogslope) * ker.ptw("log") + (expander @ scale.ptw("log"))
        op = ker.ptw("exp")
        vol0, vol1 = np.zeros(pow_spc.shape), np.zeros(pow_spc.shape)
        vol0[0] = totvol
        vol1[1:] = totvol ** 0.5
        op = DiagonalOperator(makeField(op.target, vol1))(op) + makeField(op.target, vol0)
        self._op = op
        self._domain, self._target = op.domain, op.target

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
            B.hartley_fused(p._plan(dxi), f)
            return Field(self._target, out)
        w = x.val.contiguous()
        out = torch.empty_like(w)
        abar = torch.zeros(p._nb, dtype=torch.float64, device=w.device)
        f = p._fuse()
        f.pro, f.in_ = L.PRO_PLAIN, w.data_ptr()
        f.epi, f.out = L.EPI_VJP, out.data_ptr()
        f.pidx, f.amp, f.xi, f.abar = dev["pidx"].data_ptr(), self._amp.data_ptr(), self._xi.data_ptr(), abar.data_ptr()
        B.hartley_fused(p._plan(w), f)
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
        B.hartley_fused(self._plan(xi), f)
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


# ------------------------------------------------------------------------------------------------
# maker
# ------------------------------------------------------------------------------------------------
class CorrelatedFieldMaker:
    """Construction helper (reference correlated_fields.py:389-859): one or several amplitude spectra (power-law +
    integrated Wiener process, reduced variants, Matern); the single full amplitude is fused into one device operator."""

    def __init__(self, prefix, total_N=0):
        if total_N != 0:
            raise NotImplementedError("total_N > 0 is out of scope (SURVEY 2 #15)")
        self._prefix = prefix
        self._a = []
        self._target_subdomains = []
        self._azm = None
        self._offset_mean = None
        self._hyper = {}

    def add_fluctuations(self, target_subdomain, fluctuations, flexibility, asperity, loglogavgslope, prefix="",
                         index=None, dofdex=None, harmonic_partner=None):
        if dofdex is not None or index is not None:
            raise NotImplementedError("dofdex / index are out of scope")
        if harmonic_partner is None:
            harmonic_partner = target_subdomain.get_default_codomain()
        else:
            target_subdomain.check_codomain(harmonic_partner)
            harmonic_partner.check_codomain(target_subdomain)
        for arg in (fluctuations, loglogavgslope):
            if len(arg) != 2:
                raise TypeError
        for kw, arg in (("flexibility", flexibility), ("asperity", asperity)):
            if arg is None:
                continue
            if len(arg) != 2:
                raise TypeError
            if arg[0] <= 0.0 or arg[1] <= 0.0:
                raise ValueError(f"{kw} must be strictly positive (or None)")
        if flexibility is None and asperity is not None:
            raise ValueError("flexibility may not be disabled on its own")
        pre = self._prefix + str(prefix)
        fluct = LognormalTransform(*fluctuations, pre + "fluctuations")
        flex = LognormalTransform(*flexibility, pre + "flexibility") if flexibility is not None else None
        asp = LognormalTransform(*asperity, pre + "asperity") if asperity is not None else None
        avgsl = NormalTransform(*loglogavgslope, pre + "loglogavgslope")
        tsd = makeDomain(target_subdomain)
        amp = _Amplitude(PowerSpace(harmonic_partner), fluct, flex, asp, avgsl, tsd[-1].total_volume, pre + "spectrum")
        self._a.append(amp)
        self._target_subdomains.append(tsd)
        if flexibility is None or asperity is None:
            self._amp_prefix = None  # reduced amplitude models run on the generic operator graph, never the fused node
            return
        self._hyper.update(fluctuations=tuple(fluctuations), flexibility=tuple(flexibility), asperity=tuple(asperity),
                           loglogavgslope=tuple(loglogavgslope))
        self._amp_prefix = pre

    def add_fluctuations_matern(self, target_subdomain, scale, cutoff, loglogslope, prefix="", adjust_for_volume=True,
                                harmonic_partner=None):
        """Matern-kernel amplitude A(k) = a (1 + (|k|/b)^2)^(c/4) with log-normal scale a and cutoff b and normal
        spectral index c (reference library/correlated_fields.py:577-657); runs on the generic operator graph."""
        if harmonic_partner is None:
            harmonic_partner = target_subdomain.get_default_codomain()
        else:
            target_subdomain.check_codomain(harmonic_partner)
            harmonic_partner.check_codomain(target_subdomain)
        tsd = makeDomain(target_subdomain)
        pre = self._prefix + str(prefix)
        amp = _AmplitudeMatern(PowerSpace(harmonic_partner), LognormalTransform(*scale, pre + "scale"),
                               LognormalTransform(*cutoff, pre + "cutoff"), NormalTransform(*loglogslope, pre + "loglogslope"),
                               tsd[-1].total_volume if adjust_for_volume else 1.0)
        self._a.append(amp)
        self._target_subdomains.append(tsd)
        self._amp_prefix = None  # never the fused single-amplitude operator

    def set_amplitude_total_offset(self, offset_mean, offset_std, dofdex=None):
        if dofdex is not None:
            raise NotImplementedError
        self._offset_mean = offset_mean
        if offset_std is None or np.isscalar(offset_std) or isinstance(offset_std, Operator):
            raise NotImplementedError("only a (mean, std) tuple is supported for offset_std")
        if len(offset_std) != 2:
            raise TypeError("`offset_std` of invalid type and/or shape; expected a 2D tuple of floats")
        self._azm = LognormalTransform(*offset_std, self._prefix + "zeromode")
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

    def get_normalized_amplitudes(self):
        """Amplitudes corrected for the otherwise degenerate zero mode: a_i * (1/azm on k != 0, 1 on k = 0)
        (reference correlated_fields.py:809-858)."""
        out = []
        for amp in self._a:
            pspace = amp.target[0]
            mask, unmask = np.zeros(pspace.shape), np.zeros(pspace.shape)
            mask[1:] = unmask[0] = 1.0
            zm_mask = DiagonalOperator(makeField(amp.target, mask))
            zm_unmask = makeField(amp.target, unmask)
            na = ContractionOperator(amp.target, None).adjoint @ self.azm.reciprocal()
            out.append(amp * (zm_mask(na) + zm_unmask))
        return tuple(out)

    @property
    def amplitude(self):
        if len(self._a) > 1:  # reference correlated_fields.py:860-866
            raise NotImplementedError("If more than one spectrum is present in the model, no unique set of amplitudes exist "
                                      "because only the relative scale is determined.")
        na = self.get_normalized_amplitudes()[0]
        return na * (ContractionOperator(na.target, None).adjoint @ self.azm)

    @property
    def power_spectrum(self):
        return self.amplitude ** 2

    def _generic_graph(self):
        """offset + HT( azm * prod_i a_i[pindex_i] * xi ) on the product of the harmonic spaces
        (reference correlated_fields.py:713-764)."""
        n = len(self._a)
        hspace = makeDomain([a.target[0].harmonic_partner for a in self._a])
        ht = HarmonicTransformOperator(hspace, self._target_subdomains[0][0], space=0)
        for i in range(1, n):
            ht = HarmonicTransformOperator(ht.target, self._target_subdomains[i][0], space=i) @ ht
        amps = list(self.get_normalized_amplitudes())
        for i in range(n):
            co = ContractionOperator(hspace, tuple(j for j in range(n) if j != i))
            amps[i] = co.adjoint @ PowerDistributor(co.target, amps[i].target[0]) @ amps[i]
        corr = amps[0]
        for a in amps[1:]:
            corr = corr * a
        xi = Variable(hspace, self._prefix + "xi")
        expander = ContractionOperator(hspace, None).adjoint
        azm = expander @ self.azm
        op = ht((azm * corr).real * xi)
        if self._offset_mean is not None:
            op = op + float(self._offset_mean)
        return op

    def finalize(self, prior_info=0):
        if len(self._a) < 1 or self._azm is None:
            raise NotImplementedError("add_fluctuations() and set_amplitude_total_offset() must have been called")
        generic = self._generic_graph()
        pos = self._target_subdomains[0][0]
        if len(self._a) > 1 or not isinstance(pos, RGSpace) or self._amp_prefix != self._prefix:
            return generic  # product spectra run on the generic operator graph
        return CorrelatedFieldOperator(pos, generic, self._prefix, 0.0 if self._offset_mean is None else self._offset_mean,
                                       dict(self._hyper))


def SimpleCorrelatedField(target, offset_mean, offset_std, fluctuations, flexibility, asperity, loglogavgslope,
                          prefix="", harmonic_partner=None):
    """reference library/correlated_fields_simple.py:36-133 (equal to the maker for one amplitude)."""
    cfm = CorrelatedFieldMaker(prefix)
    cfm.add_fluctuations(target, fluctuations, flexibility, asperity, loglogavgslope, harmonic_partner=harmonic_partner)
    cfm.set_amplitude_total_offset(offset_mean, offset_std)
    return cfm.finalize()
