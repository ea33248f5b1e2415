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
        nb = self._target[space].shape[0]
        dom = list(self._target)
        dom[space] = UnstructuredDomain((2, nb - 2))
        self._domain = makeDomain(dom)
        self._row_target = DomainTuple.make(self._target[space])
        self._row_domain = DomainTuple.make(dom[space])
        self._log_vol = torch.from_numpy(_log_vol(self._target[space]))
        self._capability = self.TIMES | self.ADJOINT_TIMES

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


class _Normalization(Operator):
    def __init__(self, domain, space=0):
        self._domain = self._target = DomainTuple.make(domain)
        _batched(self._domain, space, "_Normalization")
        self._space = space
        pspace = self._domain[space]
        mult = pspace.rho.astype(np.float64).copy()
        mult[0] = 0.0
        self._mult = makeField(DomainTuple.make(pspace), mult)

    def apply(self, x):
        self._check_input(x)
        spec = x.exp()
        # sum over modes with multiplicities (per copy), broadcast back
        multop = DiagonalOperator(self._mult, self._domain, self._space)
        co = ContractionOperator(self._domain, self._space)
        specsum = co.adjoint(co(multop(spec)))
        return (specsum.reciprocal() * spec).sqrt()


class _Amplitude(Operator):
    """a(k) = vol0 + vol1 fluct * normalised( slope rel_logk + SlopeRemover(TwoLogIntegrations(sigma xi_s)) ).
    dofdex (total_N > 0): max(dofdex)+1 independent spectra on a leading UnstructuredDomain, distributed to the
    len(dofdex) fields (reference correlated_fields.py:277-386)."""

    def __init__(self, target, fluctuations, flexibility, asperity, loglogavgslope, totvol, key, dofdex=()):
        dofdex = [int(i) for i in dofdex]
        distributor = None
        if len(dofdex) > 0:
            n_copies, space = max(dofdex) + 1, 1
            target = makeDomain((UnstructuredDomain(n_copies), target))
            if n_copies != len(dofdex):
                distributed_tgt = makeDomain((UnstructuredDomain(len(dofdex)), target[1]))
                distributor = _Distributor(dofdex, target, distributed_tgt)
        else:
            space = 0
            target = makeDomain(target)
        pspace = target[space]
        if not isinstance(pspace, PowerSpace):
            raise TypeError("PowerSpace required")
        twolog = _TwoLogIntegrations(target, space)
        dom = twolog.domain
        shp = dom[space].shape
        expander = ContractionOperator(dom, space).adjoint
        ps_expander = ContractionOperator(target, space).adjoint
        lv = _log_vol(pspace)
        vflex = np.zeros(shp)
        vflex[0] = vflex[1] = np.sqrt(lv)
        vflex = DiagonalOperator(makeField(dom[space], vflex), dom, space)
        vasp = np.zeros(shp)
        vasp[0] += 1
        vasp = DiagonalOperator(makeField(dom[space], vasp), dom, space)
        shift = np.ones(shp)
        shift[0] = lv ** 2 / 12.0
        shift = DiagonalOperator(makeField(dom[space], shift), dom, space)(full(dom, 1.0))
        vslope = DiagonalOperator(makeField(pspace, _relative_log_k_lengths(pspace)), target, space)
        vol0, vol1 = np.zeros(pspace.shape), np.zeros(pspace.shape)
        vol1[1:] = vol0[0] = totvol
        vol0 = DiagonalOperator(makeField(pspace, vol0), target, space)(full(target, 1.0))
        vol1 = DiagonalOperator(makeField(pspace, vol1), target, space)
        slope = vslope @ ps_expander @ loglogavgslope
        sig_flex = vflex @ expander @ flexibility if flexibility is not None else None
        sig_asp = vasp @ expander @ asperity if asperity is not None else None
        sig_fluc = vol1 @ ps_expander @ fluctuations
        xi = Variable(dom, key)
        # flexibility / asperity may be switched off (reference correlated_fields.py:351-363): pure power law, or an
        # integrated Wiener process without the asperity term
        if sig_asp is None and sig_flex is None:
            op = _Normalization(target, space) @ slope
        elif sig_asp is None:
            sigma = DiagonalOperator(shift.sqrt()) @ sig_flex
            smooth = _SlopeRemover(target, space) @ twolog @ (sigma * xi)
            op = _Normalization(target, space) @ (slope + smooth)
        elif sig_flex is None:
            raise ValueError("flexibility may not be disabled on its own")
        else:
            sigma = sig_flex * (sig_asp + shift).sqrt()
            smooth = _SlopeRemover(target, space) @ twolog @ (sigma * xi)
            op = _Normalization(target, space) @ (slope + smooth)
        if distributor is not None:
            op = ((distributor @ sig_fluc) * (distributor @ op)) + distributor(vol0)
            self._fluc = _Distributor(dofdex, fluctuations.target, distributed_tgt[0]) @ fluctuations
        else:
            op = (sig_fluc * op) + vol0
            self._fluc = fluctuations
        self._op = op
        self._domain, self._target = op.domain, op.target
        self._space = space

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
        if int(total_N) < 0:
            raise ValueError("total_N must be >= 0")
        self._total_N = int(total_N)  # > 0: total_N fields at once on a leading UnstructuredDomain(total_N)
        self._prefix = prefix
        self._a = []
        self._target_subdomains = []
        self._azm = None
        self._offset_mean = None
        self._hyper = {}

    def add_fluctuations(self, target_subdomain, fluctuations, flexibility, asperity, loglogavgslope, prefix="",
                         index=None, dofdex=None, harmonic_partner=None):
        dofdex = self._check_dofdex(dofdex)
        N = max(dofdex) + 1 if self._total_N > 0 else 0
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
        fluct = LognormalTransform(*fluctuations, pre + "fluctuations", N)
        flex = LognormalTransform(*flexibility, pre + "flexibility", N) if flexibility is not None else None
        asp = LognormalTransform(*asperity, pre + "asperity", N) if asperity is not None else None
        avgsl = NormalTransform(*loglogavgslope, pre + "loglogavgslope", N)
        tsd = makeDomain((UnstructuredDomain(N), target_subdomain)) if self._total_N > 0 else makeDomain(target_subdomain)
        amp = _Amplitude(PowerSpace(harmonic_partner), fluct, flex, asp, avgsl, tsd[-1].total_volume, pre + "spectrum",
                         dofdex if self._total_N > 0 else ())
        if index is not None:
            self._a.insert(index, amp)
            self._target_subdomains.insert(index, tsd)
        else:
            self._a.append(amp)
            self._target_subdomains.append(tsd)
        if flexibility is None or asperity is None or self._total_N > 0:
            self._amp_prefix = None  # reduced amplitude models run on the generic operator graph, never the fused node
            return
        self._hyper.update(fluctuations=tuple(fluctuations), flexibility=tuple(flexibility), asperity=tuple(asperity),
                           loglogavgslope=tuple(loglogavgslope))
        self._amp_prefix = pre

    def add_fluctuations_matern(self, target_subdomain, scale, cutoff, loglogslope, prefix="", adjust_for_volume=True,
                                harmonic_partner=None):
        """Matern-kernel amplitude A(k) = a (1 + (|k|/b)^2)^(c/4) with log-normal scale a and cutoff b and normal
        spectral index c (reference library/correlated_fields.py:577-657); runs on the generic operator graph."""
        if self._total_N > 0:
            raise NotImplementedError("the Matern amplitude only works for total_N == 0 (as in the reference)")
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

    def _check_dofdex(self, dofdex):
        if dofdex is None:
            return [0] * self._total_N
        dofdex = [int(i) for i in dofdex]
        if len(dofdex) != self._total_N:
            raise ValueError("length of dofdex needs to match total_N")
        return dofdex

    def set_amplitude_total_offset(self, offset_mean, offset_std, dofdex=None):
        self._offset_mean = offset_mean
        if offset_std is None or np.isscalar(offset_std) or isinstance(offset_std, Operator):
            raise NotImplementedError("only a (mean, std) tuple is supported for offset_std")
        if len(offset_std) != 2:
            raise TypeError("`offset_std` of invalid type and/or shape; expected a 2D tuple of floats")
        dofdex = self._check_dofdex(dofdex)
        N = max(dofdex) + 1 if self._total_N > 0 else 0
        zm = LognormalTransform(*offset_std, self._prefix + "zeromode", N)
        if self._total_N > 0:
            zm = _Distributor(dofdex, zm.target, UnstructuredDomain(self._total_N)) @ zm
        self._azm = zm
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
        space = 1 if self._total_N > 0 else 0
        out = []
        for amp in self._a:
            pspace = amp.target[space]
            mask, unmask = np.zeros(pspace.shape), np.zeros(pspace.shape)
            mask[1:] = unmask[0] = 1.0
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
        return na * (ContractionOperator(na.target, space).adjoint @ self.azm)

    @property
    def power_spectrum(self):
        return self.amplitude ** 2

    def _generic_graph(self):
        """offset + HT( azm * prod_i a_i[pindex_i] * xi ) on the product of the harmonic spaces, with a leading
        UnstructuredDomain(total_N) when several fields are modelled at once (reference correlated_fields.py:713-764)."""
        n = len(self._a)
        if self._total_N > 0:
            hspace = makeDomain([UnstructuredDomain(self._total_N)] + [a.target[-1].harmonic_partner for a in self._a])
            spaces, amp_space = tuple(range(1, n + 1)), 1
        else:
            hspace = makeDomain([a.target[0].harmonic_partner for a in self._a])
            spaces, amp_space = tuple(range(n)), 0
        ht = HarmonicTransformOperator(hspace, self._target_subdomains[0][amp_space], space=spaces[0])
        for i in range(1, n):
            ht = HarmonicTransformOperator(ht.target, self._target_subdomains[i][amp_space], space=spaces[i]) @ ht
        amps = list(self.get_normalized_amplitudes())
        for i in range(n):
            co = ContractionOperator(hspace, spaces[:i] + spaces[i + 1:])
            amps[i] = co.adjoint @ PowerDistributor(co.target, amps[i].target[amp_space], amp_space) @ amps[i]
        corr = amps[0]
        for a in amps[1:]:
            corr = corr * a
        xi = Variable(hspace, self._prefix + "xi")
        expander = ContractionOperator(hspace, spaces).adjoint
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
        if len(self._a) > 1 or self._total_N > 0 or not isinstance(pos, RGSpace) or self._amp_prefix != self._prefix:
            return generic  # product spectra and total_N > 0 run on the generic operator graph
        return CorrelatedFieldOperator(pos, generic, self._prefix, 0.0 if self._offset_mean is None else self._offset_mean,
                                       dict(self._hyper))


def SimpleCorrelatedField(target, offset_mean, offset_std, fluctuations, flexibility, asperity, loglogavgslope,
                          prefix="", harmonic_partner=None):
    """reference library/correlated_fields_simple.py:36-133 (equal to the maker for one amplitude)."""
    cfm = CorrelatedFieldMaker(prefix)
    cfm.add_fluctuations(target, fluctuations, flexibility, asperity, loglogavgslope, harmonic_partner=harmonic_partner)
    cfm.set_amplitude_total_offset(offset_mean, offset_std)
    return cfm.finalize()
