"""Likelihood energies and the standard Hamiltonian.

Counterpart of reference nifty/cl/operators/energy_operators.py: EnergyOperator (:36-41),
LikelihoodEnergyOperator (:44-164), _LikelihoodChain (:166-208), Squared2NormOperator (:306-327),
QuadraticFormOperator (:330-352), GaussianEnergy (:517-595), PoissonianEnergy (:617-640) and
StandardHamiltonian (:890-931).  Other likelihoods of the reference are out of scope.
"""
import numpy as np

from .domains import DomainTuple, MultiDomain, makeDomain
from .field import Field, MultiField, full
from .operators import (Adder, EndomorphicOperator, LinearOperator, Linearization, Operator, SamplingEnabler,
                        SandwichOperator, ScalingOperator, VdotOperator, _OpChain, _same_domain, is_linearization,
                        is_operator, makeOp)


class EnergyOperator(Operator):
    """Operator with scalar target."""

    _target = DomainTuple.scalar_domain()


class LikelihoodEnergyOperator(EnergyOperator):
    """Energy with a data residual and a Fisher metric in data space."""

    def __init__(self, data_residual, sqrt_data_metric_at):
        if data_residual is not None and not is_operator(data_residual):
            raise TypeError(f"{data_residual} is not an operator")
        self._res = data_residual
        self._sqrt_data_metric_at = sqrt_data_metric_at
        self._name = None

    def normalized_residual(self, x):
        return (self._sqrt_data_metric_at(x) @ self._res).force(x)

    @property
    def data_domain(self):
        return None if self._res is None else self._res.target

    def get_transformation(self):
        raise NotImplementedError("`get_transformation` not implemented (yet) for this operator")

    def __matmul__(self, other):
        return _LikelihoodChain(self, other)

    def __rmatmul__(self, other):
        return _LikelihoodChain(other, self)

    def __add__(self, other):
        return _LikelihoodSum.make([self, other])

    def __radd__(self, other):
        return _LikelihoodSum.make([other, self])

    def get_metric_at(self, x):
        dtp, f = self.get_transformation()
        bun = f(Linearization.make_var(x)).jac
        return SandwichOperator.make(bun, sampling_dtype=dtp)

    @property
    def name(self):
        return self._name

    @name.setter
    def name(self, x):
        if isinstance(self, _LikelihoodSum):
            raise RuntimeError("The name of a LikelihoodSum cannot be set. Set the name of each individual "
                               "LikelihoodEnergy separately.")
        self._name = x


class _LikelihoodChain(LikelihoodEnergyOperator):
    """likelihood @ model  (or scalar @ likelihood)."""

    def __init__(self, op1, op2):
        self._op = _OpChain.make((op1, op2))
        self._domain = self._op.domain
        if isinstance(op1, ScalingOperator):
            res, sqrt_met = op2._res, op2._sqrt_data_metric_at
        elif op1._res is None:
            res = sqrt_met = None
        else:
            if op2.target is not op1._res.domain:
                raise NotImplementedError("likelihood chains need model.target == data-residual domain "
                                          "(PartialExtractor is not implemented)")
            res = op1._res @ op2

            def sqrt_met(x, a=op1, b=op2):
                return a._sqrt_data_metric_at(b.force(x))
        super().__init__(res, sqrt_met)
        self.name = (op2 if isinstance(op1, ScalingOperator) else op1).name

    @property
    def likelihood(self):
        ops = self._op._ops
        return ops[1] if isinstance(ops[0], ScalingOperator) else ops[0]

    @property
    def model(self):
        """The operator chain feeding the likelihood (used by the fusion pass of optimize_kl)."""
        ops = self._op._ops
        ii = 2 if isinstance(ops[0], ScalingOperator) else 1
        return _OpChain.make(ops[ii:]) if len(ops) > ii else None

    def get_transformation(self):
        ops = self._op._ops
        scaled = isinstance(ops[0], ScalingOperator)
        ii = 1 if scaled else 0
        tr = ops[ii].get_transformation()
        if tr is None:
            return tr
        dtype, trafo = tr
        if scaled:
            trafo = trafo.scale(np.sqrt(ops[0]._factor))
        return dtype, _OpChain.make((trafo,) + tuple(ops[ii + 1:]))

    def apply(self, x):
        self._check_input(x)
        return self._op(x)

    def __repr__(self):
        return repr(self._op)


class _LikelihoodSum(LikelihoodEnergyOperator):
    """Sum of likelihoods (several data sets / instruments; reference energy_operators.py:211-303).  The data spaces
    of the summands form a disjoint union: residuals, data metrics and transformations carry the summand's name
    ("Likelihood i" by default) as key (DomainTuple data) or key prefix (MultiDomain data)."""

    def __init__(self, ops, _callingfrommake=False):
        from .operators import PrependKey

        if not _callingfrommake:
            raise NotImplementedError
        if len({isinstance(oo.domain, DomainTuple) for oo in ops}) > 1:
            raise RuntimeError("Some operators have DomainTuple and others have MultiDomain as domain. "
                               "This should not happen.")
        self._ops = list(ops)
        self._name = None
        names = self._all_names()
        if len(names) != len(set(names)):
            raise ValueError(f"Name collision in likelihoods detected: {names}")
        res, prep, data_ops = [], [], []
        for ii, oo in enumerate(self._ops):
            if oo._res is None:
                continue
            lprep = Operator.identity_operator(oo.data_domain)
            key = self._get_name(ii)
            if isinstance(lprep.target, DomainTuple):
                lprep = lprep.ducktape_left("")
            else:
                key = key + ": "
            lprep = PrependKey(lprep.target, key) @ lprep
            prep.append(lprep)
            res.append(lprep @ oo._res)
            data_ops.append(oo)

        def sqrt_data_metric_at(x):
            tot = None
            for pp, oo in zip(prep, data_ops):
                term = pp @ oo._sqrt_data_metric_at(x) @ pp.adjoint
                tot = term if tot is None else tot + term
            return tot

        data_residuals = None
        for rr in res:
            data_residuals = rr if data_residuals is None else data_residuals + rr
        self._res, self._sqrt_data_metric_at = data_residuals, sqrt_data_metric_at
        self._domain = data_residuals.domain if data_residuals is not None else self._ops[0].domain

    @classmethod
    def _unpack(cls, ops, res):
        for op in ops:
            res = cls._unpack(op._ops, res) if isinstance(op, cls) else res + [op]
        return res

    @classmethod
    def make(cls, ops):
        for op in ops:
            if not isinstance(op, LikelihoodEnergyOperator):
                raise TypeError(f"a likelihood can only be added to another likelihood, got {type(op).__name__}")
        res = cls._unpack(ops, [])
        return res[0] if len(res) == 1 else cls(res, _callingfrommake=True)

    def apply(self, x):
        from .operators import _OpSum

        self._check_input(x)
        return _OpSum._apply_operator_sum(x, self._ops)

    def get_transformation(self):
        from .operators import PrependKey

        trs = [oo.get_transformation() for oo in self._ops]
        if any(tr is None for tr in trs):
            return None
        dtype, total = {}, None
        for ii, (dtp, tr) in enumerate(trs):
            key = self._get_name(ii)
            if isinstance(tr.target, MultiDomain):
                key = key + ": "
                dtype.update({key + d: dtp[d] for d in dtp.keys()})
                tr = PrependKey(tr.target, key) @ tr
            else:
                dtype[key] = dtp
                tr = tr.ducktape_left(key)
            total = tr if total is None else total + tr
        return dtype, total

    def _all_names(self):
        return [self._get_name(ii) for ii in range(len(self._ops))]

    def _get_name(self, i):
        res = self._ops[i].name
        return f"Likelihood {i}" if res is None else res

    def __repr__(self):
        return "_LikelihoodSum:\n" + "\n".join(f"  *{self._get_name(ii)}*\n  {oo!r}" for ii, oo in enumerate(self._ops))


class Squared2NormOperator(EnergyOperator):
    def __init__(self, domain):
        self._domain = domain

    def apply(self, x):
        self._check_input(x)
        if not is_linearization(x):
            return x.vdot(x).at(x.device_id)
        res = x.val.vdot(x.val).at(x.device_id)
        return x.new(res, VdotOperator(x.val * 2.0))


class QuadraticFormOperator(EnergyOperator):
    def __init__(self, endo):
        if not isinstance(endo, EndomorphicOperator):
            raise TypeError(f"op must be an EndomorphicOperator.\nGot: {endo}")
        self._op = endo
        self._domain = endo.domain

    def apply(self, x):
        self._check_input(x)
        if not is_linearization(x):
            return (x.vdot(self._op(x)) * 0.5).at(x.device_id)
        tmp = self._op(x.val)
        res = (x.val.vdot(tmp) * 0.5).at(x.device_id)
        return x.new(res, VdotOperator(tmp))


class GaussianEnergy(LikelihoodEnergyOperator):
    """E(s) = 1/2 (s - d)^dagger N^-1 (s - d)."""

    def __init__(self, data=None, inverse_covariance=None, domain=None, sampling_dtype=None):
        if inverse_covariance is not None and not isinstance(inverse_covariance, LinearOperator):
            raise TypeError(f"inverse_covariance needs to be either None or a LinearOperator, got: {inverse_covariance}")
        if data is not None and not isinstance(data, (Field, MultiField)):
            raise TypeError(f"data needs to be a (Multi)Field or None, got: {data}")
        dom = None
        for cand in (None if inverse_covariance is None else inverse_covariance.domain,
                     None if data is None else data.domain, None if domain is None else makeDomain(domain)):
            if cand is None:
                continue
            if dom is not None:
                _same_domain(dom, makeDomain(cand))
            dom = makeDomain(cand)
        if dom is None:
            raise ValueError("no domain given")
        self._domain = dom
        if inverse_covariance is None:
            self._op = Squared2NormOperator(self._domain).scale(0.5)
            dt = sampling_dtype if data is None else data.dtype
            self._icov = ScalingOperator(self._domain, 1.0, dt)
        else:
            self._op = QuadraticFormOperator(inverse_covariance)
            self._icov = inverse_covariance
        self._data = data
        res = Operator.identity_operator(self._domain) if data is None else Adder(data, neg=True)
        super().__init__(res, lambda x: self.get_metric_at(x).get_sqrt())

    def apply(self, x):
        self._check_input(x)
        if self._data is not None and self._data.device_id != x.device_id:
            self._data = self._data.at(x.device_id)
        residual = x if self._data is None else x - self._data
        res = self._op(residual)
        if is_linearization(x) and x.want_metric:
            return res.add_metric(self._icov)
        return res

    def get_metric_at(self, x):
        return self._icov

    def get_transformation(self):
        return self._icov.sampling_dtype, self._icov.get_sqrt()

    def __repr__(self):
        return "GaussianEnergy"


class PoissonianEnergy(LikelihoodEnergyOperator):
    """E(lambda) = sum(lambda) - d^T log(lambda) for integer counts d."""

    def __init__(self, d):
        if not isinstance(d, Field) or not np.issubdtype(d.dtype, np.integer):
            raise TypeError("data is of invalid data-type; counts need to be integers")
        if bool((d.val < 0).any()):
            raise ValueError("count data is negative and thus can not be Poissonian")
        self._d = d
        self._d_float = {}
        self._domain = DomainTuple.make(d.domain)
        super().__init__(Adder(d, neg=True), lambda x: self.get_metric_at(x).get_sqrt())

    def _counts_like(self, x):
        key = (x.device_id, x.dtype)
        if key not in self._d_float:
            self._d_float[key] = self._d.at(x.device_id).astype(x.dtype)
        return self._d_float[key]

    def apply(self, x):
        self._check_input(x)
        val = x.val if is_linearization(x) else x
        d = self._counts_like(val)
        res = x.sum() - x.log().vdot(d)
        if not (is_linearization(x) and x.want_metric):
            return res
        return res.add_metric(self.get_metric_at(x.val))

    def get_transformation(self):
        return np.float64, Operator.identity_operator(self._domain).sqrt().scale(2.0)


class StudentTEnergy(LikelihoodEnergyOperator):
    """E(f) = (theta+1)/2 sum log(1 + f^2/theta), theta a scalar or a Field (reference energy_operators.py:704-746)."""

    def __init__(self, domain, theta):
        self._domain = DomainTuple.make(domain)
        self._theta = theta
        super().__init__(Operator.identity_operator(self._domain), lambda x: self.get_metric_at(x).get_sqrt())

    def apply(self, x):
        self._check_input(x)
        th = self._theta
        if isinstance(th, Field):
            th = th.at((x.val if is_linearization(x) else x).device_id)
            res = (makeOp((th + 1.0) * 0.5)(makeOp(th.reciprocal())(x ** 2).log1p())).sum()
        else:
            res = ((x ** 2) * (1.0 / th)).log1p().sum() * ((th + 1.0) / 2.0)
        if not (is_linearization(x) and x.want_metric):
            return res
        return res.add_metric(self.get_metric_at(x.val))

    def get_transformation(self):
        th = self._theta if isinstance(self._theta, Field) else full(self._domain, float(self._theta))
        return np.float64, makeOp(((th + 1.0) / (th + 3.0)).sqrt())


class BernoulliEnergy(LikelihoodEnergyOperator):
    """E(f) = -d^T log f - (1-d)^T log(1-f) for event data d in {0, 1} (reference energy_operators.py:749-792)."""

    def __init__(self, d):
        if not isinstance(d, Field) or not np.issubdtype(d.dtype, np.integer):
            raise TypeError(f"d needs to be a Field with integer values. Got:\n{d}")
        vals = set(np.unique(d.asnumpy()).tolist())
        if not vals <= {0, 1}:
            raise ValueError(f"d can only contain 0 and 1. Got: {vals}")
        self._d = d
        self._d_float = {}
        self._domain = DomainTuple.make(d.domain)
        super().__init__(Adder(d, neg=True), lambda x: self.get_metric_at(x).get_sqrt())

    def _events_like(self, x):
        key = (x.device_id, x.dtype)
        if key not in self._d_float:
            self._d_float[key] = self._d.at(x.device_id).astype(x.dtype)
        return self._d_float[key]

    def apply(self, x):
        self._check_input(x)
        d = self._events_like(x.val if is_linearization(x) else x)
        res = -x.log().vdot(d) + (1.0 - x).log().vdot(d - 1.0)
        if not (is_linearization(x) and x.want_metric):
            return res
        return res.add_metric(self.get_metric_at(x.val))

    def get_transformation(self):
        ident = Operator.identity_operator(self._domain)
        res = (ScalingOperator(self._domain, -1.0) + 1.0) * ident.reciprocal()  # (1 - f) / f
        return np.float64, res.sqrt().arctan().scale(-2.0)


class VariableCovarianceGaussianEnergy(LikelihoodEnergyOperator):
    """-log of a Gaussian in the residual s with an UNKNOWN diagonal inverse covariance C, both inferred:
    E(s, C) = 1/2 s^T C s - 1/2 tr log C on the MultiDomain {residual_key, inverse_covariance_key}
    (reference energy_operators.py:355-450; real sampling dtypes).  ``use_full_fisher``: the exact Fisher metric
    diag(C, 1/2 C^-2); otherwise the metric of the local transformation used by geoVI."""

    def __init__(self, domain, residual_key, inverse_covariance_key, sampling_dtype, use_full_fisher=True):
        self._kr, self._ki = str(residual_key), str(inverse_covariance_key)
        dom = DomainTuple.make(domain)
        self._domain = MultiDomain.make({self._kr: dom, self._ki: dom})
        if np.issubdtype(np.dtype(sampling_dtype), np.complexfloating):
            raise NotImplementedError("complex residuals are not supported")
        self._dt = {self._kr: np.dtype(sampling_dtype).type, self._ki: np.float64}
        self._use_full_fisher = bool(use_full_fisher)
        super().__init__(Operator.identity_operator(dom).ducktape(self._kr), lambda x: makeOp(x[self._ki].sqrt()))

    def apply(self, x):
        self._check_input(x)
        r, i = x[self._kr], x[self._ki]
        res = (r.vdot(r * i) - i.log().sum()) * 0.5
        if not (is_linearization(x) and x.want_metric):
            return res
        if not self._use_full_fisher:
            return res.add_metric(self.get_metric_at(x.val))
        ival = i.val
        met = MultiField.from_dict({self._kr: ival, self._ki: (ival ** (-2)) * 0.5}, self._domain)
        return res.add_metric(makeOp(met, sampling_dtype=self._dt))

    def get_transformation(self):
        """No global transformation to a Euclidean space exists for this energy; a local one invoking the residual is
        used (reference :436-450)."""
        from .operators import FieldAdapter

        r = FieldAdapter(self._domain[self._kr], self._kr)
        ivar = FieldAdapter(self._domain[self._kr], self._ki)
        f = r.adjoint @ (ivar.sqrt() * r) + ivar.adjoint @ (ivar.log().scale(0.5))
        return self._dt, f


class CategoricalEnergy(LikelihoodEnergyOperator):
    """E(x) = -sum d log x for one-hot data d and probabilities x normalised along `axis` by the caller (reference
    energy_operators.py:795-850)."""

    def __init__(self, d, axis=0):
        if not isinstance(d, Field) or not np.issubdtype(d.dtype, np.integer):
            raise TypeError(f"d needs to be a Field with integer values. Got:\n{d}")
        d_np = d.asnumpy()
        vals = set(np.unique(d_np).tolist())
        if not vals <= {0, 1}:
            raise ValueError(f"d can only contain 0 and 1. Got: {vals}")
        if not np.all(np.sum(d_np, axis=axis) == 1):
            raise ValueError("d must sum to 1 along the category axis (one-hot encoded)")
        self._d, self._axis = d, axis
        self._d_float = {}
        self._domain = DomainTuple.make(d.domain)
        super().__init__(Adder(d, neg=True), lambda x: self.get_metric_at(x).get_sqrt())

    def apply(self, x):
        self._check_input(x)
        xv = x.val if is_linearization(x) else x
        key = (xv.device_id, xv.dtype)
        if key not in self._d_float:
            self._d_float[key] = self._d.at(xv.device_id).astype(xv.dtype)
        res = -x.log().vdot(self._d_float[key])
        if not (is_linearization(x) and x.want_metric):
            return res
        return res.add_metric(self.get_metric_at(x.val))

    def get_transformation(self):
        return np.float64, Operator.identity_operator(self._domain).sqrt().scale(2.0)


class AveragedEnergy(EnergyOperator):
    """Mean of an energy over residual samples, E(x) = 1/n sum_s h(x + v_s) (reference energy_operators.py:934-971)."""

    def __init__(self, h, res_samples):
        self._h, self._domain = h, h.domain
        self._res_samples = tuple(res_samples)

    def apply(self, x):
        self._check_input(x)
        dev = (x.val if is_linearization(x) else x).device_id
        res = None
        for v in self._res_samples:
            term = self._h(x + v.at(dev))
            res = term if res is None else res + term
        return res * (1.0 / len(self._res_samples))

    def get_transformation(self):
        dtp, trafo = self._h.get_transformation()
        tot = None
        for v in self._res_samples:
            term = trafo @ Adder(v)
            tot = term if tot is None else tot + term
        return dtp, tot.scale(1.0 / np.sqrt(len(self._res_samples)))


class InverseGammaEnergy(LikelihoodEnergyOperator):
    """E(x) = sum (alpha+1) ln x + beta / x: the likelihood of a variance x given beta = |s|^2 / 2 of a field s with that
    variance (reference energy_operators.py:643-701).  alpha: a scalar or a Field."""

    def __init__(self, beta, alpha=-0.5):
        if not isinstance(beta, Field):
            raise TypeError(f"beta needs to be a `Field`. Got:\n{beta}")
        if beta.dtype != np.float64:
            raise TypeError(f"beta.dtype needs to be float64. Got: {beta.dtype}")
        self._domain = DomainTuple.make(beta.domain)
        self._beta = beta
        if np.isscalar(alpha):
            alpha = full(beta.domain, float(alpha))
        elif not isinstance(alpha, Field):
            raise TypeError(f"alpha needs to be a `Field`. Got:\n{alpha}")
        self._alphap1 = alpha + 1.0
        # residual 2 beta (a constant, as in the reference) with the metric x^-1 in data space
        super().__init__(_ConstantFieldOperator(self._domain, beta * 2.0), lambda x: makeOp(x.reciprocal().sqrt()))

    def apply(self, x):
        self._check_input(x)
        dev = (x.val if is_linearization(x) else x).device_id
        res = x.log().vdot(self._alphap1.at(dev)) + x.reciprocal().vdot(self._beta.at(dev))
        if not (is_linearization(x) and x.want_metric):
            return res
        return res.add_metric(self.get_metric_at(x.val))

    def get_transformation(self):
        return np.float64, makeOp(self._alphap1.sqrt()) @ Operator.identity_operator(self._domain).log()


class _ConstantFieldOperator(Operator):
    """x -> a fixed Field (the data residual of InverseGammaEnergy; zero Jacobian)."""

    def __init__(self, domain, value):
        self._domain = DomainTuple.make(domain)
        self._target = value.domain
        self._value = value

    def apply(self, x):
        self._check_input(x)
        if is_linearization(x):
            from .operators import NullOperator

            return x.new(self._value.at(x.val.device_id), NullOperator(self._domain, self._target))
        return self._value.at(x.device_id)


class StandardHamiltonian(EnergyOperator):
    """likelihood energy + 1/2 |x|^2; its metric can draw samples through CG (energy_operators.py:890-931)."""

    def __init__(self, lh, ic_samp=None, prior_sampling_dtype=None):
        self._lh = lh
        self._prior = GaussianEnergy(data=None, domain=lh.domain, sampling_dtype=prior_sampling_dtype)
        self._prior_sampling_dtype = prior_sampling_dtype
        self._ic_samp = ic_samp
        self._domain = lh.domain

    def apply(self, x):
        self._check_input(x)
        lhx, prx = self._lh(x), self._prior(x)
        if not (is_linearization(x) and x.want_metric) or self._ic_samp is None:
            return lhx + prx
        met = SamplingEnabler(lhx.metric, prx.metric, self._ic_samp)
        return (lhx + prx).add_metric(met)

    def _simplify_for_constant_input_nontrivial(self, c_inp):
        """Hamiltonian of the variable keys: likelihood with the constants inserted + prior on what is left
        (reference energy_operators.py:923-931)."""
        out, lh1 = self._lh.simplify_for_constant_input(c_inp)
        psdt = self._prior_sampling_dtype
        if isinstance(psdt, dict):
            psdt = {k: v for k, v in psdt.items() if k in lh1.domain.keys()}
        return out, StandardHamiltonian(lh1, self._ic_samp, psdt)

    @property
    def prior_energy(self):
        return self._prior

    @property
    def likelihood_energy(self):
        return self._lh

    @property
    def iteration_controller(self):
        return self._ic_samp

    def __repr__(self):
        return "StandardHamiltonian:\n  Likelihood energy:\n    " + repr(self._lh).replace("\n", "\n    ")
