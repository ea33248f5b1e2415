"""Likelihood energies and the standard Hamiltonian.

Same public classes, arguments and error behaviour as reference nifty/cl/operators/energy_operators.py (EnergyOperator
:36-41, LikelihoodEnergyOperator :44-164, likelihood chains and sums :166-303, Squared2NormOperator / QuadraticFormOperator
:306-352, VariableCovarianceGaussianEnergy :355-450, GaussianEnergy :517-595, PoissonianEnergy :617-640, InverseGammaEnergy
:643-701, StudentTEnergy :704-746, BernoulliEnergy :749-792, CategoricalEnergy :795-850, StandardHamiltonian :890-931,
AveragedEnergy :934-971), organised differently:

 * every likelihood with a FORMULA  E(x) = sum_i e(x_i; constants)  derives from `_FormulaLikelihood`: the subclass states the
   energy expression (`_energy`, written once and evaluated on Fields and Linearizations alike) and the transformation whose
   Jacobian gives the Fisher metric; input checks, moving the constants to the input's device and attaching the metric are
   done in one place;
 * a likelihood applied to a model, or scaled, is a `_LikelihoodChain` that keeps its three parts (factor, likelihood,
   model) instead of re-reading them from an operator chain;
 * a sum of likelihoods labels the data space of every summand through one helper used for residuals, data metrics and
   transformations.
"""
from functools import reduce

import numpy as np

from .domains import DomainTuple, MultiDomain, makeDomain
from .field import Field, MultiField, full
from .operators import (Adder, EndomorphicOperator, FieldAdapter, LinearOperator, Linearization, NullOperator, Operator,
                        PrependKey, SamplingEnabler, SandwichOperator, ScalingOperator, VdotOperator, _OpChain, _same_domain,
                        is_linearization, is_operator, makeOp, sum_of_operators)


def _value_of(x):
    """the Field / MultiField an operator input carries"""
    return x.val if is_linearization(x) else x


def _wants_metric(x):
    return is_linearization(x) and x.want_metric


class EnergyOperator(Operator):
    """Operator with scalar target."""

    _target = DomainTuple.scalar_domain()


class LikelihoodEnergyOperator(EnergyOperator):
    """Negative log-likelihood: an energy that knows its data residual and the square root of its Fisher metric in data
    space, and whose metric is the Fisher metric (pulled back through `get_transformation`)."""

    def __init__(self, data_residual, sqrt_data_metric_at):
        if not (data_residual is None or is_operator(data_residual)):
            raise TypeError(f"{data_residual} is not an operator")
        self._res, self._sqrt_data_metric_at, self._name = data_residual, sqrt_data_metric_at, None

    def normalized_residual(self, x):
        return (self._sqrt_data_metric_at(x) @ self._res).force(x)

    @property
    def data_domain(self):
        return self._res.target if self._res is not None else None

    def get_transformation(self):
        raise NotImplementedError("`get_transformation` not implemented (yet) for this operator")

    def get_metric_at(self, x):
        """Fisher metric at x: J^T J for the Jacobian J of the transformation into Euclidean coordinates."""
        dtype, to_euclidean = self.get_transformation()
        return SandwichOperator.make(to_euclidean(Linearization.make_var(x)).jac, sampling_dtype=dtype)

    def _metric_from_transformation(self):
        """x -> square root of the data-space metric, for likelihoods whose data space is their domain"""
        return lambda x: self.get_metric_at(x).get_sqrt()

    # a likelihood composed with anything stays a likelihood; likelihoods add up to likelihoods
    def __matmul__(self, other):
        return _LikelihoodChain(self, other)

    def __rmatmul__(self, other):
        return _LikelihoodChain(other, self)

    def __add__(self, other):
        return _LikelihoodSum.make([self, other])

    def __radd__(self, other):
        return _LikelihoodSum.make([other, self])

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
    """`likelihood @ model` or `ScalingOperator @ likelihood`, kept as its parts: `_factor` (None or the number),
    `_inner` (the likelihood) and `_model` (None or the operator feeding it)."""

    def __init__(self, op1, op2):
        scaled = isinstance(op1, ScalingOperator)
        self._factor = op1._factor if scaled else None
        self._inner, self._model = (op2, None) if scaled else (op1, op2)
        self._op = _OpChain.make((op1, op2))
        self._domain = self._op.domain
        residual, sqrt_metric = self._inner._res, self._inner._sqrt_data_metric_at
        if self._model is not None and residual is not None:
            if self._model.target is not residual.domain:
                raise NotImplementedError("likelihood chains need model.target == data-residual domain "
                                          "(PartialExtractor is not implemented)")
            inner, model = self._inner, self._model
            residual, sqrt_metric = residual @ model, (lambda x: inner._sqrt_data_metric_at(model.force(x)))
        elif self._model is not None:
            residual = sqrt_metric = None
        super().__init__(residual, sqrt_metric)
        self.name = self._inner.name

    likelihood = property(lambda self: self._inner)
    # the operator chain feeding the likelihood (used by the fusion pass of optimize_kl)
    model = property(lambda self: self._model)

    def get_transformation(self):
        if (found := self._inner.get_transformation()) is None:
            return None
        stages = [found[1]] if self._factor is None else [found[1].scale(np.sqrt(self._factor))]  # c E: coordinates x sqrt(c)
        return found[0], _OpChain.make(stages + ([] if self._model is None else [self._model]))

    def apply(self, x):
        self._check_input(x)
        return self._op(x)

    def __repr__(self):
        return repr(self._op)


class _LikelihoodSum(LikelihoodEnergyOperator):
    """Sum of likelihoods (several data sets / instruments; reference energy_operators.py:211-303).  The data spaces
    of the summands form a disjoint union: residuals, data metrics and transformations carry the summand's name
    ("Likelihood i" by default) as key (DomainTuple data) or key prefix (MultiDomain data) -- `_label`."""

    def __init__(self, ops, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError
        if len({isinstance(op.domain, DomainTuple) for op in ops}) > 1:
            raise RuntimeError("Some operators have DomainTuple and others have MultiDomain as domain. "
                               "This should not happen.")
        self._ops, self._name = list(ops), None
        names = self._all_names()
        if len(set(names)) < len(names):
            raise ValueError(f"Name collision in likelihoods detected: {names}")
        with_data = [(name, op) for name, op in zip(names, self._ops) if op._res is not None]
        embeds = [self._label(name, Operator.identity_operator(op.data_domain))[0] for name, op in with_data]
        pieces = [emb @ op._res for emb, (_, op) in zip(embeds, with_data)]
        residual = reduce(lambda a, b: a + b, pieces) if pieces else None

        def sqrt_data_metric_at(x):
            blocks = [emb @ op._sqrt_data_metric_at(x) @ emb.adjoint for emb, (_, op) in zip(embeds, with_data)]
            return reduce(lambda a, b: a + b, blocks) if blocks else None

        self._res, self._sqrt_data_metric_at = residual, sqrt_data_metric_at
        self._domain = self._ops[0].domain if residual is None else residual.domain

    @staticmethod
    def _label(name, op, dtype=None):
        """(`op` with its target moved into the labelled disjoint union, the sampling dtypes keyed the same way)"""
        if isinstance(op.target, MultiDomain):
            prefix = name + ": "
            return PrependKey(op.target, prefix) @ op, {prefix + sub: dt for sub, dt in (dtype or {}).items()}
        return op.ducktape_left(name), {name: dtype}

    @classmethod
    def make(cls, ops):
        flat = []
        for op in ops:
            if not isinstance(op, LikelihoodEnergyOperator):
                raise TypeError(f"a likelihood can only be added to another likelihood, got {type(op).__name__}")
            flat.extend(op._ops if isinstance(op, cls) else [op])  # (a sum's own terms are flat already)
        return cls(flat, _callingfrommake=True) if len(flat) != 1 else flat[0]

    def apply(self, x):
        self._check_input(x)
        return sum_of_operators(x, self._ops)

    def get_transformation(self):
        found = [op.get_transformation() for op in self._ops]
        if None in found:
            return None
        labelled = [self._label(name, op, dtype) for name, (dtype, op) in zip(self._all_names(), found)]
        dtypes = {key: dt for _, part in labelled for key, dt in part.items()}
        return dtypes, reduce(lambda a, b: a + b, [op for op, _ in labelled])

    def _get_name(self, i):
        return self._ops[i].name or f"Likelihood {i}"

    def _all_names(self):
        return [self._get_name(i) for i in range(len(self._ops))]

    def __repr__(self):
        return "_LikelihoodSum:\n" + "\n".join(f"  *{self._get_name(ii)}*\n  {oo!r}" for ii, oo in enumerate(self._ops))


class _ScalarProduct(EnergyOperator):
    """x -> weight * <x, K x> (K = identity when no operator is given) with the Jacobian <2 weight K x, .>."""

    def __init__(self, domain, kernel=None, weight=1.0):
        self._domain, self._kernel, self._weight = domain, kernel, weight

    def apply(self, x):
        self._check_input(x)
        point = _value_of(x)
        image = point if self._kernel is None else self._kernel(point)
        value = point.vdot(image)
        value = (value if self._weight == 1.0 else value * self._weight).at(point.device_id)
        if not is_linearization(x):
            return value
        slope = 2.0 * self._weight  # d/dx <x, K x> = 2 K x for the self-adjoint K of a quadratic form
        return x.new(value, VdotOperator(image if slope == 1.0 else image * slope))


class Squared2NormOperator(_ScalarProduct):
    def __init__(self, domain):
        super().__init__(domain)


class QuadraticFormOperator(_ScalarProduct):
    def __init__(self, endo):
        if not isinstance(endo, EndomorphicOperator):
            raise TypeError(f"op must be an EndomorphicOperator.\nGot: {endo}")
        super().__init__(endo.domain, endo, 0.5)
        self._op = endo


def _agreed_domain(*candidates):
    """The one domain all given candidates (None = not given) agree on."""
    found = None
    for cand in candidates:
        if cand is None:
            continue
        cand = makeDomain(cand)
        if found is not None:
            _same_domain(found, cand)
        found = cand
    if found is None:
        raise ValueError("no domain given")
    return found


class GaussianEnergy(LikelihoodEnergyOperator):
    """E(s) = 1/2 (s - d)^dagger N^-1 (s - d)."""

    def __init__(self, data=None, inverse_covariance=None, domain=None, sampling_dtype=None):
        icov = inverse_covariance
        if not (icov is None or isinstance(icov, LinearOperator)):
            raise TypeError(f"inverse_covariance needs to be either None or a LinearOperator, got: {icov}")
        if not (data is None or isinstance(data, (Field, MultiField))):
            raise TypeError(f"data needs to be a (Multi)Field or None, got: {data}")
        self._data = data
        self._domain = _agreed_domain(getattr(icov, "domain", None), getattr(data, "domain", None), domain)
        if icov is not None:
            self._op, self._icov = QuadraticFormOperator(icov), icov
        else:  # unit covariance; samples of the metric take the dtype of the data
            self._op = Squared2NormOperator(self._domain).scale(0.5)
            self._icov = ScalingOperator(self._domain, 1.0, sampling_dtype if data is None else data.dtype)
        residual = Adder(data, neg=True) if data is not None else Operator.identity_operator(self._domain)
        super().__init__(residual, self._metric_from_transformation())

    def _residual(self, x):
        if self._data is None:
            return x
        self._data = self._data.at(_value_of(x).device_id)  # follows the input to its device, once
        return x - self._data

    def apply(self, x):
        self._check_input(x)
        energy = self._op(self._residual(x))
        return energy.add_metric(self._icov) if _wants_metric(x) else energy

    def get_metric_at(self, x):
        return self._icov

    def get_transformation(self):
        return self._icov.sampling_dtype, self._icov.get_sqrt()

    def __repr__(self):
        return "GaussianEnergy"


class _FormulaLikelihood(LikelihoodEnergyOperator):
    """Likelihood given by a formula.  Subclasses provide

        _energy(x, *constants)   the energy, written with the arithmetic Fields and Linearizations share
        get_transformation()     (sampling dtype, map into coordinates in which the Fisher metric is Euclidean)

    and register their constant Fields with `_remember`; `apply` hands them over on the device and in the dtype of the
    input and attaches the Fisher metric when the input asks for one."""

    def _remember(self, **constants):
        self._constants, self._moved = constants, {}

    def _constants_for(self, point, cast):
        where = (point.device_id, point.dtype if cast else None)
        if where not in self._moved:
            moved = [c.at(point.device_id) for c in self._constants.values()]
            self._moved[where] = [c.astype(point.dtype) for c in moved] if cast else moved
        return self._moved[where]

    _cast_constants = False  # True: integer data, used as floats of the input's dtype

    def apply(self, x):
        self._check_input(x)
        energy = self._energy(x, *self._constants_for(_value_of(x), self._cast_constants))
        return energy.add_metric(self.get_metric_at(x.val)) if _wants_metric(x) else energy

    @staticmethod
    def _counts(d, message, allowed=None):
        """Checks that d is an integer-valued data Field; with `allowed` also that it only holds those values, and
        returns it as a numpy array (event data: small; count data stays where it is)."""
        if not (isinstance(d, Field) and np.issubdtype(d.dtype, np.integer)):
            raise TypeError(message)
        if allowed is None:
            return None
        values = d.asnumpy()
        seen = set(np.unique(values).tolist())
        if not seen <= allowed:
            raise ValueError(f"d can only contain 0 and 1. Got: {seen}")
        return values

    def _twice_sqrt(self):
        """x -> 2 sqrt(x): the variance-stabilising map of count-like data"""
        return np.float64, Operator.identity_operator(self._domain).sqrt().scale(2.0)


class PoissonianEnergy(_FormulaLikelihood):
    """E(lambda) = sum(lambda) - d^T log(lambda) for integer counts d."""

    _cast_constants = True

    def __init__(self, d):
        self._counts(d, "data is of invalid data-type; counts need to be integers")
        if bool((d.val < 0).any()):
            raise ValueError("count data is negative and thus can not be Poissonian")
        self._d, self._domain = d, DomainTuple.make(d.domain)
        self._remember(d=d)
        super().__init__(Adder(d, neg=True), self._metric_from_transformation())

    def _energy(self, lam, d):
        return lam.sum() - lam.log().vdot(d)

    get_transformation = _FormulaLikelihood._twice_sqrt


class StudentTEnergy(_FormulaLikelihood):
    """E(f) = (theta+1)/2 sum log(1 + f^2/theta), theta a scalar or a Field (reference energy_operators.py:704-746)."""

    def __init__(self, domain, theta):
        self._domain, self._theta = DomainTuple.make(domain), theta
        self._remember(**({"theta": theta} if isinstance(theta, Field) else {}))
        super().__init__(Operator.identity_operator(self._domain), self._metric_from_transformation())

    def _energy(self, f, theta=None):
        if theta is None:
            return ((f ** 2) * (1.0 / self._theta)).log1p().sum() * ((self._theta + 1.0) / 2.0)
        return makeOp((theta + 1.0) * 0.5)(makeOp(theta.reciprocal())(f ** 2).log1p()).sum()

    def get_transformation(self):
        theta = self._theta if isinstance(self._theta, Field) else full(self._domain, float(self._theta))
        return np.float64, makeOp(((theta + 1.0) / (theta + 3.0)).sqrt())


class BernoulliEnergy(_FormulaLikelihood):
    """E(f) = -d^T log f - (1-d)^T log(1-f) for event data d in {0, 1} (reference energy_operators.py:749-792)."""

    _cast_constants = True

    def __init__(self, d):
        self._counts(d, f"d needs to be a Field with integer values. Got:\n{d}", allowed={0, 1})
        self._d, self._domain = d, DomainTuple.make(d.domain)
        self._remember(d=d)
        super().__init__(Adder(d, neg=True), self._metric_from_transformation())

    def _energy(self, f, d):
        return (1.0 - f).log().vdot(d - 1.0) - f.log().vdot(d)

    def get_transformation(self):
        f = Operator.identity_operator(self._domain)
        odds_against = (ScalingOperator(self._domain, -1.0) + 1.0) * f.reciprocal()  # (1 - f) / f
        return np.float64, odds_against.sqrt().arctan().scale(-2.0)


class CategoricalEnergy(_FormulaLikelihood):
    """E(x) = -sum d log x for one-hot data d and probabilities x normalised along `axis` by the caller (reference
    energy_operators.py:795-850)."""

    _cast_constants = True

    def __init__(self, d, axis=0):
        onehot = self._counts(d, f"d needs to be a Field with integer values. Got:\n{d}", allowed={0, 1})
        if (onehot.sum(axis=axis) != 1).any():
            raise ValueError("d must sum to 1 along the category axis (one-hot encoded)")
        self._d, self._axis, self._domain = d, axis, DomainTuple.make(d.domain)
        self._remember(d=d)
        super().__init__(Adder(d, neg=True), self._metric_from_transformation())

    def _energy(self, x, d):
        return -x.log().vdot(d)

    get_transformation = _FormulaLikelihood._twice_sqrt


class InverseGammaEnergy(_FormulaLikelihood):
    """E(x) = sum (alpha+1) ln x + beta / x: the likelihood of a variance x given beta = |s|^2 / 2 of a field s with that
    variance (reference energy_operators.py:643-701).  alpha: a scalar or a Field."""

    def __init__(self, beta, alpha=-0.5):
        for what, arg, ok in (("beta", beta, isinstance(beta, Field)), ("alpha", alpha, np.isscalar(alpha) or isinstance(alpha, Field))):
            if not ok:
                raise TypeError(f"{what} needs to be a `Field`. Got:\n{arg}")
        if beta.dtype != np.float64:
            raise TypeError(f"beta.dtype needs to be float64. Got: {beta.dtype}")
        self._beta, self._domain = beta, DomainTuple.make(beta.domain)
        self._alphap1 = (full(beta.domain, float(alpha)) if np.isscalar(alpha) else alpha) + 1.0
        self._remember(alphap1=self._alphap1, beta=beta)
        # residual 2 beta (a constant, as in the reference) with the metric x^-1 in data space
        super().__init__(_ConstantFieldOperator(self._domain, beta * 2.0), lambda x: makeOp(x.reciprocal().sqrt()))

    def _energy(self, x, alphap1, beta):
        return x.log().vdot(alphap1) + x.reciprocal().vdot(beta)

    def get_transformation(self):
        return np.float64, makeOp(self._alphap1.sqrt()) @ Operator.identity_operator(self._domain).log()


class VariableCovarianceGaussianEnergy(LikelihoodEnergyOperator):
    """-log of a Gaussian in the residual s with an UNKNOWN diagonal inverse covariance C, both inferred:
    E(s, C) = 1/2 s^T C s - 1/2 tr log C on the MultiDomain {residual_key, inverse_covariance_key}
    (reference energy_operators.py:355-450; real sampling dtypes).  ``use_full_fisher``: the exact Fisher metric
    diag(C, 1/2 C^-2); otherwise the metric of the local transformation used by geoVI."""

    def __init__(self, domain, residual_key, inverse_covariance_key, sampling_dtype, use_full_fisher=True):
        space = DomainTuple.make(domain)
        keys = self._kr, self._ki = str(residual_key), str(inverse_covariance_key)
        self._use_full_fisher = bool(use_full_fisher)
        self._domain = MultiDomain.make(dict.fromkeys(keys, space))
        self._dt = dict(zip(keys, (self._real_type(sampling_dtype), np.float64)))
        super().__init__(Operator.identity_operator(space).ducktape(self._kr), lambda x: makeOp(x[self._ki].sqrt()))

    @staticmethod
    def _real_type(sampling_dtype):
        dt = np.dtype(sampling_dtype)
        if np.issubdtype(dt, np.complexfloating):
            raise NotImplementedError("complex residuals are not supported")
        return dt.type

    def _fisher(self, point):
        """metric at `point`: diag(C, 1/2 C^-2) when the full Fisher metric is asked for"""
        if not self._use_full_fisher:
            return self.get_metric_at(point)
        c = point[self._ki]
        return makeOp(MultiField.from_dict({self._kr: c, self._ki: (c ** (-2)) * 0.5}, self._domain), sampling_dtype=self._dt)

    def apply(self, x):
        self._check_input(x)
        s, c = x[self._kr], x[self._ki]
        energy = (s.vdot(s * c) - c.log().sum()) * 0.5
        return energy.add_metric(self._fisher(x.val)) if _wants_metric(x) else energy

    def get_transformation(self):
        """No global transformation to a Euclidean space exists for this energy; a local one invoking the residual is
        used (reference :436-450): (s, C) -> (sqrt(C) s, 1/2 log C)."""
        s, c = (FieldAdapter(self._domain[self._kr], key) for key in (self._kr, self._ki))
        return self._dt, s.adjoint @ (c.sqrt() * s) + c.adjoint @ c.log().scale(0.5)


class AveragedEnergy(EnergyOperator):
    """Mean of an energy over residual samples, E(x) = 1/n sum_s h(x + v_s) (reference energy_operators.py:934-971)."""

    def __init__(self, h, res_samples):
        self._h, self._domain, self._res_samples = h, h.domain, tuple(res_samples)

    def apply(self, x):
        self._check_input(x)
        here = _value_of(x).device_id
        shifted = [self._h(x + v.at(here)) for v in self._res_samples]
        return reduce(lambda a, b: a + b, shifted) * (1.0 / len(shifted))

    def get_transformation(self):
        dtype, to_euclidean = self._h.get_transformation()
        stacked = reduce(lambda a, b: a + b, [to_euclidean @ Adder(v) for v in self._res_samples])
        return dtype, stacked.scale(len(self._res_samples) ** -0.5)


class _ConstantFieldOperator(Operator):
    """x -> a fixed Field (the data residual of InverseGammaEnergy; zero Jacobian)."""

    def __init__(self, domain, value):
        self._domain, self._target, self._value = DomainTuple.make(domain), value.domain, value

    def apply(self, x):
        self._check_input(x)
        constant = self._value.at(_value_of(x).device_id)
        return x.new(constant, NullOperator(self._domain, self._target)) if is_linearization(x) else constant


class StandardHamiltonian(EnergyOperator):
    """likelihood energy + 1/2 |x|^2 (a standard-normal prior); with an iteration controller its metric can draw samples
    from its inverse through a CG (reference energy_operators.py:890-931)."""

    def __init__(self, lh, ic_samp=None, prior_sampling_dtype=None):
        self._lh, self._ic_samp, self._prior_sampling_dtype = lh, ic_samp, prior_sampling_dtype
        self._domain = lh.domain
        self._prior = GaussianEnergy(data=None, domain=self._domain, sampling_dtype=prior_sampling_dtype)

    def apply(self, x):
        self._check_input(x)
        parts = [energy(x) for energy in (self._lh, self._prior)]
        total = parts[0] + parts[1]
        if self._ic_samp is None or not _wants_metric(x):
            return total
        return total.add_metric(SamplingEnabler(parts[0].metric, parts[1].metric, self._ic_samp))

    def _simplify_for_constant_input_nontrivial(self, c_inp):
        """Hamiltonian of the variable keys: likelihood with the constants inserted + prior on what is left
        (reference energy_operators.py:923-931)."""
        constant_part, reduced_lh = self._lh.simplify_for_constant_input(c_inp)
        return constant_part, StandardHamiltonian(reduced_lh, self._ic_samp, self._prior_dtypes_on(reduced_lh.domain))

    def _prior_dtypes_on(self, domain):
        """the prior's sampling dtypes restricted to the keys of `domain`"""
        dtypes = self._prior_sampling_dtype
        return {key: dtypes[key] for key in domain.keys() if key in dtypes} if isinstance(dtypes, dict) else dtypes

    prior_energy = property(lambda self: self._prior)
    likelihood_energy = property(lambda self: self._lh)
    iteration_controller = property(lambda self: self._ic_samp)

    def __repr__(self):
        return "StandardHamiltonian:\n  Likelihood energy:\n    " + repr(self._lh).replace("\n", "\n    ")
