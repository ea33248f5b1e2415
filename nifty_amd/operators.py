"""Operator algebra, forward-mode linearisation and the linear building blocks of the MGVI path.

Counterpart of reference nifty/cl/operators/{operator,linear_operator,operator_adapter,chain_operator,
sum_operator,scaling_operator,diagonal_operator,sandwich_operator,adder,contraction_operator,
simple_linear_operators,block_diagonal_operator,distributors,harmonic_operators,sampling_enabler}.py
and nifty/cl/linearization.py -- same names, argument meaning and error behaviour at the seam
``LinearOperator.apply(x, mode)`` / ``Operator.apply(x)``, re-implemented on torch-tensor Fields whose
device arithmetic runs in libniftyk kernels.
"""
from numbers import Number

import numpy as np
import torch

from . import backend as B
from .domains import DomainTuple, MultiDomain, PowerSpace, RGSpace, UnstructuredDomain, makeDomain
from .field import Field, MultiField, full, is_fieldlike


def _same_domain(a, b):
    if a is not b:
        raise ValueError(f"Domain mismatch:\n{a!r}\nvs\n{b!r}")


def is_linearization(x):
    return isinstance(x, Linearization)


def is_operator(x):
    return isinstance(x, Operator)


def _operand_kind(x):
    """What an operator is being applied to / combined with: 'lin', 'field', 'op', 'number' or None."""
    if isinstance(x, Linearization):
        return "lin"
    if is_fieldlike(x):
        return "field"
    if isinstance(x, Operator):
        return "op"
    if isinstance(x, Number):
        return "number"
    return None


def _identity_on(dom):
    """Identity as a linear operator: one scaling by 1 per DomainTuple (a block per key of a MultiDomain)."""
    dom = makeDomain(dom)
    if isinstance(dom, MultiDomain):
        return BlockDiagonalOperator(dom, {key: ScalingOperator(sub, 1.0) for key, sub in dom.items()})
    return ScalingOperator(dom, 1.0)


def _compose(outer, inner):
    """`outer` after `inner`.  When the seam does not match (MultiDomains with different key sets) every key only one
    side knows is routed around the other side by an identity block, so that the composition acts on / returns the union
    (reference semantics of Operator.__matmul__ / partial_insert, operators/operator.py:177-212)."""
    if inner.target is outer.domain:
        for trivial, other in ((inner, outer), (outer, inner)):
            if trivial.isIdentity():
                return other
        return _OpChain.make((outer, inner))
    seam = (outer.domain, inner.target)
    if not all(isinstance(d, MultiDomain) for d in seam):
        raise TypeError("operators with different DomainTuples at the seam cannot be composed")
    union = MultiDomain.union(list(seam))

    def routed(op, own, other):
        missing = [k for k in other.keys() if k not in own.keys()]
        return op if not missing else op + _identity_on(MultiDomain.make({k: union[k] for k in missing}))

    return routed(outer, outer.domain, inner.target) @ routed(inner, inner.target, outer.domain)


# ================================================================================================
# Operator
# ================================================================================================
class Operator:
    """Possibly nonlinear map between (Multi)Domains (reference operators/operator.py:30-430).  The arithmetic of
    operators is table-driven (`_ARITHMETIC`): an operand is classified once (_operand_kind) and the table says how
    `self (+|-|*) operand` is built."""

    @property
    def domain(self):
        return self._domain

    @property
    def target(self):
        return self._target

    def isIdentity(self):
        return False

    # -- application ---------------------------------------------------------------------------
    def apply(self, x):
        raise NotImplementedError

    def force(self, x):
        return self.apply(x.extract(self.domain))

    def _check_input(self, x):
        kind = _operand_kind(x)
        if kind not in ("lin", "field"):
            raise TypeError("operators act on Fields, MultiFields or Linearizations")
        if kind == "lin" and not x.jac.isIdentity():
            raise ValueError("apply() expects a Linearization with trivial Jacobian")
        _same_domain(self._domain, x.domain)

    def __call__(self, x):
        try:
            return _CALL[_operand_kind(x)](self, x)
        except KeyError:
            raise TypeError(f"cannot apply operator to {type(x)}") from None

    # -- algebra ---------------------------------------------------------------------------------
    def __matmul__(self, x):
        from .energy_operators import LikelihoodEnergyOperator

        if not is_operator(x) or isinstance(x, LikelihoodEnergyOperator):
            return NotImplemented
        return _compose(self, x)

    def partial_insert(self, x):
        if not is_operator(x) or not isinstance(self.domain, MultiDomain) or not isinstance(x.target, MultiDomain):
            raise TypeError("partial_insert needs operators on MultiDomains")
        return _compose(self, x)

    identity_operator = staticmethod(_identity_on)

    def scale(self, factor):
        if not isinstance(factor, Number):
            raise TypeError(".scale() takes a number as input")
        return self if factor == 1 else ScalingOperator(self.target, factor)(self)

    def __neg__(self):
        return self.scale(-1)

    def _arithmetic(self, other, sym):
        rule = _ARITHMETIC.get((sym, _operand_kind(other)))
        return NotImplemented if rule is None else rule(self, other)

    def __mul__(self, x):
        return self._arithmetic(x, "*")

    def __add__(self, x):
        return self._arithmetic(x, "+")

    def __sub__(self, x):
        return self._arithmetic(x, "-")

    __rmul__ = __mul__
    __radd__ = __add__

    def __rsub__(self, x):
        return x + (-self)

    def __truediv__(self, x):
        kind = _operand_kind(x)
        if kind == "number":
            return self.scale(1.0 / x)
        return self * x.reciprocal() if kind in ("op", "field") else NotImplemented

    def __rtruediv__(self, x):
        return self.reciprocal() * x

    def __abs__(self):
        return self.ptw("abs")

    def __pow__(self, power):
        """op ** number pointwise; op ** op = exp(power * log(op)) (operator.py:288-300)"""
        if isinstance(power, Number):
            return self.ptw("power", power)
        return (power * self.log()).exp() if _operand_kind(power) in ("op", "field") else NotImplemented

    def __rpow__(self, base):
        if isinstance(base, Number):
            return self.scale(np.log(base)).exp()
        return (base.log() * self).exp() if _operand_kind(base) in ("op", "field") else NotImplemented

    def __getitem__(self, key):
        if not isinstance(self.target, MultiDomain):
            raise TypeError("Only Operators with a MultiDomain as target can be subscripted.")
        return ducktape(None, self, key) @ self

    def simplify_for_constant_input(self, c_inp):
        """(constant output or None, operator on the remaining keys) for a partially constant input
        (reference operator.py:393-441).  No algebraic simplification is attempted: the constants are inserted."""
        if c_inp is None or (isinstance(c_inp, MultiField) and len(c_inp.keys()) == 0):
            return None, self
        if not isinstance(self.domain, MultiDomain) or not isinstance(c_inp, MultiField):
            raise ValueError("partially constant input needs a MultiDomain")
        if not set(c_inp.keys()) <= set(self.domain.keys()):
            raise ValueError
        from .energy_operators import EnergyOperator

        if c_inp.domain is self.domain and not isinstance(self, EnergyOperator):
            return None, ConstantOperator(self(c_inp))  # nothing variable is left (operator.py:413-422)
        return self._simplify_for_constant_input_nontrivial(c_inp)

    def _simplify_for_constant_input_nontrivial(self, c_inp):
        return None, self @ InsertionOperator(self.domain, c_inp)

    def ptw(self, op, *args, **kwargs):
        return _OpChain.make((_FunctionApplier(self.target, op, *args, **kwargs), self))

    def ptw_pre(self, op, *args, **kwargs):
        return _OpChain.make((self, _FunctionApplier(self.domain, op, *args, **kwargs)))

    def apply_to_random_sample(self, **kwargs):
        """self(from_random(self.domain, **kwargs)) (operator.py:440-460)"""
        from .field import from_random

        return self(from_random(self.domain, **kwargs))

    # a plain operator carries no value (Fields and Linearizations do: operator.py:43-75)
    val = jac = metric = None
    want_metric = False

    @property
    def real(self):
        return Realizer(self.target)(self)

    @property
    def imag(self):
        from .selection_operators import Imaginizer

        return Imaginizer(self.target)(self)

    def sum(self, spaces=None):
        return ContractionOperator(self.target, spaces)(self)

    def vdot(self, other):
        if not is_operator(other):
            raise TypeError
        return (self * other).sum()

    def broadcast(self, index, space):
        if not isinstance(self.target, DomainTuple):
            raise RuntimeError("Broadcasting works only on DomainTuples")
        tgt = list(self.target)
        tgt.insert(index, space)
        return ContractionOperator(tgt, index).adjoint(self)

    def ducktape(self, name):
        """input from key `name` of a MultiField, or -- a domain instead of a string -- from a reshaped field on that domain
        (operator.py:351-373)"""
        if isinstance(name, str):
            return self @ ducktape(self, None, name)
        from .selection_operators import DomainChangerAndReshaper

        return self @ DomainChangerAndReshaper(makeDomain(name), self.domain)

    def ducktape_left(self, name):
        if isinstance(name, str):
            return ducktape(None, self.target, name)(self)
        from .selection_operators import DomainChangerAndReshaper

        return DomainChangerAndReshaper(self.target, DomainTuple.make(name))(self)

    def transpose(self, indices):
        from .selection_operators import TransposeOperator

        return TransposeOperator(self.target, indices)(self)

    def conjugate(self):
        from .selection_operators import ConjugationOperator

        return ConjugationOperator(self.target)(self)

    def integrate(self, spaces=None):
        return IntegrationOperator(self.target, spaces)(self)

    def squeeze(self, aggressive=False):
        from .selection_operators import SqueezeOperator

        return SqueezeOperator(self.target, aggressive)(self)

    def __repr__(self):
        return self.__class__.__name__


def _shifted(op, constant, negate):
    """x -> op(x) +/- constant (a number fills the target)."""
    if isinstance(constant, Number):
        constant = full(op.target, float(constant))
    return Adder(constant, neg=negate) @ op


# operand kind -> what `op(operand)` means
_CALL = {
    "field": lambda op, x: op.apply(x),
    # differentiate at the value, then chain the incoming Jacobian behind the result
    "lin": lambda op, x: op.apply(x.trivial_jac()).prepend_jac(x.jac),
    "op": lambda op, x: op @ x,
}

# (symbol, operand kind) -> builder of `op (symbol) operand`
_ARITHMETIC = {
    ("*", "op"): lambda op, o: _OpProd(op, o),
    ("*", "number"): lambda op, o: op.scale(o),
    ("*", "field"): lambda op, o: makeOp(o) @ op,
    ("+", "op"): lambda op, o: _OpSum(op, o),
    ("+", "number"): lambda op, o: _shifted(op, o, False),
    ("+", "field"): lambda op, o: _shifted(op, o, False),
    ("-", "op"): lambda op, o: _OpSum(op, -o),
    ("-", "number"): lambda op, o: _shifted(op, o, True),
    ("-", "field"): lambda op, o: _shifted(op, o, True),
}

_POINTWISE_METHODS = ("exp", "log", "sqrt", "tanh", "sigmoid", "reciprocal", "log1p", "expm1", "abs", "sin", "cos", "arctan", "tan",
                      "sinh", "cosh", "log10", "sinc", "sign", "unitstep", "softplus", "absolute", "power", "clip", "exponentiate")


def _install_pointwise(cls, names):
    """cls.exp(), cls.log(), ... = cls.ptw("exp"), ..."""
    def bound_to(name):
        def method(self, *args, **kwargs):
            return self.ptw(name, *args, **kwargs)

        method.__name__ = name
        return method

    for name in names:
        if name not in cls.__dict__:  # (an explicit method, e.g. Linearization.clip, stays)
            setattr(cls, name, bound_to(name))


_install_pointwise(Operator, _POINTWISE_METHODS)


def _install_pointwise_pre(cls, names):
    """cls.exp_pre(), ... = cls.ptw_pre("exp"), ...: the function applied to the INPUT (operator.py:525-535)"""
    def bound_to(name):
        def method(self, *args, **kwargs):
            return self.ptw_pre(name, *args, **kwargs)

        method.__name__ = name + "_pre"
        return method

    for name in names:
        setattr(cls, name + "_pre", bound_to(name))


_install_pointwise_pre(Operator, _POINTWISE_METHODS)


class _FunctionApplier(Operator):
    """Pointwise nonlinearity (reference operators/operator.py:476-503)."""

    def __init__(self, domain, funcname, *args, **kwargs):
        self._domain = self._target = makeDomain(domain)
        self._funcname, self._args, self._kwargs = funcname, args, kwargs

    def apply(self, x):
        self._check_input(x)
        return x.ptw(self._funcname, *self._args, **self._kwargs)

    def __repr__(self):
        return f"_FunctionApplier ('{self._funcname}')"


def _indented(title, parts):
    return title + ":\n" + "\n".join("  " + repr(p).replace("\n", "\n  ") for p in parts)


class _OpChain(Operator):
    """op_0 o op_1 o ... o op_n (the LAST one acts first); nested chains are spliced by make()."""

    def __init__(self, ops, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError
        self._ops = tuple(ops)
        for later, earlier in zip(self._ops, self._ops[1:]):
            _same_domain(later.domain, earlier.target)
        self._domain, self._target = self._ops[-1].domain, self._ops[0].target

    @classmethod
    def make(cls, ops):
        links = tuple(link for op in ops for link in (op._ops if isinstance(op, _OpChain) else (op,)))
        return links[0] if len(links) == 1 else cls(links, _callingfrommake=True)

    def apply(self, x):
        self._check_input(x)
        for op in self._ops[::-1]:
            x = op(x)
        return x

    def __repr__(self):
        return _indented("_OpChain", self._ops)


def domain_union(domains):
    domains = list(domains)
    if any(isinstance(d, MultiDomain) for d in domains):
        return MultiDomain.union(domains)
    for d in domains[1:]:
        _same_domain(domains[0], d)
    return domains[0]


def _evaluate_terms(x, ops):
    """[op(restriction of x to op.domain) for op in ops]: plain values for a field, Linearizations (each w.r.t. its own
    restriction, carrying x's want_metric) for a Linearization."""
    if not is_linearization(x):
        return [op.force(x) for op in ops]
    return [op(Linearization.make_var(x.val.extract(op.domain), x.want_metric)) for op in ops]


def sum_of_operators(x, ops):
    """sum_i ops[i](x) where every operator sees the part of x it is defined on (reference operator.py:619-641).  The
    sum rule of values, Jacobians and metrics is the one of Linearization.__add__; a metric survives only if every term
    brings one."""
    terms = _evaluate_terms(x, ops)
    total = terms[0]
    for term in terms[1:]:
        total = total + term if is_linearization(x) else total.flexible_addsub(term, False)
    return x.new(total.val, total.jac, total.metric) if is_linearization(x) else total


class _OpProd(Operator):
    """Pointwise product of two operators (reference operator.py:555-600); the product rule is Linearization.__mul__."""

    def __init__(self, op1, op2):
        if op1.target != op2.target:
            raise ValueError("target mismatch")
        self._op1, self._op2 = op1, op2
        self._domain, self._target = domain_union((op1.domain, op2.domain)), op1.target

    def apply(self, x):
        self._check_input(x)
        left, right = _evaluate_terms(x, (self._op1, self._op2))
        prod = left * right
        return x.new(prod.val, prod.jac) if is_linearization(x) else prod

    def __repr__(self):
        return _indented("_OpProd", (self._op1, self._op2))


class _OpSum(Operator):
    def __init__(self, op1, op2):
        self._op1, self._op2 = op1, op2
        self._domain, self._target = (domain_union((op1.domain, op2.domain)), domain_union((op1.target, op2.target)))

    def apply(self, x):
        self._check_input(x)
        return sum_of_operators(x, (self._op1, self._op2))

    _apply_operator_sum = staticmethod(sum_of_operators)

    def __repr__(self):
        return _indented("_OpSum", (self._op1, self._op2))


# ================================================================================================
# Linearization (forward-mode AD object)
# ================================================================================================
class Linearization:
    """Value + Jacobian (+ metric) of an operator at a point (reference linearization.py:26-435).

    All arithmetic funnels into two rules: `_plus` (sum rule; metrics add, and only if both sides have one) and `_times`
    (product rule; no metric survives a product)."""

    def __init__(self, val, jac, metric=None, want_metric=False):
        _same_domain(val.domain, jac.target)
        self._val, self._jac, self._metric, self._want_metric = val, jac, metric, want_metric

    domain = property(lambda self: self._jac.domain)
    target = property(lambda self: self._jac.target)
    val = property(lambda self: self._val)
    jac = property(lambda self: self._jac)
    want_metric = property(lambda self: self._want_metric)
    metric = property(lambda self: self._metric)
    device_id = property(lambda self: self._val.device_id)

    # -- constructors ------------------------------------------------------------------------------
    def new(self, val, jac, metric=None):
        return Linearization(val, jac, metric, self._want_metric)

    @staticmethod
    def make_var(field, want_metric=False):
        return Linearization(field, ScalingOperator(field.domain, 1.0), want_metric=want_metric)

    @staticmethod
    def make_const(field, want_metric=False):
        return Linearization(field, NullOperator(field.domain, field.domain), want_metric=want_metric)

    @staticmethod
    def make_const_empty_input(field, want_metric=False):
        """constant w.r.t. NOTHING: the Jacobian's input domain is the empty MultiDomain (linearization.py:375-400)"""
        return Linearization(field, NullOperator(MultiDomain.make({}), field.domain), want_metric=want_metric)

    @staticmethod
    def make_partial_var(field, constants, want_metric=False):
        """variable in every key of a MultiField except `constants` (unit / zero blocks; linearization.py:403-438)"""
        if len(constants) == 0:
            return Linearization.make_var(field, want_metric)
        blocks = {key: ScalingOperator(dom, 0.0 if key in constants else 1.0) for key, dom in field.domain.items()}
        return Linearization(field, BlockDiagonalOperator(field.domain, blocks), want_metric=want_metric)

    def at(self, device_id):
        """the value moved to `device_id`; the Jacobian is not changed (linearization.py:150-164)"""
        return self.new(self._val.at(device_id), self._jac)

    def scale(self, factor):
        return self if factor == 1 else self._times(factor)

    def trivial_jac(self):
        return Linearization.make_var(self._val, self._want_metric)

    def add_metric(self, metric):
        return self.new(self._val, self._jac, metric)

    def with_want_metric(self):
        return Linearization(self._val, self._jac, self._metric, True)

    def prepend_jac(self, jac):
        """Chain rule: this linearisation was taken w.r.t. a quantity whose own Jacobian is `jac`."""
        if jac.isIdentity():
            return self
        pulled_back = None if self._metric is None else SandwichOperator.make(jac, self._metric)
        return self.new(self._val, jac if self._jac.isIdentity() else self._jac @ jac, pulled_back)

    @property
    def gradient(self):
        return self._jac.adjoint_times(Field.scalar(1.0).at(self._val.device_id))

    # -- the two rules -------------------------------------------------------------------------------
    def _plus(self, other, minus):
        kind = _operand_kind(other)
        if kind in ("field", "number") or np.isscalar(other):  # a constant shifts the value only
            return self.new(self._val - other if minus else self._val + other, self._jac, self._metric)
        if kind != "lin":
            return NotImplemented
        both = self._metric is not None and other._metric is not None
        return self.new(self._val.flexible_addsub(other._val, minus), self._jac._myadd(other._jac, minus),
                        self._metric._myadd(other._metric, minus) if both else None)

    _myadd = _plus

    def _times(self, other):
        kind = _operand_kind(other)
        if kind == "number" or np.isscalar(other):
            if other == 1:
                return self
            return self.new(self._val * other, self._jac.scale(other), None if self._metric is None else self._metric.scale(other))
        if kind not in ("field", "lin"):
            return NotImplemented
        _same_domain(self.target, other.domain if kind == "field" else other.target)
        if kind == "field":
            return self.new(self._val * other, makeOp(other)(self._jac))
        cross = makeOp(other._val)(self._jac)._myadd(makeOp(self._val)(other._jac), False)  # v dJ_u + u dJ_v
        return self.new(self._val * other._val, cross)

    def __add__(self, o): return self._plus(o, False)
    def __sub__(self, o): return self._plus(o, True)
    def __rsub__(self, o): return (-self)._plus(o, False)
    __radd__ = __add__
    __mul__ = __rmul__ = _times

    def __neg__(self):
        if self._metric is not None:
            raise RuntimeError("Cannot negate operators with metric")
        return self.new(-self._val, -self._jac)

    def __truediv__(self, other):
        return self._times(1.0 / other if np.isscalar(other) else other.ptw("reciprocal"))

    def __rtruediv__(self, other):
        return self.ptw("reciprocal")._times(other)

    def __pow__(self, power):
        if np.isscalar(power):
            return self.ptw("power", power)
        return (power * self.log()).exp() if _operand_kind(power) in ("lin", "field") else NotImplemented

    def __rpow__(self, base):
        if np.isscalar(base):
            return (self * float(np.log(base))).exp()
        return (base.log() * self).exp() if _operand_kind(base) == "field" else NotImplemented

    def __abs__(self):
        return self.ptw("abs")

    def ducktape(self, name):
        raise RuntimeError("ducktape works only on operators")

    def ducktape_left(self, name):
        if isinstance(name, str):
            return ducktape(None, self.target, name)(self)
        from .selection_operators import DomainChangerAndReshaper

        return DomainChangerAndReshaper(self.target, DomainTuple.make(name))(self)

    def broadcast(self, index, space):
        if not isinstance(self.target, DomainTuple):
            raise RuntimeError("Broadcasting works only on DomainTuples")
        tgt = list(self.target)
        tgt.insert(index, space)
        return ContractionOperator(tgt, index).adjoint(self)

    def transpose(self, indices):
        from .selection_operators import TransposeOperator

        return TransposeOperator(self.target, indices)(self)

    def conjugate(self):
        from .selection_operators import ConjugationOperator

        return ConjugationOperator(self.target)(self)

    def integrate(self, spaces=None):
        return IntegrationOperator(self.target, spaces)(self)

    # -- structure -----------------------------------------------------------------------------------
    def __getitem__(self, name):
        if not isinstance(self.target, MultiDomain):
            raise TypeError("not subscriptable")
        return self.new(self._val[name], ducktape(None, self._jac.target, name)(self._jac))

    @property
    def real(self):
        return self.new(self._val.real, Realizer(self._jac.target)(self._jac))

    @property
    def imag(self):
        from .selection_operators import Imaginizer

        return self.new(self._val.imag, Imaginizer(self._jac.target)(self._jac))

    def vdot(self, other):
        if is_fieldlike(other):
            return self.new(self._val.vdot(other.at(self._val.device_id)), VdotOperator(other)(self._jac))
        return self.new(self._val.vdot(other._val),
                        VdotOperator(self._val)(other._jac) + VdotOperator(other._val)(self._jac))

    def sum(self, spaces=None):
        total = self._val.sum(spaces) if spaces is not None else Field.scalar(self._val.s_sum()).at(self.device_id)
        return self.new(total, ContractionOperator(self._jac.target, spaces)(self._jac))

    def ptw(self, op, *args, **kwargs):
        f, df = self._val.ptw_with_deriv(op, *args, **kwargs)
        return self.new(f, makeOp(df)(self._jac))

    def clip(self, a_min=None, a_max=None):
        return self.ptw("clip", a_min, a_max)


_install_pointwise(Linearization, _POINTWISE_METHODS)


# ================================================================================================
# LinearOperator
# ================================================================================================
def _mode_index(mode):
    """0 TIMES, 1 ADJOINT_TIMES, 2 INVERSE_TIMES, 3 ADJOINT_INVERSE_TIMES for the one-hot mode, else -1.  The index is a
    two-bit number (bit 0: adjoint, bit 1: inverse), so viewing an operator through a transformation is an XOR."""
    return mode.bit_length() - 1 if mode in (1, 2, 4, 8) else -1


def _mode_seen_through(mode, trafo):
    return 1 << (_mode_index(mode) ^ trafo)


class LinearOperator(Operator):
    """Linear map with TIMES / ADJOINT_TIMES / INVERSE_TIMES / ADJOINT_INVERSE_TIMES modes
    (reference operators/linear_operator.py:24-262)."""

    TIMES, ADJOINT_TIMES, INVERSE_TIMES, ADJOINT_INVERSE_TIMES = 1, 2, 4, 8
    INVERSE_ADJOINT_TIMES = 8
    ADJOINT_BIT, INVERSE_BIT = 1, 2
    _backwards = 6   # modes in which a chain runs front to back (ADJOINT_TIMES, INVERSE_TIMES)
    _all_ops = 15

    @staticmethod
    def _flip_capability(cap, trafo):
        return sum(_mode_seen_through(bit, trafo) for bit in (1, 2, 4, 8) if cap & bit)

    @staticmethod
    def _add_inverse_capability(cap):
        return cap | LinearOperator._flip_capability(cap, LinearOperator.INVERSE_BIT)

    def _dom(self, mode):
        """where the input of `mode` lives: TIMES and ADJOINT_INVERSE_TIMES read the domain"""
        return self.domain if _mode_index(mode) in (0, 3) else self.target

    def _tgt(self, mode):
        return self.target if _mode_index(mode) in (0, 3) else self.domain

    def _flip_modes(self, trafo):
        return OperatorAdapter(self, trafo) if trafo else self

    inverse = property(lambda self: self._flip_modes(LinearOperator.INVERSE_BIT))
    adjoint = property(lambda self: self._flip_modes(LinearOperator.ADJOINT_BIT))
    capability = property(lambda self: self._capability)

    # -- algebra: linear operators stay linear operators ------------------------------------------------
    def __matmul__(self, other):
        if is_operator(other) and other.isIdentity():
            return self
        if isinstance(other, LinearOperator):
            return ChainOperator.make([self, other])
        return Operator.__matmul__(self, other)

    def __rmatmul__(self, other):
        return ChainOperator.make([other, self]) if isinstance(other, LinearOperator) else NotImplemented

    def _myadd(self, other, oneg):
        return SumOperator.make((self, other), (False, oneg))

    def __add__(self, other):
        return self._myadd(other, False) if isinstance(other, LinearOperator) else Operator.__add__(self, other)

    def __sub__(self, other):
        return self._myadd(other, True) if isinstance(other, LinearOperator) else Operator.__sub__(self, other)

    __radd__ = __add__

    def scale(self, factor):
        if not isinstance(factor, Number):
            raise TypeError(".scale() takes a number as input")
        return self if factor == 1 else ChainOperator.make([ScalingOperator(self.target, factor), self])

    def __neg__(self):
        return self.scale(-1)

    # -- application ---------------------------------------------------------------------------------------
    def apply(self, x, mode):
        raise NotImplementedError

    def force(self, x):
        return self.apply(x.extract(self.domain), self.TIMES)

    def __call__(self, x):
        kind = _operand_kind(x)
        if self.isIdentity() or kind == "op":
            return x if self.isIdentity() else self @ x
        if kind == "field":
            return self.apply(x, self.TIMES)
        if kind == "lin":  # a linear map is its own Jacobian
            return x.new(self.apply(x.val, self.TIMES), self).prepend_jac(x.jac)
        raise TypeError(f"cannot apply linear operator to {type(x)}")

    def times(self, x):
        return self.apply(x, self.TIMES)

    def inverse_times(self, x):
        return self.apply(x, self.INVERSE_TIMES)

    def adjoint_times(self, x):
        return self.apply(x, self.ADJOINT_TIMES)

    def adjoint_inverse_times(self, x):
        return self.apply(x, self.ADJOINT_INVERSE_TIMES)

    inverse_adjoint_times = adjoint_inverse_times

    def _check_mode(self, mode):
        if _mode_index(mode) < 0:
            raise NotImplementedError("invalid operator mode specified")
        if not mode & self.capability:
            raise NotImplementedError("requested operator mode is not supported")

    def _check_input(self, x, mode):
        self._check_mode(mode)
        _same_domain(self._dom(mode), x.domain)

    def draw_sample(self, from_inverse=False, device_id=-1):
        raise NotImplementedError

    def get_sqrt(self):
        raise NotImplementedError


class EndomorphicOperator(LinearOperator):
    @property
    def target(self):
        return self._domain

    @property
    def sampling_dtype(self):
        return getattr(self, "_dtype", None)


class OperatorAdapter(LinearOperator):
    """adjoint / inverse view of another operator (reference operator_adapter.py:22-68)."""

    _VIEW = ("", "adjoint", "inverse", "adjoint inverse")

    def __init__(self, op, trafo):
        trafo = int(trafo)
        if trafo not in (1, 2, 3):
            raise ValueError("invalid operator transformation")
        self._op, self._trafo = op, trafo
        seen_times = 1 << trafo   # what the wrapped operator executes for this view's TIMES
        self._domain, self._target = op._dom(seen_times), op._tgt(seen_times)
        self._capability = self._flip_capability(op.capability, trafo)

    def _flip_modes(self, trafo):
        combined = trafo ^ self._trafo
        return OperatorAdapter(self._op, combined) if combined else self._op

    def apply(self, x, mode):
        return self._op.apply(x, _mode_seen_through(mode, self._trafo))

    def draw_sample(self, from_inverse=False, device_id=-1):
        inverted = bool(self._trafo & self.INVERSE_BIT)
        return self._op.draw_sample(from_inverse != inverted, device_id)

    def __repr__(self):
        return f"OperatorAdapter({self._VIEW[self._trafo]}) of\n  {self._op!r}"


class PrependKey(LinearOperator):
    """Prepends a string to every key of a MultiDomain (reference simple_linear_operators.py:447-471)."""

    def __init__(self, domain, pre):
        if not isinstance(domain, MultiDomain):
            raise ValueError("PrependKey needs a MultiDomain")
        self._domain, self._pre = domain, str(pre)
        self._target = MultiDomain.make({self._pre + k: domain[k] for k in domain.keys()})
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        if mode == self.TIMES:
            return MultiField.from_dict({self._pre + k: x[k] for k in self._domain.keys()}, self._target)
        return MultiField.from_dict({k: x[self._pre + k] for k in self._domain.keys()}, self._domain)


class _KeyEmbedding(LinearOperator):
    """MultiField on a sub-set of keys -> full MultiDomain with zeros on the other keys; adjoint extracts."""

    def __init__(self, domain, target):
        self._domain, self._target = MultiDomain.make(domain), MultiDomain.make(target)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        if mode == self.TIMES:
            zeros = {k: full(self._target[k], 0.0, x.device_id) for k in self._target.keys() if k not in self._domain.keys()}
            return x.unite(MultiField.from_dict(zeros)) if zeros else x
        return x.extract(self._domain)


class InsertionOperator(Operator):
    """x_variable -> x_variable united with a constant MultiField (reference simplify_for_const.py:115-146): how a
    partially constant input is inserted into an operator (Operator.simplify_for_constant_input)."""

    def __init__(self, target, cst_field):
        for arg, kind in ((target, MultiDomain), (cst_field, MultiField)):
            if not isinstance(arg, kind):
                raise TypeError(f"{kind.__name__} expected")
        frozen = set(cst_field.keys())
        self._cst, self._target = cst_field, MultiDomain.make(target)
        self._domain = MultiDomain.make({k: sub for k, sub in self._target.items() if k not in frozen})
        self._jac = _KeyEmbedding(self._domain, self._target)

    def apply(self, x):
        self._check_input(x)
        val = x.val if is_linearization(x) else x
        val = val.unite(self._cst.at(val.device_id))
        return x.new(val, self._jac) if is_linearization(x) else val

    def __repr__(self):
        return f"InsertionOperator\n  Constant: {self._cst.keys()}\n  Variable: {self._domain.keys()}"


class ConstantOperator(Operator):
    """Returns `output` whatever it is given; zero Jacobian (reference simplify_for_const.py:28-46)."""

    def __init__(self, output, domain={}):
        self._domain, self._target, self._output = makeDomain(domain), output.domain, output
        if isinstance(self._domain, dict):
            self._domain = MultiDomain.make(self._domain)

    def apply(self, x):
        self._check_input(x)
        out = self._output.at(x.device_id)
        return x.new(out, NullOperator(self._domain, self._target)) if is_linearization(x) else out

    def __repr__(self):
        tgt = self.target.keys() if isinstance(self.target, MultiDomain) else "()"
        return f"{tgt} <- ConstantOperator"


class NullOperator(LinearOperator):
    def __init__(self, domain, target, default_domain_device_id=-1, default_target_device_id=-1):
        # (the zero field is made on the device of the input: the reference's default device ids are accepted, not needed)
        self._domain, self._target = makeDomain(domain), makeDomain(target)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        return full(self._tgt(mode), 0.0, x.device_id)


class ScalingOperator(EndomorphicOperator):
    """Multiplication by a scalar (reference scaling_operator.py:25-139).  Every mode and every adjoint / inverse view is
    the same operation with a transformed factor (`_factor_seen_through`)."""

    def __init__(self, domain, factor, sampling_dtype=None):
        if isinstance(factor, Field) and factor.shape == ():
            factor = factor.asnumpy()[()]
        if isinstance(factor, (torch.Tensor, np.ndarray)) and tuple(factor.shape) == ():  # field.val[()] of a scalar field
            factor = factor.item()
        if not np.isscalar(factor):
            raise TypeError("Scalar required")
        self._domain, self._factor, self._dtype = makeDomain(domain), factor, sampling_dtype
        self._capability = self._all_ops

    def isIdentity(self):
        return self._factor == 1

    def _factor_seen_through(self, trafo):
        fct = np.conj(self._factor) if trafo & self.ADJOINT_BIT else self._factor
        return 1.0 / fct if trafo & self.INVERSE_BIT else fct

    def apply(self, x, mode):
        self._check_input(x, mode)
        if self._factor == 1.0:
            return x
        if self._factor == 0.0:  # (also the "inverse" modes: the reference returns zeros, scaling_operator.py:80-81)
            return full(x.domain, 0.0, x.device_id)
        return x * self._factor_seen_through(_mode_index(mode))

    def _flip_modes(self, trafo):
        return ScalingOperator(self._domain, self._factor_seen_through(trafo), self._dtype)

    def _get_fct(self, from_inverse):
        fct = self._factor
        positive = np.imag(fct) == 0.0 and np.real(fct) >= 0.0 and not (np.real(fct) == 0.0 and from_inverse)
        if not positive:
            raise ValueError("operator not positive definite")
        return 1.0 / np.sqrt(fct) if from_inverse else np.sqrt(fct)

    def draw_sample(self, from_inverse=False, device_id=-1):
        from .field import from_random

        if self._dtype is None:
            raise RuntimeError("Need to specify dtype to be able to sample from this operator:\n" + repr(self))
        return from_random(self._domain, "normal", dtype=self._dtype, device_id=device_id,
                           std=float(self._get_fct(from_inverse)))

    def get_sqrt(self):
        fct = self._get_fct(False)
        if np.iscomplexobj(fct) or fct < 0:  # a complex-TYPED factor has no real square-root operator, even 2 + 0j
            raise ValueError("get_sqrt() works only for positive definite operators.")
        return ScalingOperator(self._domain, fct)

    def __call__(self, other):
        res = EndomorphicOperator.__call__(self, other)
        if is_linearization(other) and other.metric is not None and np.isreal(self._factor) and self._factor >= 0:
            # the metric of the scaled quantity: sqrt(factor) on both sides of the incoming one
            root = ScalingOperator(other.metric.domain, np.sqrt(self._factor), self._dtype)
            res = res.add_metric(SandwichOperator.make(root, other.metric))
        return res

    def __repr__(self):
        extra = "" if self._dtype is None else f", sampling dtype {self._dtype}"
        return f"ScalingOperator ({self._factor}{extra})"


class DiagonalOperator(EndomorphicOperator):
    """Pointwise multiplication by a Field, optionally living on a sub-set of the spaces
    (reference diagonal_operator.py:51-260)."""

    def __init__(self, diagonal, domain=None, spaces=None, sampling_dtype=None, _trafo=0):
        if not isinstance(diagonal, Field):
            raise TypeError("Field object required")
        dom = diagonal.domain if domain is None else DomainTuple.make(domain)
        if spaces is None:
            _same_domain(diagonal.domain, dom)
            active, ldiag = None, diagonal.val
        else:
            active, ldiag = self._placed(diagonal, dom, (spaces,) if np.isscalar(spaces) else tuple(spaces))
        self._setup(dom, ldiag, sampling_dtype, _trafo, active)

    @staticmethod
    def _placed(diagonal, dom, spaces):
        """A diagonal that lives on the sub-spaces `spaces` of `dom`: (spaces or None when they are all of dom, the values
        reshaped so that they broadcast against fields on dom)."""
        if len(spaces) != len(diagonal.domain):
            raise ValueError("spaces and domain must have the same length")
        for own, there in enumerate(spaces):
            if diagonal.domain[own] != dom[there]:
                raise ValueError("Mismatch between:\n{}\nand:\n{}".format(diagonal.domain[own], dom[there]))
        if spaces == tuple(range(len(dom))):
            return None, diagonal.val
        covered = {axis for sp in spaces for axis in dom.axes[sp]}
        return spaces, diagonal.val.reshape([n if axis in covered else 1 for axis, n in enumerate(dom.shape)])

    def _setup(self, dom, ldiag, sampling_dtype, trafo, spaces):
        if not (ldiag.is_floating_point() or ldiag.is_complex()):
            ldiag = ldiag.to(torch.float64)  # integer-valued diagonals (ift.full(dom, 2)): torch would take their sqrt in fp32
        self._domain, self._ldiag, self._dtype, self._trafo, self._spaces = dom, ldiag, sampling_dtype, trafo, spaces
        self._complex = ldiag.is_complex()
        self._capability = self._all_ops
        self._diagmin_cache = None

    @staticmethod
    def _from_ldiag(proto, ldiag, sampling_dtype, trafo, spaces):
        res = DiagonalOperator.__new__(DiagonalOperator)
        res._setup(proto._domain, ldiag, sampling_dtype, trafo, spaces)
        return res

    @property
    def _diagmin(self):
        if self._complex:
            raise RuntimeError("complex DiagonalOperator does not have _diagmin")
        if self._diagmin_cache is None:
            self._diagmin_cache = float(self._ldiag.min())  # reduction for validation only
        return self._diagmin_cache

    def _full(self):
        return self._spaces is None

    def _mul(self, xval, divide, conj):
        d = self._ldiag
        if d.device != xval.device:
            d = self._ldiag = d.to(xval.device)
        conj = bool(conj and self._complex)
        if not xval.is_cuda:
            d = torch.conj_physical(d) if conj else d
            return xval / d if divide else xval * d
        if self._complex or xval.is_complex():
            # d, conj d, 1/d, 1/conj d (diagonal_operator.py:194-214): one launch of the complex kernel; a diagonal on a
            # sub-set of the spaces is broadcast first (a replication copy)
            full = d if self._full() else d.expand(self._domain.shape).contiguous()
            return B.cplx_muldiv(xval, full, divide=divide, conj_b=conj)
        if d.dtype != xval.dtype:
            d = d.to(xval.dtype)
        if not self._full():
            # a diagonal on a sub-set of the spaces: its broadcast over the other axes is materialised once (a copy),
            # the product itself is the element-wise kernel
            key = (str(xval.device), xval.dtype)
            cache = self.__dict__.setdefault("_expanded", {})
            if key not in cache:
                cache[key] = d.expand(self._domain.shape).contiguous()
            d = cache[key]
        from . import _lib as L

        return B.binary(L.OP_DIV if divide else L.OP_MUL, xval.contiguous(), d.contiguous())

    def apply(self, x, mode):
        self._check_input(x, mode)
        trafo = _mode_index(mode) ^ self._trafo
        return Field(x.domain, self._mul(x.val, divide=bool(trafo & 2), conj=bool(trafo & 1)))

    def _actual_diag(self):
        """the diagonal this operator multiplies by in TIMES mode (the stored one seen through the pending transformation)"""
        d = self._ldiag
        inverted, conjugated = bool(self._trafo & 2), bool(self._trafo & 1 and self._complex)
        if not d.is_cuda:
            d = 1.0 / d if inverted else d
            return torch.conj_physical(d) if conjugated else d
        if self._complex:
            return B.cplx_muldiv(1.0, d, divide=True, conj_b=conjugated) if inverted else \
                (B.cplx_pointwise("conjugate", d) if conjugated else d)
        return B.binary(3, 1.0, d) if inverted else d

    def _flip_modes(self, trafo):
        return DiagonalOperator._from_ldiag(self, self._ldiag, self._dtype, self._trafo ^ trafo, self._spaces)

    def _scale(self, fct):
        d = self._actual_diag()
        if not d.is_cuda:
            d = d * fct
        elif d.is_complex() or isinstance(fct, complex):
            d = B.cplx_muldiv(d, complex(fct))
        else:
            d = B.binary(2, d, float(fct))
        return DiagonalOperator._from_ldiag(self, d, self._dtype, 0, self._spaces)

    def _add(self, number):
        """this operator + number * identity (diagonal_operator.py:149-155)"""
        d = self._actual_diag()
        if not d.is_cuda:
            d = d + number
        elif d.is_complex() or isinstance(number, complex):
            raise NotImplementedError
        else:
            d = B.binary(0, d, float(number))
        return DiagonalOperator._from_ldiag(self, d, self._dtype, 0, self._spaces)

    def _combine_sum(self, op, selfneg, opneg):
        """(+-) self (+-) op as one diagonal (diagonal_operator.py:157-163); both on the same sub-spaces"""
        a, b = self._actual_diag(), op._actual_diag()
        if a.dtype != b.dtype:
            wide = torch.promote_types(a.dtype, b.dtype)
            a, b = a.to(wide), b.to(wide)
        if a.is_cuda and (a.is_complex() or b.is_complex()):
            raise NotImplementedError
        if not a.is_cuda:
            total = (-a if selfneg else a) + (-b if opneg else b)
        else:
            a = B.binary(2, a, -1.0) if selfneg else a
            total = B.binary(1 if opneg else 0, a.contiguous(), b.contiguous())
        return DiagonalOperator._from_ldiag(self, total, self._dtype if self._dtype == op._dtype else None, 0, self._spaces)

    def _combine_prod(self, op):
        if self._spaces != op._spaces and not (self._full() and op._full()):
            # diagonals on different sub-spaces: the product lives on their union (diagonal_operator.py:118-147)
            a, b = self._actual_diag(), op._actual_diag()
            if a.is_cuda:
                raise NotImplementedError
            union = None if (self._full() or op._full()) else tuple(sorted(set(self._spaces) | set(op._spaces)))
            if union == tuple(range(len(self._domain))):
                union = None
            prod = a * b
            if union is None:
                prod = prod.expand(self._domain.shape).contiguous()
            return DiagonalOperator._from_ldiag(self, prod, self._dtype if self._dtype == op._dtype else None, 0, union)
        a, b = self._actual_diag(), op._actual_diag()
        if a.dtype != b.dtype:  # (numpy semantics: the wider type wins, e.g. an fp32 linearisation point times fp64 tables)
            wide = torch.promote_types(a.dtype, b.dtype)
            a, b = a.to(wide), b.to(wide)
        if not a.is_cuda:
            prod = a * b
        else:
            prod = B.cplx_muldiv(a, b) if (a.is_complex() or b.is_complex()) else B.binary(2, a, b)
        return DiagonalOperator._from_ldiag(self, prod, self._dtype if self._dtype == op._dtype else None, 0, self._spaces)

    def process_sample(self, samp, from_inverse):
        inv = from_inverse ^ (self._trafo >= 2)
        if self._complex or self._diagmin < 0.0 or (self._diagmin == 0.0 and inv):
            raise ValueError("operator not positive definite")
        sq = Field(self._domain, self._ldiag.expand(self._domain.shape).contiguous()).sqrt() if not self._full() else \
            Field(self._domain, self._ldiag).sqrt()
        return samp / sq if inv else samp * sq

    def draw_sample(self, from_inverse=False, device_id=-1):
        if self._dtype is None:
            raise RuntimeError("Need to specify dtype to be able to sample from this operator:\n" + repr(self))
        res = Field.from_random(self._domain, "normal", dtype=self._dtype, device_id=device_id)
        return self.process_sample(res, from_inverse)

    def get_sqrt(self):
        if self._complex or self._diagmin < 0.0:
            raise ValueError("get_sqrt() works only for positive definite operators.")
        d = self._ldiag
        sq = torch.sqrt(d) if not d.is_cuda else B.pointwise("sqrt", d.contiguous())
        return DiagonalOperator._from_ldiag(self, sq, self._dtype, self._trafo, self._spaces)

    def __repr__(self):
        return f"DiagonalOperator (domain/target shape: {self._domain.shape})"


class BlockDiagonalOperator(EndomorphicOperator):
    """One operator per MultiDomain key (reference block_diagonal_operator.py:24-83)."""

    def __init__(self, domain, operators):
        if not isinstance(domain, MultiDomain):
            raise TypeError("MultiDomain expected")
        # a key without operator (missing or None) is the identity on that key
        self._domain, self._ops = domain, tuple(operators.get(k) for k in domain.keys())
        for op in self._ops:
            if op is not None and not isinstance(op, LinearOperator):
                raise TypeError("LinearOperator expected")
            if op is not None and op.target is not op.domain:
                raise TypeError("domain and target mismatch")
        self._capability = _common_capability([op for op in self._ops if op is not None], self._all_ops)

    def _blockwise(self, other, combine):
        """BlockDiagonalOperator of combine(own block, other's block) per key (block_diagonal_operator.py:88-103)"""
        _same_domain(self._domain, other._domain)
        blocks = {}
        for key, a, b in zip(self._domain.keys(), self._ops, other._ops):
            a = ScalingOperator(self._domain[key], 1.0) if a is None else a
            b = ScalingOperator(self._domain[key], 1.0) if b is None else b
            blocks[key] = combine(a, b)
        return BlockDiagonalOperator(self._domain, blocks)

    def apply(self, x, mode):
        self._check_input(x, mode)
        vals = tuple(op.apply(v, mode=mode) if op is not None else v for op, v in zip(self._ops, x.values()))
        return MultiField(self._domain, vals)

    def _flip_modes(self, trafo):
        return BlockDiagonalOperator(self._domain, {k: None if op is None else op._flip_modes(trafo)
                                                    for k, op in zip(self._domain.keys(), self._ops)})

    def draw_sample(self, from_inverse=False, device_id=-1):
        if any(op is None for op in self._ops):
            raise RuntimeError("Need to specify dtype for all operators that are set to None.")
        vals = tuple(op.draw_sample(from_inverse, device_id) for op in self._ops)
        return MultiField(self._domain, vals)

    def get_sqrt(self):
        return BlockDiagonalOperator(self._domain, {k: op.get_sqrt() for k, op in zip(self._domain.keys(), self._ops)
                                                    if op is not None})

    @property
    def sampling_dtype(self):
        return {k: getattr(op, "sampling_dtype", None) for k, op in zip(self._domain.keys(), self._ops)}


def _common_capability(ops, start):
    """modes every operator of `ops` supports (within `start`)"""
    for op in ops:
        start &= op.capability
    return start


def makeOp(inp, dom=None, sampling_dtype=None):
    """Diagonal operator from a scalar / Field / MultiField (reference sugar.py:410-458)."""
    if inp is None:
        return None
    if isinstance(inp, MultiField):
        if dom is not None:
            raise TypeError("dom only allowed for Fields")
        per_key = sampling_dtype if isinstance(sampling_dtype, dict) else dict.fromkeys(inp.keys(), sampling_dtype)
        return BlockDiagonalOperator(inp.domain, {k: makeOp(v, sampling_dtype=per_key[k]) for k, v in inp.items()})
    if isinstance(inp, Field):
        if dom is not None:  # the field lives on the leading spaces of `dom`
            return DiagonalOperator(inp, domain=dom, spaces=tuple(range(len(inp.domain))), sampling_dtype=sampling_dtype)
        if inp.domain is DomainTuple.scalar_domain():
            return ScalingOperator(inp.domain, inp.asnumpy()[()], sampling_dtype)
        return DiagonalOperator(inp, sampling_dtype=sampling_dtype)
    if np.isscalar(inp):
        if not isinstance(dom, (DomainTuple, MultiDomain)):
            raise TypeError("need proper `dom` argument")
        return ScalingOperator(dom, inp, sampling_dtype)
    if dom is not None:
        raise TypeError("dom only allowed for Fields")
    raise NotImplementedError


# ================================================================================================
# chains and sums
# ================================================================================================
class ChainOperator(LinearOperator):
    """Product of linear operators with scalar / diagonal merging (reference chain_operator.py:26-145)."""

    def __init__(self, ops, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError
        self._ops = ops
        self._capability = _common_capability(ops, self._all_ops)
        self._domain, self._target = ops[-1].domain, ops[0].target

    @staticmethod
    def simplify(ops):
        """One left-to-right sweep over the spliced factor list: scalars are collected into ONE number (scalings on an
        endomorphic domain commute with everything), neighbouring full-domain diagonals are multiplied, and at the end
        the number goes into the leading diagonal if there is one, else in front as a ScalingOperator."""
        links = [link for op in ops for link in (op._ops if isinstance(op, ChainOperator) else (op,))]
        number, number_dtype, kept = 1.0, None, []
        for link in links:
            if isinstance(link, ScalingOperator) and np.isscalar(link._factor) and link.domain is link.target:
                number = number * link._factor
                number_dtype = number_dtype if link._dtype is None else link._dtype
                continue
            mergeable = bool(kept) and all(isinstance(o, DiagonalOperator) for o in (kept[-1], link)) \
                and kept[-1].domain is link.domain \
                and (all(o._full() for o in (kept[-1], link)) or not (kept[-1]._ldiag.is_cuda or link._ldiag.is_cuda))
            blocks = bool(kept) and all(isinstance(o, BlockDiagonalOperator) for o in (kept[-1], link))
            if mergeable:
                kept[-1] = kept[-1]._combine_prod(link)
            elif blocks and kept[-1].domain is link.domain:
                kept[-1] = kept[-1]._blockwise(link, lambda a, b: a @ b)
            else:
                kept.append(link)
        if number == 1 and kept:
            return kept
        if kept and isinstance(kept[0], DiagonalOperator) and kept[0]._full():
            return [kept[0]._scale(number)] + kept[1:]
        return [ScalingOperator((kept[0] if kept else links[0]).target, number, number_dtype)] + kept

    @staticmethod
    def make(ops):
        ops = tuple(ops)
        if not ops:
            raise ValueError("ops is empty")
        seams = [(later.domain, earlier.target) for later, earlier in zip(ops, ops[1:])]
        for pair in seams:
            _same_domain(*pair)
        *rest, first = ChainOperator.simplify(ops)[::-1]
        return ChainOperator([first] + rest[::-1], _callingfrommake=True) if rest else first

    def _flip_modes(self, trafo):
        if not trafo:
            return self
        # the adjoint and the inverse of a product reverse it; doing both restores the order
        order = self._ops if trafo == (self.ADJOINT_BIT | self.INVERSE_BIT) else self._ops[::-1]
        return ChainOperator.make([op._flip_modes(trafo) for op in order])

    def apply(self, x, mode):
        self._check_mode(mode)
        for op in (self._ops if mode & self._backwards else self._ops[::-1]):
            x = op.apply(x, mode)
        return x

    def draw_sample(self, from_inverse=False, device_id=-1):
        raise NotImplementedError

    def __repr__(self):
        return _indented("ChainOperator", self._ops)


class SumOperator(LinearOperator):
    """Sum / difference of linear operators (reference sum_operator.py:26-225), held as signed terms."""

    def __init__(self, ops, neg, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError
        self._ops, self._neg = ops, neg
        self._capability = _common_capability(ops, self.TIMES | self.ADJOINT_TIMES)
        self._domain, self._target = domain_union(op.domain for op in ops), domain_union(op.target for op in ops)

    @staticmethod
    def _signed_terms(ops, neg):
        """(operator, minus) pairs with nested sums opened up"""
        for op, minus in zip(ops, neg):
            if isinstance(op, SumOperator):
                yield from ((inner, inner_minus != minus) for inner, inner_minus in zip(op._ops, op._neg))
            else:
                yield op, minus

    @staticmethod
    def make(ops, neg=None):
        ops = tuple(ops)
        neg = (False,) * len(ops) if neg is None else tuple(bool(n) for n in neg)
        if not ops or len(ops) != len(neg):
            raise ValueError("length mismatch")
        # multiples of the identity on ONE domain add up to a single scaling, which goes LAST like in the reference
        # (sum_operator.py:104-107): the order of the terms is the order of the random draws of draw_sample
        terms, identity = [], None
        for op, minus in SumOperator._signed_terms(ops, neg):
            if isinstance(op, ScalingOperator) and (identity is None or identity["domain"] is op.domain):
                identity = identity or dict(domain=op.domain, factor=0.0, dtype=None)
                identity["factor"] = identity["factor"] + (-op._factor if minus else op._factor)
                identity["dtype"] = identity["dtype"] if op._dtype is None else op._dtype
            else:
                terms.append((op, minus))
        def mergeable(op, other=None):
            """a real or host diagonal (on the sub-spaces and domain of `other`, with its sampling dtype)"""
            if not isinstance(op, DiagonalOperator) or (op._ldiag.is_cuda and op._complex):
                return False
            return other is None or (op.domain is other.domain and op._spaces == other._spaces and op._dtype == other._dtype)

        # a multiple of the identity goes into the first diagonal on its domain with the same sampling dtype; diagonals on the
        # same sub-spaces add up to one (sum_operator.py:107-140) -- the merged sum draws ONE sample, like the reference's
        if identity is not None and identity["factor"] != 0:
            host = next((i for i, (op, _) in enumerate(terms) if mergeable(op) and op.domain is identity["domain"]
                         and op._dtype == identity["dtype"] and not isinstance(identity["factor"], complex)), None)
            if host is not None:
                op, minus = terms[host]
                terms[host] = (op._add(-identity["factor"] if minus else identity["factor"]), minus)
                identity = None
        if identity is not None and (identity["factor"] != 0 or not terms):
            terms.append((ScalingOperator(identity["domain"], identity["factor"], identity["dtype"]), False))
        merged_terms, used = [], set()
        for i, (op, minus) in enumerate(terms):
            if i in used:
                continue
            if mergeable(op):
                for j in range(i + 1, len(terms)):
                    if j not in used and mergeable(terms[j][0], op):
                        op, minus = op._combine_sum(terms[j][0], minus, terms[j][1]), False
                        used.add(j)
            merged_terms.append((op, minus))
        terms = merged_terms
        # block-diagonal terms on one MultiDomain add block by block (sum_operator.py:141-152)
        first = next((i for i, (op, _) in enumerate(terms) if isinstance(op, BlockDiagonalOperator)), None)
        if first is not None:
            merged, others = terms[first], []
            for op, minus in terms[first + 1:]:
                if isinstance(op, BlockDiagonalOperator) and op.domain is merged[0].domain:
                    signs = (merged[1], minus)
                    merged = (merged[0]._blockwise(op, lambda a, b: SumOperator.make((a, b), signs)), False)
                else:
                    others.append((op, minus))
            terms = terms[:first] + [merged] + others
        if len(terms) == 1:
            op, minus = terms[0]
            return op.scale(-1) if minus else op
        return SumOperator(tuple(t[0] for t in terms), tuple(t[1] for t in terms), _callingfrommake=True)

    def _flip_modes(self, trafo):
        if trafo & self.INVERSE_BIT:  # the inverse of a sum is not the sum of the inverses
            return OperatorAdapter(self, trafo)
        return SumOperator.make([op._flip_modes(trafo) for op in self._ops], self._neg)

    def apply(self, x, mode):
        self._check_mode(mode)
        parts = [(op.apply(x.extract(op._dom(mode)), mode), minus) for op, minus in zip(self._ops, self._neg)]
        (first, first_minus), rest = parts[0], parts[1:]
        total = -first if first_minus else first
        for part, minus in rest:
            total = total.flexible_addsub(part, minus)
        return total

    def draw_sample(self, from_inverse=False, device_id=-1):
        from functools import reduce

        if from_inverse:
            raise NotImplementedError("cannot draw from inverse of this operator")
        # the draws happen in term order (the RNG sequence); the signs do not matter for the distribution
        return reduce(lambda acc, d: acc.flexible_addsub(d, False), [op.draw_sample(False, device_id) for op in self._ops])

    def __repr__(self):
        return _indented("SumOperator", self._ops)


class SandwichOperator(EndomorphicOperator):
    """bun^dagger cheese bun (reference sandwich_operator.py:27-110)."""

    def __init__(self, bun, cheese, op, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError
        self._bun, self._cheese, self._op = bun, cheese, op
        self._domain, self._capability = op.domain, op._capability

    @staticmethod
    def make(bun, cheese=None, sampling_dtype=None):
        while isinstance(cheese, SandwichOperator):  # a sandwich in a sandwich is one sandwich with a longer bun
            bun, cheese = cheese._bun @ bun, cheese._cheese
        for what, op, optional in (("bun", bun, False), ("cheese", cheese, True)):
            if not (isinstance(op, LinearOperator) or (optional and op is None)):
                raise TypeError(f"{what} must be a linear operator" + (" or None" if optional else ""))
        cheese = ScalingOperator(bun.target, 1.0, sampling_dtype) if cheese is None else cheese
        if not isinstance(bun, ScalingOperator):
            return SandwichOperator(bun, cheese, bun.adjoint @ cheese @ bun, _callingfrommake=True)
        weight = abs(bun._factor) ** 2  # a scalar bun commutes with the cheese
        return cheese if weight == 1.0 else SandwichOperator(bun, cheese, cheese.scale(weight), _callingfrommake=True)

    def apply(self, x, mode):
        return self._op.apply(x, mode)

    def draw_sample(self, from_inverse=False, device_id=-1):
        # xi ~ cheese       =>  cov(bun^T xi)  = bun^T cheese bun
        # xi ~ cheese^-1    =>  cov(bun^-1 xi) = (bun^T cheese bun)^-1   (needs an invertible bun)
        if from_inverse and not self._bun.capability & self._bun.INVERSE_TIMES:
            raise NotImplementedError("cannot draw from inverse of this operator")
        push = self._bun.inverse_times if from_inverse else self._bun.adjoint_times
        return push(self._cheese.draw_sample(from_inverse, device_id))

    def get_sqrt(self):
        return self._cheese.get_sqrt() @ self._bun

    def __repr__(self):
        return "SandwichOperator:\n  Cheese:\n    " + repr(self._cheese) + "\n  Bun:\n    " + repr(self._bun).replace("\n", "\n    ")


# ================================================================================================
# small linear / affine operators
# ================================================================================================
class Adder(Operator):
    """x -> x +/- a (reference adder.py:24-60)."""

    def __init__(self, a, neg=False, domain=None):
        if not is_fieldlike(a):
            raise TypeError("Field or MultiField required")
        self._a = a
        self._domain = self._target = makeDomain(a.domain if domain is None else domain)
        self._neg = bool(neg)

    def apply(self, x):
        self._check_input(x)
        if self._a.device_id != x.device_id:
            self._a = self._a.at(x.device_id)
        if self._neg:
            return x - self._a
        return x + self._a


class VdotOperator(LinearOperator):
    """<field, .> onto the scalar domain (reference simple_linear_operators.py:33-56)."""

    def __init__(self, field):
        if not is_fieldlike(field):
            raise TypeError("Field or MultiField required")
        self._field = field
        self._domain = field.domain
        self._target = DomainTuple.scalar_domain()
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        if self._field.device_id != x.device_id:
            self._field = self._field.at(x.device_id)
        if mode == self.TIMES:
            return Field.scalar(self._field.s_vdot(x)).at(x.device_id)
        scalar = x.asnumpy()[()]
        return self._field * (complex(scalar) if np.iscomplexobj(scalar) else float(scalar))


class Realizer(EndomorphicOperator):
    def __init__(self, domain):
        self._domain = makeDomain(domain)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        return x.real


class FieldAdapter(LinearOperator):
    """Extract one key of a MultiField (reference simple_linear_operators.py:152-214)."""

    def __init__(self, target, name):
        self._name = name
        self._target = DomainTuple.make(target)
        self._domain = MultiDomain.make({name: self._target})
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        if mode == self.TIMES:
            return x[self._name]
        return MultiField(self._domain, (x,))

    def __repr__(self):
        return f"FieldAdapter {self._name!r}"


def ducktape(left, right, name):
    """Convenience constructor of FieldAdapters (reference simple_linear_operators.py:238-300)."""
    if isinstance(left, str) or isinstance(right, str):
        raise TypeError("name must be the third argument")
    left = left.domain if is_operator(left) or is_fieldlike(left) else left
    right = right.target if is_operator(right) else (right.domain if is_fieldlike(right) else right)
    if left is None and right is None:
        raise ValueError("need at least one domain")
    if left is not None and right is not None:
        raise ValueError("only one of left / right may be given")
    if right is None:  # operator ending in a MultiDomain{name} ... -> feeds `left`
        left = makeDomain(left)
        if isinstance(left, DomainTuple):
            return FieldAdapter(left, name)
        return FieldAdapter(left[name], name).adjoint
    right = makeDomain(right)
    if isinstance(right, DomainTuple):
        return FieldAdapter(right, name).adjoint
    return FieldAdapter(right[name], name)


def IntegrationOperator(domain, spaces):
    """Contraction weighted with the pixel volumes: the integral over the sub-domains `spaces` (reference
    contraction_operator.py:80-95)."""
    return ContractionOperator(domain, spaces, 1)


def Variable(target, key):
    return ducktape(makeDomain(target), None, key)


class ContractionOperator(LinearOperator):
    """Sum over sub-spaces; the adjoint broadcasts (reference contraction_operator.py:25-95)."""

    def __init__(self, domain, spaces, power=0):
        self._domain = DomainTuple.make(domain)
        n = len(self._domain)
        if spaces is None:
            spaces = tuple(range(n))
        elif np.isscalar(spaces):
            spaces = (int(spaces),)
        self._spaces = tuple(sorted(int(s) for s in spaces))
        if any(s < 0 or s >= n for s in self._spaces):
            raise ValueError("invalid space index")
        self._target = DomainTuple.make([d for i, d in enumerate(self._domain) if i not in self._spaces])
        self._power = power
        self._capability = self.TIMES | self.ADJOINT_TIMES
        self._axes = tuple(a for s in self._spaces for a in self._domain.axes[s])
        self._kept_index = {}

    def apply(self, x, mode):
        self._check_input(x, mode)
        full_contraction = len(self._spaces) == len(self._domain)
        if not self._spaces:  # nothing to contract: identity (torch would read an empty `dim` as "all axes")
            return x
        if self._power != 0:
            # pixel volumes**power along the contracted sub-domains (contraction_operator.py:54-77): on the way in for
            # TIMES, on the way out for the broadcast
            plain = ContractionOperator(self._domain, self._spaces)
            if mode == self.TIMES:
                return plain.apply(x.weight(self._power, spaces=self._spaces), mode)
            return plain.apply(x, mode).weight(self._power, spaces=self._spaces)
        if mode == self.ADJOINT_TIMES:
            if x.val.is_cuda and full_contraction:
                ones = torch.ones(self._domain.shape, dtype=x.val.dtype, device=x.val.device)
                return Field(self._domain, B.binary(2, ones, float(x.val.item())))
            shp = [1 if i in self._axes else n for i, n in enumerate(self._domain.shape)]
            # broadcast = replication copy (no arithmetic), also for device tensors
            return Field(self._domain, x.val.reshape(shp).expand(self._domain.shape).contiguous())
        if x.val.is_cuda:
            if not full_contraction:
                # sum over the contracted axes = scatter-add onto the flat index of the kept coordinates
                # (static map: bins summed in a fixed order through the bin-sorted permutation, no atomics)
                key = str(x.val.device)
                nkept = int(np.prod(self._target.shape))
                if key not in self._kept_index:
                    kept = torch.arange(nkept, dtype=torch.int32, device=x.val.device)
                    shp = [1 if i in self._axes else n for i, n in enumerate(self._domain.shape)]
                    kept = kept.reshape(shp).expand(self._domain.shape).contiguous().reshape(-1)
                    self._kept_index[key] = B.bin_plan(kept, nkept) if kept.numel() <= B.BIN_PLAN_MAX else kept
                plan = self._kept_index[key]
                if isinstance(plan, tuple):
                    res = B.bin_sum(x.val.contiguous().reshape(-1), plan)
                else:
                    res = B.scatter_add(x.val.contiguous().reshape(-1), plan, nkept).to(x.val.dtype)
                return Field(self._target, res.reshape(self._target.shape))
            return Field.scalar(x.s_sum()).at(x.device_id)
        if full_contraction:
            return Field.scalar(x.s_sum())
        return Field(self._target, x.val.sum(dim=self._axes))


# ================================================================================================
# harmonic transforms and power distributors
# ================================================================================================
def _space_index(domain, space, what="space"):
    """Index of the sub-space an operator acts on: `space`, or the only one of a one-space domain."""
    if space is None:
        if len(domain) != 1:
            raise ValueError(f"need a {what} index for DomainTuples with more than one entry")
        return 0
    space = int(space)
    if not 0 <= space < len(domain):
        raise ValueError(f"invalid {what} index")
    return space


def _host_fftn(v, axes, inverse=False):
    """FFT of a HOST tensor over `axes` by scipy.fft (pocketfft) -- the library the reference falls back to without ducc0
    (ducc_dispatch.py:70-118), so host fields agree with the reference's numpy path to its own 1e-14 test tolerances;
    device tensors never come here (libniftyk's plans)."""
    import scipy.fft

    fn = scipy.fft.ifftn if inverse else scipy.fft.fftn
    return torch.from_numpy(fn(v.numpy(), axes=axes))


class _RGTransformBase(LinearOperator):
    def __init__(self, domain, target=None, space=None):
        self._domain = DomainTuple.make(domain)
        self._space = _space_index(self._domain, space)
        here = self._domain[self._space]
        if not isinstance(here, RGSpace):
            raise TypeError(f"{type(self).__name__} only works on RGSpaces")
        there = here.get_default_codomain() if target is None else target
        for a, b in ((here, there), (there, here)):  # each must accept the other as its codomain
            a.check_codomain(b)
        self._target = DomainTuple.make([there if i == self._space else sub for i, sub in enumerate(self._domain)])

    def _over_subspace(self, val, fn_host, fn_dev):
        """Apply a transform over the axes of self._space (reference harmonic_operators.py:59-75, `axes=`).  The kernels
        transform TRAILING axes of a contiguous array with everything in front as a batch, so a sub-space that is not
        last is moved there by a permutation copy (data movement only) and moved back afterwards."""
        axes = tuple(self._domain.axes[self._space])
        if not val.is_cuda:
            return fn_host(val, axes)
        nd = val.dim()
        if axes == tuple(range(nd - len(axes), nd)):
            return fn_dev(val.contiguous(), len(axes))
        perm = [i for i in range(nd) if i not in axes] + list(axes)
        inv = [perm.index(i) for i in range(nd)]
        return fn_dev(val.permute(perm).contiguous(), len(axes)).permute(inv).contiguous()

    def _factor(self, mode):
        if mode & (self.TIMES | self.ADJOINT_TIMES):
            return self._domain[self._space].scalar_dvol
        return self._target[self._space].scalar_dvol


class HartleyOperator(_RGTransformBase):
    """Genuine N-D Hartley transform times the volume factor (reference harmonic_operators.py:97-161)."""

    def __init__(self, domain, target=None, space=None):
        super().__init__(domain, target, space)
        self._capability = self._all_ops

    def apply(self, x, mode):
        self._check_input(x, mode)
        if x.val.is_complex():
            return self._cartesian(x.real, mode) + 1j * self._cartesian(x.imag, mode) if not x.val.is_cuda else \
                Field(self._tgt(mode), torch.view_as_complex(torch.stack(
                    [self._cartesian(x.real, mode).val, self._cartesian(x.imag, mode).val], dim=-1).contiguous()))
        return self._cartesian(x, mode)

    def _cartesian(self, x, mode):
        fct = self._factor(mode)
        from . import config

        def host(v, axes):
            f = _host_fftn(v, axes)
            h = f.real + f.imag if config.get("hartley_convention") == "non_canonical_hartley" else f.real - f.imag
            return h if fct == 1 else h * fct

        return Field(self._tgt(mode), self._over_subspace(x.val, host, lambda v, nd: B.hartley(v, ndim=nd, scale=fct)))


class FFTOperator(_RGTransformBase):
    """Complex FFT between an RGSpace and its codomain (reference harmonic_operators.py:35-94)."""

    def __init__(self, domain, target=None, space=None):
        super().__init__(domain, target, space)
        self._capability = self._all_ops

    def apply(self, x, mode):
        self._check_input(x, mode)
        ncells = x.domain[self._space].size
        inverse = x.domain[self._space].harmonic
        fct = self._factor(mode) * (ncells if inverse else 1.0)
        v = x.val
        if not v.is_complex():
            v = v.to(torch.complex64 if v.dtype == torch.float32 else torch.complex128)

        def host(a, axes):
            res = _host_fftn(a, axes, inverse)
            return res if fct == 1 else res * fct

        # ifftn carries 1/N: N * ifftn = unnormalised backward transform
        dev = lambda a, nd: B.fftn(a, ndim=nd, inverse=inverse, scale=fct / ncells if inverse else fct)  # noqa: E731
        return Field(self._tgt(mode), self._over_subspace(v, host, dev))


class HarmonicTransformOperator(LinearOperator):
    """Harmonic space -> position space, real to real (reference harmonic_operators.py:283-337)."""

    def __init__(self, domain, target=None, space=None):
        self._capability = self.TIMES | self.ADJOINT_TIMES
        self._op = self._real_transform(DomainTuple.make(domain), target, space)
        self._domain, self._target = self._op.domain, self._op.target

    @staticmethod
    def _real_transform(domain, target, space):
        """On an RGSpace the real-to-real harmonic transform IS the Hartley transform (restricted to two modes)."""
        source = domain[_space_index(domain, space)]
        if not source.harmonic:
            raise TypeError("HarmonicTransformOperator only works on a harmonic space")
        if not isinstance(source, RGSpace):
            raise NotImplementedError("spherical harmonic transforms are out of scope")
        return HartleyOperator(domain, target, space)

    def apply(self, x, mode):
        self._check_input(x, mode)
        return self._op.apply(x, mode)


def HarmonicSmoothingOperator(domain, sigma, space=None):
    """Smoothing with a Gaussian kernel of width `sigma` (position-space units) on a non-harmonic RGSpace:
    Hartley^-1 . diag(exp(-2 pi^2 sigma^2 k^2)) . Hartley (reference harmonic_operators.py:340-380)."""
    width = float(sigma)
    if width == 0.0:
        return ScalingOperator(domain, 1.0)
    if width < 0.0:
        raise ValueError("sigma must be non-negative")
    return _gaussian_convolution(DomainTuple.make(domain), width, space)


def _gaussian_convolution(domain, width, space):
    """transform, damp every mode by the Fourier image of the Gaussian, transform back"""
    space = _space_index(domain, space)
    if domain[space].harmonic:
        raise TypeError("domain must not be harmonic")
    to_harmonic = HartleyOperator(domain, space=space)
    kspace = to_harmonic.target[space]
    damping = kspace.get_fft_smoothing_kernel_function(width)(kspace.get_k_length_array())
    # the kernel lives on the transformed sub-space only and is broadcast over the others (diagonal_operator.py:51-120)
    return ChainOperator.make([to_harmonic.inverse, DiagonalOperator(damping, to_harmonic.target, space), to_harmonic])



class FFTShiftOperator(EndomorphicOperator):
    """Cyclic shift by half the grid along the axes of the selected RGSpaces -- numpy's ``fftshift`` forwards (TIMES,
    ADJOINT_INVERSE_TIMES), ``ifftshift`` backwards: a permutation, so adjoint = inverse (reference
    operators/harmonic_operators.py:383-423; same arguments).  On a GPU one nk_roll launch moves the data."""

    def __init__(self, domain, spaces=None):
        self._domain = DomainTuple.make(domain)
        self._capability = self._all_ops
        self._axes = tuple(ax for i in self._selected(spaces) for ax in self._domain.axes[i])

    def _selected(self, spaces):
        """Sorted, unique, non-negative positions in the domain tuple; every one of them an RGSpace."""
        count = len(self._domain)
        wanted = range(count) if spaces is None else ([spaces] if isinstance(spaces, (int, np.integer)) else list(spaces))
        picked = set()
        for entry in wanted:
            if not isinstance(entry, (int, np.integer)) or not -count <= entry < count:
                raise AssertionError("spaces: indices into the domain tuple")  # (the reference asserts, :408-415)
            picked.add(int(entry) % count)
        for i in picked:
            if not isinstance(self._domain[i], RGSpace):
                raise AssertionError("FFTShiftOperator only shifts RGSpaces")
        return sorted(picked)

    def apply(self, x, mode):
        self._check_input(x, mode)
        forward = bool(mode & (self.TIMES | self.ADJOINT_INVERSE_TIMES))
        v = x.val
        shifts = [0] * v.dim()
        for ax in self._axes:
            n = v.shape[ax]
            shifts[ax] = n // 2 if forward else -(n // 2)  # fftshift rolls by n // 2, ifftshift by -(n // 2)
        if v.is_cuda:
            return Field(self._tgt(mode), B.roll(v, shifts))
        return Field(self._tgt(mode), torch.roll(v, shifts=shifts, dims=tuple(range(v.dim()))))


class InversionEnabler(EndomorphicOperator):
    """`op` plus the modes it lacks, those computed by a conjugate-gradient solve of ``op_flipped(y) = x`` from y = 0
    (reference operators/inversion_enabler.py:28-80; same arguments).  `approximation`: a linear operator close to `op` that
    HAS the missing modes, used as the preconditioner.  A solve that does not converge logs a warning and returns its last
    iterate, like the reference.  On a GPU the solve is minimization.ConjugateGradient on the nk_cg_* kernels."""

    def __init__(self, op, iteration_controller, approximation=None):
        problems = {"Operator needs to be linear.": not isinstance(op, LinearOperator),
                    "Operator needs to be endomorphic.": isinstance(op, LinearOperator) and op.domain is not op.target
                    and op.domain != op.target}
        for text, bad in problems.items():
            if bad:
                raise TypeError(text)
        self._domain = op.domain
        self._wrapped = dict(op=op, controller=iteration_controller, preconditioner=approximation)
        self._capability = self._add_inverse_capability(op.capability)

    def _by_solve(self, rhs, mode):
        """y with  op|mode^-1 (y) = rhs : `mode` is the inverse of a mode `op` has, so flip the inverse bit and run CG from 0."""
        from .minimization import ConjugateGradient, IterationController, QuadraticEnergy, logger

        which = _mode_index(mode)
        forward = self._wrapped["op"]._flip_modes(which ^ self.INVERSE_BIT)
        approx = self._wrapped["preconditioner"]
        problem = QuadraticEnergy(rhs * 0.0, forward, rhs)
        result, status = ConjugateGradient(self._wrapped["controller"])(
            problem, preconditioner=None if approx is None else approx._flip_modes(which))
        if status != IterationController.CONVERGED:
            logger.warning("Error detected during operator inversion")
        return result.position

    def apply(self, x, mode):
        self._check_mode(mode)
        op = self._wrapped["op"]
        return op.apply(x, mode) if op.capability & mode else self._by_solve(x, mode)

    def draw_sample(self, from_inverse=False, device_id=-1):
        return self._wrapped["op"].draw_sample(from_inverse, device_id)

    def __repr__(self):
        return "InversionEnabler:\n  " + repr(self._wrapped["op"]).replace("\n", "\n  ")


class _JacCountingOperator(EndomorphicOperator):
    """Identity whose applications are tallied per mode in the owner's counter table."""

    def __init__(self, domain, tally):
        self._domain, self._tally = makeDomain(domain), tally
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        self._tally["jac" if mode == self.TIMES else "jac_adj"] += 1
        return x


class CountingOperator(Operator):
    """Identity that counts how often it is applied to fields / Linearizations and how often its Jacobian and
    adjoint Jacobian run (reference counting_operator.py:20-84; optimize_kl.py:370,441 reports them).  All four counters
    live in one table; the Jacobian handed to Linearizations writes into the same table."""

    _ROWS = (("apply", "apply: \t\t"), ("apply_lin", "apply Linearization: \t"), ("jac", "Jacobian: \t\t"),
             ("jac_adj", "Adjoint Jacobian: \t"))

    def __init__(self, domain):
        self._domain = self._target = makeDomain(domain)
        self._tally = dict.fromkeys((key for key, _ in self._ROWS), 0)
        self._derivative = _JacCountingOperator(self._domain, self._tally)

    def apply(self, x):
        self._check_input(x)
        differentiated = is_linearization(x)
        self._tally["apply_lin" if differentiated else "apply"] += 1
        return x.new(x.val, self._derivative) if differentiated else x

    count_apply = property(lambda self: self._tally["apply"])
    count_apply_lin = property(lambda self: self._tally["apply_lin"])
    count_jac = property(lambda self: self._tally["jac"])
    count_jac_adj = property(lambda self: self._tally["jac_adj"])

    def __repr__(self):
        return f"CountingOperator({self._domain!r})"

    def report(self):
        return "\n".join(f"* {label}{self._tally[key]:>7}" for key, label in self._ROWS)


class DOFDistributor(LinearOperator):
    """Gather bins -> pixels (TIMES) and scatter-add pixels -> bins (ADJOINT_TIMES)
    (reference distributors.py:28-127)."""

    def __init__(self, dofdex, target=None, space=None):
        if target is None:
            target = dofdex.domain
        self._target = DomainTuple.make(target)
        self._space = _space_index(self._target, space)
        idx = dofdex.val if isinstance(dofdex, Field) else torch.as_tensor(dofdex)
        if idx.dtype not in (torch.int32, torch.int64):
            raise TypeError("dofdex must contain integers")
        if idx.numel() and int(idx.min()) < 0:
            raise ValueError("Negative dof encountered")
        nbin = int(idx.max()) + 1
        self._init2(idx, nbin, UnstructuredDomain(nbin))

    def _init2(self, idx, nbin, other_space):
        """idx: tensor of bin indices, or a zero-argument callable producing it on first HOST use (large PowerSpaces
        never materialise the 8 N byte host index: device applications fetch PowerSpace.device_pindex instead)."""
        self._idx_src = idx
        self._idx_host = None
        self._idx32 = {}
        self._bin_plans = {}
        self._nbin = nbin
        # the other spaces of the target (e.g. the UnstructuredDomain(total_N) of several correlated fields, or spaces
        # BEHIND the distributed one, distributors.py:60-104) are carried along: batch rows of the gather / scatter
        self._domain = DomainTuple.make([other_space if i == self._space else sub for i, sub in enumerate(self._target)])
        self._nlead = int(np.prod([sub.size for i, sub in enumerate(self._target) if i != self._space])) \
            if len(self._target) > 1 else 0
        self._capability = self.TIMES | self.ADJOINT_TIMES

    @property
    def _idx(self):
        if self._idx_host is None:
            src = self._idx_src() if callable(self._idx_src) else self._idx_src
            self._idx_host = src.reshape(-1).to(torch.int64)
        return self._idx_host

    def _device_index(self, device):
        key = str(device)
        if key not in self._idx32:
            pspace = getattr(self, "_pspace", None)
            if pspace is not None:
                self._idx32[key] = pspace.device_pindex(device)
            else:
                self._idx32[key] = self._idx.to(torch.int32).to(device).contiguous()
        return self._idx32[key]

    def apply(self, x, mode):
        self._check_input(x, mode)
        if self._nlead:
            return Field(self._tgt(mode), self._apply_rows(x.val, mode))
        tshape = self._target.shape
        return Field(self._tgt(mode), self._apply1(x.val, mode, tshape).reshape(self._tgt(mode).shape))

    def _apply_rows(self, v, mode):
        """One gather / scatter per combination of the carried-along indices.  The distributed space is moved to the back
        first when other spaces follow it (a permutation copy), and back afterwards; slices and stacking are copies too."""
        src, dst = self._dom(mode), self._tgt(mode)
        grid_shape = tuple(self._target[self._space].shape)
        mine = list(src.axes[self._space])
        others = [ax for ax in range(len(src.shape)) if ax not in mine]
        rows = v.permute(others + mine).contiguous().reshape((self._nlead, -1))
        outs = torch.stack([self._apply1(rows[i], mode, grid_shape).reshape(-1) for i in range(self._nlead)])
        # back to the natural axis order of the output domain
        dst_mine = list(dst.axes[self._space])
        dst_others = [ax for ax in range(len(dst.shape)) if ax not in dst_mine]
        moved = outs.reshape([dst.shape[ax] for ax in dst_others] + [dst.shape[ax] for ax in dst_mine])
        order = dst_others + dst_mine
        return moved.permute([order.index(ax) for ax in range(len(dst.shape))]).contiguous()

    def _apply1(self, v, mode, tshape):
        if v.is_complex():  # real and imaginary parts are distributed / collected separately (the kernels are real)
            parts = [self._apply1(part.contiguous(), mode, tshape) for part in (v.real, v.imag)]
            return torch.complex(*parts)
        if mode == self.TIMES:
            if v.is_cuda:
                return B.gather(v.contiguous(), self._device_index(v.device), tshape)
            return v[self._idx.to(v.device)].reshape(tshape)
        if v.is_cuda:
            # static index map: every bin is summed in a fixed order over the bin-sorted permutation of the pixels (made
            # once per device); only maps beyond BIN_PLAN_MAX points keep the fp64 atomics
            key = str(v.device)
            if key not in self._bin_plans:
                idx = self._device_index(v.device)
                self._bin_plans[key] = B.bin_plan(idx, self._nbin) if idx.numel() <= B.BIN_PLAN_MAX else None
            plan = self._bin_plans[key]
            if plan is not None:
                return B.bin_sum(v.contiguous().reshape(-1), plan)
            bins = B.scatter_add(v.contiguous().reshape(-1), self._device_index(v.device), self._nbin)
            return bins.to(v.dtype)
        out = torch.zeros(self._nbin, dtype=torch.float64)
        out.index_add_(0, self._idx, v.reshape(-1).to(torch.float64))
        return out.to(v.dtype)


class PowerDistributor(DOFDistributor):
    """PowerSpace -> harmonic space by pindex (reference distributors.py:130-161)."""

    def __init__(self, target, power_space=None, space=None):
        self._target = DomainTuple.make(target)
        self._space = _space_index(self._target, space)
        self._pspace = bins = self._bins_of(self._target[self._space], power_space)
        # the host copy of the bin index is only built if a HOST field is ever distributed
        self._init2(lambda: torch.from_numpy(np.array(bins.pindex)), bins.shape[0], bins)

    @staticmethod
    def _bins_of(grid, power_space):
        """The PowerSpace whose bins are distributed over `grid` (the default binning when none is given)."""
        if not grid.harmonic:
            raise ValueError("Operator requires harmonic target space")
        if power_space is None:
            return PowerSpace(grid)
        if not isinstance(power_space, PowerSpace):
            raise TypeError("power_space argument must be a PowerSpace")
        if power_space.harmonic_partner != grid:
            raise ValueError("power_space does not match its partner")
        return power_space


class MaskOperator(LinearOperator):
    """Keeps the un-flagged pixels of a field in an UnstructuredDomain (reference mask_operator.py:26-58): flags are
    converted to boolean, True = flagged.  TIMES compresses, ADJOINT_TIMES expands with zeros."""

    def __init__(self, flags):
        if not isinstance(flags, Field):
            raise TypeError("flags must be a Field")
        flagged = flags.val.reshape(-1).to(torch.bool).cpu()
        self._keep = torch.nonzero(~flagged).reshape(-1)  # int64 flat indices of the kept pixels, ascending
        self._idx32 = {}
        self._capability = self.TIMES | self.ADJOINT_TIMES
        self._domain = DomainTuple.make(flags.domain)
        self._target = DomainTuple.make(UnstructuredDomain(int(self._keep.numel())))

    def _device_index(self, device):
        key = str(device)
        if key not in self._idx32:
            self._idx32[key] = self._keep.to(torch.int32).to(device).contiguous()
        return self._idx32[key]

    def apply(self, x, mode):
        self._check_input(x, mode)
        v = x.val
        if mode == self.TIMES:
            if v.is_cuda:
                return Field(self._target, B.gather(v.contiguous().reshape(-1), self._device_index(v.device),
                                                    self._target.shape))
            return Field(self._target, v.reshape(-1)[self._keep])
        if v.is_cuda:  # unique indices: the scatter-add kernel is a plain scatter here
            full = B.scatter_add(v.contiguous(), self._device_index(v.device), self._domain.size)
            return Field(self._domain, full.to(v.dtype).reshape(self._domain.shape))
        out = torch.zeros(self._domain.size, dtype=v.dtype)
        out[self._keep] = v
        return Field(self._domain, out.reshape(self._domain.shape))


# ================================================================================================
# sampling
# ================================================================================================
class SamplingEnabler(EndomorphicOperator):
    """(likelihood + prior) with sampling from its inverse via CG (reference sampling_enabler.py:27-97).

    A draw from the inverse of  A = likelihood + prior  is the solution y of  A y = b  for a right-hand side with
    covariance A: b = prior(s) + nj with s ~ prior^-1 and nj ~ likelihood (or, `start_from_zero`, one draw b ~ A solved
    from y = 0).  The CG starts at the prior draw s, where the residual A s - b = likelihood(s) - nj is known without
    another application of the prior."""

    def __init__(self, likelihood, prior, iteration_controller, approximation=None, start_from_zero=False):
        for op in (likelihood, prior):
            if not is_operator(op):
                raise TypeError("likelihood and prior must be operators")
        self._op = likelihood + prior
        self._domain, self._capability = self._op.domain, self._op.capability
        self._likelihood, self._prior = likelihood, prior
        self._ic, self._approximation, self._start_from_zero = iteration_controller, approximation, bool(start_from_zero)

    def apply(self, x, mode):
        return self._op.apply(x, mode)

    def _linear_problem(self, device_id):
        """(b, quadratic energy of A y = b at the CG's starting point)"""
        from .minimization import QuadraticEnergy

        if self._start_from_zero:
            b = self._op.draw_sample(device_id=device_id)
            return b, QuadraticEnergy(b * 0.0, self._op, b)
        start = self._prior.draw_sample(from_inverse=True, device_id=device_id)
        noise = self._likelihood.draw_sample(device_id=device_id)
        b = self._prior(start) + noise
        return b, QuadraticEnergy(start, self._op, b, _grad=self._likelihood(start) - noise)

    def special_draw_sample(self, from_inverse=False, device_id=-1):
        from .minimization import ConjugateGradient

        y = self._direct_draw(from_inverse, device_id)
        if y is None:
            b, problem = self._linear_problem(device_id)
            extra = {} if self._approximation is None else dict(preconditioner=self._approximation.inverse)
            return b, ConjugateGradient(self._ic)(problem, **extra)[0].position
        return self._op(y), y

    def _direct_draw(self, from_inverse, device_id):
        """A sample straight from the operator, or None when only the CG route is open (which needs from_inverse)."""
        try:
            return self._op.draw_sample(from_inverse, device_id)
        except NotImplementedError:
            if not from_inverse:
                raise ValueError("from_inverse must be True here") from None
            return None

    def draw_sample(self, from_inverse=False, device_id=-1):
        return self.special_draw_sample(from_inverse, device_id)[1]

    def __repr__(self):
        return "SamplingEnabler:\n  Likelihood:\n    " + repr(self._likelihood) + "\n  Prior:\n    " + repr(self._prior)


def WienerFilterCurvature(R, N, S, iteration_controller=None, iteration_controller_sampling=None):
    """R^dagger N^-1 R + S^-1, the inverse of the Wiener-filter propagator, invertible by conjugate gradient with S^-1 as
    preconditioner and -- with `iteration_controller_sampling` -- able to draw from its inverse (reference
    library/wiener_filter_curvature.py:25-66; `N` and `S` must implement `draw_sample` for that)."""
    for cov in (N, S):
        if not isinstance(cov, EndomorphicOperator):
            raise TypeError("noise and signal covariance must be endomorphic operators")
    data_term, prior_term = SandwichOperator.make(R, N.inverse), S.inverse
    if iteration_controller_sampling is not None:
        curvature = SamplingEnabler(data_term, prior_term, iteration_controller_sampling, prior_term)
    else:
        curvature = data_term + prior_term
    return InversionEnabler(curvature, iteration_controller, prior_term)
