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


# ================================================================================================
# Operator
# ================================================================================================
class Operator:
    """Possibly nonlinear map between (Multi)Domains (reference operators/operator.py:30-430)."""

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
        if not (is_fieldlike(x) or is_linearization(x)):
            raise TypeError("operators act on Fields, MultiFields or Linearizations")
        if is_linearization(x):
            if not isinstance(x.jac, ScalingOperator) or x.jac._factor != 1:
                raise ValueError("apply() expects a Linearization with trivial Jacobian")
        _same_domain(self._domain, x.domain)

    def __call__(self, x):
        if is_linearization(x):
            return self.apply(x.trivial_jac()).prepend_jac(x.jac)
        if is_fieldlike(x):
            return self.apply(x)
        if is_operator(x):
            return self @ x
        raise TypeError(f"cannot apply operator to {type(x)}")

    # -- algebra ---------------------------------------------------------------------------------
    def __matmul__(self, x):
        from .energy_operators import LikelihoodEnergyOperator

        if is_operator(x) and not isinstance(x, LikelihoodEnergyOperator):
            if x.target is self.domain:
                if x.isIdentity():
                    return self
                if self.isIdentity():
                    return x
                return _OpChain.make((self, x))
            return self.partial_insert(x)
        return NotImplemented

    def partial_insert(self, x):
        if not (isinstance(self.domain, MultiDomain) and isinstance(x.target, MultiDomain)):
            raise TypeError("partial_insert needs MultiDomains")
        bigdom = MultiDomain.union([self.domain, x.target])
        k1, k2 = set(self.domain.keys()), set(x.target.keys())
        le, ri = k2 - k1, k1 - k2
        leop, riop = self, x
        if ri:
            riop = riop + Operator.identity_operator(MultiDomain.make({k: bigdom[k] for k in ri}))
        if le:
            leop = leop + Operator.identity_operator(MultiDomain.make({k: bigdom[k] for k in le}))
        return leop @ riop

    @staticmethod
    def identity_operator(dom):
        dom = makeDomain(dom)
        if isinstance(dom, DomainTuple):
            return ScalingOperator(dom, 1.0)
        return BlockDiagonalOperator(dom, {k: ScalingOperator(d, 1.0) for k, d in dom.items()})

    def scale(self, factor):
        if not isinstance(factor, Number):
            raise TypeError(".scale() takes a number as input")
        if factor == 1:
            return self
        return ScalingOperator(self.target, factor)(self)

    def __neg__(self):
        return self.scale(-1)

    def __mul__(self, x):
        if is_operator(x):
            return _OpProd(self, x)
        if isinstance(x, Number):
            return self.scale(x)
        if is_fieldlike(x):
            return makeOp(x) @ self
        return NotImplemented

    __rmul__ = __mul__

    def __add__(self, x):
        if is_operator(x):
            return _OpSum(self, x)
        if isinstance(x, Number):
            return Adder(full(self.target, float(x))) @ self
        if is_fieldlike(x):
            return Adder(x) @ self
        return NotImplemented

    __radd__ = __add__

    def __sub__(self, x):
        if is_operator(x):
            return _OpSum(self, -x)
        if isinstance(x, Number):
            return Adder(full(self.target, float(x)), neg=True) @ self
        if is_fieldlike(x):
            return Adder(x, neg=True) @ self
        return NotImplemented

    def __rsub__(self, x):
        return x + (-self)

    def __truediv__(self, x):
        if isinstance(x, Number):
            return self.scale(1.0 / x)
        if is_operator(x):
            return self * x.reciprocal()
        return NotImplemented

    def __pow__(self, power):
        if isinstance(power, Number):
            return self.ptw("power", power)
        return NotImplemented

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
        return self._simplify_for_constant_input_nontrivial(c_inp)

    def _simplify_for_constant_input_nontrivial(self, c_inp):
        return None, self @ InsertionOperator(self.domain, c_inp)

    def ptw(self, op, *args, **kwargs):
        return _OpChain.make((_FunctionApplier(self.target, op, *args, **kwargs), self))

    def ptw_pre(self, op, *args, **kwargs):
        return _OpChain.make((self, _FunctionApplier(self.domain, op, *args, **kwargs)))

    @property
    def real(self):
        return Realizer(self.target)(self)

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
        if isinstance(name, str):
            return self @ ducktape(self, None, name)
        raise NotImplementedError("ducktape to a new domain is not implemented")

    def ducktape_left(self, name):
        if isinstance(name, str):
            return ducktape(None, self.target, name)(self)
        raise NotImplementedError

    def __repr__(self):
        return self.__class__.__name__


for _fn in ("exp", "log", "sqrt", "tanh", "sigmoid", "reciprocal", "log1p", "expm1", "abs", "sin", "cos", "arctan"):
    def _mk(fn):
        def method(self):
            return self.ptw(fn)
        method.__name__ = fn
        return method
    setattr(Operator, _fn, _mk(_fn))


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


class _OpChain(Operator):
    def __init__(self, ops, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError
        self._ops = tuple(ops)
        self._domain, self._target = self._ops[-1].domain, self._ops[0].target
        for a, b in zip(self._ops[:-1], self._ops[1:]):
            _same_domain(a.domain, b.target)

    @classmethod
    def make(cls, ops):
        flat = []
        for op in ops:
            flat.extend(op._ops if isinstance(op, _OpChain) else [op])
        return flat[0] if len(flat) == 1 else cls(flat, _callingfrommake=True)

    def apply(self, x):
        self._check_input(x)
        for op in reversed(self._ops):
            x = op(x)
        return x

    def __repr__(self):
        return "_OpChain:\n" + "\n".join("  " + repr(o).replace("\n", "\n  ") for o in self._ops)


def domain_union(domains):
    if all(isinstance(d, DomainTuple) for d in domains):
        for d in domains[1:]:
            _same_domain(domains[0], d)
        return domains[0]
    return MultiDomain.union(domains)


class _OpProd(Operator):
    """Pointwise product of two operators with the product rule (operator.py:555-600)."""

    def __init__(self, op1, op2):
        self._domain = domain_union((op1.domain, op2.domain))
        if op1.target != op2.target:
            raise ValueError("target mismatch")
        self._target = op1.target
        self._op1, self._op2 = op1, op2

    def apply(self, x):
        self._check_input(x)
        lin = is_linearization(x)
        wm = x.want_metric if lin else False
        v = x.val if lin else x
        v1, v2 = v.extract(self._op1.domain), v.extract(self._op2.domain)
        if not lin:
            return self._op1(v1) * self._op2(v2)
        l1 = self._op1(Linearization.make_var(v1, wm))
        l2 = self._op2(Linearization.make_var(v2, wm))
        jac = makeOp(l1.val)(l2.jac)._myadd(makeOp(l2.val)(l1.jac), False)
        return l1.new(l1.val * l2.val, jac)

    def __repr__(self):
        return "_OpProd:\n  " + repr(self._op1) + "\n  " + repr(self._op2)


class _OpSum(Operator):
    def __init__(self, op1, op2):
        self._domain = domain_union((op1.domain, op2.domain))
        self._target = domain_union((op1.target, op2.target))
        self._op1, self._op2 = op1, op2

    def apply(self, x):
        self._check_input(x)
        return self._apply_operator_sum(x, [self._op1, self._op2])

    @staticmethod
    def _apply_operator_sum(x, ops):
        """operator.py:619-641"""
        if not is_linearization(x):
            res = None
            for op in ops:
                tmp = op.force(x)
                res = tmp if res is None else res.flexible_addsub(tmp, False)
            return res
        wm = x.want_metric
        lins = [op(Linearization.make_var(x.val.extract(op.domain), wm)) for op in ops]
        val, jac = lins[0].val, lins[0].jac
        for ll in lins[1:]:
            val = val.flexible_addsub(ll.val, False)
            jac = jac._myadd(ll.jac, False)
        res = x.new(val, jac)
        if all(ll.metric is not None for ll in lins):
            met = lins[0].metric
            for ll in lins[1:]:
                met = met._myadd(ll.metric, False)
            res = res.add_metric(met)
        return res

    def __repr__(self):
        return "_OpSum:\n  " + repr(self._op1) + "\n  " + repr(self._op2)


# ================================================================================================
# Linearization (forward-mode AD object)
# ================================================================================================
class Linearization:
    """Value + Jacobian (+ metric) of an operator at a point (reference linearization.py:26-435)."""

    def __init__(self, val, jac, metric=None, want_metric=False):
        _same_domain(val.domain, jac.target)
        self._val, self._jac, self._metric, self._want_metric = val, jac, metric, want_metric

    def new(self, val, jac, metric=None):
        return Linearization(val, jac, metric, self._want_metric)

    def trivial_jac(self):
        return Linearization.make_var(self._val, self._want_metric)

    def prepend_jac(self, jac):
        if jac.isIdentity():
            return self
        new_jac = jac if self._jac.isIdentity() else self._jac @ jac
        if self._metric is None:
            return self.new(self._val, new_jac)
        return self.new(self._val, new_jac, SandwichOperator.make(jac, self._metric))

    domain = property(lambda self: self._jac.domain)
    target = property(lambda self: self._jac.target)
    val = property(lambda self: self._val)
    jac = property(lambda self: self._jac)
    want_metric = property(lambda self: self._want_metric)
    metric = property(lambda self: self._metric)
    device_id = property(lambda self: self._val.device_id)

    @property
    def gradient(self):
        return self._jac.adjoint_times(Field.scalar(1.0).at(self._val.device_id))

    def __getitem__(self, name):
        if not isinstance(self.target, MultiDomain):
            raise TypeError("not subscriptable")
        return self.new(self._val[name], ducktape(None, self._jac.target, name)(self._jac))

    def __neg__(self):
        if self._metric is not None:
            raise RuntimeError("Cannot negate operators with metric")
        return self.new(-self._val, -self._jac)

    @property
    def real(self):
        return self.new(self._val.real, Realizer(self._jac.target)(self._jac))

    def _myadd(self, other, neg):
        if np.isscalar(other) or is_fieldlike(other):
            return self.new(self._val - other if neg else self._val + other, self._jac, self._metric)
        if not is_linearization(other):
            return NotImplemented
        met = None
        if self._metric is not None and other._metric is not None:
            met = self._metric._myadd(other._metric, neg)
        return self.new(self._val.flexible_addsub(other._val, neg), self._jac._myadd(other._jac, neg), met)

    def __add__(self, o): return self._myadd(o, False)
    __radd__ = __add__
    def __sub__(self, o): return self._myadd(o, True)
    def __rsub__(self, o): return (-self).__add__(o)

    def __mul__(self, other):
        if np.isscalar(other):
            if other == 1:
                return self
            met = None if self._metric is None else self._metric.scale(other)
            return self.new(self._val * other, self._jac.scale(other), met)
        if is_fieldlike(other):
            _same_domain(self.target, other.domain)
            return self.new(self._val * other, makeOp(other)(self._jac))
        if is_linearization(other):
            _same_domain(self.target, other.target)
            return self.new(self._val * other._val,
                            makeOp(other._val)(self._jac)._myadd(makeOp(self._val)(other._jac), False))
        return NotImplemented

    __rmul__ = __mul__

    def __truediv__(self, other):
        if np.isscalar(other):
            return self.__mul__(1.0 / other)
        return self.__mul__(other.ptw("reciprocal"))

    def __rtruediv__(self, other):
        return self.ptw("reciprocal").__mul__(other)

    def __pow__(self, power):
        if np.isscalar(power):
            return self.ptw("power", power)
        return NotImplemented

    def vdot(self, other):
        if is_fieldlike(other):
            return self.new(self._val.vdot(other.at(self._val.device_id)), VdotOperator(other)(self._jac))
        return self.new(self._val.vdot(other._val),
                        VdotOperator(self._val)(other._jac) + VdotOperator(other._val)(self._jac))

    def sum(self, spaces=None):
        return self.new(self._val.sum(spaces) if spaces is not None else Field.scalar(self._val.s_sum()).at(self.device_id),
                        ContractionOperator(self._jac.target, spaces)(self._jac))

    def ptw(self, op, *args, **kwargs):
        f, df = self._val.ptw_with_deriv(op, *args, **kwargs)
        return self.new(f, makeOp(df)(self._jac))

    def add_metric(self, metric):
        return self.new(self._val, self._jac, metric)

    def with_want_metric(self):
        return Linearization(self._val, self._jac, self._metric, True)

    @staticmethod
    def make_var(field, want_metric=False):
        return Linearization(field, ScalingOperator(field.domain, 1.0), want_metric=want_metric)

    @staticmethod
    def make_const(field, want_metric=False):
        return Linearization(field, NullOperator(field.domain, field.domain), want_metric=want_metric)


for _fn in ("exp", "log", "sqrt", "tanh", "sigmoid", "reciprocal", "log1p", "expm1", "abs", "arctan"):
    def _mk2(fn):
        def method(self):
            return self.ptw(fn)
        method.__name__ = fn
        return method
    setattr(Linearization, _fn, _mk2(_fn))


# ================================================================================================
# LinearOperator
# ================================================================================================
class LinearOperator(Operator):
    """Linear map with TIMES / ADJOINT_TIMES / INVERSE_TIMES / ADJOINT_INVERSE_TIMES modes
    (reference operators/linear_operator.py:24-262)."""

    TIMES, ADJOINT_TIMES, INVERSE_TIMES, ADJOINT_INVERSE_TIMES = 1, 2, 4, 8
    INVERSE_ADJOINT_TIMES = 8
    ADJOINT_BIT, INVERSE_BIT = 1, 2
    _ilog = (-1, 0, 1, -1, 2, -1, -1, -1, 3)
    _validMode = (False, True, True, False, True, False, False, False, True)
    _modeTable = ((1, 2, 4, 8), (2, 1, 8, 4), (4, 8, 1, 2), (8, 4, 2, 1))
    _backwards = 6
    _all_ops = 15

    @staticmethod
    def _flip_capability(cap, trafo):
        res = 0
        for bit in (1, 2, 4, 8):
            if cap & bit:
                res |= LinearOperator._modeTable[trafo][LinearOperator._ilog[bit]]
        return res

    @staticmethod
    def _add_inverse_capability(cap):
        return cap | LinearOperator._flip_capability(cap, LinearOperator.INVERSE_BIT)

    def _dom(self, mode):
        return self.domain if (mode & 9) else self.target

    def _tgt(self, mode):
        return self.domain if (mode & 6) else self.target

    def _flip_modes(self, trafo):
        return self if trafo == 0 else OperatorAdapter(self, trafo)

    @property
    def inverse(self):
        return self._flip_modes(self.INVERSE_BIT)

    @property
    def adjoint(self):
        return self._flip_modes(self.ADJOINT_BIT)

    @property
    def capability(self):
        return self._capability

    def __matmul__(self, other):
        if is_operator(other) and other.isIdentity():
            return self
        if isinstance(other, LinearOperator):
            return ChainOperator.make([self, other])
        return Operator.__matmul__(self, other)

    def __rmatmul__(self, other):
        if isinstance(other, LinearOperator):
            return ChainOperator.make([other, self])
        return NotImplemented

    def _myadd(self, other, oneg):
        return SumOperator.make((self, other), (False, oneg))

    def __add__(self, other):
        if isinstance(other, LinearOperator):
            return self._myadd(other, False)
        return Operator.__add__(self, other)

    __radd__ = __add__

    def __sub__(self, other):
        if isinstance(other, LinearOperator):
            return self._myadd(other, True)
        return Operator.__sub__(self, other)

    def __neg__(self):
        return self.scale(-1)

    def scale(self, factor):
        if not isinstance(factor, Number):
            raise TypeError(".scale() takes a number as input")
        if factor == 1:
            return self
        return ChainOperator.make([ScalingOperator(self.target, factor), self])

    def force(self, x):
        return self.apply(x.extract(self.domain), self.TIMES)

    def apply(self, x, mode):
        raise NotImplementedError

    def __call__(self, x):
        if self.isIdentity():
            return x
        if is_linearization(x):
            return x.new(self(x.val), self).prepend_jac(x.jac)
        if is_fieldlike(x):
            return self.apply(x, self.TIMES)
        if is_operator(x):
            return self @ x
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
        if not (0 <= mode < 9 and self._validMode[mode]):
            raise NotImplementedError("invalid operator mode specified")
        if mode & self.capability == 0:
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

    def __init__(self, op, trafo):
        self._op, self._trafo = op, int(trafo)
        if self._trafo < 1 or self._trafo > 3:
            raise ValueError("invalid operator transformation")
        self._domain = op._dom(1 << self._trafo)
        self._target = op._tgt(1 << self._trafo)
        self._capability = self._flip_capability(op.capability, self._trafo)

    def _flip_modes(self, trafo):
        newtrafo = trafo ^ self._trafo
        return self._op if newtrafo == 0 else OperatorAdapter(self._op, newtrafo)

    def apply(self, x, mode):
        return self._op.apply(x, self._modeTable[self._trafo][self._ilog[mode]])

    def draw_sample(self, from_inverse=False, device_id=-1):
        if self._trafo & self.INVERSE_BIT:
            return self._op.draw_sample(not from_inverse, device_id)
        return self._op.draw_sample(from_inverse, device_id)

    def __repr__(self):
        return "OperatorAdapter({}) of\n  ".format(["", "adjoint", "inverse", "adjoint inverse"][self._trafo]) + repr(self._op)


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
        if not isinstance(target, MultiDomain):
            raise TypeError
        if not isinstance(cst_field, MultiField):
            raise TypeError
        self._target = MultiDomain.make(target)
        self._domain = MultiDomain.make({k: self._target[k] for k in self._target.keys() if k not in cst_field.keys()})
        self._cst = cst_field
        self._jac = _KeyEmbedding(self._domain, self._target)

    def apply(self, x):
        self._check_input(x)
        val = x.val if is_linearization(x) else x
        val = val.unite(self._cst.at(val.device_id))
        return x.new(val, self._jac) if is_linearization(x) else val

    def __repr__(self):
        return f"InsertionOperator\n  Constant: {self._cst.keys()}\n  Variable: {self._domain.keys()}"


class NullOperator(LinearOperator):
    def __init__(self, domain, target):
        self._domain, self._target = makeDomain(domain), makeDomain(target)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        return full(self._tgt(mode), 0.0, x.device_id)


class ScalingOperator(EndomorphicOperator):
    """Multiplication by a scalar (reference scaling_operator.py:25-139)."""

    def __init__(self, domain, factor, sampling_dtype=None):
        if isinstance(factor, Field) and factor.shape == ():
            factor = factor.asnumpy()[()]
        if not np.isscalar(factor):
            raise TypeError("Scalar required")
        self._domain = makeDomain(domain)
        self._factor = factor
        self._capability = self._all_ops
        self._dtype = sampling_dtype

    def isIdentity(self):
        return self._factor == 1

    def apply(self, x, mode):
        self._check_input(x, mode)
        fct = self._factor
        if fct == 1.0:
            return x
        if fct == 0.0:
            return full(x.domain, 0.0, x.device_id)
        if mode & (self.ADJOINT_TIMES | self.ADJOINT_INVERSE_TIMES):
            fct = np.conj(fct)
        if mode & (self.INVERSE_TIMES | self.ADJOINT_INVERSE_TIMES):
            fct = 1.0 / fct
        return x * fct

    def _flip_modes(self, trafo):
        fct = self._factor
        if trafo & self.ADJOINT_BIT:
            fct = np.conj(fct)
        if trafo & self.INVERSE_BIT:
            fct = 1.0 / fct
        return ScalingOperator(self._domain, fct, self._dtype)

    def _get_fct(self, from_inverse):
        fct = self._factor
        if np.imag(fct) != 0.0 or np.real(fct) < 0.0 or (np.real(fct) == 0.0 and from_inverse):
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
        return ScalingOperator(self._domain, fct)

    def __call__(self, other):
        res = EndomorphicOperator.__call__(self, other)
        if is_linearization(other) and np.isreal(self._factor) and self._factor >= 0 and other.metric is not None:
            sq = ScalingOperator(other.metric.domain, np.sqrt(self._factor), self._dtype)
            res = res.add_metric(SandwichOperator.make(sq, other.metric))
        return res

    def __repr__(self):
        s = f"ScalingOperator ({self._factor}"
        if self._dtype is not None:
            s += f", sampling dtype {self._dtype}"
        return s + ")"


class DiagonalOperator(EndomorphicOperator):
    """Pointwise multiplication by a Field, optionally living on a sub-set of the spaces
    (reference diagonal_operator.py:51-260)."""

    def __init__(self, diagonal, domain=None, spaces=None, sampling_dtype=None, _trafo=0):
        if not isinstance(diagonal, Field):
            raise TypeError("Field object required")
        self._dtype, self._trafo = sampling_dtype, _trafo
        self._domain = diagonal.domain if domain is None else DomainTuple.make(domain)
        if spaces is None:
            self._spaces = None
            _same_domain(diagonal.domain, self._domain)
            self._ldiag = diagonal.val
        else:
            spaces = (spaces,) if np.isscalar(spaces) else tuple(spaces)
            if len(spaces) != len(diagonal.domain):
                raise ValueError("spaces and domain must have the same length")
            for i, j in enumerate(spaces):
                if diagonal.domain[i] != self._domain[j]:
                    raise ValueError("Mismatch between:\n{}\nand:\n{}".format(diagonal.domain[i], self._domain[j]))
            self._spaces = None if spaces == tuple(range(len(self._domain))) else spaces
            if self._spaces is None:
                self._ldiag = diagonal.val
            else:
                active = [a for s in spaces for a in self._domain.axes[s]]
                shp = [n if i in active else 1 for i, n in enumerate(self._domain.shape)]
                self._ldiag = diagonal.val.reshape(shp)
        self._complex = self._ldiag.is_complex()
        self._capability = self._all_ops
        self._diagmin_cache = None

    @staticmethod
    def _from_ldiag(proto, ldiag, sampling_dtype, trafo, spaces):
        res = DiagonalOperator.__new__(DiagonalOperator)
        res._dtype, res._trafo, res._domain, res._spaces = sampling_dtype, trafo, proto._domain, spaces
        res._ldiag = ldiag
        res._complex = ldiag.is_complex()
        res._capability = proto._all_ops
        res._diagmin_cache = None
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
        if conj and self._complex:
            d = torch.conj_physical(d)
        if not xval.is_cuda:
            return xval / d if divide else xval * d
        if self._complex or xval.is_complex():
            raise NotImplementedError("complex DiagonalOperator on device")
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
        trafo = self._ilog[mode] ^ self._trafo
        return Field(x.domain, self._mul(x.val, divide=bool(trafo & 2), conj=bool(trafo & 1)))

    def _actual_diag(self):
        d = self._ldiag
        if self._trafo & 2:
            d = 1.0 / d if not d.is_cuda else B.binary(3, 1.0, d)
        if self._trafo & 1 and self._complex:
            d = torch.conj_physical(d)
        return d

    def _flip_modes(self, trafo):
        return DiagonalOperator._from_ldiag(self, self._ldiag, self._dtype, self._trafo ^ trafo, self._spaces)

    def _scale(self, fct):
        d = self._actual_diag()
        d = d * fct if not d.is_cuda else B.binary(2, d, float(fct))
        return DiagonalOperator._from_ldiag(self, d, self._dtype, 0, self._spaces)

    def _combine_prod(self, op):
        if self._spaces != op._spaces and not (self._full() and op._full()):
            a, b = self._actual_diag(), op._actual_diag()
            if a.is_cuda:
                raise NotImplementedError
            return DiagonalOperator._from_ldiag(self, a * b, self._dtype if self._dtype == op._dtype else None, 0, None
                                                if (self._full() or op._full()) else tuple(set(self._spaces) | set(op._spaces)))
        a, b = self._actual_diag(), op._actual_diag()
        prod = a * b if not a.is_cuda else B.binary(2, a, b)
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
        self._domain = domain
        self._ops = tuple(operators[k] for k in domain.keys())
        self._capability = self._all_ops
        for op in self._ops:
            if op is not None:
                self._capability &= op.capability

    def apply(self, x, mode):
        self._check_input(x, mode)
        vals = tuple(op.apply(v, mode=mode) if op is not None else v for op, v in zip(self._ops, x.values()))
        return MultiField(self._domain, vals)

    def _flip_modes(self, trafo):
        return BlockDiagonalOperator(self._domain, {k: None if op is None else op._flip_modes(trafo)
                                                    for k, op in zip(self._domain.keys(), self._ops)})

    def draw_sample(self, from_inverse=False, device_id=-1):
        vals = tuple(op.draw_sample(from_inverse, device_id) for op in self._ops)
        return MultiField(self._domain, vals)

    def get_sqrt(self):
        return BlockDiagonalOperator(self._domain, {k: op.get_sqrt() for k, op in zip(self._domain.keys(), self._ops)})

    @property
    def sampling_dtype(self):
        return {k: getattr(op, "sampling_dtype", None) for k, op in zip(self._domain.keys(), self._ops)}


def makeOp(inp, dom=None, sampling_dtype=None):
    """Diagonal operator from a scalar / Field / MultiField (reference sugar.py:410-458)."""
    if inp is None:
        return None
    if np.isscalar(inp):
        if not isinstance(dom, (DomainTuple, MultiDomain)):
            raise TypeError("need proper `dom` argument")
        return ScalingOperator(dom, inp, sampling_dtype)
    if dom is not None and not isinstance(inp, Field):
        raise TypeError("dom only allowed for Fields")
    if isinstance(inp, Field):
        if inp.domain is DomainTuple.scalar_domain() and dom is None:
            return ScalingOperator(inp.domain, inp.asnumpy()[()], sampling_dtype)
        return DiagonalOperator(inp, sampling_dtype=sampling_dtype) if dom is None else \
            DiagonalOperator(inp, domain=dom, spaces=tuple(range(len(inp.domain))), sampling_dtype=sampling_dtype)
    if isinstance(inp, MultiField):
        dts = sampling_dtype if isinstance(sampling_dtype, dict) else {k: sampling_dtype for k in inp.keys()}
        return BlockDiagonalOperator(inp.domain, {k: makeOp(v, sampling_dtype=dts[k]) for k, v in inp.items()})
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
        self._capability = self._all_ops
        for op in ops:
            self._capability &= op.capability
        self._domain, self._target = ops[-1].domain, ops[0].target

    @staticmethod
    def simplify(ops):
        # flatten
        flat = []
        for op in ops:
            flat.extend(op._ops if isinstance(op, ChainOperator) else [op])
        # pull scalars to the front and merge them
        fct, rest, sdt = 1.0, [], None
        for op in flat:
            if isinstance(op, ScalingOperator) and np.isscalar(op._factor) and op.domain is op.target:
                fct = fct * op._factor
                sdt = op._dtype if op._dtype is not None else sdt
            else:
                rest.append(op)
        # merge neighbouring full-domain diagonals
        merged = []
        for op in rest:
            if merged and isinstance(op, DiagonalOperator) and isinstance(merged[-1], DiagonalOperator) \
                    and op._full() and merged[-1]._full():
                merged[-1] = merged[-1]._combine_prod(op)
            else:
                merged.append(op)
        if fct != 1 or not merged:
            if merged and isinstance(merged[0], DiagonalOperator) and merged[0]._full():
                merged[0] = merged[0]._scale(fct)
            else:
                dom = merged[0].target if merged else flat[0].target
                merged.insert(0, ScalingOperator(dom, fct, sdt))
        return merged

    @staticmethod
    def make(ops):
        ops = tuple(ops)
        if len(ops) == 0:
            raise ValueError("ops is empty")
        for a, b in zip(ops[:-1], ops[1:]):
            _same_domain(a.domain, b.target)
        ops = ChainOperator.simplify(ops)
        if len(ops) == 1:
            return ops[0]
        return ChainOperator(ops, _callingfrommake=True)

    def _flip_modes(self, trafo):
        if trafo == 0:
            return self
        if trafo == self.ADJOINT_BIT or trafo == self.INVERSE_BIT:
            return ChainOperator.make([op._flip_modes(trafo) for op in reversed(self._ops)])
        return ChainOperator.make([op._flip_modes(trafo) for op in self._ops])

    def apply(self, x, mode):
        self._check_mode(mode)
        seq = self._ops if (mode & self._backwards) else reversed(self._ops)
        for op in seq:
            x = op.apply(x, mode)
        return x

    def draw_sample(self, from_inverse=False, device_id=-1):
        raise NotImplementedError

    def __repr__(self):
        return "ChainOperator:\n" + "\n".join("  " + repr(o).replace("\n", "\n  ") for o in self._ops)


class SumOperator(LinearOperator):
    """Sum / difference of linear operators (reference sum_operator.py:26-225)."""

    def __init__(self, ops, neg, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError
        self._ops, self._neg = ops, neg
        self._capability = self.TIMES | self.ADJOINT_TIMES
        for op in ops:
            self._capability &= op.capability
        self._domain = domain_union([op.domain for op in ops])
        self._target = domain_union([op.target for op in ops])

    @staticmethod
    def make(ops, neg=None):
        ops = tuple(ops)
        neg = (False,) * len(ops) if neg is None else tuple(bool(n) for n in neg)
        if len(ops) == 0 or len(ops) != len(neg):
            raise ValueError("length mismatch")
        flat, fneg = [], []
        for op, n in zip(ops, neg):
            if isinstance(op, SumOperator):
                flat.extend(op._ops)
                fneg.extend(nn != n for nn in op._neg)
            else:
                flat.append(op)
                fneg.append(n)
        # merge scalar multiples of the identity on identical domains
        scal, rest, rneg, sdt = None, [], [], None
        for op, n in zip(flat, fneg):
            if isinstance(op, ScalingOperator) and isinstance(op.domain, (DomainTuple, MultiDomain)) and \
                    (scal is None or scal[0] is op.domain):
                val = -op._factor if n else op._factor
                scal = (op.domain, val if scal is None else scal[1] + val)
                sdt = op._dtype if op._dtype is not None else sdt
            else:
                rest.append(op)
                rneg.append(n)
        if scal is not None:
            # at the END like the reference (sum_operator.py:104-107): the order fixes the RNG sequence of draw_sample
            rest.append(ScalingOperator(scal[0], scal[1], sdt))
            rneg.append(False)
        if len(rest) == 1:
            return rest[0] if not rneg[0] else rest[0].scale(-1)
        return SumOperator(tuple(rest), tuple(rneg), _callingfrommake=True)

    def _flip_modes(self, trafo):
        if trafo & self.INVERSE_BIT:
            return OperatorAdapter(self, trafo)
        return SumOperator.make([op._flip_modes(trafo) for op in self._ops], self._neg)

    def apply(self, x, mode):
        self._check_mode(mode)
        res = None
        for op, n in zip(self._ops, self._neg):
            tmp = op.apply(x.extract(op._dom(mode)), mode)
            if res is None:
                res = -tmp if n else tmp
            else:
                res = res.flexible_addsub(tmp, n)
        return res

    def draw_sample(self, from_inverse=False, device_id=-1):
        if from_inverse:
            raise NotImplementedError("cannot draw from inverse of this operator")
        res = self._ops[0].draw_sample(from_inverse, device_id)
        for op in self._ops[1:]:
            res = res.flexible_addsub(op.draw_sample(from_inverse, device_id), False)
        return res

    def __repr__(self):
        return "SumOperator:\n" + "\n".join("  " + repr(o).replace("\n", "\n  ") for o in self._ops)


class SandwichOperator(EndomorphicOperator):
    """bun^dagger cheese bun (reference sandwich_operator.py:27-110)."""

    def __init__(self, bun, cheese, op, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError
        self._bun, self._cheese, self._op = bun, cheese, op
        self._domain, self._capability = op.domain, op._capability

    @staticmethod
    def make(bun, cheese=None, sampling_dtype=None):
        if isinstance(cheese, SandwichOperator):
            bun = cheese._bun @ bun
            cheese = cheese._cheese
        if not isinstance(bun, LinearOperator):
            raise TypeError("bun must be a linear operator")
        if cheese is not None and not isinstance(cheese, LinearOperator):
            raise TypeError("cheese must be a linear operator or None")
        if cheese is None:
            cheese = ScalingOperator(bun.target, 1.0, sampling_dtype)
        if isinstance(bun, ScalingOperator):
            fct = abs(bun._factor) ** 2
            if fct == 1.0:
                return cheese
            op = cheese.scale(fct)
        else:
            op = bun.adjoint @ cheese @ bun
        return SandwichOperator(bun, cheese, op, _callingfrommake=True)

    def apply(self, x, mode):
        return self._op.apply(x, mode)

    def draw_sample(self, from_inverse=False, device_id=-1):
        if from_inverse:
            if self._bun.capability & self._bun.INVERSE_TIMES:
                try:
                    return self._bun.inverse_times(self._cheese.draw_sample(from_inverse, device_id))
                except NotImplementedError:
                    pass
            raise NotImplementedError("cannot draw from inverse of this operator")
        return self._bun.adjoint_times(self._cheese.draw_sample(from_inverse, device_id))

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
        return self._field * float(np.real(x.asnumpy()[()]))


class Realizer(EndomorphicOperator):
    def __init__(self, domain):
        self._domain = makeDomain(domain)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        return x.real


class FieldAdapter(LinearOperator):
    """Extract one key of a MultiField (reference simple_linear_operators.py:152-214)."""

    def __init__(self, tgt, name):
        self._name = name
        self._target = DomainTuple.make(tgt)
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


def Variable(domain, key):
    return ducktape(makeDomain(domain), None, key)


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
            raise NotImplementedError("weighted contractions")
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
class _RGTransformBase(LinearOperator):
    def __init__(self, domain, target=None, space=None):
        self._domain = DomainTuple.make(domain)
        if space is None:
            if len(self._domain) != 1:
                raise ValueError("need a space index for DomainTuples with more than one entry")
            space = 0
        self._space = int(space)
        adom = self._domain[self._space]
        if not isinstance(adom, RGSpace):
            raise TypeError(f"{type(self).__name__} only works on RGSpaces")
        if target is None:
            target = adom.get_default_codomain()
        tgt = list(self._domain)
        tgt[self._space] = target
        self._target = DomainTuple.make(tgt)
        adom.check_codomain(target)
        target.check_codomain(adom)

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
            f = torch.fft.fftn(v, dim=axes)
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
            res = torch.fft.ifftn(a, dim=axes) if inverse else torch.fft.fftn(a, dim=axes)
            return res if fct == 1 else res * fct

        # ifftn carries 1/N: N * ifftn = unnormalised backward transform
        dev = lambda a, nd: B.fftn(a, ndim=nd, inverse=inverse, scale=fct / ncells if inverse else fct)  # noqa: E731
        return Field(self._tgt(mode), self._over_subspace(v, host, dev))


class HarmonicTransformOperator(LinearOperator):
    """Harmonic space -> position space, real to real (reference harmonic_operators.py:283-337)."""

    def __init__(self, domain, target=None, space=None):
        domain = DomainTuple.make(domain)
        if space is None and len(domain) == 1:
            space = 0
        hspc = domain[space]
        if not hspc.harmonic:
            raise TypeError("HarmonicTransformOperator only works on a harmonic space")
        if not isinstance(hspc, RGSpace):
            raise NotImplementedError("spherical harmonic transforms are out of scope")
        self._op = HartleyOperator(domain, target, space)
        self._domain, self._target = self._op.domain, self._op.target
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        return self._op.apply(x, mode)


def HarmonicSmoothingOperator(domain, sigma, space=None):
    """Smoothing with a Gaussian kernel of width `sigma` (position-space units) on a non-harmonic RGSpace:
    Hartley^-1 . diag(exp(-2 pi^2 sigma^2 k^2)) . Hartley (reference harmonic_operators.py:340-380)."""
    sigma = float(sigma)
    if sigma < 0.0:
        raise ValueError("sigma must be non-negative")
    if sigma == 0.0:
        return ScalingOperator(domain, 1.0)
    domain = DomainTuple.make(domain)
    if space is None:
        if len(domain) != 1:
            raise ValueError("need to specify space")
        space = 0
    space = int(space)
    if space < 0 or space >= len(domain):
        raise ValueError("invalid space index")
    if domain[space].harmonic:
        raise TypeError("domain must not be harmonic")
    hartley = HartleyOperator(domain, space=space)
    codomain = hartley.target[space]
    kernel = codomain.get_fft_smoothing_kernel_function(sigma)(codomain.get_k_length_array())
    # the kernel lives on the transformed sub-space only and is broadcast over the others (diagonal_operator.py:51-120)
    return hartley.inverse(DiagonalOperator(kernel, hartley.target, space)(hartley))


class _JacCountingOperator(EndomorphicOperator):
    def __init__(self, domain):
        self._domain = makeDomain(domain)
        self._capability = self.TIMES | self.ADJOINT_TIMES
        self._count_times = 0
        self._count_adjoint_times = 0

    def apply(self, x, mode):
        self._check_input(x, mode)
        if mode == self.TIMES:
            self._count_times += 1
        else:
            self._count_adjoint_times += 1
        return x


class CountingOperator(Operator):
    """Identity that counts how often it is applied to fields / Linearizations and how often its Jacobian and
    adjoint Jacobian run (reference counting_operator.py:20-84; optimize_kl.py:370,441 reports them)."""

    def __init__(self, domain):
        self._domain = self._target = makeDomain(domain)
        self._count_apply = 0
        self._count_apply_lin = 0
        self._derivative = _JacCountingOperator(self._domain)

    def apply(self, x):
        self._check_input(x)
        if is_linearization(x):
            self._count_apply_lin += 1
            return x.new(x.val, self._derivative)
        self._count_apply += 1
        return x

    count_apply = property(lambda self: self._count_apply)
    count_apply_lin = property(lambda self: self._count_apply_lin)
    count_jac = property(lambda self: self._derivative._count_times)
    count_jac_adj = property(lambda self: self._derivative._count_adjoint_times)

    def __repr__(self):
        return f"CountingOperator({self._domain!r})"

    def report(self):
        return "\n".join([f"* apply: \t\t{self.count_apply:>7}", f"* apply Linearization: \t{self.count_apply_lin:>7}",
                          f"* Jacobian: \t\t{self.count_jac:>7}", f"* Adjoint Jacobian: \t{self.count_jac_adj:>7}"])


class DOFDistributor(LinearOperator):
    """Gather bins -> pixels (TIMES) and scatter-add pixels -> bins (ADJOINT_TIMES)
    (reference distributors.py:28-127)."""

    def __init__(self, dofdex, target=None, space=None):
        if target is None:
            target = dofdex.domain
        self._target = DomainTuple.make(target)
        if space is None and len(self._target) == 1:
            space = 0
        self._space = int(space)
        if self._space != len(self._target) - 1:
            raise NotImplementedError("DOFDistributor: only the LAST space of the target can be distributed "
                                      "(leading spaces are carried along)")
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
        # leading spaces of the target (e.g. the UnstructuredDomain(total_N) of several correlated fields) are batch axes
        lead = tuple(self._target[i] for i in range(len(self._target) - 1)) if hasattr(self, "_target") else ()
        self._domain = DomainTuple.make(lead + (other_space,))
        self._nlead = int(np.prod([d.size for d in lead])) if lead else 0
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
        if self._nlead:  # one gather / scatter per leading index (slices and stacking are copies)
            tshape = tuple(self._target.shape[len(self._target.shape) - len(self._target[self._space].shape):])
            rows = x.val.reshape((self._nlead, -1))
            outs = [self._apply1(rows[i], mode, tshape) for i in range(self._nlead)]
            return Field(self._tgt(mode), torch.stack(outs).reshape(self._tgt(mode).shape))
        tshape = self._target.shape
        return Field(self._tgt(mode), self._apply1(x.val, mode, tshape).reshape(self._tgt(mode).shape))

    def _apply1(self, v, mode, tshape):
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
        if space is None and len(self._target) == 1:
            space = 0
        self._space = int(space)
        if self._space != len(self._target) - 1:
            raise NotImplementedError("PowerDistributor: only the LAST space of the target can be distributed")
        hspace = self._target[self._space]
        if not hspace.harmonic:
            raise ValueError("Operator requires harmonic target space")
        if power_space is None:
            power_space = PowerSpace(hspace)
        else:
            if not isinstance(power_space, PowerSpace):
                raise TypeError("power_space argument must be a PowerSpace")
            if power_space.harmonic_partner != hspace:
                raise ValueError("power_space does not match its partner")
        self._pspace = power_space
        self._init2(lambda: torch.from_numpy(np.array(power_space.pindex)), power_space.shape[0], power_space)


class MaskOperator(LinearOperator):
    """Keeps the un-flagged pixels of a field in an UnstructuredDomain (reference mask_operator.py:26-58): flags are
    converted to boolean, True = flagged.  TIMES compresses, ADJOINT_TIMES expands with zeros."""

    def __init__(self, flags):
        if not isinstance(flags, Field):
            raise TypeError
        self._domain = DomainTuple.make(flags.domain)
        keep = torch.logical_not(flags.val.to(torch.bool)).reshape(-1)
        self._keep = torch.nonzero(keep.cpu()).reshape(-1)  # int64 flat indices of the kept pixels, ascending
        self._target = DomainTuple.make(UnstructuredDomain(int(self._keep.numel())))
        self._capability = self.TIMES | self.ADJOINT_TIMES
        self._idx32 = {}

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
    """(likelihood + prior) with sampling from its inverse via CG (reference sampling_enabler.py:27-97)."""

    def __init__(self, likelihood, prior, iteration_controller, approximation=None, start_from_zero=False):
        if not is_operator(likelihood) or not is_operator(prior):
            raise TypeError
        self._likelihood, self._prior, self._ic = likelihood, prior, iteration_controller
        self._approximation = approximation
        self._start_from_zero = bool(start_from_zero)
        self._op = likelihood + prior
        self._domain, self._capability = self._op.domain, self._op.capability

    def apply(self, x, mode):
        return self._op.apply(x, mode)

    def special_draw_sample(self, from_inverse=False, device_id=-1):
        from .minimization import ConjugateGradient, QuadraticEnergy

        try:
            res = self._op.draw_sample(from_inverse, device_id)
            return self._op(res), res
        except NotImplementedError:
            if not from_inverse:
                raise ValueError("from_inverse must be True here")
            if self._start_from_zero:
                b = self._op.draw_sample(device_id=device_id)
                energy = QuadraticEnergy(b * 0.0, self._op, b)
            else:
                s = self._prior.draw_sample(from_inverse=True, device_id=device_id)
                nj = self._likelihood.draw_sample(device_id=device_id)
                b = self._prior(s) + nj
                energy = QuadraticEnergy(s, self._op, b, _grad=self._likelihood(s) - nj)
            inverter = ConjugateGradient(self._ic)
            if self._approximation is not None:
                energy, _ = inverter(energy, preconditioner=self._approximation.inverse)
            else:
                energy, _ = inverter(energy)
            return b, energy.position

    def draw_sample(self, from_inverse=False, device_id=-1):
        return self.special_draw_sample(from_inverse, device_id)[1]

    def __repr__(self):
        return "SamplingEnabler:\n  Likelihood:\n    " + repr(self._likelihood) + "\n  Prior:\n    " + repr(self._prior)
