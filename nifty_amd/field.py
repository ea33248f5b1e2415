"""Field / MultiField: immutable value objects living on a DomainTuple / MultiDomain.

Counterpart of reference nifty/cl/field.py, multi_field.py and the array seam any_array.py.  The
value is a ``torch.Tensor``.  ``device_id == -1`` means host memory (the reference's numpy
convention, any_array.py:97-101) and uses plain host arithmetic; ``device_id >= 0`` means a GPU and
EVERY operation goes through the hand-written HIP kernels of libniftyk (``backend``) -- a device
Field never silently computes on the host or through eager PyTorch math.
"""
import numpy as np
import torch

from . import backend as B
from . import _lib as L
from . import parallel, random
from .domains import DomainTuple, MultiDomain, makeDomain

_NP2T = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
         np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128,
         np.dtype(np.int64): torch.int64, np.dtype(np.int32): torch.int32, np.dtype(bool): torch.bool}
_T2NP = {v: k for k, v in _NP2T.items()}


def torch_dtype(dt):
    if isinstance(dt, torch.dtype):
        return dt
    return _NP2T[np.dtype(dt)]


def numpy_dtype(dt):
    if isinstance(dt, torch.dtype):
        return _T2NP[dt]
    return np.dtype(dt)


def device_of(device_id):
    return torch.device("cpu") if device_id is None or device_id < 0 else torch.device("cuda", device_id)


def device_available():
    return torch.cuda.is_available()


def _as_tensor(val, device=None):
    if isinstance(val, torch.Tensor):
        return val if device is None else val.to(device)
    arr = np.asarray(val)
    if arr.dtype == np.float16:
        arr = arr.astype(np.float32)
    if arr.ndim and not arr.flags.writeable:
        arr = arr.copy()  # (read-only tables such as PowerSpace.pindex: a tensor must not alias memory numpy protects)
    t = torch.from_numpy(np.ascontiguousarray(arr)) if arr.ndim else torch.tensor(arr.item(), dtype=torch_dtype(arr.dtype))
    return t if device is None else t.to(device)


_BIN = {"add": L.OP_ADD, "sub": L.OP_SUB, "mul": L.OP_MUL, "div": L.OP_DIV}
_HOST_BIN = {"add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.true_divide}


class _AnyArrayMeta(type):
    def __instancecheck__(cls, obj):
        return isinstance(obj, torch.Tensor)


class AnyArray(metaclass=_AnyArrayMeta):
    """The array type behind `Field.val`.  The reference wraps numpy / cupy arrays in its own `AnyArray` (any_array.py); here
    the values ARE torch tensors (host or GPU), so `AnyArray(x)` converts `x` to one and `isinstance(field.val, AnyArray)` holds.
    The numpy-function protocol of the reference's class is not provided: `np.mean(field.val)` must read `field.asnumpy()`."""

    def __new__(cls, arr):
        return _as_tensor(arr.val if isinstance(arr, Field) else arr)


def _binary(op, a, b):
    """a (op) b where a, b are tensors of equal shape or python scalars (at least one tensor)."""
    ref = a if torch.is_tensor(a) else b
    if not ref.is_cuda:
        return _HOST_BIN[op](a, b)
    involved = [t for t in (a, b) if torch.is_tensor(t)]
    if any(t.is_complex() for t in involved) or any(isinstance(x, complex) for x in (a, b) if not torch.is_tensor(x)):
        return _complex_binary(op, a, b, involved)
    if torch.is_tensor(a) and torch.is_tensor(b) and a.dtype != b.dtype:
        dt = torch.promote_types(a.dtype, b.dtype)
        a, b = _cast(a, dt), _cast(b, dt)
    return B.binary(_BIN[op], a, b)


def _complex_binary(op, a, b, involved):
    """Device arithmetic with a complex operand: sums and real multiples run on the interleaved real view (nk_binary),
    products and quotients in the complex kernel (nk_cplx_muldiv)."""
    both = len(involved) == 2
    if op in ("add", "sub"):
        cdt = torch.complex128 if any(t.dtype in (torch.float64, torch.complex128) for t in involved) else torch.complex64
        if both:
            ca, cb = (t.to(cdt) for t in (a, b))  # (a real operand is promoted: a copy, no arithmetic)
            return torch.view_as_complex(B.binary(_BIN[op], torch.view_as_real(ca.contiguous()), torch.view_as_real(cb.contiguous())))
        field, scalar = (a, b) if torch.is_tensor(a) else (b, a)
        shift = torch.full(field.shape, complex(scalar), dtype=cdt, device=field.device)  # a constant field
        return _complex_binary(op, a if torch.is_tensor(a) else shift, b if torch.is_tensor(b) else shift, [field, shift])
    if not both and torch.is_tensor(a) and a.is_complex() and not isinstance(b, complex):  # complex field (*|/) real number
        return torch.view_as_complex(B.binary(_BIN[op], torch.view_as_real(a.contiguous()), float(b)))
    return B.cplx_muldiv(a, b, divide=(op == "div"))


def _cast(t, dt):
    if t.dtype == dt:
        return t
    return t.to(dt)  # dtype conversion = data movement (plumbing), not arithmetic


class Field:
    """Immutable field on a DomainTuple (reference field.py:33-803)."""

    __slots__ = ("_domain", "_val")

    def __init__(self, domain, val):
        if not isinstance(domain, DomainTuple):
            raise TypeError("domain must be of type DomainTuple")
        tensor = val if isinstance(val, torch.Tensor) else _as_tensor(val)
        if domain.shape != tuple(tensor.shape):
            raise ValueError(f"shape mismatch: {tuple(tensor.shape)} vs {domain.shape}")
        self._domain, self._val = domain, tensor

    # ---- constructors ---------------------------------------------------------------------
    @staticmethod
    def from_raw(domain, arr, device_id=None):
        domain = DomainTuple.make(domain)
        t = _as_tensor(arr, None if device_id is None else device_of(device_id))
        return Field(domain, t.reshape(domain.shape))

    @staticmethod
    def scalar(val, device_id=-1):
        return Field(DomainTuple.scalar_domain(), _as_tensor(np.asarray(val))).at(device_id)

    @staticmethod
    def full(domain, val, device_id=-1, dtype=None):
        domain = DomainTuple.make(domain)
        if not np.isscalar(val):
            raise TypeError("val must be a scalar")
        if dtype is None:
            dtype = np.complex128 if isinstance(val, complex) else np.float64 if not isinstance(val, (bool, np.bool_)) else bool
            if isinstance(val, (int, np.integer)) and not isinstance(val, (bool, np.bool_)):
                dtype = np.int64
        return Field(domain, torch.full(domain.shape, val, dtype=torch_dtype(dtype), device=device_of(device_id)))

    @staticmethod
    def from_random(domain, random_type="normal", dtype=np.float64, device_id=-1, **kwargs):
        """numpy PCG64 draw (parity with the reference's seeds): on the host -- or, for a normal / uniform / pm1 draw of a
        device field, the same stream computed on the GPU (random.Random.on_device).  field.py:128-156"""
        domain = DomainTuple.make(domain)
        if device_id >= 0:
            return Field(domain, random.Random.on_device(random_type, dtype, domain.shape, device_of(device_id), **kwargs))
        gen = getattr(random.Random, random_type)
        arr = gen(dtype=dtype, shape=domain.shape, **kwargs)
        return Field(domain, _as_tensor(arr, device_of(device_id)))

    # ---- properties -----------------------------------------------------------------------
    @property
    def domain(self):
        return self._domain

    @property
    def val(self):
        return self._val

    raw = val

    @property
    def dtype(self):
        return numpy_dtype(self._val.dtype)

    @property
    def shape(self):
        return self._domain.shape

    @property
    def size(self):
        return self._domain.size

    @property
    def device_id(self):
        return self._val.device.index if self._val.is_cuda else -1

    def at(self, device_id, *, check_fail=True):
        """This field on device `device_id` (-1: the host); `check_fail` is accepted for the reference's signature (field.py:185)"""
        if device_id == self.device_id:
            return self
        return Field(self._domain, self._val.to(device_of(device_id)))

    def asnumpy(self):
        return self._val.detach().cpu().numpy()

    def asnumpy_rw(self):
        return self.asnumpy().copy()

    def val_rw(self):
        return self._val.clone()

    @property
    def real(self):
        if not self._val.is_complex():
            return self
        return Field(self._domain, torch.view_as_real(self._val)[..., 0].contiguous())

    @property
    def imag(self):
        if not self._val.is_complex():
            raise ValueError(".imag called on a non-complex Field")
        return Field(self._domain, torch.view_as_real(self._val)[..., 1].contiguous())

    def conjugate(self):
        if not self._val.is_complex():
            return self
        return Field(self._domain, torch.conj_physical(self._val) if not self._val.is_cuda else
                     B.cplx_pointwise("conjugate", self._val))

    def astype(self, dtype):
        return Field(self._domain, _cast(self._val, torch_dtype(dtype)))

    # ---- reductions -----------------------------------------------------------------------
    def s_vdot(self, x):
        """conj(self).x as a host scalar (reference field.py:374-393)."""
        if not isinstance(x, Field):
            raise TypeError("The dot-partner must be an instance of the Field class")
        if x._domain is not self._domain:
            raise ValueError("domain mismatch")
        a, b = self._val, x._val
        if not a.is_cuda:
            if a.dtype != b.dtype:
                dt = torch.promote_types(a.dtype, b.dtype)
                a, b = a.to(dt), b.to(dt)
            r = torch.vdot(a.reshape(-1), b.reshape(-1))
            return complex(r) if r.is_complex() else float(r)
        if a.is_complex() or b.is_complex():
            # conj(a).b from four real dots on de-interleaved copies (complex device data only occurs around FFTOperator)
            def parts(t):
                if not t.is_complex():
                    return _cast(t, torch.float64).contiguous(), None
                v = torch.view_as_real(t)
                return v[..., 0].contiguous(), v[..., 1].contiguous()

            (ar, ai), (br, bi) = parts(a), parts(b)
            dot = lambda u, v: 0.0 if u is None or v is None else float(B.vdot(_cast(u, torch.float64), _cast(v, torch.float64)).item())  # noqa: E731
            return complex(dot(ar, br) + dot(ai, bi), dot(ar, bi) - dot(ai, br))
        dt = torch.promote_types(a.dtype, b.dtype)
        if not dt.is_floating_point:
            dt = torch.float64
        # no communication here: energies take rank-local per-sample values with this (the minimisers synchronise the
        # scalars that steer them, minimization._ls)
        return float(B.vdot(_cast(a, dt).contiguous(), _cast(b, dt).contiguous()).item())

    def vdot(self, x, spaces=None):
        """conj(self).x over all sub-domains (a scalar Field) or over `spaces` only (field.py:343-372)"""
        if spaces is None or len(self._domain._chosen(spaces)) == len(self._domain):
            return Field.scalar(self.s_vdot(x)).at(self.device_id)
        if not isinstance(x, Field):
            raise TypeError("The dot-partner must be an instance of the Field class")
        if x._domain is not self._domain:
            raise ValueError("domain mismatch")
        return (self.conjugate() * x).sum(spaces)

    def cast_domain(self, new_domain):
        """The same values on another domain of the same shape (field.py:110-126)"""
        from .domains import DomainTuple

        return Field(DomainTuple.make(new_domain), self._val)

    def map(self, func):
        """`func` (a numpy function keeping the shape) applied to the values; device fields go through the host"""
        return Field.from_raw(self._domain, func(self.asnumpy())).at(self.device_id)

    def _compare(self, other, name):
        """element-wise comparison -> boolean Field (a predicate, not arithmetic: evaluated by torch where the data lives)"""
        if isinstance(other, Field):
            if other._domain is not self._domain:
                raise ValueError("domains are incompatible.")
            other = other._val
        elif not np.isscalar(other):
            return NotImplemented
        return Field(self._domain, getattr(torch, name)(self._val, other))

    def __eq__(self, o): return self._compare(o, "eq")  # element-wise like numpy's (field.py:783-800); `is` compares objects
    def __ne__(self, o): return self._compare(o, "ne")
    __hash__ = object.__hash__
    def __gt__(self, o): return self._compare(o, "gt")
    def __ge__(self, o): return self._compare(o, "ge")
    def __lt__(self, o): return self._compare(o, "lt")
    def __le__(self, o): return self._compare(o, "le")

    def norm(self, ord=2):
        if ord == 2:
            v = self.s_vdot(self)
            return float(np.sqrt(np.real(v)))
        arr = self._val if not self._val.is_cuda else self._val.cpu()
        return float(torch.linalg.vector_norm(arr.reshape(-1), ord=ord))

    def s_sum(self):
        if not self._val.is_cuda:
            r = self._val.sum()
            return complex(r) if r.is_complex() else float(r)
        return float(B.vsum(self._val.contiguous()).item())

    def sum(self, spaces=None):
        if spaces is None:
            return Field.scalar(self.s_sum()).at(self.device_id)
        from .operators import ContractionOperator

        return ContractionOperator(self._domain, spaces)(self)

    def weight(self, power=1, spaces=None):
        """Every pixel times (its volume)**power along the sub-domains `spaces` (all by default; reference field.py:285-322):
        uniform volumes collapse into one factor, a sub-domain with individual volumes (PowerSpace: rho * pdvol per bin)
        contributes a broadcast factor field."""
        ndom = len(self._domain)
        chosen = range(ndom) if spaces is None else ((int(spaces),) if np.isscalar(spaces) else tuple(int(s) for s in spaces))
        out, uniform = self, 1.0
        for ind in chosen:
            vol = self._domain[ind].dvol
            if np.isscalar(vol):
                uniform *= vol
                continue
            axes = self._domain.axes[ind]
            shape = [n if ax in axes else 1 for ax, n in enumerate(self._domain.shape)]
            factor = np.broadcast_to((np.asarray(vol, dtype=np.float64) ** power).reshape(shape), self._domain.shape)
            out = out * Field(self._domain, _as_tensor(np.ascontiguousarray(factor))).at(self.device_id).astype(
                np.float64 if not self._val.is_complex() else self.dtype)
        uniform = uniform ** power
        return out * uniform if uniform != 1.0 else (out if out is not self else Field(self._domain, self._val.clone()))

    # ---- volumes and statistics (reference field.py:257-283, 419-664).  Small diagnostics: whole-field scalars of device
    # fields go through the fixed-order device sums (s_sum / s_vdot); partial reductions through ContractionOperator.
    def scalar_weight(self, spaces=None):
        return self._domain.scalar_weight(spaces)

    def total_volume(self, spaces=None):
        return self._domain.total_volume(spaces)

    def integrate(self, spaces=None):
        uniform = self.scalar_weight(spaces)
        if uniform is not None:
            return self.sum(spaces) * uniform
        return self.weight(1, spaces=spaces).sum(spaces)

    def s_integrate(self):
        uniform = self.scalar_weight()
        return self.s_sum() * uniform if uniform is not None else self.weight(1).s_sum()

    def mean(self, spaces=None):
        """integrate(spaces) / total_volume(spaces): the plain mean for uniform pixels"""
        if self.scalar_weight(spaces) is not None:
            count = int(np.prod([self._domain[c].size for c in self._domain._chosen(spaces)], dtype=np.int64))
            return self.sum(spaces) * (1.0 / count)
        return self.integrate(spaces) * (1.0 / self.total_volume(spaces))

    def s_mean(self):
        return self.s_integrate() / self.total_volume()

    def _spread(self, spaces):
        """|self - mean|^2 with the mean over `spaces` broadcast back"""
        from .operators import ContractionOperator

        centre = self.mean(spaces)
        if len(self._domain._chosen(spaces)) != len(self._domain) or centre.domain is not self._domain:
            centre = ContractionOperator(self._domain, spaces).adjoint_times(centre)
        dev = self - centre
        return (dev.conjugate() * dev).real if self._val.is_complex() else dev * dev

    def var(self, spaces=None):
        return self._spread(spaces).mean(spaces)

    def s_var(self):
        return self._spread(None).s_mean()

    def std(self, spaces=None):
        return self.var(spaces).ptw("sqrt")

    def s_std(self):
        return float(np.sqrt(self.s_var()))

    def _host_contraction(self, name, spaces):
        """prod / all / any over sub-domains: diagnostics, evaluated by torch on whatever device holds the data"""
        chosen = self._domain._chosen(spaces)
        if len(chosen) == len(self._domain):
            return Field.scalar(getattr(self, "s_" + name)()).at(self.device_id)
        axes = tuple(a for c in chosen for a in self._domain.axes[c])
        out = self._val
        for ax in sorted(axes, reverse=True):
            out = getattr(torch, name)(out, dim=ax)
        from .domains import DomainTuple

        return Field(DomainTuple.make([d for i, d in enumerate(self._domain) if i not in chosen]), out)

    def prod(self, spaces=None):
        return self._host_contraction("prod", spaces)

    def s_prod(self):
        r = self._val.prod()
        return complex(r) if r.is_complex() else float(r)

    def all(self, spaces=None):
        return self._host_contraction("all", spaces)

    def any(self, spaces=None):
        return self._host_contraction("any", spaces)

    def s_all(self):
        return bool(self._val.all())

    def s_any(self):
        return bool(self._val.any())

    def outer(self, x):
        """The field self (x) x on the product of both domains (field.py:324-341)"""
        if not isinstance(x, Field):
            raise TypeError("The multiplier must be an instance of the Field class")
        from .domains import DomainTuple

        a = self._val.reshape(self._val.shape + (1,) * x._val.dim())
        return Field(DomainTuple.make(tuple(self._domain) + tuple(x._domain)), _binary("mul", a.expand(self._val.shape + x._val.shape).contiguous(), x._val.expand(self._val.shape + x._val.shape).contiguous()))

    def scale(self, factor):
        return self if factor == 1 else factor * self

    def ducktape(self, name):
        raise RuntimeError("ducktape works only on operators")

    def ducktape_left(self, name):
        """The MultiField {name: self}, or -- a domain instead of a string -- the same values on that domain
        (operator.py:364-373)"""
        if isinstance(name, str):
            return MultiField.from_dict({name: self})
        from .domains import DomainTuple

        new = DomainTuple.make(name)
        if new.size != self._domain.size:
            raise ValueError("Domain and target do not have the same number of pixels")
        return Field(new, self._val.reshape(new.shape))

    def transpose(self, indices):
        from .selection_operators import TransposeOperator

        return TransposeOperator(self._domain, indices)(self)

    def squeeze(self, aggressive=False):
        from .selection_operators import SqueezeOperator

        return SqueezeOperator(self._domain, aggressive)(self)

    def broadcast(self, index, space):
        """self repeated along a new sub-domain `space` at position `index` (field.py:434-441)"""
        from .operators import ContractionOperator

        tgt = list(self._domain)
        tgt.insert(index, space)
        return ContractionOperator(tgt, index).adjoint_times(self)

    def abs(self):
        return self.ptw("abs")

    def __bool__(self):
        raise TypeError("Field does not support implicit conversion to bool")

    # ---- arithmetic -----------------------------------------------------------------------
    def _bin(self, other, op, reverse=False):
        if isinstance(other, Field):
            if other._domain is not self._domain:
                raise ValueError("domains are incompatible.")
            a, b = (other._val, self._val) if reverse else (self._val, other._val)
            return Field(self._domain, _binary(op, a, b))
        if np.isscalar(other) or (isinstance(other, np.ndarray) and other.shape == ()):
            other = other.item() if isinstance(other, (np.ndarray, np.generic)) else other
            a, b = (other, self._val) if reverse else (self._val, other)
            return Field(self._domain, _binary(op, a, b))
        return NotImplemented

    def __add__(self, o): return self._bin(o, "add")
    def __radd__(self, o): return self._bin(o, "add", True)
    def __sub__(self, o): return self._bin(o, "sub")
    def __rsub__(self, o): return self._bin(o, "sub", True)
    def __mul__(self, o): return self._bin(o, "mul")
    def __rmul__(self, o): return self._bin(o, "mul", True)
    def __truediv__(self, o): return self._bin(o, "div")
    def __rtruediv__(self, o): return self._bin(o, "div", True)

    def __neg__(self):
        return self * (-1.0)

    def __pos__(self):
        return self

    def __abs__(self):
        return self.ptw("abs")

    def __pow__(self, p):
        if np.isscalar(p):
            return self.ptw("power", p)
        return (p * self.ptw("log")).ptw("exp") if isinstance(p, Field) else NotImplemented  # operator.py:288-293

    def __rpow__(self, base):
        return (self * float(np.log(base))).ptw("exp") if np.isscalar(base) else NotImplemented

    def _inplace(self, *a, **k):
        raise TypeError("In-place operations are deliberately not supported")

    __iadd__ = __isub__ = __imul__ = __itruediv__ = __ipow__ = _inplace

    # ---- pointwise nonlinearities (reference pointwise.py, any_array.py:472-532) ---------------
    def ptw(self, op, *args, **kwargs):
        return Field(self._domain, _ptw(self._val, op, False, *args, **kwargs))

    def ptw_with_deriv(self, op, *args, **kwargs):
        f, d = _ptw(self._val, op, True, *args, **kwargs)
        return Field(self._domain, f), Field(self._domain, d)

    def __repr__(self):
        return f"<nifty_amd.Field on {self._domain!r}, device {self.device_id}>"

    def extract(self, dom):
        if dom is not self._domain:
            raise ValueError("domain mismatch")
        return self

    def extract_part(self, dom):
        return self.extract(dom)

    def unite(self, other):
        return self + other

    def flexible_addsub(self, other, neg):
        return self - other if neg else self + other

    def clip(self, a_min=None, a_max=None):
        return self.ptw("clip", a_min, a_max)


def _host_ptw(x, op, deriv, *args):
    if not (x.is_floating_point() or x.is_complex()):
        x = x.to(torch.float64)  # numpy evaluates integer input in float64; torch would pick float32
    if op == "exp":
        f = torch.exp(x); return (f, f) if deriv else f
    if op == "log":
        return (torch.log(x), 1.0 / x) if deriv else torch.log(x)
    if op == "sqrt":
        f = torch.sqrt(x); return (f, 0.5 / f) if deriv else f
    if op == "tanh":
        f = torch.tanh(x); return (f, 1.0 - f * f) if deriv else f
    if op == "sigmoid":
        t = torch.tanh(x); f = 0.5 + 0.5 * t; return (f, 0.5 - 0.5 * t * t) if deriv else f
    if op == "reciprocal":
        f = 1.0 / x; return (f, -f * f) if deriv else f
    if op == "power":
        p = args[0]; f = torch.pow(x, p); return (f, p * torch.pow(x, p - 1)) if deriv else f
    if op in ("abs", "absolute"):
        f = torch.abs(x)
        if not deriv:
            return f
        if x.is_complex():
            raise TypeError("Argument must not be complex because abs(z) is not holomorphic")
        d = torch.sign(x); d = torch.where(x == 0, torch.full_like(d, float("nan")), d)
        return f, d
    if op == "log1p":
        return (torch.log1p(x), 1.0 / (1.0 + x)) if deriv else torch.log1p(x)
    if op == "expm1":
        f = torch.expm1(x); return (f, f + 1.0) if deriv else f
    if op == "clip":
        if x.is_complex():
            raise TypeError("Argument must not be complex")
        lo, hi = args
        f = torch.clamp(x, lo, hi)
        if not deriv:
            return f
        d = torch.ones_like(x)
        if lo is not None:
            d = torch.where(f == lo, torch.zeros_like(d), d)
        if hi is not None:
            d = torch.where(f == hi, torch.zeros_like(d), d)
        return f, d
    if op == "arctan":
        return (torch.arctan(x), 1.0 / (1.0 + x * x)) if deriv else torch.arctan(x)
    if op in ("sin", "cos"):
        f = getattr(torch, op)(x)
        if not deriv:
            return f
        return f, (torch.cos(x) if op == "sin" else -torch.sin(x))
    if op == "tan":
        f = torch.tan(x); return (f, 1.0 / torch.cos(x) ** 2) if deriv else f
    if op in ("sinh", "cosh"):
        f = getattr(torch, op)(x)
        return (f, torch.cosh(x) if op == "sinh" else torch.sinh(x)) if deriv else f
    if op == "log10":
        return (torch.log10(x), (1.0 / np.log(10.0)) / x) if deriv else torch.log10(x)
    if op == "sinc":  # sin(pi x) / (pi x); the derivative (cos(pi x) - sinc x) / x, 0 at the origin
        f = torch.sinc(x)
        if not deriv:
            return f
        safe = torch.where(x == 0, torch.ones_like(x), x)
        return f, torch.where(x == 0, torch.zeros_like(x), (torch.cos(np.pi * safe) - f) / safe)
    if op == "sign":
        if x.is_complex():
            raise TypeError("Argument must not be complex")
        f = torch.sign(x)
        return (f, torch.where(x == 0, torch.full_like(x, float("nan")), torch.zeros_like(x))) if deriv else f
    if op == "unitstep":
        if x.is_complex():
            raise TypeError("Argument must not be complex")
        f = (x >= 0).to(x.dtype)
        return (f, torch.zeros_like(x)) if deriv else f
    if op == "softplus":  # log(1 + e^x), linear above 33 and zero below -33 (pointwise.py:99-122); complex: by the real part
        re = x.real if x.is_complex() else x
        mid = torch.where((re > 33) | (re < -33), torch.zeros_like(x), x)
        f = torch.where(re > 33, x, torch.where(re < -33, torch.zeros_like(x), torch.log(1 + torch.exp(mid))))
        if not deriv:
            return f
        d = torch.where(re > 33, torch.ones_like(x), torch.where(re < -33, torch.zeros_like(x), 1 / (1 + torch.exp(-mid))))
        return f, d
    if op == "exponentiate":
        base = args[0]
        f = torch.pow(torch.as_tensor(base, dtype=x.dtype), x)
        return (f, np.log(base) * f) if deriv else f
    raise NotImplementedError(f"pointwise operation {op!r}")


def _ptw(x, op, deriv, *args, **kwargs):
    if not x.is_cuda:
        return _host_ptw(x, op, deriv, *args)
    if x.is_complex():
        if deriv or op not in B.CPLX_POINTWISE:
            raise NotImplementedError(f"complex pointwise operation {op!r} (with derivative: {deriv}) has no device kernel")
        return B.cplx_pointwise(op, x)
    if op not in B.POINTWISE:
        raise NotImplementedError(f"pointwise operation {op!r} has no device kernel yet")
    if op == "clip":
        return B.pointwise(op, x.contiguous(), args[0], want_derivative=deriv, param2=args[1])
    param = float(args[0]) if op in ("power", "exponentiate") else 0.0
    return B.pointwise(op, x.contiguous(), param, want_derivative=deriv)


def _pointwise_method(name):
    def method(self, *args, **kwargs):
        return self.ptw(name, *args, **kwargs)

    method.__name__ = name
    return method


for _name in ("exp", "log", "sqrt", "tanh", "sigmoid", "reciprocal", "log1p", "expm1", "sin", "cos", "absolute", "arctan", "tan",
              "sinh", "cosh", "log10", "sinc", "sign", "unitstep", "softplus", "power", "exponentiate"):
    setattr(Field, _name, _pointwise_method(_name))


class MultiField:
    """Dictionary key -> Field over a MultiDomain (reference multi_field.py)."""

    __slots__ = ("_domain", "_val", "_flat")

    # Element-wise arithmetic of DEVICE MultiFields runs on one packed buffer per operand when the whole MultiField is at most
    # this many numbers: a product-spectrum model has a dozen keys, eleven of them a few hundred numbers long, and an operation
    # per key is a kernel launch per key (the CG of the generic graph: ~70 launches per iteration for six vector updates; the
    # GPU idle 80 % of the time, tools/gpu_product_probe.py).  Beyond it the packing copy costs more than the launches.
    # The values of the result are views into its buffer; element-wise results are the same bits either way.
    PACK_MAX = 1 << 23
    PACK_MIN_KEYS = 3

    def __init__(self, domain, val, _flat=None):
        if not isinstance(domain, MultiDomain):
            raise TypeError("domain must be of type MultiDomain")
        if not (isinstance(val, tuple) and len(val) == len(domain)):
            raise ValueError("length mismatch")
        if not all(isinstance(f, Field) and f.domain is sub for f, sub in zip(val, domain.domains())):
            raise ValueError("domain mismatch")  # one Field per key, each on that key's (interned) DomainTuple
        self._domain, self._val, self._flat = domain, val, _flat

    def __reduce__(self):
        return (MultiField, (self._domain, self._val))  # (the packed buffer is a cache: never written to files)

    @staticmethod
    def _from_flat(domain, flat):
        """the MultiField whose values are consecutive views of `flat` (key order)"""
        vals, at = [], 0
        for sub in domain.domains():
            vals.append(Field(sub, flat[at:at + sub.size].reshape(sub.shape)))
            at += sub.size
        return MultiField(domain, tuple(vals), _flat=flat)

    def _packed(self):
        """all values as ONE contiguous device buffer (made once: MultiFields are immutable), or None: host data, mixed or
        complex dtypes, few keys, or more than PACK_MAX numbers"""
        if self._flat is None:
            vals = self._val
            if len(vals) < MultiField.PACK_MIN_KEYS or self._domain.size > MultiField.PACK_MAX:
                return None
            first = vals[0]._val
            if not (first.is_cuda and first.is_floating_point()) or any(v._val.dtype != first.dtype or not v._val.is_cuda
                                                                        for v in vals):
                return None
            self._flat = torch.cat([v._val.reshape(-1) for v in vals])
        return self._flat

    @staticmethod
    def from_dict(dct, domain=None):
        if domain is None:
            if not all(isinstance(f.domain, DomainTuple) for f in dct.values()):
                raise TypeError("Values of dictionary need to be Fields defined on DomainTuples.")
            domain = MultiDomain.make({k: f.domain for k, f in dct.items()})
        try:
            return MultiField(domain, tuple(dct[k] for k in domain.keys()))
        except KeyError:
            raise ValueError(f"missing keys: {[k for k in domain.keys() if k not in dct]}") from None

    @staticmethod
    def from_raw(domain, arr, device_id=None):
        domain = MultiDomain.make(domain)
        return MultiField(domain, tuple(Field.from_raw(domain[k], arr[k], device_id) for k in domain.keys()))

    @staticmethod
    def full(domain, val, device_id=-1):
        domain = MultiDomain.make(domain)
        return MultiField(domain, tuple(Field.full(d, val, device_id) for d in domain.domains()))

    @staticmethod
    def from_random(domain, random_type="normal", dtype=np.float64, device_id=-1, **kwargs):
        """One draw per key in ALPHABETICAL order (the parity-relevant RNG order, multi_field.py:109-153)."""
        domain = MultiDomain.make(domain)
        per_key = dtype if isinstance(dtype, dict) else dict.fromkeys(domain.keys(), dtype)
        draws = [Field.from_random(sub, random_type, per_key[key], device_id, **kwargs) for key, sub in domain.items()]
        return MultiField(domain, tuple(draws))

    def to_dict(self):
        return {k: v for k, v in zip(self._domain.keys(), self._val)}

    def keys(self):
        return self._domain.keys()

    def items(self):
        return zip(self._domain.keys(), self._val)

    def values(self):
        return self._val

    def __getitem__(self, key):
        return self._val[self._domain.idx[key]]

    def __contains__(self, key):
        return key in self._domain

    @property
    def domain(self):
        return self._domain

    @property
    def val(self):
        return {k: v.val for k, v in self.items()}

    @property
    def dtype(self):
        dts = {v.dtype for v in self._val}
        if len(dts) == 1:
            return dts.pop()
        return {k: v.dtype for k, v in self.items()}

    @property
    def size(self):
        return sum(v.size for v in self._val)

    @property
    def device_id(self):
        ids = {v.device_id for v in self._val}
        if len(ids) > 1:
            raise RuntimeError("MultiField spread over several devices")
        return ids.pop() if ids else -1

    def at(self, device_id):
        return MultiField(self._domain, tuple(v.at(device_id) for v in self._val))

    def asnumpy(self):
        return {k: v.asnumpy() for k, v in self.items()}

    @property
    def real(self):
        return MultiField(self._domain, tuple(v.real for v in self._val))

    @property
    def imag(self):
        return self._map(lambda v: v.imag)

    def conjugate(self):
        return MultiField(self._domain, tuple(v.conjugate() for v in self._val))

    def astype(self, dtype):
        """every component cast to `dtype`, or to dtype[key] for a dict (multi_field.py:98-113)"""
        per_key = dtype if isinstance(dtype, dict) else dict.fromkeys(self.keys(), dtype)
        return MultiField(self._domain, tuple(v.astype(per_key[k]) for k, v in self.items()))

    def clip(self, a_min=None, a_max=None):
        return self.ptw("clip", a_min, a_max)

    def scale(self, factor):
        return self if factor == 1 else self * factor

    def s_all(self):
        return all(v.s_all() for v in self._val)

    def s_any(self):
        return any(v.s_any() for v in self._val)

    def val_rw(self):
        return {k: v.val_rw() for k, v in self.items()}

    def asnumpy_rw(self):
        return {k: v.asnumpy_rw() for k, v in self.items()}

    def s_vdot(self, x):
        if x._domain is not self._domain:
            raise ValueError("domain mismatch")
        pairs = [(a._val, b._val) for a, b in zip(self._val, x._val)]
        plain = all(u.is_cuda and v.is_cuda and u.dtype == v.dtype and u.is_floating_point() for u, v in pairs)
        if plain and len(pairs) > 1:
            # one dot kernel per key into ONE device buffer, one transfer for all of them (a read-back per key costs a
            # synchronisation each: twelve per dot product of a product-spectrum model); added up in key order on the host
            # like the per-key results were: the same bits
            slots = torch.zeros(len(pairs), dtype=torch.float64, device=pairs[0][0].device)
            for k, (u, v) in enumerate(pairs):
                B.vdot(u.contiguous(), v.contiguous(), result=slots[k:k + 1])
            res = 0.0
            for part in slots.cpu().tolist():
                res = res + part
            return res
        res = 0.0
        for a, b in zip(self._val, x._val):
            res = res + a.s_vdot(b)
        return res

    def vdot(self, x):
        return Field.scalar(self.s_vdot(x)).at(self.device_id)

    def norm(self, ord=2):
        if ord == 2:
            return float(np.sqrt(np.real(self.s_vdot(self))))
        nrm = np.asarray([f.norm(ord) for f in self._val])
        return float(np.linalg.norm(nrm, ord=ord))

    def s_sum(self):
        return sum(v.s_sum() for v in self._val)

    def _map(self, fn):
        return MultiField(self._domain, tuple(fn(v) for v in self._val))

    def _bin(self, other, op, name=None, reverse=False):
        """op(value, other value) per key -- or, for packable device MultiFields and a named operation, ONE launch on the
        packed buffers (`reverse`: other (op) self)"""
        if isinstance(other, MultiField):
            if other._domain is not self._domain:
                raise ValueError("domain mismatch")
            if name is not None:
                mine = self._packed()
                theirs = other._packed() if mine is not None else None
                if theirs is not None and theirs.dtype == mine.dtype:
                    a, b = (theirs, mine) if reverse else (mine, theirs)
                    return MultiField._from_flat(self._domain, _binary(name, a, b))
            return MultiField(self._domain, tuple(op(a, b) for a, b in zip(self._val, other._val)))
        if np.isscalar(other):
            if name is not None and not isinstance(other, complex):
                mine = self._packed()
                if mine is not None:
                    number = other.item() if isinstance(other, np.generic) else other
                    a, b = (number, mine) if reverse else (mine, number)
                    return MultiField._from_flat(self._domain, _binary(name, a, b))
            return self._map(lambda v: op(v, other))
        return NotImplemented

    def __add__(self, o): return self._bin(o, lambda a, b: a + b, "add")
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a, "add", True)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b, "sub")
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a, "sub", True)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b, "mul")
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a, "mul", True)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b, "div")
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: b / a, "div", True)
    def __pow__(self, p): return self._map(lambda v: v ** p)
    def __neg__(self): return self._map(lambda v: -v)
    def __abs__(self): return self._map(abs)

    def ptw(self, op, *args, **kwargs):
        return self._map(lambda v: v.ptw(op, *args, **kwargs))

    def ptw_with_deriv(self, op, *args, **kwargs):
        pairs = [v.ptw_with_deriv(op, *args, **kwargs) for v in self._val]
        return (MultiField(self._domain, tuple(p[0] for p in pairs)),
                MultiField(self._domain, tuple(p[1] for p in pairs)))


    # ---- key plumbing (multi_field.py:281-376) --------------------------------------------------
    def extract(self, subset):
        if subset is self._domain:
            return self
        subset = MultiDomain.make(subset)
        return MultiField(subset, tuple(self[k] for k in subset.keys()))

    def extract_by_keys(self, keys):
        dom = MultiDomain.make({k: v for k, v in self._domain.items() if k in keys})
        return self.extract(dom)

    def extract_part(self, subset):
        keys = [k for k in subset.keys() if k in self._domain]
        return self.extract_by_keys(keys)

    def unite(self, other):
        return self.flexible_addsub(other, False)

    @staticmethod
    def union(fields, domain=None):
        res = {}
        for f in fields:
            res.update(f.to_dict())
        return MultiField.from_dict(res, domain)

    def flexible_addsub(self, other, neg):
        if self._domain is other._domain:
            return self - other if neg else self + other
        # different key sets: keys both know are combined, the others pass through (with the sign for `other`'s)
        mine = self.to_dict()
        theirs = {k: -f if neg else f for k, f in other.items()}
        merged = {k: mine[k] + theirs[k] if (k in mine and k in theirs) else mine.get(k, theirs.get(k))
                  for k in {**mine, **theirs}}
        return MultiField.from_dict(merged)

    def __repr__(self):
        return "<nifty_amd.MultiField keys=" + ", ".join(self.keys()) + ">"


def is_fieldlike(obj):
    return isinstance(obj, (Field, MultiField))


def full(domain, val, device_id=-1):
    domain = makeDomain(domain)
    return (MultiField if isinstance(domain, MultiDomain) else Field).full(domain, val, device_id)


def from_random(domain, random_type="normal", dtype=np.float64, device_id=-1, **kwargs):
    domain = makeDomain(domain)
    cls = MultiField if isinstance(domain, MultiDomain) else Field
    return cls.from_random(domain, random_type, dtype, device_id, **kwargs)


def makeField(domain, arr, device_id=None):
    domain = makeDomain(domain)
    cls = MultiField if isinstance(domain, MultiDomain) else Field
    return cls.from_raw(domain, arr, device_id)


for _name in ("exp", "log", "sqrt", "tanh", "sigmoid", "reciprocal", "log1p", "expm1", "sin", "cos", "absolute", "arctan", "tan",
              "sinh", "cosh", "log10", "sinc", "sign", "unitstep", "softplus", "power", "exponentiate", "abs"):
    if _name not in MultiField.__dict__:
        setattr(MultiField, _name, _pointwise_method(_name))
