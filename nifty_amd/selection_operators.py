"""Linear operators that only move, select or relabel values -- no arithmetic beyond adding values that land on the same pixel
(reference operators/simple_linear_operators.py, selection_operators.py, transpose_operator.py, outer_product_operator.py,
value_inserter.py, field_zero_padder.py, domain_tuple_field_inserter.py).  They sit either side of the hot path in user models
(padding a field before a convolution, picking measured pixels, stacking slices); they are tensor views and copies on whatever
device holds the field, there is nothing here for a hand-written kernel to win."""
import numpy as np
import torch

from .domains import DomainTuple, MultiDomain, RGSpace, UnstructuredDomain, makeDomain
from .field import Field, MultiField
from .operators import EndomorphicOperator, LinearOperator, _same_domain, _space_index


def _first_axis(domain, space):
    return sum(len(sub.shape) for sub in domain[:space])


def _embedded(domain, where, values):
    """A Field of zeros on `domain` with `values` written at index / slice tuple `where`"""
    out = torch.zeros(domain.shape, dtype=values.dtype, device=values.device)
    out[where] = values
    return Field(domain, out)


def _selected(domain, where, values):
    """The Field on `domain` holding a copy of values[where]"""
    return Field(domain, values[where].clone())


class ConjugationOperator(EndomorphicOperator):
    """x -> conj(x): its own adjoint and inverse as a real-linear map (simple_linear_operators.py:59-74)"""

    def __init__(self, domain):
        self._domain = DomainTuple.make(domain)
        self._capability = self._all_ops

    def apply(self, x, mode):
        self._check_input(x, mode)
        return x.conjugate()


class Imaginizer(EndomorphicOperator):
    """x -> Im(x); the adjoint places a real field on the imaginary axis (simple_linear_operators.py:126-149)"""

    def __init__(self, domain):
        self._domain = makeDomain(domain)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        dtypes = x.dtype.values() if isinstance(x.dtype, dict) else [x.dtype]
        if mode == self.TIMES:
            if not all(np.issubdtype(dt, np.complexfloating) for dt in dtypes):
                raise ValueError("Imaginizer needs complex input")
            return x.imag
        if not all(dt in (np.float64, np.float32) for dt in dtypes):
            raise ValueError("the adjoint of Imaginizer needs real input")
        return 1j * x


class GeometryRemover(LinearOperator):
    """Relabels sub-domain `space` (default: all) as an UnstructuredDomain of the same shape; no volume factors
    (simple_linear_operators.py:316-347)"""

    def __init__(self, domain, space=None):
        self._domain = DomainTuple.make(domain)
        chosen = range(len(self._domain)) if space is None else (int(space),)
        self._target = DomainTuple.make([UnstructuredDomain(sub.shape) if i in chosen else sub
                                         for i, sub in enumerate(self._domain)])
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        return Field(self._tgt(mode), x.val)


class DomainChangerAndReshaper(LinearOperator):
    """Same values, another DomainTuple with the same number of pixels (simple_linear_operators.py:474-512)"""

    def __init__(self, domain, target):
        self._domain, self._target = makeDomain(domain), makeDomain(target)
        if isinstance(self._domain, MultiDomain) or isinstance(self._target, MultiDomain):
            raise NotImplementedError("MultiDomains are not supported yet")
        if self._domain.size != self._target.size:
            raise ValueError("Domain and target do not have the same number of pixels\n"
                             f"Domain: {self._domain.shape}\nTarget: {self._target.shape}")
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        tgt = self._tgt(mode)
        return Field(tgt, x.val.reshape(tgt.shape))

    def __repr__(self):
        shapes = lambda dom: " ".join(str(sub.shape) for sub in dom)  # noqa: E731
        return f"Reshape {shapes(self._target)} <- {shapes(self._domain)}"


class PartialExtractor(LinearOperator):
    """The keys of `target` out of a MultiField on `domain`; the adjoint fills the other keys with zeros
    (simple_linear_operators.py:420-444)"""

    def __init__(self, domain, target):
        if not isinstance(domain, MultiDomain) or not isinstance(target, MultiDomain):
            raise TypeError("MultiDomain expected")
        self._domain, self._target = domain, target
        for key in target.keys():
            _same_domain(domain[key], target[key])
        self._rest = MultiDomain.make({k: domain[k] for k in domain.keys() if k not in target.keys()})
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        if mode == self.TIMES:
            return x.extract(self._target)
        return x.unite(MultiField.full(self._rest, 0.0, device_id=x.device_id))

    def __repr__(self):
        return f"{self._target.keys()} <- {self._domain.keys()}"


class ExtractAtIndices(LinearOperator):
    """The values at a list of pixels of sub-domain `space` (an index may occur several times), as an UnstructuredDomain
    there; the adjoint adds values back onto their pixels (simple_linear_operators.py:515-573)"""

    def __init__(self, domain, indices, space=0):
        self._domain = makeDomain(domain)
        if not isinstance(indices, tuple):
            raise TypeError("indices need to be a tuple")
        if len(self._domain[space].shape) != len(indices):
            raise ValueError("Shape of indices don't match dimension of space")
        self._target = makeDomain([UnstructuredDomain(len(indices[0])) if i == space else sub
                                   for i, sub in enumerate(self._domain)])
        grid = self._domain[space].shape
        nlead = _first_axis(self._domain, space)
        self._lead = int(np.prod(self._domain.shape[:nlead], dtype=np.int64))
        self._cells = int(np.prod(grid, dtype=np.int64))
        self._trail = int(np.prod(self._domain.shape[nlead + len(grid):], dtype=np.int64))
        # one linear index per requested pixel of the sub-domain
        self._linear = torch.as_tensor(np.ravel_multi_index(tuple(np.asarray(ix, dtype=np.int64) for ix in indices), grid),
                                       dtype=torch.int64)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        v = x.val
        if mode == self.TIMES:
            rows = v.reshape(self._lead, self._cells, self._trail)
            return Field(self._target, rows[:, self._linear.to(v.device)].reshape(self._target.shape))
        # colliding indices: summed on the host in index order, so the result does not depend on the device's atomics
        rows = v.cpu().reshape(self._lead, len(self._linear), self._trail)
        out = torch.zeros((self._lead, self._cells, self._trail), dtype=v.dtype)
        out.index_add_(1, self._linear, rows)
        return Field(self._domain, out.reshape(self._domain.shape).to(v.device))


class SqueezeOperator(LinearOperator):
    """Drops sub-domains of shape (1,) -- `aggressive`: every length-1 axis of RGSpaces and UnstructuredDomains as well
    (simple_linear_operators.py:576-632)"""

    def __init__(self, domain, aggressive=False):
        self._domain = DomainTuple.make(domain)
        kept, dropped, axis = [], [], 0
        for sub in self._domain:
            ones = [axis + i for i, n in enumerate(sub.shape) if n == 1]
            if sub.shape == (1,):
                dropped += ones
            elif aggressive and isinstance(sub, (RGSpace, UnstructuredDomain)):
                dropped += ones
                keep = [i for i, n in enumerate(sub.shape) if n != 1]
                shape = [sub.shape[i] for i in keep]
                kept.append(RGSpace(shape, [sub.distances[i] for i in keep], sub.harmonic) if isinstance(sub, RGSpace)
                            else UnstructuredDomain(shape))
            else:
                kept.append(sub)
            axis += len(sub.shape)
        if not dropped:
            raise RuntimeError("Nothing found to be squeezed")
        self._target = DomainTuple.make(kept)
        self._capability = self._all_ops

    def apply(self, x, mode):
        self._check_input(x, mode)
        return Field(self._tgt(mode), x.val.reshape(self._tgt(mode).shape))


class TransposeOperator(LinearOperator):
    """Reorders the sub-domains: target[i] = domain[indices[i]] (transpose_operator.py:25-52)"""

    def __init__(self, domain, indices):
        self._domain = DomainTuple.make(domain)
        self._indices = tuple(int(i) for i in indices)
        if len(self._indices) != len(self._domain):
            raise IndexError("Either too many or too few indices given.")
        self._target = DomainTuple.make([self._domain[i] for i in self._indices])
        if self._domain.size != self._target.size:
            raise ValueError("List of indices not complete")
        self._forward = tuple(ax for i in self._indices for ax in self._domain.axes[i])
        self._backward = tuple(int(i) for i in np.argsort(self._forward))
        self._capability = self._all_ops

    def apply(self, x, mode):
        self._check_input(x, mode)
        order = self._forward if mode & (self.TIMES | self.ADJOINT_INVERSE_TIMES) else self._backward
        return Field(self._tgt(mode), x.val.permute(order).contiguous())

    def __repr__(self):
        return f"Transpose (indices={self._indices})"


class OuterProduct(LinearOperator):
    """x -> field (x) x on the product of both domains; the adjoint contracts with the field, not conjugated
    (outer_product_operator.py:25-56)"""

    def __init__(self, domain, field):
        self._domain = DomainTuple.make(domain)
        self._field = field
        self._target = DomainTuple.make(tuple(field.domain) + tuple(self._domain))
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        if self._field.device_id != x.device_id:
            self._field = self._field.at(x.device_id)
        if mode == self.TIMES:
            return self._field.outer(x)
        from .operators import ContractionOperator

        spread = ContractionOperator(self._target, tuple(range(len(self._field.domain), len(self._target)))).adjoint_times
        return (spread(self._field) * x).sum(tuple(range(len(self._field.domain))))


class ValueInserter(LinearOperator):
    """A scalar into pixel `index` of an otherwise zero field; the adjoint reads that pixel (value_inserter.py:25-64)"""

    def __init__(self, target, index):
        self._domain = DomainTuple.scalar_domain()
        self._target = DomainTuple.make(target)
        index = tuple(index)
        if not all(isinstance(n, int) for n in index) or len(index) > len(self._target.shape) or \
                not all(0 <= n < self._target.shape[i] for i, n in enumerate(index)):
            raise TypeError("index: integers inside the target's shape expected")
        if len(index) != len(self._target.shape):
            raise ValueError("index must address exactly one pixel")
        self._index = index
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        move = _embedded if mode == self.TIMES else _selected
        return move(self._tgt(mode), self._index, x.val)


class DomainTupleFieldInserter(LinearOperator):
    """A field on the target without sub-domain `space`, written into the slice of the target at `index` along that
    sub-domain (domain_tuple_field_inserter.py:26-75)"""

    def __init__(self, target, space, index):
        if not 0 <= space <= len(target):
            raise ValueError("invalid space index")
        self._target = DomainTuple.make(target)
        self._domain = DomainTuple.make([sub for i, sub in enumerate(self._target) if i != space])
        shape = self._target[space].shape
        if len(index) != len(shape):
            raise ValueError("shape mismatch between new_space and position")
        if any(p < 0 or p >= n for n, p in zip(shape, index)):
            raise ValueError("bad position value")
        first = _first_axis(self._target, space)
        self._slice = (slice(None),) * first + tuple(int(p) for p in index)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        move = _embedded if mode == self.TIMES else _selected
        return move(self._tgt(mode), self._slice, x.val)


class FieldZeroPadder(LinearOperator):
    """Zero-pads an RGSpace sub-domain to `new_shape`: at the end of every axis, or -- `central` -- in the middle, which keeps
    the low frequencies of a harmonic field at both ends (field_zero_padder.py:27-119; like there the Nyquist entry of an
    even axis is not split)"""

    def __init__(self, domain, new_shape, space=0, central=False):
        self._domain = DomainTuple.make(domain)
        self._space = _space_index(self._domain, space)
        grid = self._domain[self._space]
        if not isinstance(grid, RGSpace):
            raise TypeError("RGSpace required")
        if len(new_shape) != len(grid.shape):
            raise ValueError("Shape mismatch")
        if any(a < b for a, b in zip(new_shape, grid.shape)):
            raise ValueError("New shape must not be smaller than old shape")
        self._target = DomainTuple.make([RGSpace(new_shape, grid.distances, grid.harmonic) if i == self._space else sub
                                         for i, sub in enumerate(self._domain)])
        self._central = bool(central)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        v, wanted = x.val, self._tgt(mode).shape
        for ax in self._target.axes[self._space]:
            have, want = v.shape[ax], wanted[ax]
            if have == want:
                continue
            small = min(have, want)
            shape = list(v.shape)
            shape[ax] = want
            out = torch.zeros(shape, dtype=v.dtype, device=v.device)
            if not self._central:
                out.narrow(ax, 0, small).copy_(v.narrow(ax, 0, small))
            else:
                # frequencies 0 .. Nyquist of the small axis stay at the front, the Nyquist negative ones at the back; on an
                # even small axis the Nyquist entry is in both runs: padding writes it twice, the adjoint adds both up
                head, back = small // 2 + 1, small // 2
                out.narrow(ax, 0, head).copy_(v.narrow(ax, 0, head))
                tail_in, tail_out = v.narrow(ax, have - back, back), out.narrow(ax, want - back, back)
                if mode == self.TIMES:
                    tail_out.copy_(tail_in)
                else:
                    tail_out.add_(tail_in)
            v = out
        return Field(self._tgt(mode), v)


class SliceOperator(LinearOperator):
    """Cuts every sub-domain to `new_shape[i]` (None: unchanged) from the start or -- `center` -- from the middle, keeping
    the pixel distances of RGSpaces unless `preserve_dist` is off; the adjoint zero-fills (selection_operators.py:31-121)"""

    def __init__(self, domain, new_shape, center=False, preserve_dist=True):
        self._domain = DomainTuple.make(domain)
        if len(new_shape) != len(self._domain):
            raise ValueError(f"shape ({new_shape}) is incompatible with the shape of the domain ({self._domain.shape})")
        target, cuts = [], []
        for i, (sub, shape) in enumerate(zip(self._domain, new_shape)):
            shape = sub.shape if shape is None else tuple(int(n) for n in np.atleast_1d(shape))
            if len(shape) != len(sub.shape):
                raise ValueError(f"shape of subspace ({i}) is incompatible with the domain")
            if shape == tuple(sub.shape):
                target.append(sub)
            elif any(n > m for n, m in zip(shape, sub.shape)):
                raise ValueError(f"domain axes ({sub}) is smaller than the target shape{shape}")
            elif isinstance(sub, RGSpace):
                target.append(RGSpace(shape, sub.distances if preserve_dist else None, sub.harmonic))
            elif isinstance(sub, UnstructuredDomain):
                target.append(UnstructuredDomain(shape))
            else:
                raise ValueError(f"{type(sub).__name__} can not be sliced")
            for have, want in zip(sub.shape, shape):
                start = (have - want) // 2 if center else 0
                cuts.append(slice(start, start + want))
        self._cuts = tuple(cuts)
        self._target = DomainTuple.make(target)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    def apply(self, x, mode):
        self._check_input(x, mode)
        move = _selected if mode == self.TIMES else _embedded
        return move(self._tgt(mode), self._cuts, x.val)

    def __str__(self):
        return f"{type(self).__name__}({self._domain.shape} -> {self._target.shape})"


class SplitOperator(LinearOperator):
    """One Field into a MultiField of selections: per key a tuple with one selector per sub-domain -- None (everything), a
    slice, a boolean mask, a list of indices (each turns the sub-domain into an UnstructuredDomain of the selected length) or
    an integer (drops the sub-domain).  Selections of different keys may overlap: the adjoint adds them up
    (selection_operators.py:124-222; selectors address the first axis of their sub-domain, as there)."""

    def __init__(self, domain, slices_by_key, intersecting_slices=True):
        self._domain = DomainTuple.make(domain)
        self._overlap = bool(intersecting_slices)
        targets, self._select = {}, {}
        for key, sel in slices_by_key.items():
            if len(sel) > len(self._domain):
                raise ValueError(f"slice at key {key!r} has more dimensions than the input")
            subs, index = [], []
            for i, sub in enumerate(self._domain):
                one = sel[i] if i < len(sel) else None
                rest = (slice(None),) * (len(sub.shape) - 1)
                if one is None or (isinstance(one, slice) and one == slice(None)):
                    subs.append(sub)
                    index += [slice(None)] * len(sub.shape)
                    continue
                if isinstance(one, slice):
                    count = len(range(*one.indices(sub.shape[0])))
                elif isinstance(one, np.ndarray) and one.dtype == np.dtype(bool):
                    if one.size != sub.size:
                        raise ValueError(f"shape mismatch between desired slice {one} and the shape of the domain {sub.size}")
                    count = int(one.sum())
                    one = torch.as_tensor(np.flatnonzero(one), dtype=torch.int64)
                elif isinstance(one, (tuple, list, np.ndarray)):
                    count = len(one)
                    one = torch.as_tensor(np.asarray(one), dtype=torch.int64)
                elif isinstance(one, (int, np.integer)):
                    count, one = None, int(one)
                else:
                    raise ValueError(f"invalid type for specifying a slice; got {one}")
                if count is not None:
                    subs.append(UnstructuredDomain((count,) + tuple(sub.shape[1:])))
                index += [one, *rest]
            targets[key] = DomainTuple.make(subs)
            self._select[key] = tuple(index)
        self._target = MultiDomain.make(targets)
        self._capability = self.TIMES | self.ADJOINT_TIMES

    @staticmethod
    def _on(index, device):
        return tuple(ix.to(device) if isinstance(ix, torch.Tensor) else ix for ix in index)

    def apply(self, x, mode):
        self._check_input(x, mode)
        if mode == self.TIMES:
            v = x.val
            return MultiField.from_dict({key: Field(self._target[key], v[self._on(ix, v.device)].reshape(self._target[key].shape))
                                         for key, ix in self._select.items()}, self._target)
        first = next(iter(x.values())).val
        out = torch.zeros(self._domain.shape, dtype=first.dtype, device=first.device)
        for key, ix in self._select.items():
            ix = self._on(ix, first.device)
            part = x[key].val.reshape(out[ix].shape)
            out[ix] = out[ix] + part if self._overlap else part
        return Field(self._domain, out)

    def __str__(self):
        return f"{type(self).__name__} {self._target.keys()!r} <-"
