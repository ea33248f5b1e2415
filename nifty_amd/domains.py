"""Geometry objects of the path: RGSpace, PowerSpace, UnstructuredDomain, DomainTuple, MultiDomain.

Counterpart of reference nifty/cl/domains/{domain,structured_domain,unstructured_domain,rg_space,
power_space}.py, nifty/cl/domain_tuple.py and nifty/cl/multi_domain.py.  DomainTuple / MultiDomain
objects are interned so that identity (``is``) is the equality test, as in the reference
(domain_tuple.py:74-85, multi_domain.py:38-60).  The spherical domains (LMSpace, HPSpace, GLSpace) are geometry descriptors
only -- sizes, volumes, |k| tables, so that fields, diagonal / selection operators and power spaces over them work; the
spherical harmonic transforms between them are out of scope and HarmonicTransformOperator says so.
"""
from functools import reduce

import numpy as np


class Domain:
    """Abstract base: something with a shape and a size that can be hashed and compared by value."""

    def _key(self):
        raise NotImplementedError

    def __hash__(self):
        return hash((type(self).__name__,) + self._key())

    def __eq__(self, other):
        if self is other:
            return True
        return type(self) is type(other) and self._key() == other._key()

    def __ne__(self, other):
        return not self == other

    @property
    def shape(self):
        raise NotImplementedError

    @property
    def size(self):
        return int(reduce(lambda a, b: a * b, self.shape, 1))


class UnstructuredDomain(Domain):
    """A plain index set without geometry (reference domains/unstructured_domain.py)."""

    def __init__(self, shape):
        if np.isscalar(shape):
            shape = (shape,)
        self._shape = tuple(int(i) for i in shape)

    def _key(self):
        return (self._shape,)

    def __repr__(self):
        return f"UnstructuredDomain(shape={self._shape})"

    @property
    def shape(self):
        return self._shape

    # an index set has no geometry: unit weights, so that Field.weight / IntegrationOperator may run over mixed domains
    # (the reference defines no volume here and raises AttributeError when such a space is weighted)
    scalar_dvol = dvol = property(lambda self: 1.0)
    total_volume = property(lambda self: float(self.size))


class StructuredDomain(Domain):
    """A domain with volume information (reference domains/structured_domain.py)."""

    @property
    def scalar_dvol(self):
        raise NotImplementedError

    @property
    def dvol(self):
        return self.scalar_dvol

    @property
    def total_volume(self):
        d = self.dvol
        return self.size * d if np.isscalar(d) else float(np.sum(d))

    @property
    def harmonic(self):
        raise NotImplementedError


class RGSpace(StructuredDomain):
    """Regular Cartesian grid with periodic boundaries (reference domains/rg_space.py:59-227).

    `distances` are the pixel distances of THIS space; for a harmonic space they are stored
    internally via the position-space distances of its partner, exactly like the reference, so that
    ``space.get_default_codomain().get_default_codomain() == space`` bit for bit.
    """

    def __init__(self, shape, distances=None, harmonic=False, _realdistances=None):
        self._harmonic = bool(harmonic)
        self._shape = tuple(int(n) for n in np.atleast_1d(shape))
        if any(n < 0 for n in self._shape):
            raise ValueError("Negative number of pixels encountered")
        cells = np.array(self._shape, dtype=np.float64)
        if _realdistances is not None:       # handed over by the partner space: bit-identical round trips
            position = np.array(_realdistances, dtype=np.float64)
        elif distances is None:              # unit total volume in position space
            position = 1.0 / cells
        else:                                # given for THIS space: a harmonic space stores its partner's distances
            own = np.broadcast_to(np.asarray(distances, dtype=np.float64), cells.shape)
            position = 1.0 / (cells * own) if self._harmonic else own
        if position.shape != cells.shape or (position <= 0).any():
            raise ValueError("Non-positive distances encountered")
        self._rdistances = tuple(float(d) for d in position)
        self._hdistances = tuple(float(d) for d in 1.0 / (cells * position))
        self._dvol = float(reduce(lambda a, b: a * b, self.distances))

    def _key(self):
        return (self._shape, self._rdistances, self._harmonic)

    def __repr__(self):
        return f"RGSpace(shape={self._shape}, distances={self.distances}, harmonic={self._harmonic})"

    @property
    def harmonic(self):
        return self._harmonic

    @property
    def shape(self):
        return self._shape

    @property
    def distances(self):
        return self._hdistances if self._harmonic else self._rdistances

    @property
    def scalar_dvol(self):
        return self._dvol

    @property
    def extents(self):
        return tuple(n * d for n, d in zip(self._shape, self.distances))

    def get_default_codomain(self):
        return RGSpace(self._shape, None, not self._harmonic, self._rdistances)

    def check_codomain(self, codomain):
        if not isinstance(codomain, RGSpace):
            raise TypeError("domain is not a RGSpace")
        # a grid and its Fourier partner: same cells, opposite character, n_i * d_i * d'_i = 1 on every axis
        unit = [n * d * dk for n, d, dk in zip(self._shape, self.distances, codomain.distances)]
        problems = (("The shapes of domain and codomain must be identical.", self._shape != codomain.shape),
                    ("domain.harmonic and codomain.harmonic must not be the same.", self._harmonic == codomain.harmonic),
                    ("The grid-distances of domain and codomain do not match.", any(abs(u - 1.0) >= 1e-7 for u in unit)))
        for message, wrong in problems:
            if wrong:
                raise AttributeError(message)

    # --- harmonic-space geometry ---------------------------------------------------------------
    def _dist_array(self):
        """|k| for every pixel (numpy, host)."""
        out = None
        for n, d in zip(self._shape, self.distances):
            ax = np.arange(n, dtype=np.float64)
            ax = np.minimum(ax, n - ax) * d
            if len(self._shape) == 1:
                return ax
            ax *= ax
            out = ax if out is None else np.add.outer(out, ax)
        return np.sqrt(out)

    def get_k_length_array(self):
        if not self._harmonic:
            raise NotImplementedError
        from .field import Field

        return Field.from_raw(self, self._dist_array())

    def get_fft_smoothing_kernel_function(self, sigma):
        """k -> exp(-2 pi^2 sigma^2 k^2): Fourier image of a Gaussian of width sigma (reference rg_space.py:164-171)."""
        if not self._harmonic:
            raise NotImplementedError
        return lambda x: (x * x * (-2.0 * np.pi * np.pi * sigma * sigma)).ptw("exp")

    def equal_distances(self):
        return bool(np.all(np.array(self.distances) == self.distances[0]))

    def _k2_flags(self):
        """bool table over integer k^2 (equal distances): which squared radii occur on the grid."""
        half = np.asarray(self._shape) // 2
        flags = np.zeros(int(np.sum(half * half)) + 1, dtype=bool)
        acc = None
        for h in half:
            sq = np.arange(h + 1, dtype=np.int64) ** 2
            acc = sq if acc is None else np.add.outer(acc, sq)
        flags[acc] = True
        return flags

    def get_unique_k_lengths(self):
        if not self._harmonic:
            raise NotImplementedError
        if len(self._shape) == 1:
            return np.arange(self._shape[0] // 2 + 1, dtype=np.float64) * self.distances[0]
        if self.equal_distances():
            return np.sqrt(np.nonzero(self._k2_flags())[0]) * self.distances[0]
        vals = np.unique(self._dist_array())
        tol = 1e-12 * vals[-1]
        return vals[np.diff(np.r_[vals, 2 * vals[-1]]) > tol]


class PowerSpace(StructuredDomain):
    """Radial bins of a harmonic RGSpace (reference domains/power_space.py:155-198).

    ``pindex`` (host int64, like the reference) is materialised lazily and only for moderate grid
    sizes; large equal-distance grids get their bin statistics from the integer-k^2 table and the
    per-pixel index is produced directly on the device (``device_pindex``), never as an 8N-byte host
    array.
    """

    _cache = {}
    HOST_PINDEX_LIMIT = 1 << 27

    def __init__(self, harmonic_partner, binbounds=None):
        if not (isinstance(harmonic_partner, StructuredDomain) and harmonic_partner.harmonic):
            raise ValueError("harmonic_partner must be a harmonic space.")
        if harmonic_partner.scalar_dvol is None:
            raise ValueError("harmonic partner must have scalar volume factors")
        self._hp = harmonic_partner
        if binbounds is not None:
            binbounds = tuple(binbounds)
            if min(binbounds) < 0:
                raise ValueError("Negative binbounds encountered")
        self._binbounds = binbounds
        key = (harmonic_partner, binbounds)
        data = PowerSpace._cache.get(key)
        if data is None:
            data = PowerSpace._cache[key] = self._compute(harmonic_partner, binbounds)
        self._data = data

    @staticmethod
    def _compute(hp, binbounds):
        pdvol = hp.scalar_dvol
        natural_k2 = binbounds is None and isinstance(hp, RGSpace) and hp.equal_distances() and len(hp.shape) <= 3
        if natural_k2 and hp.size > PowerSpace.HOST_PINDEX_LIMIT:
            # statistics from the k^2 histogram without touching a full-grid array
            flags = hp._k2_flags()
            k2 = np.nonzero(flags)[0]
            table = (np.cumsum(flags) - 1).astype(np.int32)
            rho = PowerSpace._rho_from_axes(hp.shape, table, len(k2))
            kl = np.sqrt(k2.astype(np.float64)) * hp.distances[0]
            return dict(pindex=None, k2table=table, rho=rho, k_lengths=kl, dvol=rho * pdvol)
        karr = hp._dist_array()
        if binbounds is None:
            u = hp.get_unique_k_lengths()
            tbb = 0.5 * (u[:-1] + u[1:])
        else:
            tbb = binbounds
        pindex = np.searchsorted(tbb, karr)
        nbin = len(tbb) + 1
        rho = np.bincount(pindex.ravel(), minlength=nbin)
        if (rho == 0).any():
            raise ValueError("empty bins detected")
        kl = np.bincount(pindex.ravel(), weights=karr.ravel(), minlength=nbin).astype(np.float64) / rho
        pindex.flags.writeable = False
        kl.flags.writeable = False
        table = None
        if natural_k2:
            table = (np.cumsum(hp._k2_flags()) - 1).astype(np.int32)
        return dict(pindex=pindex, k2table=table, rho=rho, k_lengths=kl, dvol=rho * pdvol)

    @staticmethod
    def _rho_from_axes(shape, table, nbin):
        """Bin multiplicities by convolving per-axis k^2 histograms (no N-sized array)."""
        hist = None
        for n in shape:
            ax = np.arange(n, dtype=np.int64)
            ax = np.minimum(ax, n - ax) ** 2
            h = np.bincount(ax)
            if hist is None:
                hist = h
            else:
                nz_a, nz_b = np.nonzero(hist)[0], np.nonzero(h)[0]
                out = np.zeros(len(hist) + len(h) - 1, dtype=np.int64)
                np.add.at(out, np.add.outer(nz_a, nz_b).ravel(), np.multiply.outer(hist[nz_a], h[nz_b]).ravel())
                hist = out
        rho = np.zeros(nbin, dtype=np.int64)
        nz = np.nonzero(hist)[0]
        np.add.at(rho, table[nz], hist[nz])
        return rho

    def _key(self):
        return (self._hp, self._binbounds)

    def __repr__(self):
        return f"PowerSpace(harmonic_partner={self._hp}, binbounds={self._binbounds})"

    @property
    def harmonic(self):
        return False

    @property
    def shape(self):
        return self._data["k_lengths"].shape

    @property
    def scalar_dvol(self):
        return None

    @property
    def dvol(self):
        return self._data["dvol"]

    @property
    def harmonic_partner(self):
        return self._hp

    @property
    def binbounds(self):
        return self._binbounds

    @property
    def k_lengths(self):
        return self._data["k_lengths"]

    @property
    def rho(self):
        return self._data["rho"]

    @property
    def pindex(self):
        p = self._data["pindex"]
        if p is None:
            raise MemoryError("pindex of this large grid is only available on the device (device_pindex)")
        return p

    def device_pindex(self, device):
        """int32 bin index per pixel as a flat device tensor (cached per device)."""
        import torch

        device = torch.device(device)
        cache = self._data.setdefault("dev", {})
        key = str(device)
        if key in cache:
            return cache[key]
        if self._data["pindex"] is not None:
            t = torch.from_numpy(np.ascontiguousarray(self._data["pindex"], dtype=np.int32).ravel()).to(device)
        else:
            import ctypes

            from . import _lib as L

            shape = self._hp.shape
            table = torch.from_numpy(self._data["k2table"]).to(device)
            t = torch.empty(self._hp.size, dtype=torch.int32, device=device)
            shp = (ctypes.c_int64 * len(shape))(*shape)
            with torch.cuda.device(device):
                L.check(L.load().nk_pindex_from_k2(len(shape), shp, table.data_ptr(), t.data_ptr(), 0,
                                                   torch.cuda.current_stream().cuda_stream), "nk_pindex_from_k2")
        cache[key] = t
        return t

    @staticmethod
    def linear_binbounds(nbin, first_bound, last_bound):
        nbin = int(nbin)
        if nbin < 3:
            raise ValueError("nbin must be at least 3")
        return np.linspace(float(first_bound), float(last_bound), nbin - 1)

    @staticmethod
    def logarithmic_binbounds(nbin, first_bound, last_bound):
        nbin = int(nbin)
        if nbin < 3:
            raise ValueError("nbin must be at least 3")
        return np.logspace(np.log(float(first_bound)), np.log(float(last_bound)), nbin - 1, base=np.e)


    @staticmethod
    def useful_binbounds(space, logarithmic, nbin=None):
        """Bin bounds that leave no bin of `space` empty (power_space.py:105-153).  The outermost bounds sit half-way into
        the first and the last gap between distinct |k|; in between the bins are equally wide in |k| -- or, `logarithmic`,
        in log |k| -- and at most so many that the widest gap on that scale still catches a point.  `nbin` None: as many as
        possible.  (logarithmic, nbin) = (None, None) asks for the natural binning: returns None."""
        if not (isinstance(space, StructuredDomain) and space.harmonic):
            raise ValueError("first argument must be a harmonic space.")
        if (logarithmic, nbin) == (None, None):
            return None
        lengths = np.asarray(space.get_unique_k_lengths(), dtype=np.float64)
        if lengths.size < 3:
            raise ValueError("Space does not have enough unique k lengths")
        outer = (lengths[:2].mean(), lengths[-2:].mean())
        marks = np.concatenate(([outer[0]], lengths[1:-1], [outer[1]]))
        on_scale = np.log(marks) if logarithmic else marks
        finest = int((on_scale[-1] - on_scale[0]) / np.diff(on_scale).max()) + 2
        wanted = finest if nbin is None else int(nbin)
        if wanted < 3:
            raise ValueError("nbin must be at least 3")
        if wanted > finest:
            raise ValueError("nbin is too large")
        return (PowerSpace.logarithmic_binbounds if logarithmic else PowerSpace.linear_binbounds)(wanted, *outer)


class LMSpace(StructuredDomain):
    """Spherical-harmonic coefficients a_lm up to lmax / mmax, stored as reals: the m = 0 column (lmax+1 values), then for
    every m >= 1 the (real, imaginary) pairs of l = m..lmax (reference domains/lm_space.py:24-165).  Geometry only."""

    def __init__(self, lmax, mmax=None):
        self._lmax = int(lmax)
        if self._lmax < 0:
            raise ValueError("lmax must be >=0.")
        self._mmax = self._lmax if mmax is None else int(mmax)
        if not 0 <= self._mmax <= self._lmax:
            raise ValueError("mmax must be >=0 and <=lmax.")

    def _key(self):
        return (self._lmax, self._mmax)

    def __repr__(self):
        return f"LMSpace(lmax={self._lmax}, mmax={self._mmax})"

    harmonic = property(lambda self: True)
    lmax = property(lambda self: self._lmax)
    mmax = property(lambda self: self._mmax)
    scalar_dvol = property(lambda self: 1.0)

    @property
    def size(self):
        # one real per (l, m = 0), two per (l, m >= 1): sum over m of the column lengths lmax + 1 - m
        cols = self._lmax + 1 - np.arange(self._mmax + 1)
        return int(2 * cols.sum() - cols[0])

    @property
    def shape(self):
        return (self.size,)

    def _dist_array(self):
        """l of every stored coefficient (float64)"""
        ls = np.arange(self._lmax + 1, dtype=np.float64)
        return np.concatenate([ls] + [np.repeat(ls[m:], 2) for m in range(1, self._mmax + 1)])

    def get_k_length_array(self):
        from .field import Field

        return Field.from_raw(self, self._dist_array())

    def get_unique_k_lengths(self):
        return np.arange(self._lmax + 1, dtype=np.float64)

    def get_fft_smoothing_kernel_function(self, sigma):
        """l -> exp(-sigma^2 l (l+1) / 2), the spherical image of a Gaussian beam (lm_space.py:86-94)"""
        return lambda x: ((x + 1.0) * x * (-0.5 * sigma * sigma)).ptw("exp")

    def get_default_codomain(self):
        return GLSpace(self._lmax + 1, 2 * self._mmax + 1)

    def check_codomain(self, codomain):
        if not isinstance(codomain, (GLSpace, HPSpace)):
            raise TypeError("codomain must be a GLSpace or HPSpace.")


class HPSpace(StructuredDomain):
    """The sphere in 12 nside^2 equal-area HEALPix pixels (reference domains/hp_space.py:23-93).  Geometry only."""

    def __init__(self, nside):
        self._nside = int(nside)
        if self._nside < 1:
            raise ValueError("nside must be >=1.")

    def _key(self):
        return (self._nside,)

    def __repr__(self):
        return f"HPSpace(nside={self._nside})"

    harmonic = property(lambda self: False)
    nside = property(lambda self: self._nside)
    size = property(lambda self: 12 * self._nside * self._nside)
    shape = property(lambda self: (self.size,))
    scalar_dvol = property(lambda self: np.pi / (3 * self._nside * self._nside))

    def get_default_codomain(self):
        return LMSpace(lmax=2 * self._nside)

    def check_codomain(self, codomain):
        if not isinstance(codomain, LMSpace):
            raise TypeError("codomain must be a LMSpace.")


class GLSpace(StructuredDomain):
    """The sphere on nlat Gauss-Legendre rings of nlon pixels each (reference domains/gl_space.py:23-126).  Geometry only;
    the ring weights are numpy's Gauss-Legendre weights times the azimuthal pixel width 2 pi / nlon."""

    def __init__(self, nlat, nlon=None):
        counts = dict(nlat=int(nlat))
        counts["nlon"] = 2 * counts["nlat"] - 1 if nlon is None else int(nlon)  # default: just enough pixels per ring
        for name, n in counts.items():
            if n < 1:
                raise ValueError(f"{name} must be a positive number.")
        self._nlat, self._nlon, self._ring_weights = counts["nlat"], counts["nlon"], None

    def _key(self):
        return (self._nlat, self._nlon)

    def __repr__(self):
        return f"GLSpace(nlat={self._nlat}, nlon={self._nlon})"

    harmonic = property(lambda self: False)
    nlat = property(lambda self: self._nlat)
    nlon = property(lambda self: self._nlon)
    size = property(lambda self: self._nlat * self._nlon)
    shape = property(lambda self: (self.size,))
    scalar_dvol = property(lambda self: None)
    total_volume = property(lambda self: 4 * np.pi)

    @property
    def dvol(self):
        if self._ring_weights is None:
            self._ring_weights = np.polynomial.legendre.leggauss(self._nlat)[1] * (2 * np.pi / self._nlon)
        return np.repeat(self._ring_weights, self._nlon)

    def get_default_codomain(self):
        mmax = self._nlon // 2
        return LMSpace(lmax=max(mmax, self._nlat - 1), mmax=mmax)

    def check_codomain(self, codomain):
        if not isinstance(codomain, LMSpace):
            raise TypeError("codomain must be a LMSpace.")


def _interned(cache, key, build):
    """cache[key], built on first request: one object per distinct domain, so identity compares domains"""
    if key not in cache:
        cache[key] = build()
    return cache[key]


class DomainTuple:
    """Ordered product of Domains, interned (reference domain_tuple.py)."""

    _cache = {}
    _scalar = None

    def __init__(self, domain, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError("To create a DomainTuple call `DomainTuple.make()`.")
        self._dom = self._parse(domain)
        ranks = [len(d.shape) for d in self._dom]
        firsts = [sum(ranks[:i]) for i in range(len(ranks))]  # first array axis of every sub-domain
        self._axes = tuple(tuple(range(lo, lo + n)) for lo, n in zip(firsts, ranks))
        self._shape = tuple(s for d in self._dom for s in d.shape)
        self._size = int(np.prod(self._shape, dtype=np.int64)) if self._shape else 1

    @staticmethod
    def _parse(domain):
        if domain is None:
            return ()
        if isinstance(domain, DomainTuple):
            return domain._dom
        if isinstance(domain, Domain):
            return (domain,)
        if not isinstance(domain, (tuple, list)):
            raise TypeError(f"Given object {domain!r} contains something that is not an instance of Domain class.")
        for d in domain:
            if not isinstance(d, Domain):
                raise TypeError(f"Given object {d!r} is not an instance of Domain class.")
        return tuple(domain)

    @staticmethod
    def make(domain):
        """The ONE DomainTuple object of these sub-domains (interning makes `is` the domain comparison)."""
        if isinstance(domain, (DomainTuple, dict)):
            return domain if isinstance(domain, DomainTuple) else MultiDomain.make(domain)
        parts = DomainTuple._parse(domain)
        return _interned(DomainTuple._cache, parts, lambda: DomainTuple(parts, _callingfrommake=True))

    @staticmethod
    def scalar_domain():
        if DomainTuple._scalar is None:
            DomainTuple._scalar = DomainTuple.make(())
        return DomainTuple._scalar

    def __getitem__(self, i):
        return self._dom[i]

    def __len__(self):
        return len(self._dom)

    def __iter__(self):
        return iter(self._dom)

    @property
    def shape(self):
        return self._shape

    @property
    def size(self):
        return self._size

    @property
    def axes(self):
        return self._axes

    def _chosen(self, spaces):
        n = len(self._dom)
        if spaces is None:
            return tuple(range(n))
        chosen = (int(spaces),) if np.isscalar(spaces) else tuple(int(x) for x in spaces)
        if any(c < 0 or c >= n for c in chosen) or len(set(chosen)) != len(chosen):
            raise ValueError("invalid sub-domain index")
        return tuple(sorted(chosen))

    def scalar_weight(self, spaces=None):
        """The uniform pixel volume over the sub-domains `spaces`, None when one of them has individual volumes
        (domain_tuple.py:137-147)"""
        vol = 1.0
        for c in self._chosen(spaces):
            one = self._dom[c].scalar_dvol
            if one is None:
                return None
            vol *= one
        return vol

    def total_volume(self, spaces=None):
        """domain_tuple.py:149-159"""
        vol = 1.0
        for c in self._chosen(spaces):
            vol *= self._dom[c].total_volume
        return vol


    def __hash__(self):
        return hash(self._dom)

    def __eq__(self, other):
        if self is other:
            return True
        return isinstance(other, DomainTuple) and self._dom == other._dom

    def __ne__(self, other):
        return not self == other

    def __reduce__(self):
        return (DomainTuple.make, (self._dom,))

    def __repr__(self):
        body = "\n".join(f"* {d!r}" for d in self._dom)
        return f"DomainTuple, len: {len(self._dom)}" + ("\n" + body if body else "")


class MultiDomain:
    """Alphabetically ordered dictionary key -> DomainTuple, interned (reference multi_domain.py)."""

    _cache = {}

    def __init__(self, dct, _callingfrommake=False):
        if not _callingfrommake:
            raise NotImplementedError("To create a MultiDomain call `MultiDomain.make()`.")
        self._keys = tuple(sorted(dct.keys()))
        self._domains = tuple(dct[k] for k in self._keys)
        self._idx = {k: i for i, k in enumerate(self._keys)}

    @staticmethod
    def make(inp):
        """The ONE MultiDomain object with these keys and sub-domains (interned like DomainTuple)."""
        if isinstance(inp, MultiDomain):
            return inp
        if not (isinstance(inp, dict) and all(isinstance(key, str) for key in inp)):
            raise TypeError("dict expected" if not isinstance(inp, dict) else "keys must be strings")
        entries = tuple((key, DomainTuple.make(inp[key])) for key in sorted(inp))
        return _interned(MultiDomain._cache, entries, lambda: MultiDomain(dict(entries), _callingfrommake=True))

    def keys(self):
        return self._keys

    def values(self):
        return self._domains

    def domains(self):
        return self._domains

    def items(self):
        return zip(self._keys, self._domains)

    @property
    def idx(self):
        return self._idx

    def __getitem__(self, key):
        return self._domains[self._idx[key]]

    def __len__(self):
        return len(self._keys)

    def __contains__(self, key):
        return key in self._idx

    @property
    def size(self):
        return sum(d.size for d in self._domains)

    def __hash__(self):
        return hash((self._keys, self._domains))

    def __eq__(self, other):
        if self is other:
            return True
        return isinstance(other, MultiDomain) and self._keys == other._keys and self._domains == other._domains

    def __ne__(self, other):
        return not self == other

    @staticmethod
    def union(inp):
        merged = {}
        for key, sub in (item for dom in inp for item in dom.items()):
            if merged.setdefault(key, sub) != sub:
                raise ValueError(f"domain mismatch for key {key!r}")
        return MultiDomain.make(merged)

    def __reduce__(self):
        return (MultiDomain.make, (dict(self.items()),))

    def __repr__(self):
        lines = ["MultiDomain:"]
        for k, d in self.items():
            lines.append("  " + k + ": " + repr(d).replace("\n", "\n  "))
        return "\n".join(lines)


def makeDomain(domain):
    """Reference sugar.makeDomain: dict -> MultiDomain, anything else -> DomainTuple."""
    if isinstance(domain, (MultiDomain, dict)):
        return MultiDomain.make(domain)
    return DomainTuple.make(domain)
