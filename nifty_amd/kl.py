"""Sampled KL energy, MGVI / geoVI sample drawing and sample lists on the generic operator graph.

Counterpart of reference nifty/cl/minimization/kl_energies.py (draw_samples :91-159, SampledKLEnergy
:162-296, SampledKLEnergyClass :299-360), minimization/sample_list.py (SampleListBase :46-383,
ResidualSampleList :386-498, SampleList :501-597) and minimization/energy_adapter.py.  The fused
single-GPU engine (engine.py) implements the same mathematics for the recognised CorrelatedField +
likelihood pattern; optimize_kl picks it automatically.

Difference by design: the linearisation of the Hamiltonian at every sample is cached when the KL
energy is built, so one metric application costs one forward + one adjoint Jacobian per sample (the
reference re-linearises inside every apply_metric call; mathematically identical).
"""
import functools
import os
import pickle

import numpy as np
import torch

from .energy_operators import GaussianEnergy, StandardHamiltonian
from .field import Field, MultiField
from .minimization import DescentMinimizer, Energy
from .operators import (EndomorphicOperator, Linearization, Operator, SamplingEnabler, SandwichOperator, ScalingOperator,
                        makeDomain)
from . import parallel
from .parallel import SamplePlan, get_MPI_params_from_comm, shareRange


def _scalar_value(field):
    v = field.asnumpy()[()]
    return float(np.real(v))


def _nan_to_inf(value, nanisinf):
    """an energy that cannot be evaluated counts as infinitely bad when the caller asks for that"""
    return np.inf if (nanisinf and np.isnan(value)) else value


class EnergyAdapter(Energy):
    """Energy protocol on top of an EnergyOperator (reference energy_adapter.py:28-110)."""

    def __init__(self, position, op, constants=[], want_metric=False, nanisinf=False):
        if constants:  # constant keys are frozen into the operator; the energy lives on the remaining keys
            position, op = _reduce_by_keys(position, op, constants)
        super().__init__(position)
        self._op, self._want_metric, self._nanisinf = op, want_metric, nanisinf
        here = op(Linearization.make_var(position, want_metric))
        self._grad, self._metric = here.gradient, here.metric
        self._val = _nan_to_inf(_scalar_value(here.val), nanisinf)

    def at(self, position):
        return EnergyAdapter(position, self._op, want_metric=self._want_metric, nanisinf=self._nanisinf)

    @property
    def value(self):
        return self._val

    @property
    def gradient(self):
        return self._grad

    @property
    def metric(self):
        return self._metric

    def apply_metric(self, x):
        return self._metric(x)


class _SelfAdjointOperatorWrapper(EndomorphicOperator):
    def __init__(self, domain, func):
        self._func = func
        self._capability = self.TIMES | self.ADJOINT_TIMES
        self._domain = makeDomain(domain)

    def apply(self, x, mode):
        self._check_input(x, mode)
        return self._func(x)


# ------------------------------------------------------------------------------------------------
# sample lists
# ------------------------------------------------------------------------------------------------
class SampleListBase:
    """A set of samples distributed over the ranks of ``comm`` (sample_list.py:46-383)."""

    def __init__(self, comm, domain):
        self._comm = comm
        self._domain = makeDomain(domain)
        ntask, rank, _ = get_MPI_params_from_comm(comm)
        self._n_total = None
        self._ntask, self._rank = ntask, rank

    @property
    def comm(self):
        return self._comm

    @property
    def MPI_master(self):
        """True on the rank that writes files (sample_list.py:95-97)"""
        return get_MPI_params_from_comm(self._comm)[2]

    @property
    def domain(self):
        return self._domain

    def n_local_samples(self):
        raise NotImplementedError

    def local_item(self, i):
        raise NotImplementedError

    @property
    def n_samples(self):
        """Global sample count (one scalar all-reduce, cached: the local count of a sample list never changes)."""
        if self._n_total is None:
            n = self.n_local_samples()
            self._n_total = n if self._comm is None else int(round(self._comm.sum_float(n)))
        return self._n_total

    def local_iterator(self, op=None):
        op = _as_function(op)
        for i in range(self.n_local_samples()):
            yield op(self.local_item(i))

    def iterator(self, op=None):
        """All samples in global order on EVERY rank (sample_list.py:186-210): each sample is handed over by the rank that
        holds it -- one object broadcast per sample (through the host: pickled Fields), so this is for diagnostics and
        small statistics, not for the hot path (average() / local_iterator() never move samples)."""
        comm = self._comm
        if comm is None or comm.size == 1:
            yield from self.local_iterator(op)
            return
        op = _as_function(op)
        counts = comm.allgather_object(self.n_local_samples())
        device_id = self._device_id()
        for owner, count in enumerate(counts):
            for i in range(count):
                mine = _to_host(self.local_item(i)) if owner == comm.rank else None
                yield op(comm.bcast_object(mine, root=owner).at(device_id))

    def _device_id(self):
        return -1

    def _sum_over_ranks(self, local_terms, like=None):
        """Sum of Fields / MultiFields / floats (or tuples of those) over the local terms and over all ranks, added in the
        rank-count-independent order of the reference (utilities.py:349-414 -> parallel.pair_tree): the same bits for every
        split of the samples over ranks, ranks WITHOUT samples included (shareRange leaves ranks empty when there are fewer
        samples than ranks).  `like`: an object shaped like one term (floats, fields on this rank's device) for such a rank
        to receive the total into -- the caller usually knows it; without it the first rank that holds a term broadcasts a
        host template (a pickled collective per call: diagnostics only).  NK_TREE_SUM=0: local running sum + one
        all-reduce (rounds 1-3)."""
        terms, comm = list(local_terms), self._comm
        tree = parallel.tree_sum_enabled()
        alone = comm is None or comm.size == 1
        if alone and not terms:
            raise ValueError("sum over an empty sample list")
        if alone:
            return parallel.tree_fold(terms, _add) if tree else functools.reduce(_add, terms)
        # how the terms are spread over the ranks never changes for a list: asked once (one pickled all-gather), not per call
        if getattr(self, "_counts", None) is None:
            self._counts = comm.term_counts(len(terms))
        if not any(self._counts):
            raise ValueError("sum over an empty sample list")
        if not terms:
            if like is None:
                root = next(r for r, c in enumerate(self._counts) if c)
                like = _place_like(comm.bcast_object(None, root=root), self._device_id())
            cache = self.__dict__.setdefault("_zeros", {})
            key = _map(like, _zero_key)
            if key not in cache:
                cache[key] = _map(like, lambda o: 0.0 if isinstance(o, float) else _zeros_like(o))
            like = cache[key]
        elif like is None and not all(self._counts):
            root = next(r for r, c in enumerate(self._counts) if c)
            if comm.rank == root:
                comm.bcast_object(_zero_like_host(terms[0]), root=root)
            else:
                comm.bcast_object(None, root=root)
        if not tree:
            # running sum of the local terms, then ONE all-reduce of a private copy (the collective writes in place)
            box = _Flat(functools.reduce(_add, terms) if terms else like, comm)
            return box.rebuilt(comm.allreduce_sum_(box.tensors))
        boxes = [_Flat(t, comm) for t in terms]
        spare = _Flat(like, comm) if not terms else None
        total = comm.tree_allreduce([b.tensors for b in boxes], self._counts, like=None if spare is None else spare.tensors)
        return (boxes[0] if boxes else spare).rebuilt(total)

    def average(self, op=None):
        """Mean of op(sample) over ALL samples (sample_list.py:212-237)."""
        total = self._sum_over_ranks(self.local_iterator(op))
        return total * (1.0 / self.n_samples)

    def _average_2tuple(self, op):
        """Mean of a (float, field) pair (sample_list.py:239-270)."""
        v, f = self._sum_over_ranks((float(a), b) for a, b in self.local_iterator(op))
        n = self.n_samples
        return v / n, f * (1.0 / n)

    def sample_stat(self, op=None):
        """(mean, variance) over the samples (sample_list.py:272-293); distributed lists go through iterator()."""
        n = self.n_samples
        if n == 1:  # (sample_list.py:287-289: the mean and a zero variance)
            only = self.average(op)
            return only, only * 0.0
        samples = list(self.iterator(op))
        mean = samples[0]
        for s in samples[1:]:
            mean = mean + s
        mean = mean * (1.0 / n)
        var = None
        for s in samples:
            d = s - mean
            var = d * d if var is None else var + d * d
        return mean, var * (1.0 / (n - 1))

    def save(self, file_name_base, overwrite=False):
        raise NotImplementedError

    def save_to_hdf5(self, file_name, op=None, samples=False, mean=False, std=False, overwrite=False):
        """Export to one HDF5 file (sample_list.py:104-184): group ``samples`` with one entry ``"0"``, ``"1"``, ... per sample
        of op(sample) in global order, group ``stats`` with ``mean`` and / or ``standard deviation``; a MultiField entry is
        a sub-group with one data set per key.  File attributes: ``nifty domain`` (and for an Operator `op` its string
        representation, domain and target).  Rank 0 writes; every rank takes part in the sample hand-over.  `h5py` is an
        optional dependency imported here, as in the reference: without it this raises ImportError."""
        import h5py

        if not (samples or mean or std):
            raise ValueError("Neither samples nor mean nor standard deviation shall be written.")
        writer = self._rank == 0
        if os.path.isfile(file_name):
            if not overwrite:
                raise RuntimeError(f"File {file_name} already exists. Delete it or use `overwrite=True`")
            if writer:
                os.remove(file_name)
        out = h5py.File(file_name, "w") if writer else _NullGroup()
        described = op if isinstance(op, Operator) else None
        out.attrs["nifty domain"] = repr(self.domain if described is None else described.target)
        if described is not None:
            out.attrs.update({"nifty operator string representation": str(op), "nifty operator domain": repr(op.domain),
                              "nifty operator target": repr(op.target)})
        entries = []  # (group, name, field) in the order they are written; every rank walks the same collectives
        if samples:
            entries += [("samples", str(number), sample) for number, sample in enumerate(self.iterator(op))]
        if std:
            mu, variance = self.sample_stat(op)
            entries += ([("stats", "mean", mu)] if mean else []) + [("stats", "standard deviation", variance.sqrt())]
        elif mean:
            entries.append(("stats", "mean", self.average(op)))
        groups = {}
        for group, name, field in entries:
            if group not in groups:
                groups[group] = out.create_group(group)
            _write_hdf5_entry(groups[group], name, field)
        out.close()
        if self._comm is not None:
            self._comm.barrier()


class _NullGroup:
    """What a rank that does not write holds instead of the h5py file: accepts everything, stores nothing."""

    def __init__(self):
        self.attrs = {}

    def create_group(self, name):
        return self

    def create_dataset(self, name, data=None):
        return None

    def close(self):
        pass


def _write_hdf5_entry(group, name, obj):
    """A Field as a data set, a MultiField as a sub-group of data sets (reference sample_list.py:632-642)."""
    if not isinstance(name, str):
        raise TypeError("HDF5 entry names are strings")
    if isinstance(obj, MultiField):
        sub = group.create_group(name)
        for key, fld in obj.items():
            _write_hdf5_entry(sub, key, fld)
    elif isinstance(obj, Field):
        group.create_dataset(name, data=obj.asnumpy())
    else:
        raise TypeError("only Fields and MultiFields can be written")


class ResidualSampleList(SampleListBase):
    """mean +/- residual_i (sample_list.py:386-498)."""

    def __init__(self, mean, residuals, neg, comm=None):
        super().__init__(comm, mean.domain)
        self._m = mean
        self._r = tuple(residuals)
        self._n = tuple(neg)
        if len(self._r) != len(self._n):
            raise ValueError("length mismatch between residuals and neg")
        r_dom = self._r[0].domain if self._r else None
        for r in self._r:
            if r.domain is not r_dom:
                raise ValueError("all residuals must live on the same domain")

    def n_local_samples(self):
        return len(self._r)

    def local_item(self, i):
        return self._m.flexible_addsub(self._r[i], self._n[i])

    def _device_id(self):
        return self._m.device_id

    def at(self, mean):
        """Only the entries present in `mean` are updated (sample_list.py:436-455)."""
        if isinstance(self._m, MultiField) and self.domain is not mean.domain:
            mean = MultiField.union([self._m, mean])
        sl = ResidualSampleList(mean, self._r, self._n, self._comm)
        sl._n_total = self._n_total  # same residuals: no new count reduction per evaluation
        return sl

    @property
    def mean(self):
        return self._m

    def save(self, file_name_base, overwrite=False):
        """``<base>.<global number>.pickle`` = [residual, neg] per sample, written by the rank that holds it, and
        ``<base>.mean.pickle`` by rank 0 (sample_list.py:467-484; host Fields).  The number of samples of a saved list is the
        length of the run of consecutive sample files, so a stale ``<base>.<n_samples>.pickle`` of a longer earlier list is
        removed first (:673-683)."""
        total = self.n_samples
        first = shareRange(total, self._ntask, self._rank)[0]  # global number of this rank's first sample
        _end_sample_run(file_name_base, total, overwrite, self._comm, self._rank)
        files = {_sample_file(file_name_base, first + i): [_to_host(res), neg] for i, (res, neg) in enumerate(zip(self._r, self._n))}
        if self._rank == 0:
            files[f"{file_name_base}.mean.pickle"] = _to_host(self._m)
        for name, content in files.items():
            _dump(name, content, overwrite)

    @staticmethod
    def load_mean(file_name_base):
        with open(f"{file_name_base}.mean.pickle", "rb") as f:
            return pickle.load(f)

    @staticmethod
    def load(file_name_base, comm=None, device_id=-1):
        res, neg = [], []
        for name in _local_sample_files(file_name_base, comm):
            with open(name, "rb") as f:
                r, n = pickle.load(f)
            res.append(r.at(device_id))
            neg.append(n)
        mean = ResidualSampleList.load_mean(file_name_base).at(device_id)
        return ResidualSampleList(mean, res, neg, comm)


class SampleList(SampleListBase):
    """Explicit list of samples (sample_list.py:501-597)."""

    def __init__(self, samples, comm=None, domain=None):
        if domain is None:
            if not samples:
                raise ValueError("need a domain for an empty SampleList")
            domain = samples[0].domain
        super().__init__(comm, domain)
        self._s = tuple(samples)

    def n_local_samples(self):
        return len(self._s)

    def local_item(self, i):
        return self._s[i]

    def _device_id(self):
        return self._s[0].device_id if self._s else -1

    def save(self, file_name_base, overwrite=False):
        """``<base>.<global number>.pickle`` per sample (sample_list.py:541-551)."""
        total = self.n_samples
        first = shareRange(total, self._ntask, self._rank)[0]
        _end_sample_run(file_name_base, total, overwrite, self._comm, self._rank)
        for i, s in enumerate(self._s):
            _dump(_sample_file(file_name_base, first + i), _to_host(s), overwrite)

    @staticmethod
    def load(file_name_base, comm=None, device_id=-1):
        samples = []
        for name in _local_sample_files(file_name_base, comm):
            with open(name, "rb") as f:
                samples.append(pickle.load(f).at(device_id))
        domain = None
        if comm is not None and comm.size > 1:  # ranks without a sample still need the domain (sample_list.py:563-568)
            domain = comm.bcast_object(samples[0].domain if comm.rank == 0 else None, root=0)
        return SampleList(samples, comm, domain)


def _sample_file(file_name_base, number):
    return f"{file_name_base}.{int(number)}.pickle"


def _saved_sample_count(file_name_base):
    """How many samples ``<base>.0.pickle, <base>.1.pickle, ...`` hold: the length of the run of consecutive numbers
    that starts at 0 (sample_list.py:350-355, 663-670)."""
    import re

    directory, stem = os.path.split(os.path.abspath(file_name_base))
    pattern = re.compile(re.escape(stem) + r"\.([0-9]+)\.pickle$")
    present = {int(m.group(1)) for m in map(pattern.match, os.listdir(directory)) if m}
    if 0 not in present:
        raise RuntimeError(f"No files matching `{file_name_base}.*.pickle`")
    count = 1
    while count in present:
        count += 1
    return count


def _local_sample_files(file_name_base, comm):
    """The sample files of this rank under the shareRange distribution, all ranks looking at complete files."""
    if comm is not None:
        comm.barrier()
    ntask, rank, _ = get_MPI_params_from_comm(comm)
    names = [_sample_file(file_name_base, i) for i in range(*shareRange(_saved_sample_count(file_name_base), ntask, rank))]
    missing = [name for name in names if not os.path.isfile(name)]
    if missing:
        raise RuntimeError(f"File {missing[0]} not found")
    return names


def _end_sample_run(file_name_base, total, overwrite, comm, rank):
    """rank 0 removes a left-over ``<base>.<total>.pickle`` so that the files about to be written END the run; nobody
    writes before that happened."""
    nxt = _sample_file(file_name_base, total)
    clash = False
    if rank == 0:
        if overwrite:
            try:
                os.remove(nxt)
            except FileNotFoundError:
                pass
        else:  # the "next" sample of a longer, older run must not exist (sample_list.py:673-690)
            clash = os.path.isfile(nxt)
    if comm is not None:
        clash = bool(comm.bcast_object(clash if rank == 0 else None, root=0))  # every rank learns the verdict (and waits for it)
    if clash:
        raise RuntimeError(f"{nxt} already exists. You may want to remove it or specify overwrite=True")


def _to_host(f):
    return f.at(-1)


def _as_function(op):
    """What is applied to every sample: nothing, a callable, or an Operator -- which sees the part of the sample that
    lies in its domain (`force`), so operators on a sub-domain of the latent space work (sample_list.py:571-581)."""
    if op is None:
        return lambda x: x
    return op.force if isinstance(op, Operator) else op


class _Flat:
    """One term of a distributed sum -- a float, Field, MultiField or a tuple of those -- as a list of tensors that a
    collective may overwrite: private copies of the field values, python floats as one-element fp64 tensors where the
    backend wants them."""

    def __init__(self, term, comm):
        self._parts = []  # ("number", None) or ("field", private copy)
        self._tuple = isinstance(term, tuple)
        self.tensors = []
        for o in (term if self._tuple else (term,)):
            if isinstance(o, float):
                self._parts.append(None)
                self.tensors.append(torch.tensor([o], dtype=torch.float64, device=comm.scalar_device()))
            else:
                mine = o * 1.0
                self._parts.append(mine)
                self.tensors += [f.val for f in (mine.values() if isinstance(mine, MultiField) else [mine])]

    def rebuilt(self, tensors):
        """The term again once `tensors` (this object's own, in order) hold the result."""
        if any(a is not b for a, b in zip(tensors, self.tensors)):
            raise RuntimeError("_Flat.rebuilt: foreign tensors")
        numbers = iter(float(t.item()) for t, p in zip(self._number_slots(), [p for p in self._parts if p is None]))
        out = tuple(next(numbers) if p is None else p for p in self._parts)
        return out if self._tuple else out[0]

    def _number_slots(self):
        slots, at = [], 0
        for p in self._parts:
            if p is None:
                slots.append(self.tensors[at])
                at += 1
            else:
                at += len(p.values()) if isinstance(p, MultiField) else 1
        return slots


def _map(obj, fn):
    return tuple(fn(o) for o in obj) if isinstance(obj, tuple) else fn(obj)


def _add(a, b):
    return tuple(x + y for x, y in zip(a, b)) if isinstance(a, tuple) else a + b


def _zeros_like(o):
    """A zero Field / MultiField on the domain, dtype and device of `o` (not `o * 0`: that keeps NaN / inf)."""
    if isinstance(o, MultiField):
        return MultiField.from_dict({k: _zeros_like(f) for k, f in o.items()}, o.domain)
    return Field(o.domain, torch.zeros_like(o.val))


def _zero_key(o):
    """What a cached zero term must share with the object it stands in for: domain, dtype and device of every part."""
    if isinstance(o, float):
        return float
    parts = o.values() if isinstance(o, MultiField) else [o]
    return (id(o.domain),) + tuple((str(f.val.dtype), str(f.val.device)) for f in parts)


def _zero_like_host(obj):
    return _map(obj, lambda o: 0.0 if isinstance(o, float) else _to_host(o) * 0.0)


def _place_like(template, device_id):
    return _map(template, lambda o: o if isinstance(o, float) else o.at(device_id))


def _dump(fname, obj, overwrite):
    if os.path.isfile(fname) and not overwrite:
        raise RuntimeError(f"{fname} already exists")
    with open(fname, "wb") as f:
        pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)


# ------------------------------------------------------------------------------------------------
# sampling
# ------------------------------------------------------------------------------------------------
class _LinearSampler:
    """Draws (b, y) with y = M^-1 b for the metric M of the sampling problem at `position`: the Hamiltonian's own
    metric (MGVI), or 1 + J_f^T J_f of the likelihood's Gaussianising transformation f (geoVI), which also provides
    the coordinate transformation g(x) = x + J_f(p)^T f(x) and g(p) for the non-linear fit (kl_energies.py:105-128)."""

    def __init__(self, ham, position, geometric, napprox, device_id):
        self.position, self.device_id = position, device_id
        self.g = self.g_of_position = None
        if geometric:
            tr = ham.likelihood_energy.get_transformation()
            if tr is None:
                raise ValueError("Geometric sampling only works for likelihoods")
            dtype, f = tr
            f_lin = f(Linearization.make_var(position))
            jac = f_lin.jac
            self.g = ScalingOperator(f.domain, 1.0) + jac.adjoint @ f
            self.g_of_position = position + jac.adjoint(f_lin.val)
            self.metric = SamplingEnabler(SandwichOperator.make(jac, ScalingOperator(f.target, 1.0, dtype)),
                                          ScalingOperator(f_lin.domain, 1.0, float), ham.iteration_controller)
        else:
            self.metric = ham(Linearization.make_var(position, want_metric=True)).metric
        if napprox >= 1:
            # sampled diagonal of the metric as the preconditioner of the sampling solves (kl_energies.py:127-128); its
            # draws come from the CURRENT stream, before the per-sample seeds are spawned -- the reference's RNG order
            from .operators import makeOp
            from .probing import approximation2endo

            self.metric._approximation = makeOp(approximation2endo(self.metric, napprox, device_id))

    def draw(self, _seed=None):
        return self.metric.special_draw_sample(True, device_id=self.device_id)


def draw_samples(position, H, minimizer, n_samples, mirror_samples, napprox=0, want_error=False, comm=None,
                 device_id=-1):
    """MGVI (minimizer None) or geoVI residual samples around ``position`` (kl_energies.py:91-159), as a
    ResidualSampleList distributed over ``comm`` by a parallel.SamplePlan (ranks may end up without samples)."""
    for arg, kind in ((n_samples, int), (mirror_samples, bool), (H, StandardHamiltonian)):
        if not isinstance(arg, kind):
            raise TypeError
    at = position.extract(H.domain) if isinstance(position, MultiField) else position
    sampler = _LinearSampler(H, at, minimizer is not None, napprox, device_id)
    plan = SamplePlan(n_samples, mirror_samples, comm)
    plan.check_synchronised()

    def linear_residual(pair, mirrored):
        return pair[1], mirrored

    def fitted_residual(pair, mirrored):
        # geoVI: fit x to  g(x) = g(p) +- b  starting at  p +- y; the fit's displacement is stored unmirrored
        b, y = pair
        target = sampler.g_of_position - b if mirrored else sampler.g_of_position + b
        start = at - y if mirrored else at + y
        fit, _ = minimizer(EnergyAdapter(start, GaussianEnergy(target) @ sampler.g, nanisinf=True, want_metric=True))
        return fit.position - at, False

    drawn = plan.run(sampler.draw, linear_residual if minimizer is None else fitted_residual)
    return ResidualSampleList(position, [r for r, _ in drawn], [n for _, n in drawn], comm)


def _checked_key_sets(position, constants, point_estimates):
    """(constants, point estimates) as sets, checked against the keys of the latent space"""
    frozen, estimated = set(constants), set(point_estimates)
    if isinstance(position, MultiField):
        keys = set(position.keys())
        for what, subset, given in (("Constants", frozen, constants), ("Point estimates", estimated, point_estimates)):
            if not subset <= keys:
                raise ValueError(f"{what} are not a subset of the keys of the latent space: {given}")
        if estimated == keys:
            raise RuntimeError("Point estimates for whole domain. Use EnergyAdapter instead.")
    return frozen, estimated


def SampledKLEnergy(position, hamiltonian, n_samples, minimizer_sampling, mirror_samples=True, constants=[],
                    point_estimates=[], napprox=0, comm=None, nanisinf=True, device_id=-1):
    """Draw samples at ``position`` and return the sampled KL energy (kl_energies.py:162-296)."""
    expected = {"hamiltonian": (hamiltonian, StandardHamiltonian), "n_samples": (n_samples, int),
                "mirror_samples": (mirror_samples, bool), "minimizer_sampling": (minimizer_sampling, (DescentMinimizer, type(None)))}
    wrong = [name for name, (arg, kind) in expected.items() if not isinstance(arg, kind)]
    if wrong:
        raise TypeError(f"invalid type of: {', '.join(wrong)}")
    if hamiltonian.domain is not position.domain:
        raise ValueError
    frozen, estimated = _checked_key_sets(position, constants, point_estimates)
    # keys that are constant AND point estimates leave the problem entirely (kl_energies.py:281-287) ...
    gone = sorted(frozen & estimated)
    left_out = position.extract_by_keys(gone) if (gone and isinstance(position, MultiField)) else None
    position, hamiltonian = _reduce_by_keys(position, hamiltonian, gone)
    # ... and nothing is sampled along the point estimates: they are inserted into the sampling Hamiltonian (:289-293)
    sampling_hamiltonian = _reduce_by_keys(position, hamiltonian, point_estimates)[1]
    drawn = draw_samples(position, sampling_hamiltonian, minimizer_sampling, n_samples, mirror_samples, napprox=napprox,
                         comm=comm, device_id=device_id)
    return SampledKLEnergyClass(drawn, hamiltonian, constants, left_out, nanisinf)


def _reduce_field(field, keys):
    if isinstance(field, MultiField) and len(keys) > 0:
        return field.extract_by_keys(set(field.keys()) - set(keys))
    return field


def _reduce_by_keys(field, operator, keys):
    """(variable part of the field, operator with the constant part inserted) (kl_energies.py:49-76)."""
    keys = set(keys)
    if not isinstance(field, MultiField):
        if keys:
            raise ValueError("constants need a MultiField position")
        return field, operator
    frozen = operator.simplify_for_constant_input(field.extract_by_keys(keys))[1]
    return field.extract_by_keys(set(field.keys()) - keys), frozen


class SampledKLEnergyClass(Energy):
    """KL(p) = 1/S sum_s H(p +/- r_s)  (kl_energies.py:299-360).  The Hamiltonian is linearised ONCE per local sample
    when the energy is built; value, gradient and every later metric application reuse those linearisations, and the
    sums over samples go through SampleListBase._sum_over_ranks (ranks without samples contribute zeros)."""

    def __init__(self, sample_list, hamiltonian, constants, invariants, nanisinf):
        if not isinstance(sample_list, ResidualSampleList):
            raise TypeError
        if sample_list.domain is not hamiltonian.domain:
            raise ValueError("domain mismatch")
        super().__init__(_reduce_field(sample_list._m, constants))
        self._sample_list, self._hamiltonian = sample_list, hamiltonian
        self._constants, self._invariants, self._nanisinf = constants, invariants, bool(nanisinf)

        def linearised(sample):  # the constant keys of THIS sample are inserted, the rest is differentiated (:318-321)
            variable, ham = _reduce_by_keys(sample, hamiltonian, constants)
            return ham(Linearization.make_var(variable, want_metric=True))

        self._lins = [linearised(sample) for sample in sample_list.local_iterator()]
        weight = 1.0 / sample_list.n_samples
        value, gradient = sample_list._sum_over_ranks(((_scalar_value(lin.val), lin.gradient) for lin in self._lins),
                                                      like=(0.0, self._position))
        self._val, self._grad = _nan_to_inf(value * weight, self._nanisinf), gradient * weight

    @property
    def value(self):
        return self._val

    @property
    def gradient(self):
        return self._grad

    def at(self, position):
        return SampledKLEnergyClass(self._sample_list.at(position), self._hamiltonian, self._constants,
                                    self._invariants, self._nanisinf)

    def apply_metric(self, x):
        total = self._sample_list._sum_over_ranks((lin.metric(x) for lin in self._lins), like=x)
        return total * (1.0 / self._sample_list.n_samples)

    @property
    def metric(self):
        return _SelfAdjointOperatorWrapper(self.position.domain, self.apply_metric)

    @property
    def samples(self):
        if self._invariants is None:
            return self._sample_list
        return self._sample_list.at(self._invariants)
