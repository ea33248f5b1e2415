"""Convenience functions of the reference's top level (nifty/cl/sugar.py) that are built from the operators of this package:
power-spectrum helpers, codomain lookup, the pointwise functions as free functions, a pre-image finder and a timing helper.
Plotting and the operator-tree profiler are out of scope."""
import time

import numpy as np

from .domains import DomainTuple, PowerSpace, StructuredDomain
from .field import Field, MultiField, from_random
from .minimization import logger
from .operators import DiagonalOperator, Linearization, Operator, PowerDistributor, ScalingOperator, _space_index

POINTWISE_NAMES = ("sqrt", "sin", "cos", "tan", "sinc", "exp", "expm1", "log", "log10", "log1p", "sinh", "cosh", "tanh",
                   "sigmoid", "reciprocal", "abs", "absolute", "sign", "power", "clip", "softplus", "exponentiate", "arctan",
                   "unitstep")


def _free_function(name):
    def fn(x, *args, **kwargs):
        return x.ptw(name, *args, **kwargs)

    fn.__name__ = name
    fn.__doc__ = f"x.ptw({name!r}, ...) for a Field, MultiField, Linearization or Operator (sugar.py:474-486)"
    return fn


for _name in POINTWISE_NAMES:
    globals()[_name] = _free_function(_name)


def PS_field(pspace, function, device_id=-1):
    """`function` evaluated at the bin centres |k| of a PowerSpace, as a Field on it (sugar.py:54-73)"""
    if not isinstance(pspace, PowerSpace):
        raise TypeError("PowerSpace expected")
    return Field.from_raw(DomainTuple.make(pspace), np.asarray(function(pspace.k_lengths))).at(device_id)


def _spectrum_on_grid(grid, power_spectrum):
    """The spectrum (a callable of |k| or a Field on a PowerSpace of `grid`) spread over the harmonic grid"""
    if callable(power_spectrum):
        bins = PowerSpace(grid)
        values = PS_field(bins, power_spectrum)
    else:
        if not isinstance(power_spectrum, Field):
            raise TypeError("Field object expected")
        if len(power_spectrum.domain) != 1:
            raise ValueError("exactly one domain required")
        bins, values = power_spectrum.domain[0], power_spectrum
        if not isinstance(bins, PowerSpace):
            raise TypeError("PowerSpace required")
    return PowerDistributor(grid, bins)(values)


def get_signal_variance(spec, space):
    """Expected point variance of a field with power spectrum `spec` on the position-space partner of `space`: the spectrum
    integrated over the harmonic grid with the squared pixel volume (sugar.py:76-102)"""
    if isinstance(space, StructuredDomain) and space.harmonic:
        space = PowerSpace(space)
    if not isinstance(space, PowerSpace):
        raise ValueError("space must be either a harmonic space or Power space.")
    return _spectrum_on_grid(space.harmonic_partner, PS_field(space, spec)).weight(2).s_sum()


def create_power_operator(domain, power_spectrum, space=None, sampling_dtype=None):
    """DiagonalOperator with the given spectrum along sub-domain `space` of `domain` (sugar.py:200-227)"""
    domain = DomainTuple.make(domain)
    space = _space_index(domain, space)
    return DiagonalOperator(_spectrum_on_grid(domain[space], power_spectrum), domain, space, sampling_dtype)


def create_harmonic_smoothing_operator(domain, space, sigma):
    """The Fourier-space factor of a Gaussian smoothing of width `sigma` along `space` (sugar.py:290-310)"""
    domain = DomainTuple.make(domain)
    kernel = domain[space].get_fft_smoothing_kernel_function(sigma)
    return DiagonalOperator(kernel(domain[space].get_k_length_array()), domain, space)


def power_analyze(field, spaces=None, binbounds=None, keep_phase_information=False):
    """Bin-averaged |field|^2 along the harmonic sub-domains `spaces` (sugar.py:113-180): every analysed sub-domain is replaced
    by its PowerSpace.  `keep_phase_information`: real and imaginary parts analysed separately, returned as re + i im."""
    for sp in field.domain:
        if not sp.harmonic and not isinstance(sp, PowerSpace):
            logger.warning("WARNING: Field has a space in `domain` which is neither harmonic nor a PowerSpace.")
    chosen = field.domain._chosen(spaces)
    if not chosen:
        raise ValueError("No space for analysis specified.")
    is_complex = field.val.is_complex()
    if keep_phase_information:
        if not is_complex:
            raise ValueError("cannot keep phase from real-valued input Field")
        parts = [field.real * field.real, field.imag * field.imag]
    elif is_complex:
        parts = [field.real * field.real + field.imag * field.imag]
    else:
        parts = [field * field]

    def binned(part, idx):
        spread = PowerDistributor(part.domain, PowerSpace(part.domain[idx], binbounds), idx)
        return spread.adjoint_times(part.weight(1, spaces=idx)).weight(-1, spaces=idx)

    for idx in chosen:
        parts = [binned(part, idx) for part in parts]
    return parts[0] + 1j * parts[1] if keep_phase_information else parts[0]


def get_default_codomain(domainoid, space=None):
    """The harmonic partner of a space, or the DomainTuple with sub-domain `space` replaced by its partner (sugar.py:489-518)"""
    if isinstance(domainoid, StructuredDomain) and not isinstance(domainoid, PowerSpace):
        return domainoid.get_default_codomain()
    if not isinstance(domainoid, DomainTuple):
        raise TypeError("Works only on RGSpaces and DomainTuples containing those")
    space = _space_index(domainoid, space)
    if not isinstance(domainoid[space], StructuredDomain):
        raise TypeError("can only codomain structrued spaces")
    parts = list(domainoid)
    parts[space] = domainoid[space].get_default_codomain()
    return DomainTuple.make(parts)


def calculate_position(operator, output):
    """An approximate pre-image of `output` under `operator` (sugar.py:564-603): three short MGVI rounds on a Gaussian
    likelihood around the slightly noised output, started near zero."""
    from .energy_operators import GaussianEnergy, StandardHamiltonian
    from .kl import SampledKLEnergy
    from .minimization import GradientNormController, NewtonCG

    if not isinstance(operator, Operator):
        raise TypeError("operator expected")
    if output.domain != operator.target:
        raise TypeError("output does not live on the operator's target")
    if isinstance(output, MultiField):
        dtypes = {f.dtype for f in output.values()}
        if len(dtypes) != 1:
            raise ValueError("Only MultiFields with one dtype supported.")
        dtype, peak = dtypes.pop(), max(np.max(np.abs(v)) for v in output.asnumpy().values())
    else:
        dtype, peak = output.dtype, np.max(np.abs(output.asnumpy()))
    noise = ScalingOperator(output.domain, 1e-3 * peak ** 2, dtype)
    data = output + noise.draw_sample()
    pos = 0.1 * from_random(operator.domain)
    hamiltonian = StandardHamiltonian(GaussianEnergy(data, noise.inverse) @ operator,
                                      ic_samp=GradientNormController(iteration_limit=200), prior_sampling_dtype=pos.dtype)
    minimizer = NewtonCG(GradientNormController(iteration_limit=10, name="findpos"))
    for it in range(3):
        logger.info(f"Start iteration {it + 1}/3")
        kl, _ = minimizer(SampledKLEnergy(pos, hamiltonian, 3, None))
        pos = kl.position
    return pos


class _Stopwatch:
    """Times callables: three untimed calls, then `ntries` timed ones with the device synchronised after each; optionally a
    cProfile of the timed calls (logged / dumped)."""

    def __init__(self, ntries, device_id, verbose, dump_prefix):
        self.ntries, self.device_id, self.verbose, self.dump_prefix = ntries, device_id, verbose, dump_prefix
        self.seconds = {}

    def _sync(self):
        if self.device_id > -1:
            import torch

            torch.cuda.synchronize(self.device_id)

    def __call__(self, key, label, call):
        import cProfile
        import io
        import pstats

        for _ in range(3):
            out = call()
        self._sync()
        profile = cProfile.Profile() if (self.verbose or self.dump_prefix is not None) else None
        start = time.time()
        if profile is not None:
            profile.enable()
        for _ in range(self.ntries):
            out = call()
            self._sync()
        if profile is not None:
            profile.disable()
        self.seconds[key] = (time.time() - start) / self.ntries
        logger.info(f"{label.ljust(33)}: {self.seconds[key] * 1000:>8.3f} ms")
        if self.verbose:
            text = io.StringIO()
            pstats.Stats(profile, stream=text).sort_stats(pstats.SortKey.TIME).print_stats(5)
            logger.info(text.getvalue())
        if self.dump_prefix is not None:
            profile.dump_stats(f"{self.dump_prefix}_{key}.prof")
        return out


def exec_time(obj, want_metric=True, verbose=False, domain_dtype=np.float64, ntries=1, device_id=-1, dump_prefix=None):
    """Wall-clock seconds per call of the pieces of an Operator (value, linearisation, Jacobian, adjoint, gradient, metric) or
    of an Energy (at, value, gradient, metric), logged and returned as a dict under the reference's keys (sugar.py:606-693)."""
    from .minimization import Energy

    watch = _Stopwatch(ntries, device_id, verbose, dump_prefix)
    if isinstance(obj, Energy):
        nearby = 0.99 * obj.position
        pieces = [("energy.at", "Energy.at()", lambda: obj.at(nearby)), ("value", "Energy.value", lambda: obj.value),
                  ("gradient", "Energy.gradient", lambda: obj.gradient), ("metric", "Energy.metric", lambda: obj.metric)]
        if obj.metric is not None:
            pieces += [("apply_metric", "Energy.apply_metric", lambda: obj.apply_metric(obj.position)),
                       ("metric()", "Energy.metric(position)", lambda: obj.metric(obj.position))]
        for piece in pieces:
            watch(*piece)
    elif isinstance(obj, Operator):
        pos = from_random(obj.domain, "normal", dtype=domain_dtype, device_id=device_id)
        at_pos = Linearization.make_var(pos, want_metric=bool(want_metric))
        watch("apply", "Operator call with field", lambda: obj(pos))
        lin = watch("apply_lin", "Operator call with linearization", lambda: obj(at_pos))
        watch("jac", "Apply linearization", lambda: lin.jac(pos))
        watch("jac.adjoint", "Apply linearization (adjoint)", lambda: lin.jac.adjoint(lin.val))
        if obj.target is DomainTuple.scalar_domain():
            watch("gradient", "Gradient evaluation", lambda: lin.gradient)
            if want_metric:
                watch("metric_apply", "Metric apply", lambda: lin.metric(pos))
    else:
        raise TypeError("Operator or Energy expected")
    return watch.seconds
