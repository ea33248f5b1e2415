"""nifty_amd -- MI355X-native implementation of the nifty.cl MGVI/geoVI hot path.

``import nifty_amd as ift`` exposes the names of ``nifty.cl`` that the path needs (domains, Field /
MultiField, the operator algebra, energies, minimizers, CorrelatedFieldMaker, SampledKLEnergy,
optimize_kl).  Host code is Python on PyTorch-ROCm tensors; all numerics on device Fields run in
hand-written HIP kernels (libniftyk, C ABI in include/niftyk.h) -- there is no CPU or eager-PyTorch
fallback for device data.
"""
__version__ = "0.1.0"

from . import config, extra, parallel, random  # noqa: F401
from .correlated_fields import (CorrelatedFieldMaker, CorrelatedFieldOperator, LognormalTransform,  # noqa: F401
                                NormalTransform, SimpleCorrelatedField)
from .domains import (DomainTuple, GLSpace, HPSpace, LMSpace, MultiDomain, PowerSpace, RGSpace, StructuredDomain, UnstructuredDomain,  # noqa: F401
                      makeDomain)
from .energy_operators import (AveragedEnergy, BernoulliEnergy, CategoricalEnergy, EnergyOperator, GaussianEnergy, InverseGammaEnergy, LikelihoodEnergyOperator, PoissonianEnergy,  # noqa: F401
                               QuadraticFormOperator, Squared2NormOperator, StandardHamiltonian, StudentTEnergy,
                               VariableCovarianceGaussianEnergy)
from .field import AnyArray, Field, MultiField, device_available, from_random, full, is_fieldlike, makeField  # noqa: F401
from .kl import (EnergyAdapter, ResidualSampleList, SampledKLEnergy, SampledKLEnergyClass, SampleList,  # noqa: F401
                 draw_samples)
from .minimization import (AbsDeltaEnergyController, ConjugateGradient, DeltaEnergyController, DescentMinimizer,  # noqa: F401
                           Energy, EnergyHistory, GradientNormController, GradInfNormController, IterationController,
                           L_BFGS, LineSearch, Minimizer, NewtonCG, QuadraticEnergy, RelaxedNewton, SteepestDescent,
                           StochasticAbsDeltaEnergyController, VL_BFGS)
from .probing import StatCalculator, approximation2endo  # noqa: F401
from .operators import PrependKey  # noqa: F401
from .operators import (Adder, BlockDiagonalOperator, ChainOperator, ContractionOperator, CountingOperator,  # noqa: F401
                        DiagonalOperator, HarmonicSmoothingOperator, IntegrationOperator,
                        DOFDistributor, EndomorphicOperator, FFTOperator, FFTShiftOperator, FieldAdapter,
                        HarmonicTransformOperator, HartleyOperator, InversionEnabler, Linearization, LinearOperator, MaskOperator, NullOperator, Operator, OperatorAdapter,
                        PowerDistributor, Realizer, SamplingEnabler, SandwichOperator, ScalingOperator, SumOperator,
                        Variable, VdotOperator, WienerFilterCurvature, ducktape, makeOp)
from .los_response import LOSResponse  # noqa: F401
from .optimize_kl import optimize_kl  # noqa: F401
from .parallel import shareRange  # noqa: F401
from . import utilities  # noqa: E402,F401
from . import sugar  # noqa: E402,F401
from . import plot  # noqa: E402,F401
from .plot import Plot, plot_priorsamples, single_plot  # noqa: E402,F401
from .sugar import (PS_field, calculate_position, create_harmonic_smoothing_operator, create_power_operator, exec_time,  # noqa: E402,F401
                    get_default_codomain, get_signal_variance, power_analyze)
from .sugar import (abs, absolute, arctan, clip, cos, cosh, exp, expm1, exponentiate, log, log10, log1p, power, reciprocal,  # noqa: E402,F401,A004
                    sigmoid, sign, sin, sinc, sinh, softplus, sqrt, tan, tanh, unitstep)
from .domains import Domain  # noqa: E402,F401
from .selection_operators import (ConjugationOperator, DomainChangerAndReshaper, DomainTupleFieldInserter, ExtractAtIndices,  # noqa: E402,F401
                                  FieldZeroPadder, GeometryRemover, Imaginizer, OuterProduct, PartialExtractor, SliceOperator,
                                  SplitOperator, SqueezeOperator, TransposeOperator, ValueInserter)
from . import correlated_fields as _cf, energy_operators as _eo, los_response as _los, operators as _ops  # noqa: E402
import types as _types  # noqa: E402

# the reference's `ift.library.<module>` / `ift.operators.energy_operators` attribute paths (compat.py has the import paths)
library = _types.SimpleNamespace(correlated_fields=_cf, correlated_fields_simple=_cf, los_response=_los)
_ops.energy_operators = _eo
from . import kl as _kl, minimization as _min  # noqa: E402
for _name in ("conjugate_gradient", "descent_minimizers", "energy", "iteration_controllers", "line_search", "minimizer",
              "quadratic_energy"):
    setattr(_min, _name, _min)
for _name in ("kl_energies", "sample_list", "energy_adapter"):
    setattr(_min, _name, _kl)
from . import selection_operators as _sel  # noqa: E402
for _name in ("selection_operators", "transpose_operator", "outer_product_operator", "value_inserter", "field_zero_padder",
              "domain_tuple_field_inserter", "simple_linear_operators"):
    setattr(_ops, _name, _sel if _name != "simple_linear_operators" else _ops)
from .operators import domain_union  # noqa: E402,F401


_nthreads = 1


def set_nthreads(nthr):
    """Accepted for source compatibility with nifty.cl (ducc_dispatch.py:35-46): remembered for `nthreads()`, ignored by the
    device kernels."""
    global _nthreads
    _nthreads = int(nthr)


def nthreads():
    """ducc_dispatch.py:31-32."""
    return _nthreads


# ---- small helpers of the reference's top level that user scripts lean on ------------------------------------------
from .minimization import logger  # noqa: E402,F401  (one logger for the package; the reference's is "NIFTy", logger.py:21-33)


def logger_init(level=None):
    """logger.py:21-31: the package logger with a stream handler at `level` (default INFO), not propagating."""
    import logging

    level = logging.INFO if level is None else level
    logger.setLevel(level)
    logger.propagate = False
    if not any(isinstance(h, logging.StreamHandler) for h in logger.handlers):
        handler = logging.StreamHandler()
        logger.addHandler(handler)
    for h in logger.handlers:
        h.setLevel(level)
    return logger


def myassert(val):
    """utilities.py:516-520: an assertion that stays active under ``python -O``."""
    if not val:
        raise AssertionError


def is_operator(obj):
    """operators/operator.py:659-668: operator-like and neither a field nor a linearization (here those are separate
    classes; in the reference they derive from Operator, hence the helper)."""
    return isinstance(obj, Operator) and not isinstance(obj, Linearization) and not is_fieldlike(obj)


def is_linearization(obj):
    """operators/operator.py:671-673."""
    return isinstance(obj, Linearization)


def is_likelihood_energy(obj):
    """operators/operator.py:653-656: an operator that knows its variance-stabilising transformation."""
    get = getattr(obj, "get_transformation", None)
    return isinstance(obj, Operator) and callable(get) and get() is not None


from .kl import SampleListBase  # noqa: E402,F401
from .field import is_fieldlike as _is_field_or_multifield  # noqa: E402


def is_fieldlike(obj):  # noqa: F811
    """operators/operator.py:676-684: Fields, MultiFields and Linearizations -- what an operator can be applied to and that
    carries a value (inside the package `field.is_fieldlike` keeps meaning Field / MultiField)."""
    return _is_field_or_multifield(obj) or isinstance(obj, Linearization)
