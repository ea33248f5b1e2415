"""Run a script written for the reference unchanged: ``import nifty_amd.compat; nifty_amd.compat.install()`` (before the
script's own imports, e.g. in ``sitecustomize`` or a pytest plugin) makes ``import nifty.cl as ift`` resolve to this package
and the reference's sub-module paths -- ``from nifty.cl.minimization.kl_energies import SampledKLEnergyClass``,
``from nifty.cl.library.correlated_fields import CorrelatedFieldMaker``, ``nifty.cl.operators.harmonic_operators`` ... --
resolve to the module of this package that holds the same names (this package has few, larger modules: the map below).
A name the reference defines and this package does not raises AttributeError / ImportError at the import that asks for
it, never silently later.  tools/run_reference_tests.py runs the reference's own test files this way."""
import importlib
import sys
import types

# reference module path (below nifty.cl) -> module of this package (below nifty_amd) holding those names
LAYOUT = {
    "field": "field", "multi_field": "field", "linearization": "operators", "sugar": "sugar", "utilities": "utilities",
    "random": "random", "extra": "extra", "probing": "probing", "logger": "minimization", "pointwise": "sugar", "plot": "plot",
    "domain_tuple": "domains", "multi_domain": "domains",
    "domains": "domains", "domains.domain": "domains", "domains.structured_domain": "domains",
    "domains.unstructured_domain": "domains", "domains.rg_space": "domains", "domains.power_space": "domains",
    "domains.lm_space": "domains", "domains.hp_space": "domains", "domains.gl_space": "domains",
    "operators": "operators",
    "minimization": "minimization", "minimization.conjugate_gradient": "minimization",
    "minimization.descent_minimizers": "minimization", "minimization.energy": "minimization",
    "minimization.iteration_controllers": "minimization", "minimization.line_search": "minimization",
    "minimization.minimizer": "minimization", "minimization.quadratic_energy": "minimization",
    "minimization.energy_adapter": "kl", "minimization.kl_energies": "kl", "minimization.sample_list": "kl",
    "minimization.optimize_kl": "optimize_kl",
    "library": "correlated_fields", "library.correlated_fields": "correlated_fields",
    "library.correlated_fields_simple": "correlated_fields", "library.los_response": "los_response",
    "library.wiener_filter_curvature": "operators",
}
# every file of the reference's operators/ package whose classes live in operators.py / energy_operators.py here
for _name in ("adder", "block_diagonal_operator", "chain_operator", "contraction_operator", "counting_operator",
              "diagonal_operator", "distributors", "endomorphic_operator", "harmonic_operators", "inversion_enabler",
              "linear_operator", "mask_operator", "operator", "operator_adapter", "sampling_enabler", "sandwich_operator",
              "scaling_operator", "simple_linear_operators", "simplify_for_const", "sum_operator"):
    LAYOUT["operators." + _name] = "operators"
LAYOUT["operators.energy_operators"] = "energy_operators"
for _name in ("selection_operators", "transpose_operator", "outer_product_operator", "value_inserter", "field_zero_padder",
              "domain_tuple_field_inserter"):
    LAYOUT["operators." + _name] = "selection_operators"
LAYOUT["operators.normal_operators"] = "correlated_fields"


class _Alias(types.ModuleType):
    """A module object under the reference's name whose attributes are those of one module of this package"""

    def __init__(self, name, backing, is_package):
        super().__init__(name, f"nifty_amd.compat alias of {backing.__name__}")
        self.__dict__["_backing"] = backing
        if is_package:
            self.__path__ = []

    def __getattr__(self, attr):
        try:
            return getattr(self.__dict__["_backing"], attr)
        except AttributeError:
            raise AttributeError(f"{self.__name__}.{attr}: not provided by nifty_amd "
                                 f"(looked in {self.__dict__['_backing'].__name__})") from None


def _anyarray_conveniences():
    """`Field.val` is a torch tensor here, an `AnyArray` in the reference (any_array.py): give tensors the three AnyArray
    methods scripts call on it -- `asnumpy()`, `device_id`, `at(device_id)` -- and numpy's function protocol, so that
    `np.mean(field.val)` works (on a host copy).  Opt-in: only `install()` touches torch.Tensor."""
    import torch

    from .field import device_of, torch_dtype

    if hasattr(torch.Tensor, "asnumpy"):
        return
    torch.Tensor.asnumpy = lambda self: self.detach().cpu().numpy()
    torch.Tensor.device_id = property(lambda self: self.device.index if self.is_cuda else -1)
    torch.Tensor.at = lambda self, device_id: self.to(device_of(device_id))
    torch.Tensor.astype = lambda self, dtype: self.to(torch_dtype(dtype))

    def as_arrays(obj):
        if isinstance(obj, torch.Tensor):
            return obj.detach().cpu().numpy()
        if isinstance(obj, (list, tuple)):
            return type(obj)(as_arrays(o) for o in obj)
        if isinstance(obj, dict):
            return {k: as_arrays(v) for k, v in obj.items()}
        return obj

    def array_function(self, func, types, args, kwargs):
        """numpy FUNCTIONS on `field.val` (`np.mean(f.val)`, `np.nansum(...)`): evaluated by numpy on host copies, returning
        numpy results -- what scripts written against the reference's AnyArray expect from diagnostics"""
        return func(*as_arrays(args), **as_arrays(kwargs))

    torch.Tensor.__array_function__ = array_function


def install(top="nifty"):
    """Register the aliases in sys.modules (idempotent).  Returns the nifty_amd package."""
    import nifty_amd

    _anyarray_conveniences()

    if top in sys.modules and getattr(sys.modules[top], "_nifty_amd_alias", False):
        return nifty_amd
    if top in sys.modules:
        raise RuntimeError(f"a module named {top!r} is already imported; install the alias before it")
    root = types.ModuleType(top, "alias package created by nifty_amd.compat.install()")
    root.__path__ = []
    root._nifty_amd_alias = True
    root.cl = nifty_amd
    sys.modules[top] = root
    sys.modules[top + ".cl"] = nifty_amd
    packages = {path.rsplit(".", 1)[0] for path in LAYOUT if "." in path}
    for path, backing in LAYOUT.items():
        module = importlib.import_module("nifty_amd." + backing)
        alias = _Alias(f"{top}.cl.{path}", module, path in packages)
        sys.modules[alias.__name__] = alias
    for path in LAYOUT:  # children as attributes of their parents, so that `nifty.cl.operators.operator` also works by attribute
        if "." in path:
            parent, child = path.rsplit(".", 1)
            sys.modules[f"{top}.cl.{parent}"].__dict__[child] = sys.modules[f"{top}.cl.{path}"]
    return nifty_amd
