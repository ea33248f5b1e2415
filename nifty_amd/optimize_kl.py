"""optimize_kl: the MGVI / geoVI driver loop.

Counterpart of reference nifty/cl/minimization/optimize_kl.py:51-453 (same keyword surface).  Per
global iteration: push the iteration's SeedSequence, initialise missing latent keys with 0.1*N(0,1),
draw samples, minimise the sampled KL, optionally save / call back.

Fusion pass (``fuse=True``, device runs only): when the likelihood is
``GaussianEnergy | PoissonianEnergy  @  [[MaskOperator @] LOSResponse @]  [exp | sigmoid]  @  CorrelatedFieldOperator``
the iteration runs on the fused engine (engine.py: one forward + one adjoint transform per sample and metric application,
CG with device-resident scalars) and the results are handed back as MultiFields / ResidualSampleList.
Everything else walks the generic operator graph.

Files of a run (``output_directory``; reference optimize_kl.py:300-344, 425-441, 467-525, 571-615, 716-740):
``pickle/<latest|iteration_N>.*`` sample lists, ``pickle/nifty_random_state`` (the reference's layout),
``pickle/energy_history_*``, ``pickle/minisanity_history_*``, ``last_finished_iteration``, ``minisanity.txt``,
``counting_report.txt`` and ``<name>/<latest|iteration_N>.hdf5`` for every entry of ``export_operator_outputs``.
The PNG plots of the reference (energy / minisanity history) are diagnostics outside the path and not drawn
(SURVEY 2 #26); ``plot_*`` are accepted and ignored.
"""
import os
import pickle
from datetime import datetime
from inspect import signature
from warnings import warn

import numpy as np

from . import config, parallel, random
from .domains import DomainTuple, MultiDomain, makeDomain
from .energy_operators import GaussianEnergy, PoissonianEnergy, StandardHamiltonian, _LikelihoodChain
from .field import Field, MultiField, from_random, full
from .kl import EnergyAdapter, ResidualSampleList, SampledKLEnergy, SampleList
from .minimization import DescentMinimizer, EnergyHistory, IterationController, Minimizer, logger
from .los_response import LOSResponse, SparseResponse
from .operators import (ChainOperator, CountingOperator, DiagonalOperator, MaskOperator, Operator, ScalingOperator,
                        _FunctionApplier, _OpChain)
from .parallel import get_MPI_params_from_comm


def _make_callable(obj):
    if callable(obj) and not isinstance(obj, (Minimizer, IterationController, Operator)):
        return obj
    return lambda x: obj


def _nargs(func):
    return len(signature(func).parameters)


def _want_metric(minimizer):
    """first-order minimisers never ask for the metric (optimize_kl.py:769-775)"""
    from .minimization import L_BFGS, VL_BFGS, SteepestDescent

    return not isinstance(minimizer, (SteepestDescent, L_BFGS, VL_BFGS))


def _normal_initialize(mf, domain, std=0.1, device_id=-1):
    """Draw the keys of ``domain`` missing in ``mf`` (optimize_kl.py:792-805)."""
    if MultiDomain.union([domain, mf.domain]) != domain:
        raise RuntimeError("Domain of MultiField and final domain are not compatible")
    diff = makeDomain({k: domain[k] for k in domain.keys() if k not in mf.domain})
    if len(diff) == 0:
        return mf.extract(domain)
    fld = from_random(diff, std=std, device_id=device_id)
    res = mf.unite(fld) if len(mf.domain) else fld
    return res.extract(domain) if res.domain is not domain else res


# ------------------------------------------------------------------------------------------------
# fusion pass
# ------------------------------------------------------------------------------------------------
def match_fused(lh):
    """Return the FusedModel keyword arguments for a recognised likelihood graph, else None."""
    from .correlated_fields import CorrelatedFieldOperator

    if not isinstance(lh, _LikelihoodChain):
        return None
    like, model = lh.likelihood, lh.model
    nonlin = None
    response = None
    if isinstance(model, _OpChain):
        ops = list(model._ops)
        # leading linear part: [MaskOperator @] LOSResponse (BASELINE config 4, reference demos/cl/getting_started_3.py:98-100),
        # possibly merged into one ChainOperator -- it becomes ONE sparse matrix of the fused engine
        lin = []
        while ops and isinstance(ops[0], (MaskOperator, LOSResponse, ChainOperator)):
            op = ops.pop(0)
            lin.extend(op._ops if isinstance(op, ChainOperator) else [op])
        if lin:
            if len(lin) == 1 and isinstance(lin[0], LOSResponse):
                response = (lin[0], None)
            elif len(lin) == 2 and isinstance(lin[0], MaskOperator) and isinstance(lin[1], LOSResponse):
                response = (lin[1], lin[0])
            else:
                return None
        if len(ops) == 2 and isinstance(ops[0], _FunctionApplier) and ops[0]._funcname in ("exp", "sigmoid") \
                and not ops[0]._args and isinstance(ops[1], CorrelatedFieldOperator):
            nonlin, model = ops[0]._funcname, ops[1]
        elif len(ops) == 1 and isinstance(ops[0], CorrelatedFieldOperator) and response is not None:
            model = ops[0]
        else:
            return None
    if not isinstance(model, CorrelatedFieldOperator) or model._prefix != "":
        return None
    kw = dict(model.fused_parameters)
    kw.pop("prefix")
    kw["nonlin"] = nonlin
    if response is not None:
        if response[0].domain[0] is not model.target[0] and response[0].domain != model.target:
            return None
        kw["response"] = SparseResponse.from_operators(*response)
    if isinstance(like, PoissonianEnergy):
        kw.update(likelihood="poisson", data=like._d.val)
    elif isinstance(like, GaussianEnergy) and like._data is not None:
        icov = like._icov
        if isinstance(icov, ScalingOperator) and np.isreal(icov._factor):
            kw.update(likelihood="gaussian", data=like._data.val, icov=float(np.real(icov._factor)))
        elif isinstance(icov, DiagonalOperator) and icov._full() and icov._trafo == 0 and not icov._complex:
            kw.update(likelihood="gaussian", data=like._data.val, icov=icov._ldiag)
        else:
            return None
    else:
        return None
    return kw


_fused_cache = {}  # at most _FUSED_CACHE_MAX entries: an entry pins its likelihood and the model's device buffers
_FUSED_CACHE_MAX = 2


def _fused_model(lh, device_id, dtype):
    import torch

    from . import backend as B
    from .engine import FusedModel

    dtype = np.dtype(dtype)  # (np.float32 and dtype('float32') must find the same model)
    key = (id(lh), device_id, dtype)
    if key not in _fused_cache:
        while len(_fused_cache) >= _FUSED_CACHE_MAX:  # oldest first (dicts keep insertion order)
            _fused_cache.pop(next(iter(_fused_cache)))
        kw = match_fused(lh)
        tdt = torch.float32 if np.dtype(dtype) == np.float32 else torch.float64
        if kw is not None and not B.plan_supported(kw["shape"], tdt, 1, f"cuda:{device_id}"):
            kw = None  # grid outside the native planner's lengths: generic operator graph
        if kw is None:
            _fused_cache[key] = None
        else:
            _fused_cache[key] = (lh, FusedModel(kw.pop("shape"), kw.pop("distances"), dtype=tdt,
                                                device=f"cuda:{device_id}", **kw))
    ent = _fused_cache[key]
    return None if ent is None else ent[1]


def _mf_to_latent(model, mf):
    import torch

    from .engine import SMALL_KEYS, LatentVec

    small = torch.cat([mf[k].val.reshape(1).to(torch.float64) for k in SMALL_KEYS] +
                      [mf["spectrum"].val.reshape(-1).to(torch.float64)]).to(model.device).contiguous()
    return LatentVec(mf["xi"].val.to(model.device).to(model.tdtype).contiguous(), small)


def _latent_to_mf(domain, lv, dtype):
    import torch

    from .engine import SMALL_KEYS
    from .field import torch_dtype

    tdt = torch_dtype(dtype)
    vals = {k: Field(domain[k], lv.small[i].reshape(()).to(tdt)) for i, k in enumerate(SMALL_KEYS)}
    vals["spectrum"] = Field(domain["spectrum"], lv.small[5:].reshape(2, -1).to(tdt))
    vals["xi"] = Field(domain["xi"], lv.xi.to(tdt))
    return MultiField.from_dict(vals, domain)


def _fused_iteration(model, lh, mean, n_samples, minimizer, ic_sampling, comm, geo_minimizer=None):
    from .engine import FusedKL, draw_samples

    dtype = mean["xi"].dtype
    mean_lv = _mf_to_latent(model, mean)
    device_rng = None
    if config.get("sampling_rng") == "device":
        import torch

        device_rng = torch.Generator(device=model.device)
    residuals, negs, n_total = draw_samples(model, mean_lv, n_samples, True, lambda: ic_sampling, comm,
                                            device_rng=device_rng, geo_minimizer=geo_minimizer)
    kl = FusedKL(model, mean_lv, residuals, negs, n_total, comm)
    with parallel.lockstep(comm):  # replicated minimiser: identical decisions on every rank
        kl, _ = minimizer(kl)
    new_mean = _latent_to_mf(lh.domain, kl.position, dtype)
    res_mf = [_latent_to_mf(lh.domain, r, dtype) for r in residuals]
    return new_mean, ResidualSampleList(new_mean, res_mf, negs, comm), kl.value



# ------------------------------------------------------------------------------------------------
# files of a run
# ------------------------------------------------------------------------------------------------
class _RunFiles:
    """Everything optimize_kl writes below ``output_directory`` (None: nothing is written, reports go to the logger).
    File names and contents follow the reference (optimize_kl.py:460-525, 571-615, 716-740) so that its tools -- and a
    resumed reference run, for the random state -- read them."""

    _VALUE_TYPES = ("redchisq", "scmean")
    _CATEGORIES = ("data_residuals", "latent_variables")

    def __init__(self, directory, save_strategy, exports):
        self.directory, self.save_strategy, self.exports = directory, save_strategy, dict(exports)

    def tag(self, ig):
        return "latest" if self.save_strategy == "latest" else f"iteration_{ig}"

    def pickle_path(self, *parts):
        return os.path.join(self.directory, "pickle", *parts)

    def make_directories(self):
        for sub in ["pickle"] + list(self.exports):
            os.makedirs(os.path.join(self.directory, sub), exist_ok=True)

    def dump(self, name, ig, value):
        with open(self.pickle_path(f"{name}_{self.tag(ig)}"), "wb") as f:
            pickle.dump(value, f)

    def load(self, name, ig):
        with open(self.pickle_path(f"{name}_{self.tag(ig)}"), "rb") as f:
            return pickle.load(f)

    def save_random_state(self):
        with open(self.pickle_path("nifty_random_state"), "wb") as f:
            f.write(random.getState())

    def load_random_state(self):
        with open(self.pickle_path("nifty_random_state"), "rb") as f:
            random.setState(f.read())

    # ---- reports ---------------------------------------------------------------------------------
    def report(self, text, file_name, ig, comm, to_logger, every_rank):
        """`text` to the logger and / or appended to ``file_name`` under a "Finished index" header; `every_rank`: the
        ranks' texts are collected and written one "Task r" block each (optimize_kl.py:721-740)."""
        if every_rank and comm is not None:
            text = "\n".join(f"Task {r}\n{t}" for r, t in enumerate(comm.allgather_object(text)))
        elif every_rank:
            text = f"Task 0\n{text}"
        if not get_MPI_params_from_comm(comm)[2]:
            return
        if to_logger:
            logger.info(text)
        if self.directory is not None:
            with open(os.path.join(self.directory, file_name), "a", encoding="utf-8") as f:
                f.write(f"Finished index: {ig}\nCurrent datetime: {datetime.now()}\n{text}\n\n")

    def minisanity_history(self, ig, values, comm):
        """Append this iteration's fit-quality numbers to the pickled history: value type -> category -> key ->
        {index, mean, std} lists (optimize_kl.py:580-613); the history of iteration ig-1 is the starting point."""
        if self.directory is None or not get_MPI_params_from_comm(comm)[2]:
            return None
        try:
            history = self.load("minisanity_history", ig - 1) if ig > 0 else None
        except FileNotFoundError:  # a run that started (or was resumed) without one: begin here
            history = None
        if history is None:
            history = {vt: {cat: {} for cat in self._CATEGORIES} for vt in self._VALUE_TYPES}
        for vt in self._VALUE_TYPES:
            for cat in self._CATEGORIES:
                for key, stat in values[vt][cat].items():
                    track = history[vt][cat].setdefault(key, {"index": [], "mean": [], "std": []})
                    track["index"].append(ig)
                    track["mean"].append(stat["mean"])
                    track["std"].append(stat["std"])
        self.dump("minisanity_history", ig, history)
        return history

    def export(self, ig, sample_list):
        """``<name>/<tag>.hdf5`` with op(sample) for all samples, and mean / standard deviation when there is more than one
        (optimize_kl.py:500-525); operators whose domain is not part of the sample domain are skipped."""
        for name, op in self.exports.items():
            if not _is_subdomain(op.domain, sample_list.domain):
                continue
            many = sample_list.n_samples > 1
            sample_list.save_to_hdf5(os.path.join(self.directory, name, self.tag(ig) + ".hdf5"), op=op, overwrite=True,
                                     samples=True, mean=many, std=many)


def _is_subdomain(sub, total):
    if isinstance(sub, DomainTuple):
        return sub == total
    if not isinstance(sub, MultiDomain):
        raise TypeError
    return isinstance(total, MultiDomain) and all(k in total.keys() and total[k] == dom for k, dom in sub.items())


def _fused_counting_report(before, after, n_local):
    """The four rows of CountingOperator.report() for an iteration of the fused engine, from FusedModel.counters: one
    value/gradient evaluation is one Linearization + one adjoint Jacobian of the likelihood's input, one metric
    application one Jacobian + one adjoint Jacobian (per local sample; nothing is applied to plain fields)."""
    vg, met = after["value_grad"] - before["value_grad"], after["metric"] - before["metric"]
    rows = (("apply: \t\t", 0), ("apply Linearization: \t", vg), ("Jacobian: \t\t", met), ("Adjoint Jacobian: \t", vg + met))
    return "\n".join(f"* {label}{count:>7}" for label, count in rows) + \
        f"\n  (fused engine, {n_local} local samples: {after['transforms'] - before['transforms']} transforms)"

# ------------------------------------------------------------------------------------------------
# driver
# ------------------------------------------------------------------------------------------------
def optimize_kl(likelihood_energy, total_iterations, n_samples, kl_minimizer, sampling_iteration_controller,
                nonlinear_sampling_minimizer=None, constants=[], point_estimates=[], transitions=None,
                export_operator_outputs={}, output_directory=None, initial_position=None, initial_index=0, comm=None,
                inspect_callback=None, terminate_callback=None, plot_energy_history=True, plot_minisanity_history=True,
                save_strategy="latest", return_final_position=False, resume=False, sanity_checks=True, dry_run=False,
                fresh_stochasticity=True, device_id=-1, fuse=True):
    """reference optimize_kl.py:51-453 -- same arguments in the same order (they may be passed positionally), plus `fuse`.
    With ``device_id >= 0`` the whole run executes with that GPU as torch's current device: every libniftyk kernel is launched
    on the current device's current stream (backend._stream)."""
    import contextlib

    options = dict(nonlinear_sampling_minimizer=nonlinear_sampling_minimizer, constants=constants,
                   point_estimates=point_estimates, transitions=transitions, export_operator_outputs=export_operator_outputs,
                   output_directory=output_directory, initial_position=initial_position, initial_index=initial_index, comm=comm,
                   inspect_callback=inspect_callback, terminate_callback=terminate_callback,
                   plot_energy_history=plot_energy_history, plot_minisanity_history=plot_minisanity_history,
                   save_strategy=save_strategy, return_final_position=return_final_position, resume=resume,
                   sanity_checks=sanity_checks, dry_run=dry_run, fresh_stochasticity=fresh_stochasticity, device_id=device_id,
                   fuse=fuse)
    on_gpu = device_id is not None and device_id >= 0
    if on_gpu:
        import torch
    with (torch.cuda.device(device_id) if on_gpu else contextlib.nullcontext()):
        return _optimize_kl(likelihood_energy, total_iterations, n_samples, kl_minimizer, sampling_iteration_controller,
                            **options)


def _optimize_kl(likelihood_energy, total_iterations, n_samples, kl_minimizer, sampling_iteration_controller, *,
                 nonlinear_sampling_minimizer=None, constants=[], point_estimates=[], transitions=None,
                 export_operator_outputs={}, output_directory=None, initial_position=None, initial_index=0, comm=None,
                 inspect_callback=None, terminate_callback=None, plot_energy_history=True,
                 plot_minisanity_history=True, save_strategy="latest", return_final_position=False, resume=False,
                 sanity_checks=True, dry_run=False, fresh_stochasticity=True, device_id=-1, fuse=True):
    if not isinstance(export_operator_outputs, dict):
        raise TypeError
    if "pickle" in export_operator_outputs:
        raise ValueError("The key `pickle` in `export_operator_outputs` is reserved.")
    if not isinstance(initial_index, int):
        raise TypeError
    if save_strategy not in ("all", "latest"):
        raise ValueError(f"Save strategy '{save_strategy}' not supported.")
    if output_directory is None and resume:
        raise ValueError("Can only resume minimization if output_directory is not None")
    if export_operator_outputs:
        if output_directory is None:
            warn("`output_directory=None`, thus no operator outputs will be exported.")
        else:
            for name, op in export_operator_outputs.items():
                if not isinstance(name, str) or not isinstance(op, Operator):
                    raise TypeError("export_operator_outputs maps directory names to Operators")
            # without h5py the reference skips the export silently (optimize_kl.py:508-511, 523-525); so does this run, but it
            # says so once, before the first iteration is spent
            try:
                import h5py  # noqa: F401
            except ImportError:
                logger.warning("optimize_kl(export_operator_outputs=...): `h5py` is not importable here, no HDF5 files "
                               "will be written for %s", sorted(export_operator_outputs))
                export_operator_outputs = {}
    likelihood_energy = _make_callable(likelihood_energy)
    kl_minimizer = _make_callable(kl_minimizer)
    sampling_iteration_controller = _make_callable(sampling_iteration_controller)
    nonlinear_sampling_minimizer = _make_callable(nonlinear_sampling_minimizer)
    constants, point_estimates = _make_callable(constants), _make_callable(point_estimates)
    transitions, n_samples, comm = _make_callable(transitions), _make_callable(n_samples), _make_callable(comm)
    inspect_callback = _make_callable(inspect_callback)
    terminate_callback = _make_callable(False) if terminate_callback is None else terminate_callback
    fresh_stochasticity = _make_callable(fresh_stochasticity)
    if initial_index >= total_iterations:
        raise ValueError(f"Initial index is bigger than total iterations: {initial_index} >= {total_iterations}")
    if likelihood_energy(initial_index).target is not DomainTuple.scalar_domain():
        raise TypeError
    if sanity_checks:
        for ig in range(initial_index, total_iterations):
            for obj, cls in ((likelihood_energy, Operator), (kl_minimizer, DescentMinimizer),
                             (nonlinear_sampling_minimizer, (DescentMinimizer, type(None))),
                             (constants, (list, tuple)), (point_estimates, (list, tuple)), (n_samples, int)):
                if not isinstance(obj(ig), cls):
                    raise TypeError(f"{obj(ig)} is not instance of {cls} but rather {type(obj(ig))}")
            if sampling_iteration_controller(ig) is None:
                if n_samples(ig) != 0:
                    raise ValueError("sampling controller missing")
            elif not isinstance(sampling_iteration_controller(ig), IterationController):
                raise TypeError

    mean = full(makeDomain({}), 0.0) if initial_position is None else initial_position.at(device_id)
    sl = None
    energy_history = EnergyHistory()
    master = get_MPI_params_from_comm(comm(initial_index))[2]

    files = _RunFiles(output_directory, save_strategy, export_operator_outputs)
    if output_directory is not None:
        if master:
            files.make_directories()
        lfile = os.path.join(output_directory, "last_finished_iteration")
        if resume and os.path.isfile(lfile):
            with open(lfile) as f:
                last = int(f.read())
            initial_index = last + 1
            base = files.pickle_path(files.tag(last))
            if os.path.isfile(base + ".mean.pickle"):
                mean = ResidualSampleList.load_mean(base).at(device_id)
                sl = ResidualSampleList.load(base, comm=comm(last), device_id=device_id)
            else:
                sl = SampleList.load(base, device_id=device_id)
                mean = sl.local_item(0)
            if initial_index == total_iterations:
                return (sl, mean) if return_final_position else sl
            files.load_random_state()
            energy_history = files.load("energy_history", last)
        else:
            parallel.check_MPI_synced_random_state(comm(initial_index))
            if master:
                files.save_random_state()

    sseqs = random.spawn_sseq(total_iterations)
    for ig in range(total_iterations):
        if not fresh_stochasticity(ig):
            if ig == 0:
                raise ValueError("fresh_stochasticity needs to be True for initial iteration")
            prev = sseqs[ig - 1]
            sseqs[ig] = np.random.SeedSequence(prev.entropy, spawn_key=prev.spawn_key, pool_size=prev.pool_size)

    for ig in range(initial_index, total_iterations):
        random.push_sseq(sseqs[ig])
        lh = likelihood_energy(ig)
        if not isinstance(lh.domain, MultiDomain):
            raise TypeError(f"Domain of likelihood_energy needs to be a MultiDomain, got\n{lh.domain}")
        t = transitions(ig)
        mean = mean if t is None else t(sl)
        mean = _normal_initialize(mean, lh.domain, device_id=device_id)
        minimizer = kl_minimizer(ig)
        mean_iter = mean.extract(lh.domain)
        cst, pes = list(constants(ig)), list(point_estimates(ig))
        if dry_run:
            logger.info(f"Iteration {ig} checked")
            random.pop_sseq()
            continue
        ns = n_samples(ig)
        model = None
        # the ranks of this iteration's communicator must agree on seeds, domains and the mean before anything is drawn
        # (reference optimize_kl.py:382-385); the mean is compared through a device-side fingerprint (parallel._fingerprint)
        c = comm(ig)
        if c is not None:
            parallel.check_MPI_synced_random_state(c)
            parallel.check_MPI_equality(repr(lh.domain), c)
            parallel.check_MPI_equality(repr(mean.domain), c)
            parallel.check_MPI_equality(mean, c, hash=True)
        sl = None  # the old samples go before new ones are drawn (optimize_kl.py:388)
        # partial constants and a preconditioned NewtonCG (napprox) run on the generic graph
        if fuse and device_id >= 0 and ns > 0 and not cst and not pes and getattr(minimizer, "_napprox", 0) <= 1:
            model = _fused_model(lh, device_id, mean_iter["xi"].dtype if "xi" in mean_iter else np.float64)
        if model is not None:
            tally = dict(model.counters)
            new_mean, sl, value = _fused_iteration(model, lh, mean_iter, ns, minimizer, sampling_iteration_controller(ig),
                                                   c, nonlinear_sampling_minimizer(ig))
            mean = MultiField.union([mean, new_mean])
            sl = sl.at(mean)
            energy_history.append((ig, value))
            counting_report = _fused_counting_report(tally, model.counters, sl.n_local_samples())
        else:
            # the likelihood's input passes a counter (optimize_kl.py:370-373); its tally is this iteration's report
            count = CountingOperator(lh.domain)
            ham = StandardHamiltonian(lh @ count, sampling_iteration_controller(ig), prior_sampling_dtype=mean_iter.dtype)
            if ns == 0:
                e = EnergyAdapter(mean_iter, ham, constants=cst, want_metric=_want_metric(minimizer))
                with parallel.lockstep(c):
                    e, _ = minimizer(e)
                mean = MultiField.union([mean, e.position])
                sl = SampleList([mean])
            else:
                e = SampledKLEnergy(mean_iter, ham, ns, nonlinear_sampling_minimizer(ig), constants=cst,
                                    point_estimates=pes, comm=c, device_id=device_id)
                with parallel.lockstep(c):
                    e, _ = minimizer(e)
                mean = MultiField.union([mean, e.position])
                sl = e.samples.at(mean)
            energy_history.append((ig, e.value))
            counting_report = count.report()
            del e, ham
        is_master = get_MPI_params_from_comm(c)[2]
        if output_directory is not None:
            # every rank's sample files are complete before rank 0 declares the iteration finished, and the marker is
            # on disk before anybody moves on (optimize_kl.py:425-435: save under ensure_all_tasks_succeed + barriers)
            files.export(ig, sl)
            sl.save(files.pickle_path(files.tag(ig)), overwrite=True)
            if c is not None:
                c.barrier()
            if is_master:
                with open(os.path.join(output_directory, "last_finished_iteration"), "w") as f:
                    f.write(str(ig))
                files.dump("energy_history", ig, energy_history)
            if c is not None:
                c.barrier()
        # fit-quality table of this iteration (optimize_kl.py:438, 571-615): logged, appended to minisanity.txt, and its
        # numbers appended to the pickled minisanity history
        from .extra import minisanity

        checked = minisanity(lh, sl, terminal_colors=False, return_values=True)
        if checked:  # ("" for an energy without normalised residuals)
            table, ms_values = checked
            if c is not None:
                parallel.check_MPI_equality(repr(ms_values), c)
            files.report(table, "minisanity.txt", ig, c, to_logger=True, every_rank=False)
            files.minisanity_history(ig, ms_values, c)
        files.report(counting_report, "counting_report.txt", ig, c, to_logger=output_directory is None, every_rank=True)
        cb = inspect_callback
        if cb is not None and callable(cb):
            try:
                na = _nargs(cb)
            except (TypeError, ValueError):
                na = 1
            res = cb(sl) if na == 1 else cb(sl, ig)
            del res
        terminate = terminate_callback(ig)
        if c is not None:
            parallel.check_MPI_equality(bool(terminate), c)
        random.pop_sseq()
        if terminate:
            break
    return (sl, mean) if return_final_position else sl
