"""optimize_kl: the MGVI / geoVI driver loop.

Counterpart of reference nifty/cl/minimization/optimize_kl.py:51-453 (same keyword surface).  Per
global iteration: push the iteration's SeedSequence, initialise missing latent keys with 0.1*N(0,1),
draw samples, minimise the sampled KL, optionally save / call back.

Fusion pass (``fuse=True``, device runs only): when the likelihood is
``GaussianEnergy | PoissonianEnergy  @  [[MaskOperator @] LOSResponse @]  [exp | sigmoid]  @  CorrelatedFieldOperator``
the iteration runs on the fused engine (engine.py: one forward + one adjoint transform per sample and metric application,
CG with device-resident scalars) and the results are handed back as MultiFields / ResidualSampleList.
Everything else walks the generic operator graph.  Plotting and HDF5 export of the
reference are diagnostics outside the hot path and not implemented.
"""
import os
import pickle
from inspect import signature

import numpy as np

from . import config, parallel, random
from .domains import DomainTuple, MultiDomain, makeDomain
from .energy_operators import GaussianEnergy, PoissonianEnergy, StandardHamiltonian, _LikelihoodChain
from .field import Field, MultiField, from_random, full
from .kl import EnergyAdapter, ResidualSampleList, SampledKLEnergy, SampleList
from .minimization import DescentMinimizer, EnergyHistory, IterationController, Minimizer, logger
from .los_response import LOSResponse, SparseResponse
from .operators import (ChainOperator, DiagonalOperator, MaskOperator, Operator, ScalingOperator, _FunctionApplier,
                        _OpChain)
from .parallel import get_MPI_params_from_comm


def _make_callable(obj):
    if callable(obj) and not isinstance(obj, (Minimizer, IterationController, Operator)):
        return obj
    return lambda x: obj


def _nargs(func):
    return len(signature(func).parameters)


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
# driver
# ------------------------------------------------------------------------------------------------
def optimize_kl(likelihood_energy, total_iterations, n_samples, kl_minimizer, sampling_iteration_controller, *,
                device_id=-1, **kwargs):
    """reference optimize_kl.py:51-453.  With ``device_id >= 0`` the whole run executes with that GPU as torch's
    current device: every libniftyk kernel is launched on the current device's current stream (backend._stream)."""
    if device_id is not None and device_id >= 0:
        import torch

        with torch.cuda.device(device_id):
            return _optimize_kl(likelihood_energy, total_iterations, n_samples, kl_minimizer,
                                sampling_iteration_controller, device_id=device_id, **kwargs)
    return _optimize_kl(likelihood_energy, total_iterations, n_samples, kl_minimizer, sampling_iteration_controller,
                        device_id=device_id, **kwargs)


def _optimize_kl(likelihood_energy, total_iterations, n_samples, kl_minimizer, sampling_iteration_controller, *,
                 nonlinear_sampling_minimizer=None, constants=[], point_estimates=[], transitions=None,
                 export_operator_outputs={}, output_directory=None, initial_position=None, initial_index=0, comm=None,
                 inspect_callback=None, terminate_callback=None, plot_energy_history=True,
                 plot_minisanity_history=True, save_strategy="latest", return_final_position=False, resume=False,
                 sanity_checks=True, dry_run=False, fresh_stochasticity=True, device_id=-1, fuse=True):
    if not isinstance(export_operator_outputs, dict):
        raise TypeError
    if not isinstance(initial_index, int):
        raise TypeError
    if save_strategy not in ("all", "latest"):
        raise ValueError(f"Save strategy '{save_strategy}' not supported.")
    if output_directory is None and resume:
        raise ValueError("Can only resume minimization if output_directory is not None")
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

    def fname(ig):
        return "latest" if save_strategy == "latest" else f"iteration_{ig}"

    if output_directory is not None:
        if master:
            os.makedirs(os.path.join(output_directory, "pickle"), exist_ok=True)
        lfile = os.path.join(output_directory, "last_finished_iteration")
        if resume and os.path.isfile(lfile):
            with open(lfile) as f:
                last = int(f.read())
            initial_index = last + 1
            base = os.path.join(output_directory, "pickle", fname(last))
            if os.path.isfile(base + ".mean.pickle"):
                mean = ResidualSampleList.load_mean(base).at(device_id)
                sl = ResidualSampleList.load(base, comm=comm(last), device_id=device_id)
            else:
                sl = SampleList.load(base, device_id=device_id)
                mean = sl.local_item(0)
            if initial_index == total_iterations:
                return (sl, mean) if return_final_position else sl
            with open(os.path.join(output_directory, "pickle", "nifty_random_state"), "rb") as f:
                random.setState(f.read())
            with open(os.path.join(output_directory, "pickle", "energy_history_" + fname(last)), "rb") as f:
                energy_history = pickle.load(f)
        elif master:
            with open(os.path.join(output_directory, "pickle", "nifty_random_state"), "wb") as f:
                f.write(random.getState())

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
        # partial constants and a preconditioned NewtonCG (napprox) run on the generic graph
        if fuse and device_id >= 0 and ns > 0 and not cst and not pes and getattr(minimizer, "_napprox", 0) <= 1:
            model = _fused_model(lh, device_id, mean_iter["xi"].dtype if "xi" in mean_iter else np.float64)
        if model is not None:
            new_mean, sl, value = _fused_iteration(model, lh, mean_iter, ns, minimizer, sampling_iteration_controller(ig),
                                                   comm(ig), nonlinear_sampling_minimizer(ig))
            mean = MultiField.union([mean, new_mean])
            sl = sl.at(mean)
            energy_history.append((ig, value))
        else:
            ham = StandardHamiltonian(lh, sampling_iteration_controller(ig), prior_sampling_dtype=mean_iter.dtype)
            if ns == 0:
                e = EnergyAdapter(mean_iter, ham, constants=cst, want_metric=True)
                with parallel.lockstep(comm(ig)):
                    e, _ = minimizer(e)
                mean = MultiField.union([mean, e.position])
                sl = SampleList([mean])
            else:
                e = SampledKLEnergy(mean_iter, ham, ns, nonlinear_sampling_minimizer(ig), constants=cst,
                                    point_estimates=pes, comm=comm(ig), device_id=device_id)
                with parallel.lockstep(comm(ig)):
                    e, _ = minimizer(e)
                mean = MultiField.union([mean, e.position])
                sl = e.samples.at(mean)
            energy_history.append((ig, e.value))
        if output_directory is not None:
            # every rank's sample files are complete before rank 0 declares the iteration finished, and the marker is
            # on disk before anybody moves on (optimize_kl.py:425-435: save under ensure_all_tasks_succeed + barriers)
            c = comm(ig)
            sl.save(os.path.join(output_directory, "pickle", fname(ig)), overwrite=True)
            if c is not None:
                c.barrier()
            if get_MPI_params_from_comm(c)[2]:
                with open(os.path.join(output_directory, "pickle", "energy_history_" + fname(ig)), "wb") as f:
                    pickle.dump(energy_history, f)
                with open(os.path.join(output_directory, "last_finished_iteration"), "w") as f:
                    f.write(str(ig))
            if c is not None:
                c.barrier()
        # fit-quality table of this iteration (optimize_kl.py:438, 571-578): logged, and appended to minisanity.txt
        from .extra import minisanity

        table = minisanity(lh, sl, terminal_colors=False)
        if table and get_MPI_params_from_comm(comm(ig))[2]:
            logger.info(f"Iteration {ig}: minisanity\n{table}")
            if output_directory is not None:
                with open(os.path.join(output_directory, "minisanity.txt"), "a") as f:
                    f.write(f"Iteration {ig}\n{table}\n\n")
        cb = inspect_callback
        if cb is not None and callable(cb):
            try:
                na = _nargs(cb)
            except (TypeError, ValueError):
                na = 1
            res = cb(sl) if na == 1 else cb(sl, ig)
            del res
        terminate = terminate_callback(ig)
        random.pop_sseq()
        if terminate:
            break
    return (sl, mean) if return_final_position else sl
