"""Energies, iteration controllers, conjugate gradient, line search and descent minimizers.

Counterpart of reference nifty/cl/minimization/{energy,quadratic_energy,conjugate_gradient,
line_search,descent_minimizers,iteration_controllers}.py.  Everything here is host control flow over a
*vector protocol* (``+ - * scalar``, ``s_vdot``, ``norm``) that both ``MultiField`` and the fused
engine's ``LatentVec`` implement; the numerics of every vector operation happen in HIP kernels.
``ConjugateGradient`` additionally has an in-place path with device-resident scalars (one host
synchronisation per iteration) for vectors that provide it.
"""
import functools
import logging
from time import time

import numpy as np

logger = logging.getLogger("nifty_amd")

CONVERGED, CONTINUE, ERROR = 0, 1, 2
class _Counters(dict):
    """work counters (bench.py): CG iterations = metric applications inside ConjugateGradient.  `counters[k] += n` from
    several host threads (the geoVI fits of engine._fit_on_lanes) must not lose increments: use `add`."""

    def __init__(self, *a, **kw):
        import threading

        super().__init__(*a, **kw)
        self._lock = threading.Lock()

    def add(self, key, n=1):
        with self._lock:
            self[key] += n


counters = _Counters(cg_iterations=0)


def _lockstep_flush():
    from . import parallel

    parallel.lockstep_flush()


def _ls(value):
    """Decision-point synchronisation: inside a multi-rank ``parallel.lockstep`` scope (the replicated KL
    minimisation) the scalar of rank 0, otherwise ``value`` itself.  Applied to every reduction of REPLICATED vectors
    that steers control flow here (CG / line-search / L-BFGS dot products, gradient norms); per-sample energies are
    rank-local and must not pass through it."""
    from . import parallel

    if parallel.lockstep_comm() is None:
        return value
    if isinstance(value, complex):
        return complex(parallel.lockstep_float(value.real), parallel.lockstep_float(value.imag))
    return parallel.lockstep_float(float(value))


# ------------------------------------------------------------------------------------------------
# Energy protocol
# ------------------------------------------------------------------------------------------------
class Energy:
    """Value, gradient and metric of a scalar function at ``position`` (reference energy.py:45-136)."""

    def __init__(self, position):
        self._position = position
        self._gradnorm = None

    def at(self, position):
        raise NotImplementedError

    @property
    def position(self):
        return self._position

    @property
    def value(self):
        raise NotImplementedError

    @property
    def gradient(self):
        raise NotImplementedError

    @property
    def gradient_norm(self):
        if self._gradnorm is None:
            self._gradnorm = _ls(self.gradient.norm())
        return self._gradnorm

    @property
    def metric(self):
        raise NotImplementedError

    def apply_metric(self, x):
        raise NotImplementedError

    def longest_step(self, direction):
        return None


class QuadraticEnergy(Energy):
    """E(x) = 1/2 x^T A x - b^T x   (reference quadratic_energy.py:27-78)."""

    def __init__(self, position, A, b, _grad=None, _value=None):
        super().__init__(position)
        self._A, self._b, self._grad, self._value = A, b, _grad, _value
        if _grad is None:
            self._grad = self._residual()
        # the value costs three passes over the vectors: taken on first use (a CG result is usually only asked for its
        # position), or handed in by a caller that knows it (x = 0 -> 0).  Inside a multi-rank lockstep scope the dots
        # are collectives, so there it is taken right away -- never lazily by whichever rank happens to ask
        if _value is None and self._in_lockstep():
            self._value = self._compute_value()

    def _residual(self):
        """gradient A x - b at the position"""
        image = self._A(self._position)
        return image if self._b is None else image - self._b

    @staticmethod
    def _in_lockstep():
        from . import parallel

        return parallel.lockstep_comm() is not None

    def _compute_value(self):
        b = self._b
        Ax = self._grad if b is None else self._grad + b
        val = 0.5 * np.real(self._position.s_vdot(Ax))
        if b is not None:
            val -= np.real(b.s_vdot(self._position))
        return _ls(val)

    def at(self, position):
        return QuadraticEnergy(position, self._A, self._b)

    def at_with_grad(self, position, grad):
        return QuadraticEnergy(position, self._A, self._b, grad)

    @property
    def value(self):
        if self._value is None:
            self._value = self._compute_value()
        return self._value

    @property
    def gradient(self):
        return self._grad

    @property
    def metric(self):
        return self._A

    def apply_metric(self, x):
        return self._A(x)


# ------------------------------------------------------------------------------------------------
# iteration controllers
# ------------------------------------------------------------------------------------------------
class EnergyHistory:
    """Time-stamped energy values of a minimisation, kept as two parallel columns."""

    def __init__(self, stamps=(), values=()):
        self._stamps, self._values = list(stamps), list(values)

    def append(self, x):
        stamp, value = x  # (anything but a pair: ValueError, like the reference)
        self._stamps.append(float(stamp))
        self._values.append(float(value))

    def reset(self):
        del self._stamps[:], self._values[:]

    def __getitem__(self, i):
        return self._stamps[i], self._values[i]

    def __len__(self):
        return len(self._values)

    time_stamps = property(lambda self: list(self._stamps))
    energy_values = property(lambda self: list(self._values))

    def _joined(self, other, into):
        if not isinstance(other, EnergyHistory):
            return NotImplemented
        into._stamps, into._values = self._stamps + other._stamps, self._values + other._values
        return into

    def __add__(self, other):
        return self._joined(other, EnergyHistory())

    def __iadd__(self, other):
        return self._joined(other, self)


def _logged(fn):
    @functools.wraps(fn)
    def wrapper(self, energy):
        if isinstance(self._history, EnergyHistory):
            self._history.append((time(), energy.value))
        return fn(self, energy)

    return wrapper


class IterationController:
    """start(energy) / check(energy) -> CONVERGED | CONTINUE | ERROR (iteration_controllers.py:49-95)."""

    CONVERGED, CONTINUE, ERROR = CONVERGED, CONTINUE, ERROR

    def __init__(self):
        self._history = None

    def start(self, energy):
        raise NotImplementedError

    def check(self, energy):
        raise NotImplementedError

    def enable_logging(self):
        if self._history is None:
            self._history = EnergyHistory()

    def disable_logging(self):
        self._history = None

    @property
    def history(self):
        return self._history


class _LevelController(IterationController):
    """Shared bookkeeping: iteration counter, convergence level, iteration limit."""

    def __init__(self, convergence_level, iteration_limit, name):
        super().__init__()
        self._convergence_level = convergence_level
        self._iteration_limit = iteration_limit
        self._name = name

    def _reset(self):
        self._itcount, self._ccount = -1, 0

    def _criterion(self, energy):  # -> (bool increase_level, str log)
        raise NotImplementedError

    @_logged
    def start(self, energy):
        self._reset()
        self._on_start(energy)
        return self.check.__wrapped__(self, energy)

    def _on_start(self, energy):
        pass

    def stops_at_next_check(self):
        """True when the next check() ends the iteration whatever the energy is: the iteration limit is reached then."""
        return self._iteration_limit is not None and self._itcount + 1 >= self._iteration_limit

    @_logged
    def check(self, energy):
        self._itcount += 1
        inc, msg = self._criterion(energy)
        self._ccount = self._ccount + 1 if inc else max(0, self._ccount - 1)
        if self._name is not None:
            logger.info(f"{self._name}: Iteration #{self._itcount} energy={energy.value:.6E} {msg} clvl={self._ccount}")
        if self._iteration_limit is not None and self._itcount >= self._iteration_limit:
            logger.warning(("" if self._name is None else self._name + ": ") + "Iteration limit reached. Assuming convergence")
            return CONVERGED
        if self._ccount >= self._convergence_level:
            return CONVERGED
        return CONTINUE


class GradientNormController(_LevelController):
    """iteration_controllers.py:170-221"""

    def __init__(self, tol_abs_gradnorm=None, tol_rel_gradnorm=None, convergence_level=1, iteration_limit=None, name=None):
        super().__init__(convergence_level, iteration_limit, name)
        self._tol_abs, self._tol_rel = tol_abs_gradnorm, tol_rel_gradnorm

    def _on_start(self, energy):
        if self._tol_rel is not None:
            self._tol_rel_now = self._tol_rel * energy.gradient_norm

    def _criterion(self, energy):
        inc = False
        if self._tol_abs is not None and energy.gradient_norm <= self._tol_abs:
            inc = True
        if self._tol_rel is not None and energy.gradient_norm <= self._tol_rel_now:
            inc = True
        return inc, ""


class GradInfNormController(_LevelController):
    """iteration_controllers.py:242-283"""

    def __init__(self, tol, convergence_level=1, iteration_limit=None, name=None):
        super().__init__(convergence_level, iteration_limit, name)
        self._tol = tol

    def _criterion(self, energy):
        crit = _ls(energy.gradient.norm(np.inf)) / abs(energy.value)
        return (self._tol is not None and crit <= self._tol), f"crit={crit:.2E}"


class DeltaEnergyController(_LevelController):
    """iteration_controllers.py:305-355 (relative energy change)."""

    def __init__(self, tol_rel_deltaE, convergence_level=1, iteration_limit=None, name=None):
        super().__init__(convergence_level, iteration_limit, name)
        self._tol = tol_rel_deltaE

    def _on_start(self, energy):
        self._Eold = 0.0

    def _criterion(self, energy):
        E = energy.value
        rel = abs(self._Eold - E) / max(abs(self._Eold), abs(E))
        inc = self._itcount > 0 and rel < self._tol
        self._Eold = E
        return inc, f"reldiff={rel:.6E}"


class AbsDeltaEnergyController(_LevelController):
    """iteration_controllers.py:375-423 (absolute energy change)."""

    def __init__(self, deltaE, convergence_level=1, iteration_limit=None, name=None):
        super().__init__(convergence_level, iteration_limit, name)
        self._deltaE = deltaE

    def _on_start(self, energy):
        self._Eold = 0.0

    def _criterion(self, energy):
        E = energy.value
        diff = abs(self._Eold - E)
        inc = self._itcount > 0 and diff < self._deltaE
        self._Eold = E
        return inc, f"diff={diff:.6E} crit={self._deltaE:.1E}"


class StochasticAbsDeltaEnergyController(_LevelController):
    """Convergence from the standard deviation of the last ``memory_length`` energies (for stochastic minimisers;
    reference iteration_controllers.py:426-497)."""

    def __init__(self, deltaE, convergence_level=1, iteration_limit=None, name=None, memory_length=10):
        super().__init__(convergence_level, iteration_limit, name)
        self._deltaE = deltaE
        self.memory_length = memory_length

    def _on_start(self, energy):
        self._memory = []

    def _criterion(self, energy):
        self._memory = (self._memory + [energy.value])[-self.memory_length:]
        diff = float(np.std(self._memory))
        return self._itcount > 0 and diff < self._deltaE, f"diff={diff:.6E} crit={self._deltaE:.1E}"


# ------------------------------------------------------------------------------------------------
# conjugate gradient
# ------------------------------------------------------------------------------------------------
class Minimizer:
    def __call__(self, energy, preconditioner=None):
        raise NotImplementedError

    @property
    def controller(self):
        """the iteration controller that steers this minimiser (minimizer.py:40-42)"""
        return self._controller


def _forced_stop(controller):
    """Will the controller's next check end the solve at its iteration limit, whatever the energy (see _inplace_steps)?"""
    probe = getattr(controller, "stops_at_next_check", None)
    return bool(probe()) if probe is not None else False


def _config_track_energy():
    """Default (round 4): the quadratic energy of the iterate advances by dE = -alpha d.r + alpha^2/2 d.q, both dots taken
    on the way by nk_cg_update_dr -- one input stream less than re-evaluating it from x.r and x.b as the reference does
    (conjugate_gradient.py:100-101, quadratic_energy.py:31-39; -0.8 ms per iteration at 1024^3 fp32, -1.1 % of a benchmark
    step).  The same quantity in exact arithmetic, to 1e-16 for fp64 vectors; for fp32 vectors the recurrence is the value the
    reference's fp64 evaluation would see, while the re-evaluation sees the energy of the ROUNDED iterate.  The controllers
    took identical decisions both ways in 2 x 25 benchmark steps at 1024^3 fp32 (profiles/r04_trajectory_variants.txt), and
    every periodic residual refresh (nreset) restarts the recurrence from the vectors.  NK_CG_ENERGY_RECURRENCE=0: re-evaluate."""
    import os

    return os.environ.get("NK_CG_ENERGY_RECURRENCE", "1") != "0"


class _ScalarEnergyView:
    """What a controller needs from the in-place CG state: value and gradient norm as host floats."""

    def __init__(self, value, gradnorm):
        self.value, self.gradient_norm = value, gradnorm


class _HostCg:
    """State of a conjugate-gradient solve on IMMUTABLE vectors (every update makes new ones): residual r = A x - b,
    preconditioned residual s = P r (P = identity without a preconditioner), search direction d, and gamma = <r, s>.
    One `iterate` = one metric application; it returns None to go on, or the status that ends the solve."""

    def __init__(self, energy, preconditioner, refresh_every):
        self.energy, self._precondition = energy, (lambda v: v) if preconditioner is None else preconditioner
        self._refresh_every, self._since_refresh = refresh_every, 0
        self.r = energy.gradient
        self.d = self._precondition(self.r)
        self.gamma = None

    @staticmethod
    def _dot(u, v):
        return _ls(np.real(u.s_vdot(v)))

    def _bad(self, what):
        logger.error(f"ConjugateGradient stops: {what}")
        return ERROR

    def begin(self):
        self.gamma = self._dot(self.r, self.d)
        if np.isnan(self.gamma):
            return self._bad("the first <r, P r> is not a number")
        return CONVERGED if self.gamma == 0 else None

    def iterate(self, controller):
        q = self.energy.apply_metric(self.d)
        counters.add("cg_iterations")
        curvature = self._dot(self.d, q)
        if np.isnan(curvature) or curvature == 0.0:
            return self._bad("the curvature <d, A d> is zero or not a number")
        step = self.gamma / curvature
        if step < 0:
            return self._bad("negative step length: the operator is not positive definite along the search direction")
        where = self.energy.position - step * self.d
        self._since_refresh += 1
        if self._since_refresh < self._refresh_every:
            self.r = self.r - q * step  # the recurrence; every `refresh_every` iterations the residual is recomputed instead
            self.energy = self.energy.at_with_grad(where, self.r)
        else:
            self.energy = self.energy.at(where)
            self.r, self._since_refresh = self.energy.gradient, 0
        s = self._precondition(self.r)
        gamma = self._dot(self.r, s)
        if np.isnan(gamma):
            return self._bad("<r, P r> is not a number")
        if gamma < 0:
            return self._bad("<r, P r> < 0: the preconditioner is not positive definite")
        if gamma == 0:
            return CONVERGED
        _lockstep_flush()
        status = controller.check(self.energy)
        if status != CONTINUE:
            return status
        self.d = self.d * max(0, gamma / self.gamma) + s
        self.gamma = gamma
        return None


# residual refresh period of every conjugate-gradient solve of the package (reference conjugate_gradient.py:38: nreset = 20): ONE
# constant for ConjugateGradient's default, the batched sampling solves (batched.solve_together) and the sharded CG, whose
# bit-identity with each other depends on refreshing at the same iterations
CG_NRESET = 20


class ConjugateGradient(Minimizer):
    """Linear CG on a QuadraticEnergy (reference conjugate_gradient.py:48-126).

    If the vectors implement ``cg_workspace()`` (the fused engine's LatentVec) the iteration runs in
    place with the fused kernels nk_cg_curv / nk_cg_update / nk_cg_direction: alpha, beta, gamma and
    the quadratic energy value stay on the device and are fetched with ONE copy per iteration.
    """

    def __init__(self, controller, nreset=None):
        nreset = CG_NRESET if nreset is None else nreset
        self._controller = controller
        self._nreset = nreset

    def __call__(self, energy, preconditioner=None):
        if preconditioner is None and hasattr(energy.position, "cg_workspace") and isinstance(energy, QuadraticEnergy):
            return self._solve_inplace(energy)
        return self._solve_generic(energy, preconditioner)

    def _solve_generic(self, energy, preconditioner):
        """Vectors without in-place kernels (host Fields, MultiFields of the generic graph) or a preconditioned solve: the
        same iteration on a `_HostCg` state object -- curvature, step, residual, new direction -- whose checks end the
        solve with a status instead of an exception (conjugate_gradient.py:76-96)."""
        status = self._controller.start(energy)
        if status != CONTINUE:
            return energy, status
        cg = _HostCg(energy, preconditioner, self._nreset)
        status = cg.begin()
        while status is None:
            status = cg.iterate(self._controller)
        return cg.energy, status

    def _solve_inplace(self, energy):
        A, b = energy._A, energy._b
        sm = getattr(A, "sharded", None)
        if sm is not None and b is not None:
            status = self._controller.start(energy)
            if status != CONTINUE:
                return energy, status
            return self._solve_inplace_sharded(energy, sm)
        steps = self._inplace_steps(energy, self._controller)
        while True:
            try:
                next(steps)
            except StopIteration as done:
                return done.value

    def solve_many(self, problems):
        """Several independent in-place solves advanced TOGETHER: `problems` = [(energy, controller, stream or None)], one
        CG each with its own vectors, scalars and stopping rule; returns [(energy, status)].  Every round enqueues one
        iteration of every unfinished solve on that solve's stream and only then waits for the scalars of the previous
        round, so the kernel chains of the solves overlap on the device (small grids: one chain leaves most CUs idle --
        the linear samples of an MGVI iteration are such independent solves).  The arithmetic of every solve is exactly
        that of a solve on its own."""
        import contextlib

        import torch

        runs = [self._inplace_steps(energy, controller) for energy, controller, _ in problems]
        results, active = [None] * len(problems), list(range(len(problems)))
        while active:
            for k in list(active):
                stream = problems[k][2]
                with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                    try:
                        next(runs[k])
                    except StopIteration as done:
                        results[k] = done.value
                        active.remove(k)
        return results

    def _inplace_steps(self, energy, controller):
        """The in-place iteration as a generator: it yields after the device work of an iteration is enqueued and before
        the host waits for its scalars; the generator's return value is (energy, status)."""
        status = controller.start(energy)
        if status != CONTINUE:
            return energy, status
        A, b = energy._A, energy._b
        # the iterate and the residual are updated in place: on private copies -- unless the caller has handed the vectors
        # over (QuadraticEnergy.consumable: nobody else holds the start position / gradient, two N-sized copies less)
        if getattr(energy, "consumable", False):
            x, r = energy.position, energy.gradient
        else:
            x, r = energy.position.clone(), energy.gradient.clone()
        d = r.clone()
        ws = x.cg_workspace()
        gamma_prev = _ls(r.s_vdot(r))
        if np.isnan(gamma_prev):
            return energy, ERROR
        if gamma_prev == 0:
            return energy, CONVERGED
        ws.set_gamma(gamma_prev)
        since_reset = 0

        def finish(status):
            return energy.at_with_grad(x, r), status

        # the energy of the iterate advances by a two-dot recurrence instead of being recomputed from x.r and x.b
        track_energy = hasattr(ws, "update_dr") and _config_track_energy()
        value = float(energy.value) if track_energy else None
        fused_dot = bool(getattr(A, "fused_dot", False)) and hasattr(ws, "curv_slot")
        # ... and the first pass of its transform can take over the previous iteration's direction update
        # (either shortcut without the other: a KL metric that sums its samples pairwise has no place for the dot)
        fused_dir = bool(getattr(A, "fused_direction", False)) and hasattr(ws, "direction_small")

        def device_iteration(with_direction):
            """All device work of one iteration up to the fused update: [d <- beta d + r;] q = A d; d.q; x, r update."""
            extra = {}
            if with_direction and fused_dir:
                ws.direction_small(d, r)
                extra["cg_direction"] = (r, ws)
            elif with_direction:
                ws.direction(d, r)
            if fused_dot:
                # the operator's last epilogue takes d.q (xi part) while it writes q: one BLAS-1 pass less
                q = A(d, dot_out=ws.curv_slot(), **extra)
                ws.curv_small(d, q)
            else:
                q = A(d, **extra)
                ws.curv(d, q)
            if track_energy:
                ws.update_dr(x, r, d, q)
            else:
                ws.update(x, r, d, q, b)

        # (replaying this body as a captured HIP graph was measured on the launch-heavy 2048^2 fp64 Poisson configuration:
        # 162.1 vs 161.9 ms per MGVI iteration -- the chain of small kernels itself is the limit there, not the launches.
        # Round 4, with four chains side by side on lanes (where the Python launch path IS the limit, ~14 us per launch):
        # capturing the steady-state iteration at iteration 2 and replaying it gave 131 ms against 115 ms per MGVI
        # iteration -- every solve of this recipe has new vectors and at most 20 iterations, and a capture + instantiation
        # costs more than 18 replays save.  A graph would have to outlive the solve: persistent CG and linearisation buffers.)
        # Without the fused direction update the NEXT iteration's d <- beta d + r is enqueued before the host waits for
        # this iteration's scalars (it reads beta from the device scalars and steers nothing): the device works through it
        # while the host decides whether to go on -- on small grids that wait was 65 us of idle GPU per iteration.  A solve
        # that stops has updated a direction nobody reads any more.
        ahead = hasattr(ws, "fetch_begin") and not fused_dir
        direction_done = False
        iteration = 0
        while True:
            iteration += 1
            device_iteration(iteration > 1 and not direction_done)
            direction_done = False
            counters.add("cg_iterations")
            since_reset += 1
            refreshed = False
            # periodic residual refresh (conjugate_gradient.py:103-106): r = A x - b -- except when the controller is about to
            # stop at its iteration limit anyway (20 iterations, refresh every 20: the recipe of the sampling solves ends on
            # exactly this case): the iterate is final, nobody reads that residual, and the refresh is a whole metric
            # application (1 of 22 per sampling solve)
            if since_reset >= self._nreset and not _forced_stop(controller):
                Ax = A(x)
                r = Ax - b if b is not None else Ax
                ws.refresh(x, r, b)
                since_reset = 0
                refreshed = True
            if ahead:
                ws.fetch_begin()
                ws.direction(d, r)
                direction_done = True
            yield  # (solve_many: the other solves enqueue their iterations here)
            sc = ws.fetch_end() if ahead else ws.fetch()  # the single host synchronisation of this iteration
            curv, gamma, alpha = sc["curv"], sc["gamma"], sc["alpha"]
            if np.isnan(curv) or curv == 0.0 or np.isnan(alpha) or alpha < 0:
                logger.error("Error: ConjugateGradient: bad curvature / step")
                return finish(ERROR)
            if np.isnan(gamma) or gamma < 0:
                return finish(ERROR)
            if gamma == 0:
                return finish(CONVERGED)
            if track_energy and not refreshed:
                # E(x - alpha d) = E(x) - alpha d.r + alpha^2/2 d.q, both dots taken on the way (nk_cg_update_dr)
                value = value - alpha * sc["dr"] + 0.5 * alpha * alpha * curv
            else:
                value = 0.5 * sc["xr"] - 0.5 * sc["xb"] if b is not None else 0.5 * sc["xr"]
            status = controller.check(_ScalarEnergyView(value, float(np.sqrt(gamma))))
            if status != CONTINUE:
                return finish(status)
            # d <- beta d + r opens the next iteration (inside the next A(d) where the operator can take it over)


    def _solve_inplace_sharded(self, energy, sm):
        """Same iteration as _solve_inplace with x, r, d, q held as per-rank shards (engine.ShardedMetric): one
        reduce-scatter + one all-gather of the latent vector and two scalar all-reduces per iteration."""
        controller = self._controller
        b_full = energy._b
        b = sm.shard(b_full)
        x = sm.shard(energy.position, copy=True)
        r = sm.shard(energy.gradient, copy=True)
        d = r.clone()
        d_full = energy.gradient.clone()
        ws = sm.workspace()
        track_energy = hasattr(ws, "update_dr") and _config_track_energy()
        value = float(energy.value) if track_energy else None
        gamma_prev = float(ws.dot(r, r, 0).item())
        if np.isnan(gamma_prev):
            return energy, ERROR
        if gamma_prev == 0:
            return energy, CONVERGED
        since_reset = 0

        def finish(status):
            return energy.at_with_grad(sm.gather(x), sm.gather(r)), status

        while True:
            q = sm.apply_shard(d, d_full)  # all-gather of d, local metric, reduce-scatter (overlapped chunk by chunk)
            ws.curv(d, q)
            if track_energy:
                ws.update_dr(x, r, d, q)
            else:
                ws.update(x, r, d, q, b)
            counters.add("cg_iterations")
            since_reset += 1
            refreshed = False
            if since_reset >= self._nreset and not _forced_stop(controller):
                Ax = sm.apply(sm.gather(x))
                r = Ax - b
                ws.refresh(x, r, b)
                since_reset = 0
                refreshed = True
            sc = ws.fetch()
            curv, gamma, alpha = sc["curv"], sc["gamma"], sc["alpha"]
            if np.isnan(curv) or curv == 0.0 or np.isnan(alpha) or alpha < 0:
                logger.error("Error: ConjugateGradient: bad curvature / step")
                return finish(ERROR)
            if np.isnan(gamma) or gamma < 0:
                return finish(ERROR)
            if gamma == 0:
                return finish(CONVERGED)
            if track_energy and not refreshed:
                value = value - alpha * sc["dr"] + 0.5 * alpha * alpha * curv
            else:
                value = 0.5 * sc["xr"] - 0.5 * sc["xb"]
            status = controller.check(_ScalarEnergyView(value, float(np.sqrt(gamma))))
            if status != CONTINUE:
                return finish(status)
            ws.direction(d, r)


# ------------------------------------------------------------------------------------------------
# line search (strong Wolfe conditions, Nocedal & Wright alg. 3.5 / 3.6)
# ------------------------------------------------------------------------------------------------
class LineEnergy:
    """Restriction of an Energy to the line position + t * direction (line_search.py:52-100)."""

    def __init__(self, line_position, energy, line_direction, offset=0.0):
        self._t = float(line_position)
        self._dir = line_direction
        if self._t == float(offset):
            self._energy = energy
        else:
            self._energy = energy.at(position=energy.position + (self._t - float(offset)) * self._dir)

    def at(self, line_position):
        return LineEnergy(line_position, self._energy, self._dir, offset=self._t)

    @property
    def energy(self):
        return self._energy

    @property
    def value(self):
        return self._energy.value

    @property
    def directional_derivative(self):
        return _ls(np.real(self._energy.gradient.s_vdot(self._dir)))


def _interp_cubic(a, fa, fpa, b, fb, c, fc):
    """Minimiser of the cubic through (a,fa,fpa), (b,fb), (c,fc); None if it does not exist."""
    with np.errstate(divide="raise", over="raise", invalid="raise"):
        try:
            db, dc = b - a, c - a
            denom = (db * dc) ** 2 * (db - dc)
            m = np.array([[dc * dc, -db * db], [-dc ** 3, db ** 3]])
            A, B = m @ np.array([fb - fa - fpa * db, fc - fa - fpa * dc])
            A, B = A / denom, B / denom
            xmin = a + (-B + np.sqrt(B * B - 3 * A * fpa)) / (3 * A)
        except ArithmeticError:
            return None
    return xmin if np.isfinite(xmin) else None


def _interp_quadratic(a, fa, fpa, b, fb):
    with np.errstate(divide="raise", over="raise", invalid="raise"):
        try:
            db = b - a * 1.0
            B = (fb - fa - fpa * db) / (db * db)
            xmin = a - fpa / (2.0 * B)
        except ArithmeticError:
            return None
    return xmin if np.isfinite(xmin) else None


class _Wolfe:
    """The two strong-Wolfe tests for a search line with value phi0 and slope dphi0 < 0 at t = 0."""

    def __init__(self, phi0, dphi0, c1, c2):
        self.phi0, self.dphi0, self._c1, self._c2 = phi0, dphi0, c1, c2

    def too_high(self, t, phi):
        """sufficient-decrease (Armijo) condition violated at step t"""
        return phi > self.phi0 + self._c1 * t * self.dphi0

    def flat_enough(self, dphi):
        """curvature condition: |phi'(t)| <= c2 |phi'(0)|"""
        return abs(dphi) <= -self._c2 * self.dphi0


class LineSearch:
    """Step length satisfying the strong Wolfe conditions (reference line_search.py:103-416; Nocedal & Wright,
    algorithms 3.5 and 3.6): `perform_line_search` walks outwards from t = 0 until it either meets both conditions or
    holds a bracket, which `_zoom` then shrinks by safeguarded cubic / quadratic interpolation."""

    def __init__(self, preferred_initial_step_size=None, c1=1e-4, c2=0.9, max_step_size=1e30, max_iterations=100,
                 max_zoom_iterations=100):
        self.preferred_initial_step_size = preferred_initial_step_size
        self.c1, self.c2 = float(c1), float(c2)
        self.max_step_size = max_step_size
        self.max_iterations = int(max_iterations)
        self.max_zoom_iterations = int(max_zoom_iterations)

    def _opening_step(self, wolfe, direction, previous_value):
        """First trial step: the caller's preference, else the step a quadratic model through the previous energy
        would take, else a unit displacement."""
        if self.preferred_initial_step_size is not None:
            return self.preferred_initial_step_size
        if previous_value is None:
            return 1.0 / _ls(direction.norm())
        guess = min(1.0, 1.01 * 2 * (wolfe.phi0 - previous_value) / wolfe.dphi0)
        return guess if guess >= 0 else 1.0

    @staticmethod
    def _probe(origin, t):
        """(line energy, value) at step t, or None where the energy cannot be evaluated (overflow, NaN)"""
        try:
            there = origin.at(t)
            phi = there.value
        except FloatingPointError:
            return None
        return None if (np.isnan(phi) or abs(phi) > 1e100) else (there, phi)

    def _descending(self, origin):
        """the Wolfe tests of the line, or None (with a log entry) when it does not start downhill"""
        wolfe = _Wolfe(origin.value, origin.directional_derivative, self.c1, self.c2)
        if wolfe.dphi0 < 0:
            return wolfe
        if wolfe.dphi0 == 0:
            logger.warning("Directional derivative is zero; assuming convergence")
        else:
            logger.error("Error: search direction is not a descent direction")
        return None

    def perform_line_search(self, energy, pk, f_k_minus_1=None):
        origin = LineEnergy(0.0, energy, pk, 0.0)
        wolfe = self._descending(origin)
        if wolfe is None:
            return energy, False
        limit = min(cap for cap in (energy.longest_step(pk), self.max_step_size) if cap is not None)
        t = min(self._opening_step(wolfe, pk, f_k_minus_1), 0.99 * limit)
        behind, there = (0.0, wolfe.phi0, wolfe.dphi0), None  # behind: last accepted point on the way out (t, phi, phi')
        for attempt in range(self.max_iterations):
            if t == 0:
                return origin.energy, False
            # only (t, phi, phi') of the previous trial are needed from here on: let its energy go BEFORE the next one is
            # built (a sampled KL holds one latent vector per sample; three of them alive at once were 57 GB more at
            # 1024^3 x 8 samples -- 305 of the 309 GB of HBM on the trajectories that probe twice)
            there = None
            probed = self._probe(origin, t)
            if probed is None:  # not evaluable: come half way back
                t = 0.5 * (behind[0] + t)
                continue
            there, phi = probed
            if wolfe.too_high(t, phi) or (attempt > 0 and phi >= behind[1]):
                there = None
                return self._zoom(origin, wolfe, behind, (t, phi))
            dphi = there.directional_derivative
            if wolfe.flat_enough(dphi):
                return there.energy, True
            if dphi >= 0:  # walked past a minimum: it lies between here and the previous point
                there = None
                return self._zoom(origin, wolfe, (t, phi, dphi), behind[:2])
            behind, t = (t, phi, dphi), min(2 * t, limit)
            if t == limit:
                logger.warning("max step size reached")
                return there.energy, False
        logger.warning("max iterations reached")
        return (origin if there is None else there).energy, False

    def _zoom(self, origin, wolfe, low, high):
        """Shrinks the bracket between `low` = (t, phi, phi') -- the end with the smaller value, its slope pointing into
        the bracket -- and `high` = (t, phi) until a point meets both Wolfe conditions."""
        lo, phi_lo, dphi_lo = low
        hi, phi_hi = high
        if wolfe.too_high(lo, phi_lo) or dphi_lo * (hi - lo) >= 0.0:
            raise ValueError("inconsistent data")
        dropped = None  # the end point given up last: third support point of the cubic
        there = None
        for shrink in range(self.max_zoom_iterations):
            t = self._interpolate(lo, phi_lo, dphi_lo, hi, phi_hi, dropped)
            there = None  # (see perform_line_search: the previous trial's energy goes first)
            there = origin.at(t)
            phi = there.value
            if wolfe.too_high(t, phi) or phi >= phi_lo:
                dropped, (hi, phi_hi) = (hi, phi_hi), (t, phi)
                continue
            dphi = there.directional_derivative
            if wolfe.flat_enough(dphi):
                return there.energy, True
            if dphi * (hi - lo) >= 0:  # the minimum is on the other side of t: the old low end becomes the high end
                dropped, (hi, phi_hi) = (hi, phi_hi), (lo, phi_lo)
            else:
                dropped = (lo, phi_lo)
            lo, phi_lo, dphi_lo = t, phi, dphi
        logger.warning("The line search algorithm (zoom) did not converge.")
        return there.energy, False

    @staticmethod
    def _interpolate(lo, phi_lo, dphi_lo, hi, phi_hi, third):
        """Trial point inside the bracket: minimiser of the cubic through both ends and `third` if it keeps 20 % of the
        (signed) bracket width away from the ends, else of the quadratic with a 10 % margin, else the midpoint."""
        width = hi - lo
        left, right = (lo, hi) if width >= 0 else (hi, lo)

        def inside(t, margin):
            return t is not None and left + margin * width <= t <= right - margin * width

        if third is not None:
            t = _interp_cubic(lo, phi_lo, dphi_lo, hi, phi_hi, *third)
            if inside(t, 0.2):
                return t
        t = _interp_quadratic(lo, phi_lo, dphi_lo, hi, phi_hi)
        return t if inside(t, 0.1) else lo + 0.5 * width


# ------------------------------------------------------------------------------------------------
# descent minimizers
# ------------------------------------------------------------------------------------------------
class DescentMinimizer(Minimizer):
    """Descent along directions a subclass proposes, each followed by a line search (reference
    descent_minimizers.py:52-108)."""

    def __init__(self, controller, line_searcher=None):
        self._controller = controller
        self.line_searcher = LineSearch() if line_searcher is None else line_searcher

    def _step(self, energy, previous_value):
        """One direction + line search from `energy`: (energy to continue with, verdict), verdict None = go on."""
        direction = self.get_descent_direction(energy, previous_value)
        found, success = self.line_searcher.perform_line_search(energy=energy, pk=direction, f_k_minus_1=previous_value)
        if not success:
            self.reset()
        if found.value > energy.value:
            logger.error("Error: Energy has increased")
            return energy, ERROR
        if found.value == energy.value:
            logger.warning("Warning: Energy has not changed. Assuming convergence...")
            return found, CONVERGED
        return found, None

    def __call__(self, energy):
        previous_value = None
        verdict = self._controller.start(energy)
        while verdict == CONTINUE:
            _lockstep_flush()  # multi-rank: one agreement check per outer step (no-op otherwise)
            if energy.gradient_norm == 0:
                return energy, CONVERGED
            value_here = energy.value
            energy, early = self._step(energy, previous_value)
            previous_value = value_here
            verdict = self._controller.check(energy) if early is None else early
        return energy, verdict

    def reset(self):
        pass

    def get_descent_direction(self, energy, old_value=None):
        raise NotImplementedError

    @property
    def controller(self):
        return self._controller


class SteepestDescent(DescentMinimizer):
    def get_descent_direction(self, energy, _=None):
        return -energy.gradient


class RelaxedNewton(DescentMinimizer):
    """Newton direction -M^{-1} g with the energy's metric (reference descent_minimizers.py:149-163)."""

    def __init__(self, controller, line_searcher=None):
        super().__init__(controller, LineSearch(preferred_initial_step_size=1.0) if line_searcher is None else line_searcher)

    def get_descent_direction(self, energy, _=None):
        return -energy.metric.inverse_times(energy.gradient)


class L_BFGS(DescentMinimizer):
    """Limited-memory BFGS: two-loop recursion over the last `max_history_length` position / gradient differences
    (reference descent_minimizers.py:213-262; pure vector algebra on the Field / LatentVec protocol)."""

    def __init__(self, controller, line_searcher=None, max_history_length=5):
        super().__init__(controller, line_searcher)
        self.max_history_length = max_history_length
        self.reset()

    def __call__(self, energy):
        self.reset()
        return super().__call__(energy)

    def reset(self):
        from collections import deque

        self._pairs = deque(maxlen=self.max_history_length)  # (s, y) = (position, gradient) differences, oldest first
        self._last = None

    def get_descent_direction(self, energy, _=None):
        point = (energy.position, energy.gradient)
        if self._last is not None:
            self._pairs.append((point[0] - self._last[0], point[1] - self._last[1]))
        self._last = point
        p = -point[1]
        if not self._pairs:
            return p
        # first loop, newest pair first: peel the curvature pairs off the gradient
        peeled = []
        for s, y in reversed(self._pairs):
            sy = _ls(s.s_vdot(y))
            weight = _ls(s.s_vdot(p)) / sy
            p = p - weight * y
            peeled.append((s, y, sy, weight))
        s, y = self._pairs[-1]
        scale = _ls(s.s_vdot(y)) / _ls(y.s_vdot(y))  # initial Hessian guess: gamma * identity
        if scale <= 0.0:
            logger.error("L-BFGS curvature not positive definite!")
        p = p * scale
        # second loop, oldest first
        for s, y, sy, weight in reversed(peeled):
            p = p + (weight - _ls(y.s_vdot(p)) / sy) * s
        return p


class VL_BFGS(DescentMinimizer):
    """Vector-free L-BFGS (Chen, Wang, Zhou 2014; reference descent_minimizers.py:264-468): the two-loop recursion runs
    on the (2m+1) x (2m+1) Gram matrix of the basis b = (s_1..s_m, y_1..y_m, g) instead of on vectors, so one descent
    direction costs 3m+... dot products (cached across iterations) and ONE linear combination of the basis -- on the GPU
    2m+1 axpy passes instead of the 4m of L_BFGS."""

    def __init__(self, controller, line_searcher=None, max_history_length=5):
        super().__init__(controller, line_searcher)
        self.max_history_length = max_history_length
        self.reset()

    def __call__(self, energy):
        self.reset()
        return super().__call__(energy)

    def reset(self):
        m = self.max_history_length
        self._k = 0  # number of stored updates so far
        self._s, self._y = [None] * m, [None] * m
        self._ss, self._sy, self._yy = (np.zeros((m, m)) for _ in range(3))
        self._lastx = self._lastgrad = None

    def _dot(self, a, b):
        return _ls(np.real(a.s_vdot(b)))

    def get_descent_direction(self, energy, _=None):
        mmax = self.max_history_length
        x, g = energy.position, energy.gradient
        if self._lastx is not None:
            self._s[self._k % mmax] = x - self._lastx
            self._y[self._k % mmax] = g - self._lastgrad
            self._k += 1
        self._lastx, self._lastgrad = x, g
        k, m = self._k, min(self._k, mmax)
        slot = [(k - m + i) % mmax for i in range(m)]  # oldest ... newest
        # scalar products involving the newest pair (everything older is cached)
        if m:
            new = slot[-1]
            for i in slot:
                self._ss[i, new] = self._ss[new, i] = self._dot(self._s[i], self._s[new])
                self._yy[i, new] = self._yy[new, i] = self._dot(self._y[i], self._y[new])
                self._sy[i, new] = self._dot(self._s[i], self._y[new])
            for j in slot[:-1]:
                self._sy[new, j] = self._dot(self._s[new], self._y[j])
        n = 2 * m + 1
        B = np.empty((n, n))
        for i, si in enumerate(slot):
            for j, sj in enumerate(slot):
                B[i, j] = self._ss[si, sj]
                B[i, m + j] = B[m + j, i] = self._sy[si, sj]
                B[m + i, m + j] = self._yy[si, sj]
            B[2 * m, i] = B[i, 2 * m] = self._dot(self._s[si], g)
            B[2 * m, m + i] = B[m + i, 2 * m] = self._dot(self._y[si], g)
        B[2 * m, 2 * m] = g.norm()  # (never read below; the reference stores the norm here, not its square)
        # two-loop recursion in coefficient space
        delta = np.zeros(n)
        delta[2 * m] = -1.0
        alpha = np.empty(m)
        for j in range(m - 1, -1, -1):
            alpha[j] = float(np.dot(delta, B[:, j])) / B[j, m + j]
            delta[m + j] -= alpha[j]
        if m:
            delta *= B[m - 1, 2 * m - 1] / B[2 * m - 1, 2 * m - 1]
        for j in range(m):
            beta = float(np.dot(delta, B[m + j, :])) / B[j, m + j]
            delta[j] += alpha[j] - beta
        basis = [self._s[i] for i in slot] + [self._y[i] for i in slot] + [g]
        direction = basis[0] * delta[0]
        for coeff, vec in zip(delta[1:], basis[1:]):
            direction = direction + vec * coeff
        return direction


class NewtonCG(DescentMinimizer):
    """Inexact Newton: CG on metric * delta = gradient (reference descent_minimizers.py:166-210).  The CG is stopped
    early -- after 5 iterations on the first step, afterwards once it gains less than `energy_reduction_factor` times the
    outer iteration's last energy decrease per iteration."""

    def __init__(self, controller, napprox=0, line_searcher=None, name=None, nreset=20, max_cg_iterations=200,
                 energy_reduction_factor=0.1, enable_logging=False):
        unit_step = LineSearch(preferred_initial_step_size=1.0)  # a Newton step has its natural length
        super().__init__(controller, unit_step if line_searcher is None else line_searcher)
        self._cg = dict(nreset=nreset, limit=max_cg_iterations, reduction=energy_reduction_factor, name=name)
        self._napprox, self._name, self._nreset = napprox, name, nreset
        self._history = EnergyHistory() if enable_logging else None

    def _inner_controller(self, energy, old_value):
        if old_value is None:
            inner = GradientNormController(iteration_limit=5)
        else:
            inner = AbsDeltaEnergyController(self._cg["reduction"] * (old_value - energy.value),
                                             iteration_limit=self._cg["limit"], name=self._cg["name"])
        if self._history is not None:  # the CG's energy trace is appended to `inversion_history` afterwards
            inner.enable_logging()
        return inner

    def _preconditioner(self, energy):
        """inverse of the sampled diagonal of the metric (descent_minimizers.py:201-203), or None"""
        if self._napprox <= 1:
            return None
        from .operators import makeOp
        from .probing import approximation2endo

        where = getattr(energy.position, "device_id", -1)
        return makeOp(approximation2endo(energy.metric, self._napprox, where)).inverse

    def get_descent_direction(self, energy, old_value=None):
        inner = self._inner_controller(energy, old_value)
        solver = ConjugateGradient(inner, nreset=self._cg["nreset"])
        solved, outcome = solver(self._newton_model(energy), self._preconditioner(energy))
        if inner.history is not None:
            self._history += inner.history
        if outcome == ERROR:
            raise ValueError("Cannot find descent direction")
        return -solved.position

    @staticmethod
    def _newton_model(energy):
        """1/2 d.M d - g.d at d = 0.  M 0 = 0 exactly, so gradient (-g) and value (0) at the start are known and the
        reference's metric application to the zero vector is skipped."""
        g = energy.gradient
        model = QuadraticEnergy(energy.position * 0.0, energy.metric, g, _grad=-g, _value=0.0)
        model.consumable = True  # the start vectors are temporaries of this call: the CG may iterate on them in place
        return model

    @property
    def inversion_history(self):
        return self._history
