"""Consistency checks for operators (reference extra.py:41-131 ``check_linear_operator``, :134-183
``check_operator``): the identities the reference's own test-suite is built on -- linearity, adjointness
<y, A x> = <A^† y, x>, inverse round trips, Jacobian versus finite differences, and host/device agreement."""
import numpy as np

from .field import Field, MultiField, device_available, from_random
from .operators import Linearization, LinearOperator, Operator

__all__ = ["check_linear_operator", "check_operator", "assert_allclose", "assert_equal", "minisanity"]


def assert_allclose(f1, f2, atol=0, rtol=1e-7):
    """Field / MultiField comparison (reference extra.py:186-192)."""
    if not isinstance(f1, Field):  # MultiFields: same domain object, then key by key
        if f1.domain is not f2.domain:
            raise AssertionError
        pairs = [(f1[key], f2[key]) for key in f1.keys()]
    else:
        pairs = [(f1, f2)]
    for a, b in pairs:
        np.testing.assert_allclose(a.asnumpy(), b.asnumpy(), atol=atol, rtol=rtol)


def assert_equal(f1, f2, *, atol=0.0, rtol=0.0):
    """Exact agreement by default (reference extra.py:198-204)."""
    assert_allclose(f1, f2, atol=atol, rtol=rtol)


def _same_answer_twice(op, x):
    """An operator is a pure function of its input: applying it twice gives the same bits (extra.py:368-380; the device
    reductions of libniftyk are built in a fixed order, so this holds on the GPU too)."""
    assert_equal(op(x), op(x))


def _largest_entry(f):
    """max |f| over all entries (and keys) of a Field / MultiField"""
    arrays = f.asnumpy().values() if isinstance(f, MultiField) else [f.asnumpy()]
    return max((float(np.abs(a).max()) for a in arrays if a.size), default=0.0)


def _device_ids(force_device_ids):
    ids = {-1, *force_device_ids}
    if device_available():
        ids.add(0)
    if not all(isinstance(i, int) and i >= -1 for i in ids):
        raise TypeError("Device ids need to be int and >= -1")
    return sorted(ids)


def _flips(op):
    """The operator and its adjoint / inverse / adjoint-inverse views with (domain dtype, target dtype) swapped
    accordingly."""
    return [(op, 0), (op.adjoint, 1), (op.inverse, 1), (op.adjoint.inverse, 0)]


def _has(op, cap):
    return (op.capability & cap) == cap


def _check_one(op, dt_dom, dt_tgt, atol, rtol, only_r_linear, dev):
    rnd = lambda dom, dt: from_random(dom, "normal", dtype=dt, device_id=dev)  # noqa: E731
    if _has(op, op.TIMES):
        x, y = rnd(op.domain, dt_dom), rnd(op.domain, dt_dom)
        _same_answer_twice(op, x)
        res = op(x)
        if res.domain is not op.target or x.domain is not op.domain:
            raise AssertionError("operator does not map its domain to its target")
        if res.device_id != dev:
            raise AssertionError("operator moved the field to another device")
        assert_allclose(op(0.42 * x + y), 0.42 * op(x) + op(y), atol=atol, rtol=rtol)  # linearity (:263-272)
    if _has(op, op.TIMES | op.ADJOINT_TIMES):  # adjointness (:220-231)
        f1, f2 = rnd(op.domain, dt_dom), rnd(op.target, dt_tgt)
        a, b = f1.s_vdot(op.adjoint_times(f2)), op.times(f1).s_vdot(f2)
        if only_r_linear:
            a, b = np.real(a), np.real(b)
        np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)
    if _has(op, op.TIMES | op.INVERSE_TIMES):  # inverse round trips (:234-245)
        foo = rnd(op.target, dt_tgt)
        assert_allclose(op(op.inverse_times(foo)), foo, atol=atol, rtol=rtol)
        foo = rnd(op.domain, dt_dom)
        assert_allclose(op.inverse_times(op(foo)), foo, atol=atol, rtol=rtol)


def check_linear_operator(op, domain_dtype=np.float64, target_dtype=np.float64, atol=1e-14, rtol=1e-14,
                          only_r_linear=False, force_device_ids=[-1], assert_fixed_device=True, no_device_copies=True,
                          _device_ids_override=None):
    """Algebraic consistency of every capability of a LinearOperator on the host and, if present, on GPU 0
    (same arguments as the reference, extra.py:41-131; `assert_fixed_device` / `no_device_copies` are accepted and not
    needed: an operator that moved a field to another device fails the device check of every application here)."""
    if not isinstance(op, LinearOperator):
        raise TypeError("This test tests only linear operators.")
    devs = _device_ids(force_device_ids) if _device_ids_override is None else _device_ids_override
    dts = (domain_dtype, target_dtype)
    # the same input on every device gives the same output (:520-560)
    for view, swap in _flips(op):
        if not _has(view, view.TIMES):
            continue
        x = from_random(view.domain, "normal", dtype=dts[swap], device_id=-1)
        ref = view(x.at(devs[0]))
        for dev in devs[1:]:
            assert_allclose(view(x.at(dev)).at(-1), ref.at(-1), atol=max(atol, 1e-12), rtol=max(rtol, 1e-10))
    for dev in devs:
        for view, swap in _flips(op):
            _check_one(view, dts[swap], dts[1 - swap], atol, rtol, only_r_linear, dev)


def check_operator(op, loc, tol=1e-12, ntries=100, perf_check=True, only_r_differentiable=True, metric_sampling=True,
                   force_device_ids=[-1], assert_fixed_device=True, no_device_copies=True):
    """Value consistency of ``op(Linearization)`` and Jacobian against finite differences at ``loc``, then
    ``check_linear_operator`` on the Jacobian (reference extra.py:134-183, 411-496)."""
    if not isinstance(op, Operator):
        raise TypeError("This test tests only (nonlinear) operators.")
    if not isinstance(loc, (Field, MultiField)) or loc.domain is not op.domain:
        raise AssertionError("loc must live on the operator's domain")
    ftol = np.sqrt(tol)
    for dev in _device_ids(force_device_ids):
        pos = loc.at(dev)
        _same_answer_twice(op, pos)
        for wm in (False, True):
            lin = op(Linearization.make_var(pos, wm))
            assert_allclose(op(pos), lin.val, 0, 1e-7)
            if lin.jac.domain is not op.domain or lin.jac.target is not op.target:
                raise AssertionError("Jacobian domain/target mismatch")
        for _ in range(ntries):
            lin = op(Linearization.make_var(pos))
            direction = from_random(pos.domain, dtype=pos.dtype, device_id=dev)  # (complex points move in complex directions)
            dirder = lin.jac(direction)
            scale = lin.val.norm() * 1e-6 / dirder.norm() if dirder.norm() != 0 else lin.val.norm() * 1e-6
            direction = direction * scale
            pos2 = pos + direction
            lin2 = op(Linearization.make_var(pos2))
            nxt = pos2
            for _ in range(50):
                mid = pos + 0.5 * direction
                linmid = op(Linearization.make_var(mid))
                dirder = linmid.jac(direction)
                numgrad = lin2.val - lin.val
                xtol = ftol * dirder.norm() / np.sqrt(dirder.size)
                if _largest_entry(numgrad - dirder) <= xtol:
                    break
                direction = direction * 0.5
                pos2, lin2 = mid, linmid
            else:
                raise ValueError("gradient and value seem inconsistent")
            pos = nxt
            check_linear_operator(linmid.jac, domain_dtype=pos.dtype, target_dtype=dirder.dtype,
                                  only_r_linear=only_r_differentiable, atol=tol, rtol=tol, _device_ids_override=[dev])


# ------------------------------------------------------------------------------------------------
# minisanity (reference extra.py:552-758)
# ------------------------------------------------------------------------------------------------
class _Running:
    """Mean and unbiased variance of a stream of numbers (reference probing.py:24-80, Welford)."""

    def __init__(self):
        self.n, self.mean, self.m2 = 0, 0.0, 0.0

    def add(self, v):
        self.n += 1
        d = v - self.mean
        self.mean += d / self.n
        self.m2 += d * (v - self.mean)

    def result(self):
        std = np.sqrt(self.m2 / (self.n - 1)) if self.n > 1 else None  # (complex residuals: the reference's complex "variance")
        return {"mean": self.mean, "std": float(std) if std is not None and np.isrealobj(std) else std}


def _key_stats(field):
    """(sum, sum of squares, #ignored, size) of one Field, ignoring NaN and exact zeros; device Fields are reduced by
    libniftyk (nk_stats), host Fields by numpy."""
    from . import backend as B

    v = field.val
    if v.is_cuda and not v.is_complex():
        s, s2, nign = B.stats(v)
        return s, s2, nign, v.numel()
    a = field.asnumpy()
    nign = int(np.sum(np.isnan(a)) + np.sum(a == 0))
    return np.nansum(a), float(np.nansum(np.abs(a) ** 2)), nign, a.size


def minisanity(likelihood_energy, samples, terminal_colors=True, return_values=False):
    """Table of reduced chi^2, mean and degrees of freedom of the normalised data residuals and of the latent
    variables, averaged over the samples (same arguments, table layout and return values as the reference,
    extra.py:552-723).  Values far from 1 (chi^2) / 0 (mean) point at a bad fit or a mis-specified prior."""
    from .domains import MultiDomain
    from .energy_operators import LikelihoodEnergyOperator
    from .kl import SampleListBase

    if not isinstance(samples, SampleListBase):
        raise TypeError("Minisanity takes only SampleLists as input. Wrap a single field via `ift.SampleList([field])`.")
    if not isinstance(likelihood_energy, LikelihoodEnergyOperator):
        return ""
    name = likelihood_energy.name if likelihood_energy.name is not None else "<None>"

    def as_dict(f, single_key):
        return {k: f[k] for k in f.keys()} if isinstance(f.domain, MultiDomain) else {single_key: f}

    # per local sample and key: (reduced chi^2, mean, ndof, #ignored); distributed sample lists exchange these few
    # numbers so that every rank reports the statistics of ALL samples in global sample order
    local = []
    for x in samples.local_iterator():
        parts = (as_dict(likelihood_energy.normalized_residual(x), name), as_dict(x, "<None>"))
        row = []
        for part in parts:
            ent = {}
            for k in sorted(part):
                s, s2, nign, size = _key_stats(part[k])
                ndof = size - nign
                ent[k] = (s2 / ndof if ndof > 0 else s2, s / ndof if ndof > 0 else s, ndof, nign)
            row.append(ent)
        local.append(row)
    comm = getattr(samples, "_comm", None)
    if comm is not None and comm.size > 1:
        local = [row for part in comm.allgather_object(local) for row in part]
    groups = [{}, {}]  # data residuals, latent variables: key -> [chi^2 accumulator, mean accumulator, ndof, nign]
    for row in local:
        for grp, ent in zip(groups, row):
            for k, (chi, mean, ndof, nign) in ent.items():
                acc = grp.setdefault(k, [_Running(), _Running(), 0, 0])
                acc[0].add(chi)
                acc[1].add(mean)
                acc[2], acc[3] = ndof, nign
    keylen = min(max([18] + [len(k) for g in groups for k in g]), 42)
    col = (lambda c: c) if terminal_colors else (lambda c: "")
    warn, fail, bold, endc = col("\033[33m"), col("\033[31m"), col("\033[1m"), col("\033[0m")

    def table(grp):
        lines = []
        for k, (chi, mean, ndof, nign) in grp.items():
            chi, mean = chi.result(), mean.result()
            out = "  " + (k[:keylen - 1] + "…" if len(k) > keylen else k.ljust(keylen))
            foo = f"{chi['mean']:.1f}" + ("" if chi["std"] is None else f" ± {chi['std']:.1f}")
            if chi["mean"] > 5 or chi["mean"] < 1 / 5:
                out += fail + bold + f"{foo:>11}" + endc
            elif chi["mean"] > 2 or chi["mean"] < 1 / 2:
                out += warn + bold + f"{foo:>11}" + endc
            else:
                out += f"{foo:>11}"
            foo = f"{mean['mean']:.1f}" + ("" if mean["std"] is None else f" ± {mean['std']:.1f}")
            out += f"{foo:>14}" + f"{ndof:>11}" + f"{'-' if nign == 0 else nign:>11}"
            lines.append(out)
        return "\n".join(lines)

    n = 49 + keylen
    head = (keylen + 2) * " " + "{:>11}".format("reduced χ²") + "{:>14}".format("mean") + "{:>11}".format("# dof") \
        + "{:>11}".format("# ign. dof")
    res = "\n".join([n * "=", head, n * "-", "Data residuals", table(groups[0]), "Latent space", table(groups[1]), n * "="])
    if not return_values:
        return res
    pick = lambda g, i: {k: (v[i].result() if i < 2 else v[i]) for k, v in g.items()}  # noqa: E731
    values = {"redchisq": {"data_residuals": pick(groups[0], 0), "latent_variables": pick(groups[1], 0)},
              "scmean": {"data_residuals": pick(groups[0], 1), "latent_variables": pick(groups[1], 1)},
              "ndof": {"data_residuals": pick(groups[0], 2), "latent_variables": pick(groups[1], 2)},
              "nigndof": {"data_residuals": pick(groups[0], 3), "latent_variables": pick(groups[1], 3)}}
    return res, values
