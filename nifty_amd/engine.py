"""Fused MGVI/geoVI engine: CorrelatedField model + likelihood + sampled KL on one GPU.

This is the hot path of BASELINE.json's north_star with every N-sized operation fused into the
transform kernels (nk_hartley_fused prologue/epilogue):

    value+gradient of one sample  = amp_forward | HT[a(k) xi -> likelihood epilogue] | HT[dE/ds -> VJP epilogue] | amp_vjp
    metric application, 1 sample  = amp_jvp | HT[a dxi + da xi -> x metric weight] | HT[-> VJP epilogue (+d)] | amp_vjp

i.e. exactly one forward and one adjoint Hartley transform per sample (the reference re-linearises and
needs 2 + 1), no other N-sized pass, and the sum over local samples accumulates inside the
VJP epilogue.  It implements what the reference computes in
nifty/cl/library/correlated_fields.py:713-764 (finalize), operators/energy_operators.py:517-640,
890-931 (Gaussian/Poisson energy, StandardHamiltonian), operators/sampling_enabler.py:64-86 and
minimization/kl_energies.py:91-159, 299-360, behind the Energy protocol of minimization.py.

Latent vectors are ``LatentVec``: ``xi`` (the harmonic-space excitations, field dtype) and ``small``
(float64: asperity, flexibility, fluctuations, loglogavgslope, zeromode, spectrum[2, nb-2]).
"""
import contextlib
import ctypes
import math
import os
import threading

import numpy as np
import torch

from . import _lib as L
from . import backend as B
from . import batched, parallel, random
from .domains import PowerSpace, RGSpace
from .minimization import ConjugateGradient, Energy, QuadraticEnergy

SMALL_KEYS = ("asperity", "flexibility", "fluctuations", "loglogavgslope", "zeromode")
LATENT_KEYS = ("asperity", "flexibility", "fluctuations", "loglogavgslope", "spectrum", "xi", "zeromode")
_NONLIN = {None: L.NL_ID, "": L.NL_ID, "exp": L.NL_EXP, "sigmoid": L.NL_SIGMOID}
# grids up to this size draw the spectrum excitations of a prior sample on the device (FusedModel.draw_prior)
SPECTRUM_DEVICE_DRAW_MAX_POINTS = int(os.environ.get("NK_SPECTRUM_DEVICE_DRAW_MAX_POINTS", str(1 << 25)))


def lognormal_moments(mean, sigma):
    """reference nifty/cl/utilities.py:500-513"""
    checked = {}
    for name, value in (("mean", mean), ("sig", sigma)):
        checked[name] = float(value)
        if not checked[name] > 0:
            raise ValueError(f"{name} must be greater 0; got {checked[name]!r}")
    # a log-normal with these moments: variance of the log = log(1 + (sigma/mean)^2), mean of the log below the log-mean
    logsigma = math.sqrt(math.log1p((checked["sig"] / checked["mean"]) ** 2))
    return math.log(checked["mean"]) - logsigma ** 2 / 2, logsigma  # (the reference's rounding: the square of the root)


# ------------------------------------------------------------------------------------------------
# latent vectors
# ------------------------------------------------------------------------------------------------
class CgWorkspace:
    """Device-resident CG scalars (see nk_cg_* in include/niftyk.h)."""

    def __init__(self, device):
        self.scal = torch.zeros(8, dtype=torch.float64, device=device)
        self._host = torch.empty(8, dtype=torch.float64, pin_memory=True) if device.type == "cuda" else None
        # which reduction slots are known to be zero (fresh; the update's slots also after the roll of nk_cg_direction):
        # their next reduction accumulates instead of paying a memset launch first
        self._clean = {"curv": True, "update": True}

    def set_gamma(self, gamma):
        self.scal[0] = gamma

    def _first(self, which):
        """`accumulate` flag of the FIRST segment of a reduction into the slots `which` (0: the kernel zeroes them first)"""
        clean, self._clean[which] = self._clean[which], False
        return 1 if clean else 0

    def _segments(self, *vecs):
        return [[v.xi.reshape(-1) if v is not None else None for v in vecs],
                [v.small if v is not None else None for v in vecs]]

    def curv(self, d, q):
        lib, st = L.load(), B._stream()
        for i, (dd, qq) in enumerate(self._segments(d, q)):
            L.check(lib.nk_cg_curv(dd.numel(), dd.data_ptr(), qq.data_ptr(), B.dtype_code(dd), self.scal.data_ptr(),
                                   i if i else self._first("curv"), st))
        parallel.lockstep_sync_(self.scal[1:2])

    def curv_slot(self):
        """Zeroed device slot of d.q for an operator that deposits the xi part itself (``fused_dot``)."""
        slot = self.scal[1:2]
        if not self._first("curv"):
            slot.zero_()
        return slot

    def curv_small(self, d, q):
        """Adds the small-part of d.q to the slot an operator has already filled with the xi part."""
        L.check(L.load().nk_cg_curv(d.small.numel(), d.small.data_ptr(), q.small.data_ptr(), B.dtype_code(d.small),
                                    self.scal.data_ptr(), 1, B._stream()))
        parallel.lockstep_sync_(self.scal[1:2])

    def update(self, x, r, d, q, b):
        lib, st = L.load(), B._stream()
        for i, (xx, rr, dd, qq, bb) in enumerate(self._segments(x, r, d, q, b)):
            L.check(lib.nk_cg_update(xx.numel(), xx.data_ptr(), rr.data_ptr(), dd.data_ptr(), qq.data_ptr(), B.ptr(bb),
                                     B.dtype_code(xx), self.scal.data_ptr(), i if i else self._first("update"), st))
        parallel.lockstep_sync_(self.scal[2:5])

    def update_dr(self, x, r, d, q):
        """x -= alpha d, r -= alpha q; gamma = r.r and d.r (old residual) land in scal[2], scal[3] (nk_cg_update_dr)."""
        lib, st = L.load(), B._stream()
        for i, (xx, rr, dd, qq) in enumerate(self._segments(x, r, d, q)):
            L.check(lib.nk_cg_update_dr(xx.numel(), xx.data_ptr(), rr.data_ptr(), dd.data_ptr(), qq.data_ptr(),
                                        B.dtype_code(xx), self.scal.data_ptr(), i if i else self._first("update"), st))
        parallel.lockstep_sync_(self.scal[2:4])

    def refresh(self, x, r, b):
        """After a residual refresh: gamma = r.r, x.r and x.b recomputed (device side)."""
        self._clean["update"] = False
        for slot, (u, v) in ((2, (r, r)), (3, (x, r)), (4, (x, b))):
            if v is None:
                continue
            res = self.scal[slot:slot + 1]
            B.vdot(u.xi.reshape(-1), v.xi.reshape(-1), result=res, accumulate=False)
            B.vdot(u.small, v.small, result=res, accumulate=True)
            parallel.lockstep_sync_(res)

    def direction_small(self, d, r):
        """d.small <- beta d.small + r.small WITHOUT rolling the scalars: the xi part follows inside the next metric
        application (nk_fuse.cg_r), which then calls ``roll``."""
        L.check(L.load().nk_cg_direction(d.small.numel(), d.small.data_ptr(), r.small.data_ptr(), B.dtype_code(d.small),
                                         self.scal.data_ptr(), 0, B._stream()))

    def roll(self):
        L.check(L.load().nk_cg_direction(0, 0, 0, L.NK_F64, self.scal.data_ptr(), 1, B._stream()))
        self._clean["update"] = True  # (the roll leaves the slots of the vector update at zero)

    def direction(self, d, r):
        lib, st = L.load(), B._stream()
        segs = self._segments(d, r)
        for i, (dd, rr) in enumerate(segs):
            L.check(lib.nk_cg_direction(dd.numel(), dd.data_ptr(), rr.data_ptr(), B.dtype_code(dd), self.scal.data_ptr(),
                                        1 if i == len(segs) - 1 else 0, st))
        self._clean["update"] = True

    def fetch_begin(self):
        """Enqueue the copy of the scalars to the host and mark the spot in the stream: kernels enqueued after this call
        (the next iteration's direction update) run while the host waits for and looks at the scalars in fetch_end()."""
        self._host.copy_(self.scal, non_blocking=True)
        if getattr(self, "_landed", None) is None:
            self._landed = torch.cuda.Event()
        self._landed.record(torch.cuda.current_stream(self.scal.device))

    def fetch(self):
        self.fetch_begin()
        return self.fetch_end()

    def fetch_end(self):
        self._landed.synchronize()
        s = self._host.numpy()
        # inside a lockstep scope: these five scalars steer the CG -- ONE agreement check per iteration (parallel.lockstep_flush)
        parallel.lockstep_note(s[:5])
        parallel.lockstep_flush()
        return dict(gamma_prev=float(s[0]), curv=float(s[1]), gamma=float(s[2]), xr=float(s[3]), xb=float(s[4]), dr=float(s[3]),
                    alpha=float(s[0] / s[1]) if s[1] != 0 else float("nan"))


class ShardedCgWorkspace(CgWorkspace):
    """CG scalars when every rank holds 1/size of xi (and a replicated copy of the small part): the xi partial sums
    are all-reduced (one small collective per reduction), then the replicated small part is added on every rank.

    `units` = (unit length, units of this rank, units of the full vector, units per segment, segment stride, segment
    offset) when the share consists of whole reduction UNITS of the full vector (include/niftyk.h, nk_red_layout): the
    kernels then deliver the unit sums of this rank, the ranks exchange them (every unit is non-zero on exactly one rank:
    the all-reduce is exact) and add them in unit order -- the bits of the single-process reduction of the whole vector,
    whatever the number of ranks.  Without it the per-rank totals are all-reduced (rank-count dependent rounding)."""

    def __init__(self, device, comm, units=None):
        super().__init__(device)
        self.comm = comm
        self._broadcast = parallel._lockstep_mode() == "broadcast"
        self._units = units
        self._unit_sums = None if units is None else torch.zeros(4 * units[2], dtype=torch.float64, device=device)

    def _agree(self, t):
        """The replicated part of a reduction is computed per rank by fixed-order kernels on identical data: identical bits.
        NK_LOCKSTEP=broadcast forces rank 0's copy anyway (one tiny collective each, as in rounds 1-3); by default the
        scalars are compared once per iteration after their fetch (CgWorkspace.fetch)."""
        if self._broadcast:
            self.comm.bcast_(t)

    def _over_ranks(self, slot, count, launch):
        """`launch()` reduces the xi share into scal[slot : slot + count]; afterwards those scalars hold the sums over the
        full vector on every rank."""
        res = self.scal[slot:slot + count]
        if self._units is None:
            launch()
            self.comm.allreduce_sum_([res])
            return res
        lib, k = L.load(), self._units[2]
        L.check(lib.nk_red_layout(*self._units, self._unit_sums.data_ptr()), "nk_red_layout")
        try:
            launch()
        finally:
            lib.nk_red_layout(0, 0, 0, 0, 0, 0, 0)
        sums = self._unit_sums[:count * k]
        self.comm.allreduce_sum_([sums])
        L.check(lib.nk_red_finish(sums.data_ptr(), k, count, res.data_ptr(), 0, B._stream()), "nk_red_finish")
        return res

    def dot(self, u, v, slot):
        res = self._over_ranks(slot, 1, lambda: B.vdot(u.xi.reshape(-1), v.xi.reshape(-1), result=self.scal[slot:slot + 1],
                                                       accumulate=False))
        B.vdot(u.small, v.small, result=res, accumulate=True)
        self._agree(res)  # the replicated small part is reduced per rank
        return res

    def curv(self, d, q):
        lib, st = L.load(), B._stream()
        (dx, qx), (ds, qs) = self._segments(d, q)
        self._over_ranks(1, 1, lambda: L.check(lib.nk_cg_curv(dx.numel(), dx.data_ptr(), qx.data_ptr(), B.dtype_code(dx),
                                                              self.scal.data_ptr(), 0, st)))
        L.check(lib.nk_cg_curv(ds.numel(), ds.data_ptr(), qs.data_ptr(), B.dtype_code(ds), self.scal.data_ptr(), 1, st))
        self._agree(self.scal[1:2])

    def update(self, x, r, d, q, b):
        lib, st = L.load(), B._stream()
        xi, small = ([t for t in seg] for seg in self._segments(x, r, d, q, b))

        def run(seg, accumulate):
            xx, rr, dd, qq, bb = seg
            L.check(lib.nk_cg_update(xx.numel(), xx.data_ptr(), rr.data_ptr(), dd.data_ptr(), qq.data_ptr(), B.ptr(bb),
                                     B.dtype_code(xx), self.scal.data_ptr(), accumulate, st))

        self._over_ranks(2, 3, lambda: run(xi, 0))
        run(small, 1)
        self._agree(self.scal[2:5])

    def update_dr(self, x, r, d, q):
        lib, st = L.load(), B._stream()
        xi, small = ([t for t in seg] for seg in self._segments(x, r, d, q))

        def run(seg, accumulate):
            xx, rr, dd, qq = seg
            L.check(lib.nk_cg_update_dr(xx.numel(), xx.data_ptr(), rr.data_ptr(), dd.data_ptr(), qq.data_ptr(),
                                        B.dtype_code(xx), self.scal.data_ptr(), accumulate, st))

        self._over_ranks(2, 2, lambda: run(xi, 0))
        run(small, 1)
        self._agree(self.scal[2:4])

    def refresh(self, x, r, b):
        for slot, (u, v) in ((2, (r, r)), (3, (x, r)), (4, (x, b))):
            if v is not None:
                self.dot(u, v, slot)


class LatentVec:
    """A point / tangent / cotangent of the latent space (vector protocol of minimization.py)."""

    __slots__ = ("xi", "small", "sqnorm")

    def __init__(self, xi, small, sqnorm=None):
        self.xi, self.small = xi, small
        self.sqnorm = sqnorm  # optional device double: |self|^2, taken while the vector was written (shifted())

    # -- construction -------------------------------------------------------------------------
    @staticmethod
    def zeros(model):
        return LatentVec(torch.zeros(model.shape, dtype=model.tdtype, device=model.device),
                         torch.zeros(model.nsmall, dtype=torch.float64, device=model.device))

    @staticmethod
    def from_dict(model, dct):
        """Host dict (numpy arrays keyed like the reference's MultiField) -> device LatentVec."""
        nb = model.nb
        small = np.concatenate([[float(np.asarray(dct[k])) for k in SMALL_KEYS],
                                np.asarray(dct["spectrum"], dtype=np.float64).reshape(2 * (nb - 2))])
        xi = torch.from_numpy(np.ascontiguousarray(np.asarray(dct["xi"]).reshape(model.shape))).to(model.tdtype)
        return LatentVec(xi.to(model.device), torch.from_numpy(small).to(model.device))

    def to_dict(self):
        s = self.small.cpu().numpy()
        out = {k: np.array(s[i]) for i, k in enumerate(SMALL_KEYS)}
        out["spectrum"] = s[5:].reshape(2, -1).copy()
        out["xi"] = self.xi.cpu().numpy()
        return out

    def clone(self):
        return LatentVec(self.xi.clone(), self.small.clone())

    def cg_workspace(self):
        return CgWorkspace(self.xi.device)

    # -- vector protocol -----------------------------------------------------------------------
    def shifted(self, sign, other):
        """self + sign * other together with its squared norm (one pass over xi): a KL sample position p +- r and the prior
        term of its Hamiltonian."""
        sq = torch.zeros(1, dtype=torch.float64, device=self.xi.device)
        xi = B.axpby_sqnorm(1.0, self.xi, float(sign), other.xi, sq)
        small = B.axpby_sqnorm(1.0, self.small, float(sign), other.small, sq, accumulate=True)
        return LatentVec(xi, small, sq)

    def _lin(self, alpha, other, beta):
        """alpha*self + beta*other"""
        return LatentVec(B.axpby(alpha, self.xi, beta, other.xi), B.axpby(alpha, self.small, beta, other.small))

    def __add__(self, o):
        return self._lin(1.0, o, 1.0)

    def __sub__(self, o):
        return self._lin(1.0, o, -1.0)

    def __mul__(self, a):
        if not np.isscalar(a):
            return NotImplemented
        return LatentVec(B.axpby(float(a), self.xi), B.axpby(float(a), self.small))

    __rmul__ = __mul__

    def __neg__(self):
        return self * (-1.0)

    def axpy(self, a, x):
        """self + a*x"""
        return x._lin(float(a), self, 1.0)

    def dot_device(self, o, result=None):
        res = B.vdot(self.xi.reshape(-1), o.xi.reshape(-1), result=result, accumulate=result is not None)
        return B.vdot(self.small, o.small, result=res, accumulate=True)

    def s_vdot(self, o):
        return float(self.dot_device(o).item())  # rank-local; minimisers synchronise their decisions (minimization._ls)

    def norm(self, ord=2):
        if ord != 2:
            raise NotImplementedError
        return math.sqrt(self.s_vdot(self))


# ------------------------------------------------------------------------------------------------
# the fused model
# ------------------------------------------------------------------------------------------------
class _Dest:
    """Where the xi part of ONE sample's contribution to a sum over samples goes (FusedModel._vjp): `xi` is stored into, or
    -- `accumulate` -- holds a partial sum the contribution joins; `carries` are further partial sums that join on the way,
    innermost first (nk_fuse.carry1 / carry2).  `small`: list that receives the sample's small part (summed by the caller),
    or None = the old running sum in the output vector.  `after`: (target, source) additions still to be done explicitly."""

    __slots__ = ("xi", "accumulate", "carries", "small", "after", "fresh")

    def __init__(self, xi, accumulate=False, carries=(), small=None, after=(), fresh=False):
        self.xi, self.accumulate, self.carries, self.small, self.after, self.fresh = xi, accumulate, carries, small, after, fresh


class _PairTree:
    """Bookkeeping of a sum over this rank's samples in the order of parallel.pair_tree (utilities.py:349-414) WITHOUT
    keeping the terms: a binary counter of partial sums.  Sample i joins the partial sums that end at i -- one per
    trailing one-bit of i, the last sample all that are left -- innermost first, inside its own VJP epilogue (`_Dest`:
    up to MAX_CARRIES partial sums ride along as carries, deeper ones are added explicitly afterwards).  Eight samples
    need the output and two scratch vectors, and move the bytes of the plain running sum: the even samples store instead
    of read-modify-write, samples 3 and 7 read one / two partial sums more."""

    MAX_CARRIES = 2

    def __init__(self, out, scratch):
        self._stack = []        # (number of samples, tensor): sizes decrease towards the top
        self._first = out       # the first fresh vector is the output: the deepest partial sum ends up there
        self._pool = scratch    # callable -> a scratch vector shaped like `out`
        self._free = []

    def place(self, last, small=None):
        size, joined = 1, []
        while self._stack and (last or self._stack[-1][0] == size):
            n, t = self._stack.pop()
            joined.append(t)
            size += n
        if not joined:
            t = self._first if self._first is not None else (self._free.pop() if self._free else self._pool())
            self._first = None
            self._stack.append((1, t))
            return _Dest(t, small=small, fresh=True)
        ride = joined[:self.MAX_CARRIES] if len(joined) > 1 else []
        target = joined[len(ride)] if len(joined) > len(ride) else None
        if target is None:  # (exactly MAX_CARRIES partial sums: the deepest one is the running sum)
            target, ride = ride[-1], ride[:-1]
        rest = joined[len(ride) + 1:]
        after, inner = [], target
        for t in rest:  # deeper partial sums than the epilogue can take: t <- t + inner, explicitly
            after.append((t, inner))
            inner = t
        self._free += ride + ([target] + rest[:-1] if rest else [])
        self._stack.append((size, inner))
        return _Dest(target, accumulate=True, carries=tuple(ride), small=small, after=tuple(after))

    @staticmethod
    def settle(dest):
        for target, source in dest.after:
            B.axpby(1.0, target, 1.0, source, out=target)

    def total(self):
        if len(self._stack) != 1:
            raise RuntimeError("_PairTree.total: the last sample has not been placed")
        return self._stack[0][1]


class LinPoint:
    """Everything cached about one latent point: amplitude tables and the s-space metric weight."""

    __slots__ = ("x", "amp", "afield", "state", "mid", "mid_scalar", "value", "grad", "f", "tf", "gp", "wd", "tfd")

    def __init__(self):
        self.mid = None
        self.afield = None
        self.mid_scalar = 1.0
        self.f = self.tf = None
        self.gp = self.wd = self.tfd = None  # response models: g'(s) [grid], data-space metric weight, df/dmu [data]
        self.value = self.grad = None  # (stay None for a FusedModel.metric_point)


class FusedModel:
    """offset + HT(a[pindex] xi) -> nonlinearity [-> linear response] -> Gaussian / Poisson likelihood, plus the standard
    prior.  ``response`` (optional): a sparse linear map signal space -> data space with ``times`` / ``adjoint`` /
    ``n_data`` (los_response.SparseResponse: masked LOSResponse, BASELINE config 4); the likelihood then lives on the
    n_data-sized data space, and a metric application is JVP transform (x g'), R, data-space weight, R^T, (x g') VJP
    transform -- two three-pass transforms and two nk_csr_rowsum launches."""

    def __init__(self, shape, distances=None, *, offset_mean=0.0, offset_std=(1e-1, 3e-2), fluctuations=(1.0, 5e-1),
                 loglogavgslope=(-3.0, 2e-1), flexibility=(1.0, 2e-1), asperity=(5e-1, 5e-2),
                 likelihood="gaussian", data=None, icov=1.0, nonlin=None, response=None, dtype=torch.float64,
                 device="cuda:0"):
        L.load()  # fail loudly without the HIP extension
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FusedModel runs on a GPU only (no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self.device.index != torch.cuda.current_device():
            # every kernel is launched on the CURRENT device's stream (backend._stream)
            raise RuntimeError(f"FusedModel(device={self.device}): make it the current device first "
                               f"(torch.cuda.set_device / `with torch.cuda.device({self.device.index})`)")
        self.tdtype = dtype
        self.npdtype = np.float32 if dtype == torch.float32 else np.float64
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.N = int(np.prod(self.shape))
        pos = RGSpace(self.shape, distances)
        self.position_space = pos
        self.harmonic_space = hsp = pos.get_default_codomain()
        self.power_space = ps = PowerSpace(hsp)
        self.nb = nb = ps.shape[0]
        if nb < 3:
            raise ValueError("need at least 3 power bins")
        self.nsmall = 5 + 2 * (nb - 2)
        self.V = pos.total_volume
        self.h_dvol = hsp.scalar_dvol
        self.offset_mean = float(offset_mean)
        with torch.cuda.device(self.device):
            self.pidx = ps.device_pindex(self.device)
            logk = np.log(ps.k_lengths[1:])
            rel = np.insert(logk - logk[0], 0, 0.0)
            delta = np.concatenate([logk[1:] - logk[:-1], [0.0, 0.0]])
            mult = ps.rho.astype(np.float64).copy()
            mult[0] = 0.0
            geo = np.concatenate([rel, rel / rel[-1], mult, delta])
            hyp = np.array([*lognormal_moments(*fluctuations), *lognormal_moments(*flexibility),
                            *lognormal_moments(*asperity), *lognormal_moments(*offset_std),
                            float(loglogavgslope[0]), float(loglogavgslope[1]), self.V])
            self.geo = torch.from_numpy(geo).to(self.device)
            self.hyp = torch.from_numpy(hyp).to(self.device)
            self.plan = B.get_plan(self.shape, dtype, 1, self.device)
            # VJP scatter accumulators: one private copy per XCD (L2-scope atomics), folded into self.abar
            self.abar_stride = (nb + 31) // 32 * 32
            self.abar_copies = 8
            self.scatter_scratch = torch.empty(64 * (nb + 32), dtype=torch.float64, device=self.device)
            self.wfull = self.full_plan = None  # plans without the octant pipeline: see below
            self.abar_priv = torch.zeros(self.abar_copies * self.abar_stride, dtype=torch.float64, device=self.device)
            self.abar = torch.zeros(nb, dtype=torch.float64, device=self.device)
            self.damp = torch.empty(nb, dtype=torch.float64, device=self.device)
            self.latbar = torch.empty(self.nsmall, dtype=torch.float64, device=self.device)
            self.tmp = torch.empty(self.shape, dtype=dtype, device=self.device)
            # octant arrays: the VJP scatter sums (sign-flip images merged in the final transform pass) and the
            # amplitude fields a[pindex], da[pindex] (identical on all sign-flip images: 1/8 of the bytes)
            self.octant_vjp = bool(L.load().nk_plan_octant_vjp(self.plan.handle))
            # metric applications as ONE five-pass Hartley sandwich (NK_SANDWICH=0: two three-pass transforms).  Default
            # for 3-D grids only: in 2-D it replaces four passes by three, but the fused pass with its two line transforms
            # costs more than the pass it saves (2048^2 fp64 Poisson: 0.197 vs 0.162 ms per metric application;
            # NK_SANDWICH=2 forces it for every eligible grid)
            sw = os.environ.get("NK_SANDWICH", "1")
            self.sandwich = (self.octant_vjp and B.plan_sandwich(self.plan) and sw != "0"
                             and (len(self.shape) == 3 or sw == "2"))
            # the CG search-direction update d <- beta d + r rides in the first pass of the next metric application
            self.fused_direction = self.sandwich and os.environ.get("NK_CG_FUSED_DIRECTION", "1") != "0"
            oct_n = int(np.prod([n // 2 + 1 for n in self.shape]))
            self.field_shape = (oct_n,) if self.octant_vjp else self.shape
            # bin index of the octant points as its own contiguous array: octant fields are plain gathers from it
            self.pidx8 = None
            if self.octant_vjp:
                osl = tuple(slice(0, n // 2 + 1) for n in self.shape)
                self.pidx8 = self.pidx.view(self.shape)[osl].contiguous().reshape(-1)
            self.dafield = torch.empty(self.field_shape, dtype=dtype, device=self.device)
            self.w8 = torch.empty(oct_n, dtype=torch.float64, device=self.device) if self.octant_vjp else None
            self.w8max = torch.zeros(1, dtype=torch.float64, device=self.device)
            self.merge_swapped = int(len(self.shape) == 3 and self.shape[0] == self.shape[1]
                                     and hsp.distances[0] == hsp.distances[1])
            # natural binning on an equal-distance grid: bins are the ascending distinct integer k^2, which lets the
            # octant sums be reduced shell by shell in LDS (nk_octant_scatter_k2) instead of with global atomics
            self.bin_k2 = None
            # grids without the shell structure below (2-D, 1-D, non-natural binning, unequal distances): the octant points
            # sorted by bin, once -- every bin is then summed in a fixed order by nk_csr_rowsum instead of with global fp64
            # atomics (NK_SEGMENT_SUM=0: atomics, for A/B)
            self.seg_plan = None
            # (3-D only: on a 2-D quadrant plain atomics win over the shell walk, 2048^2: 49 vs 78 us, 4096^2: 184 vs 318 us --
            # tools/gpu_scatter2d_probe.py)
            if self.octant_vjp and len(self.shape) == 3 and ps._data.get("k2table") is not None and max(self.shape) >= 64:
                k2 = np.nonzero(hsp._k2_flags())[0].astype(np.int32)
                if len(k2) == nb and int(k2[-1]) < (1 << 24):
                    self.bin_k2 = torch.from_numpy(k2).to(self.device)
            # fixed-point accumulation of the shell scatter: a workgroup = (shell of 4096 bins, first-axis index mod 16) must
            # add < 2^18 points (overflow bound of the 64-bit sums) -- counted exactly here, once (a cube heuristic missed
            # anisotropic grids); beyond the bound the floating-point LDS atomics take over
            self.scatter_fixed_point = False
            if self.bin_k2 is not None:
                a_idx = torch.arange(self.shape[0] // 2 + 1, device=self.device, dtype=torch.int64)
                key = (self.pidx8.view(self.shape[0] // 2 + 1, -1) // 4096).to(torch.int64) * 16 + (a_idx % 16)[:, None]
                self.scatter_busiest = int(torch.bincount(key.reshape(-1)).max().item())
                self.scatter_fixed_point = self.scatter_busiest < (1 << 18)
                del key, a_idx
            self.k2_dense = self.k2_line_order = None
            if self.bin_k2 is not None and os.environ.get("NK_EXPAND_K2", "1") != "0":
                self.k2_dense = torch.zeros(int(self.bin_k2[-1].item()) + 1, dtype=dtype, device=self.device)
                if len(self.shape) == 3 and os.environ.get("NK_EXPAND_ORDER", "1") != "0":
                    # octant lines (a, b) sorted by a^2 + b^2: neighbours in this order read the same table entries
                    a2 = torch.arange(self.shape[0] // 2 + 1, device=self.device, dtype=torch.int64) ** 2
                    b2 = torch.arange(self.shape[1] // 2 + 1, device=self.device, dtype=torch.int64) ** 2
                    self.k2_line_order = torch.argsort((a2[:, None] + b2[None, :]).reshape(-1), stable=True).to(torch.int32)
            if self.octant_vjp and self.bin_k2 is None and os.environ.get("NK_SEGMENT_SUM", "1") != "0":
                self.seg_plan = B.bin_plan(self.pidx8, self.nb)
            # WIDE forward transform of value + gradient for fp32 fields (nk_fuse.io32): the reference promotes fp32
            # excitations to fp64 at their product with the fp64 amplitude and transforms in fp64
            # (library/correlated_fields.py:755-764).  An fp32 forward transform leaves a coherent gain error of ~6e-8 on the
            # signal; the residual N^-1 (s - d) amplifies it by sqrt(N) x signal-to-noise (1e-5 .. 1e-3 of the gradient at
            # 256^3 .. 1024^3).  So this one transform per evaluation runs on the fp64 plan -- fp64 amplitude field,
            # fp64 work array, residual and energy from the fp64 signal -- with the fp32 arrays at both ends; the adjoint and
            # every transform of a metric application stay fp32 (their error is not amplified: 1e-7).  NK_WIDE_FORWARD=0: the
            # all-fp32 evaluation, for A/B.
            self.wide = (dtype == torch.float32 and self.octant_vjp and os.environ.get("NK_WIDE_FORWARD", "1") != "0")
            # ... and on grids WITHOUT the register-resident pipeline (mixed radix, short axes) the same transform runs on the
            # generic fp64 plan with fp64 COPIES at both ends (xi widened, residual and metric weight rounded back): four
            # N x 8-byte arrays more, three extra streaming passes -- next to generic kernels at a third of the fast
            # path's bandwidth
            self.wide_generic = (dtype == torch.float32 and not self.octant_vjp
                                 and os.environ.get("NK_WIDE_FORWARD", "1") != "0")
            self._wide_state = None
            # plans WITHOUT the octant pipeline (mixed-radix grids, short axes): the generic kernels deposit xi . t per grid
            # point in a full-grid fp64 array (nk_fuse.wfull) that is summed bin by bin in a fixed order -- instead of fp64
            # atomics into per-XCD copies of abar (NK_SEGMENT_SUM=0: the atomics, for A/B)
            if (not self.octant_vjp and self.N <= B.BIN_PLAN_MAX and os.environ.get("NK_SEGMENT_SUM", "1") != "0"):
                self.full_plan = B.bin_plan(self.pidx, self.nb)
                self.wfull = torch.empty(self.N, dtype=torch.float64, device=self.device)
        # likelihood
        if likelihood not in ("gaussian", "poisson"):
            raise ValueError("likelihood must be 'gaussian' or 'poisson'")
        self.lh_kind = L.LH_GAUSS if likelihood == "gaussian" else L.LH_POISSON
        self.nonlin = _NONLIN[nonlin]
        self.icov_field, self.icov_scalar = None, 1.0
        self.response = response
        self.wide_response = False
        self.data_shape = self.shape if response is None else (int(response.n_data),)
        if response is not None:
            if int(response.n_pix) != self.N:
                raise ValueError("response does not act on this grid")
            if getattr(response, "grid_shape", 0) is None:
                response.grid_shape = tuple(self.shape)  # its columns are the points of THIS grid: TIMES may run tile by tile
            self.sandwich = self.fused_direction = False  # the middle of J^T M J is not diagonal in position space
            # the likelihood lives behind the response: no likelihood epilogue to widen.  fp32 fields take the same step
            # with fp64 copies instead (`_linearize_response`): xi widened, g(s) from the fp64 transform, the response, the
            # residual and the energy in fp64 on the data space; g'(s), the data-space weight and the residual that enters the
            # adjoint come back rounded to fp32
            self.wide_response = self.wide or self.wide_generic
            self.wide = self.wide_generic = False
        if data is not None:
            self.set_data(data, icov)
        self.counters = dict(value_grad=0, metric=0, transforms=0, cg_iterations=0)
        self._count_lock = threading.Lock()  # (lanes driven by host threads count into the same dictionary)

    # -- data ---------------------------------------------------------------------------------
    def set_data(self, data, icov=1.0):
        data = torch.as_tensor(data)
        if tuple(data.shape) != self.data_shape:
            raise ValueError("data shape mismatch")
        if self.lh_kind == L.LH_POISSON:
            if data.dtype.is_floating_point:
                raise TypeError("data is of invalid data-type; counts need to be integers")
            if bool((data < 0).any()):
                raise ValueError("count data is negative and thus can not be Poissonian")
            self.data = data.to(torch.int64).to(self.device).contiguous()
        else:
            self.data = data.to(self.tdtype).to(self.device).contiguous()
            if np.isscalar(icov):
                self.icov_scalar, self.icov_field = float(icov), None
            else:
                self.icov_field = torch.as_tensor(icov).to(self.tdtype).to(self.device).contiguous()
        self.const_mid = (self.lh_kind == L.LH_GAUSS and self.nonlin == L.NL_ID and self.icov_field is None
                          and self.response is None)
        # (static operands of the generic wide forward transform, shared by the lanes)
        self._data64 = self._icov64 = None
        if (self.wide_generic or self.wide_response) and self.lh_kind == L.LH_GAUSS:
            self._data64 = self.data.double()
            self._icov64 = None if self.icov_field is None else self.icov_field.double()
        # data-space metric weight of a Gaussian with scalar N^-1: a number, not a field (response models)
        self.const_wd = self.lh_kind == L.LH_GAUSS and self.icov_field is None

    # -- kernels --------------------------------------------------------------------------------
    def _count(self, key, n):
        with self._count_lock:
            self.counters[key] += n

    def _amp_forward(self, small):
        amp = torch.empty(self.nb, dtype=torch.float64, device=self.device)
        state = torch.empty(8 * self.nb + 16, dtype=torch.float64, device=self.device)
        L.check(L.load().nk_amp_forward(self.nb, self.geo.data_ptr(), self.hyp.data_ptr(), small.data_ptr(),
                                        state.data_ptr(), amp.data_ptr(), B._stream()), "nk_amp_forward")
        return amp, state

    def _amp_field(self, amp, out=None, dense=None):
        """table[pindex] materialised in the field dtype (or the dtype of `out`) with one gather per OCTANT point (|k| bins
        are invariant under the sign flip of every axis): every later prologue / epilogue streams the field instead of
        gathering.  dense: the k^2 scratch table matching out's dtype when it is not the field dtype."""
        out = torch.empty(self.field_shape, dtype=self.tdtype, device=self.device) if out is None else out
        if self.k2_dense is not None:
            # natural binning: the bin is a function of k^2 -- no index stream (nk_octant_expand_k2; NK_EXPAND_K2=0: gather)
            dense = self.k2_dense if dense is None else dense
            shp = (ctypes.c_int64 * len(self.shape))(*self.shape)
            L.check(L.load().nk_octant_expand_k2(len(self.shape), shp, amp.data_ptr(), self.bin_k2.data_ptr(), self.nb,
                                                 dense.data_ptr(), out.data_ptr(), B.dtype_code(out),
                                                 B.ptr(self.k2_line_order), B._stream()), "nk_octant_expand_k2")
            return out
        table = amp if out.dtype == torch.float64 else amp.to(out.dtype)
        if self.pidx8 is not None:
            return B.gather(table, self.pidx8, self.field_shape, out=out)
        shp = (ctypes.c_int64 * len(self.shape))(*self.shape)
        L.check(L.load().nk_octant_expand(len(self.shape), shp, table.data_ptr(), self.pidx.data_ptr(), out.data_ptr(),
                                          B.dtype_code(out), 1 if self.octant_vjp else 0, B._stream()), "nk_octant_expand")
        return out

    def _fuse(self):
        f = L.Fuse()
        f.scale = self.h_dvol
        f.mul_scalar = 1.0
        f.addend_scale = 1.0
        f.field_octant = 1 if self.octant_vjp else 0
        return f

    def signal(self, x, want_derivative=False):
        """nonlin(offset + HT(a xi)) as a device tensor (and optionally nonlin')."""
        amp, _ = self._amp_forward(x.small)
        f = self._fuse()
        f.pro, f.in_, f.pidx, f.amp = L.PRO_AMP, x.xi.data_ptr(), self.pidx.data_ptr(), amp.data_ptr()
        afield = self._amp_field(amp)
        f.afield = afield.data_ptr()
        out = torch.empty(self.shape, dtype=self.tdtype, device=self.device)
        d = torch.empty_like(out) if want_derivative else None
        f.epi, f.out, f.out2, f.offset, f.nonlin = L.EPI_NONLIN, out.data_ptr(), B.ptr(d), self.offset_mean, self.nonlin
        B.hartley_fused(self.plan, f)
        self._count("transforms", 1)
        return (out, d) if want_derivative else out

    def _vjp(self, lp, w, scale, addend, addend_scale, accumulate, out_xi, dot_out=None, w2=None, sandwich=None, carries=()):
        """out_xi (+)= a t + addend_scale*addend,  self.abar += scatter(xi t),  t = scale * HT(w) (HT(w * w2) with w2);
        carries: up to two partial sums of other samples that join out_xi innermost first (nk_fuse.carry1 / carry2);
        dot_out (device fp64 scalar, needs addend): += sum addend * out_xi, taken in the same epilogue.
        sandwich = (fill_prologue, scale_first, mid, mid_scalar): t = scale * HT(mid_scalar * mid . scale_first *
        HT(prologue)) through nk_hartley_sandwich instead (w is ignored)."""
        f = self._fuse()

        def run(f):
            if sandwich is None:
                B.hartley_fused(self.plan, f)
            else:
                B.hartley_sandwich(self.plan, f, sandwich[1])

        if dot_out is not None:
            if addend is None or not self.octant_vjp:
                raise ValueError("dot_out needs an addend and the register-resident transform pipeline")
            f.value = dot_out.data_ptr()
        if sandwich is not None:
            sandwich[0](f)
            f.mul, f.mul_scalar = B.ptr(sandwich[2]), sandwich[3]
        else:
            f.pro, f.in_ = L.PRO_PLAIN, w.data_ptr()
            if w2 is not None:
                f.pro, f.in2 = L.PRO_MUL, w2.data_ptr()
        f.epi, f.out, f.scale = L.EPI_VJP, out_xi.data_ptr(), self.h_dvol * scale
        f.pidx, f.amp, f.xi = self.pidx.data_ptr(), lp.amp.data_ptr(), lp.x.xi.data_ptr()
        f.afield = B.ptr(lp.afield)
        f.addend, f.addend_scale, f.accumulate = B.ptr(addend), addend_scale, 1 if accumulate else 0
        if len(carries) > 2:
            raise ValueError("the VJP epilogue takes at most two carried partial sums")
        f.carry1, f.carry2 = (B.ptr(c) for c in (tuple(carries) + (None, None))[:2])
        if self.octant_vjp:
            # the final pass stores one merged sum per octant point; nk_octant_scatter reduces them into the bins
            f.abar, f.w8 = self.abar.data_ptr(), self.w8.data_ptr()
            if self.scatter_fixed_point:  # max |w8| of this launch: the scale of the fixed-point shell scatter
                f.w8max = self.w8max.data_ptr()
            run(f)
            shp = (ctypes.c_int64 * len(self.shape))(*self.shape)
            if self.bin_k2 is not None:
                L.check(L.load().nk_octant_scatter_k2(len(self.shape), shp, self.w8.data_ptr(), self.pidx.data_ptr(),
                                                      self.bin_k2.data_ptr(), self.nb, self.scatter_scratch.data_ptr(),
                                                      self.abar.data_ptr(),
                                                      self.w8max.data_ptr() if self.scatter_fixed_point else 0, B._stream()),
                        "nk_octant_scatter_k2")
            elif self.seg_plan is not None:
                rowptr, perm, lanes = self.seg_plan
                L.check(L.load().nk_csr_rowsum(self.nb, rowptr.data_ptr(), perm.data_ptr(), 0, self.w8.data_ptr(),
                                               self.abar.data_ptr(), L.NK_F64, lanes, B._stream()), "nk_csr_rowsum")
            else:
                self.abar.zero_()
                L.check(L.load().nk_octant_scatter(len(self.shape), shp, self.w8.data_ptr(), self.pidx.data_ptr(),
                                                   self.abar.data_ptr(), self.merge_swapped, B._stream()),
                        "nk_octant_scatter")
        elif self.full_plan is not None:
            f.abar, f.wfull = self.abar.data_ptr(), self.wfull.data_ptr()
            run(f)
            rowptr, perm, lanes = self.full_plan
            L.check(L.load().nk_csr_rowsum(self.nb, rowptr.data_ptr(), perm.data_ptr(), 0, self.wfull.data_ptr(),
                                           self.abar.data_ptr(), L.NK_F64, lanes, B._stream()), "nk_csr_rowsum")
        else:
            f.abar, f.abar_copies, f.abar_stride = self.abar_priv.data_ptr(), self.abar_copies, self.abar_stride
            self.abar_priv.zero_()
            run(f)
            L.check(L.load().nk_fold_copies(self.nb, self.abar_copies, self.abar_stride, self.abar_priv.data_ptr(),
                                            self.abar.data_ptr(), B._stream()), "nk_fold_copies")
        self._count("transforms", 1 if sandwich is None else 2)

    def _amp_vjp(self, lp):
        L.check(L.load().nk_amp_vjp(self.nb, self.geo.data_ptr(), self.hyp.data_ptr(), lp.x.small.data_ptr(),
                                    lp.state.data_ptr(), self.abar.data_ptr(), self.latbar.data_ptr(), B._stream()),
                "nk_amp_vjp")

    def metric_point(self, x):
        """What metric applications at x need and nothing else -- amplitude tables / field / state and the s-space (or
        data-space) weights: the linearisation point of the SAMPLING solves (kl_energies.py:105-128 builds the metric at the
        mean there; its value and gradient are never asked for).  A Gaussian likelihood on the field itself has a constant
        weight: no transform at all; otherwise the forward transform with the likelihood epilogue fills the weight field and
        the adjoint transform of the gradient is skipped."""
        if self.response is None and self.const_mid:
            lp = LinPoint()
            lp.x = x
            lp.amp, lp.state = self._amp_forward(x.small)
            lp.afield = self._amp_field(lp.amp)
            lp.mid_scalar = self.icov_scalar
            return lp
        return self.linearize(x, want_gradient=False)

    def linearize(self, x, grad_acc=None, n_total=1, value_acc=None, dest=None, want_gradient=True):
        """Value and gradient of H = lh + 1/2|x|^2 at x, cached for metric applications.

        With ``grad_acc``/``value_acc`` the sample average is accumulated in place (1/n_total weights); with ``dest`` (a
        `_Dest` with a `small` list) the xi part of the weighted gradient goes where the pairwise sum over samples wants
        it and the small part is appended to the list (FusedKL).  want_gradient=False: only what metric applications need
        (`metric_point`): the adjoint transform of the gradient is not run, `value` / `grad` stay None.
        """
        lp = LinPoint()
        lp.x = x
        lp.amp, lp.state = self._amp_forward(x.small)
        lp.afield = self._amp_field(lp.amp)
        w = 1.0 / n_total
        value = torch.zeros(1, dtype=torch.float64, device=self.device) if value_acc is None else value_acc
        lhval = torch.zeros(1, dtype=torch.float64, device=self.device)
        if self.response is not None:
            return self._linearize_response(lp, x, grad_acc if dest is None else dest, w, value, lhval, want_gradient)
        f = self._fuse()
        f.pro, f.in_, f.pidx, f.amp = L.PRO_AMP, x.xi.data_ptr(), self.pidx.data_ptr(), lp.amp.data_ptr()
        f.afield = lp.afield.data_ptr()
        gs = self.tmp
        if not self.const_mid:
            lp.mid = torch.empty(self.shape, dtype=self.tdtype, device=self.device)
        else:
            lp.mid_scalar = self.icov_scalar
        f.epi, f.out, f.out2 = L.EPI_LIKELIHOOD, gs.data_ptr(), B.ptr(lp.mid)
        f.offset, f.lh_kind, f.nonlin = self.offset_mean, self.lh_kind, self.nonlin
        f.data, f.icov, f.icov_scalar, f.value = self.data.data_ptr(), B.ptr(self.icov_field), self.icov_scalar, lhval.data_ptr()
        if self.wide:
            # fp32 arrays at both ends of an fp64 transform (see __init__): a(k) xi(k) with the fp64 amplitude
            plan64, afield64, dense64 = self._wide_buffers()
            f.afield, f.io32 = self._amp_field(lp.amp, out=afield64, dense=dense64).data_ptr(), 1
            B.hartley_fused(plan64, f)
        elif self.wide_generic:
            # the same on the generic fp64 plan: xi widened, a(k) gathered from the fp64 table, fp64 likelihood operands;
            # the residual gradient and the metric weight come back rounded to the field type
            plan64, xi64, gs64, mid64 = self._wide_generic_buffers(lp.mid is not None)
            xi64.copy_(x.xi)
            f.in_, f.afield, f.out, f.out2 = xi64.data_ptr(), None, gs64.data_ptr(), B.ptr(mid64)
            if self.lh_kind == L.LH_GAUSS:
                f.data, f.icov = self._data64.data_ptr(), B.ptr(self._icov64)
            B.hartley_fused(plan64, f)
            gs.copy_(gs64)
            if lp.mid is not None:
                lp.mid.copy_(mid64)
        else:
            B.hartley_fused(self.plan, f)
        self._count("transforms", 1)
        if not want_gradient:
            return lp
        return self._finish_linearize(lp, x, gs, None, grad_acc if dest is None else dest, w, value, lhval)

    # -- lanes: independent scratch sets, so that the chains of several samples run on several streams at once ----------
    _SCRATCH = ("scatter_scratch", "abar_priv", "abar", "damp", "latbar", "tmp", "dafield", "w8", "w8max", "wfull", "k2_dense")

    def lanes(self, count):
        """[self, clone, clone, ...]: `count` views of this model that share every static array (bin index, geometry, data,
        plan tables, counters) and own what a transform chain scribbles on (workspace, octant sums, amplitude tangents ...),
        each with its own stream (None for the model itself = the caller's stream).  On small grids one sample's chain of
        ~15 short kernels leaves most of the chip idle (2048^2 fp64: 128 workgroups per pass on 256 CUs, 80 % of a step
        spent in kernels of 5-50 us); the samples of a KL evaluation are independent, so their chains run side by side."""
        import copy

        made = self.__dict__.setdefault("_lanes", [self])
        with torch.cuda.device(self.device):
            while len(made) < count:
                lane = copy.copy(self)
                lane.plan = B.PlanView(self.plan)
                for name in self._SCRATCH:
                    t = getattr(self, name)
                    setattr(lane, name, None if t is None else torch.zeros_like(t))
                lane._wide_state = lane._pair = None
                lane._lanes = [lane]
                lane.stream = torch.cuda.Stream(device=self.device)
                made.append(lane)
        return made[:count]

    stream = None

    def _wide_buffers(self):
        """fp64 plan (its work array: 2 N x 4 bytes more), fp64 octant amplitude field and k^2 table of the wide forward
        transform, created at the first value / gradient evaluation of an fp32 model."""
        if self._wide_state is None:
            plan64 = B.get_plan(self.shape, torch.float64, 1, self.device)
            if not L.load().nk_plan_octant_vjp(plan64.handle):
                raise RuntimeError("the fp64 plan of this grid has no register-resident pipeline")
            if self.stream is not None:  # a lane (FusedModel.lanes): the cached plan's WORKSPACE belongs to the main chain
                plan64 = B.PlanView(plan64)
            afield64 = torch.empty(self.field_shape, dtype=torch.float64, device=self.device)
            dense64 = None if self.k2_dense is None else torch.zeros_like(self.k2_dense, dtype=torch.float64)
            self._wide_state = (plan64, afield64, dense64)
        return self._wide_state

    def _wide_generic_buffers(self, with_mid):
        """fp64 plan and the fp64 copies at both ends of the wide forward transform on a grid without the register-resident
        pipeline (`wide_generic`), created at the first value / gradient evaluation."""
        if self._wide_state is None:
            plan64 = B.get_plan(self.shape, torch.float64, 1, self.device)
            if self.stream is not None:  # a lane: the cached plan's workspace belongs to the main chain
                plan64 = B.PlanView(plan64)
            xi64, gs64 = (torch.empty(self.shape, dtype=torch.float64, device=self.device) for _ in range(2))
            self._wide_state = [plan64, xi64, gs64, None]
        if with_mid and self._wide_state[3] is None:
            self._wide_state[3] = torch.empty(self.shape, dtype=torch.float64, device=self.device)
        return self._wide_state[0], self._wide_state[1], self._wide_state[2], self._wide_state[3] if with_mid else None

    def _finish_linearize(self, lp, x, gs, gs2, grad_acc, w, value, lhval):
        """Gradient J^T (gs * gs2) + x and the value lh + 1/2 x.x, accumulated with weight w (grad_acc: a LatentVec that
        holds the running sum, None, or a `_Dest` of a pairwise sum over samples)."""
        # gradient: J^T gs + x
        if isinstance(grad_acc, _Dest):
            dest, grad = grad_acc, None
            self._vjp(lp, gs, w, x.xi, w, dest.accumulate, dest.xi, w2=gs2, carries=dest.carries)
            _PairTree.settle(dest)
            self._amp_vjp(lp)
            dest.small.append(B.axpby(1.0, self.latbar, w, x.small))  # (abar, hence latbar, carries the 1/n_total weight)
        else:
            first = grad_acc is None
            grad = LatentVec(torch.empty_like(x.xi), None) if first else grad_acc
            self._vjp(lp, gs, w, x.xi, w, not first, grad.xi, w2=gs2)
            self._amp_vjp(lp)
            if first:
                grad.small = B.axpby(1.0, self.latbar, w, x.small)
            else:
                B.axpby(1.0, self.latbar, 1.0, grad.small, out=grad.small)
                B.axpby(w, x.small, 1.0, grad.small, out=grad.small)
        # value: lh + 1/2 x.x
        prior = x.sqnorm if x.sqnorm is not None else x.dot_device(x)
        B.axpby(w, lhval, 1.0, value, out=value)
        B.axpby(0.5 * w, prior, 1.0, value, out=value)
        lp.value, lp.grad = value, grad
        self._count("value_grad", 1)
        return lp

    # -- models with a linear response between the signal and the data (BASELINE config 4) -------------------------
    def _forward_nonlin(self, lp, x, g_out):
        """g(s) -> g_out and g'(s) -> lp.gp for s = offset + HT(a xi): one transform with the pointwise epilogue."""
        f = self._fuse()
        f.pro, f.in_, f.pidx, f.amp = L.PRO_AMP, x.xi.data_ptr(), self.pidx.data_ptr(), lp.amp.data_ptr()
        f.afield = lp.afield.data_ptr()
        lp.gp = torch.empty(self.shape, dtype=self.tdtype, device=self.device)
        f.epi, f.out, f.out2, f.offset, f.nonlin = L.EPI_NONLIN, g_out.data_ptr(), lp.gp.data_ptr(), self.offset_mean, self.nonlin
        B.hartley_fused(self.plan, f)
        self._count("transforms", 1)

    def _forward_nonlin_wide(self, lp, x):
        """g(s) as an fp64 tensor for a model with fp32 fields (`wide_response`): the transform runs on the fp64 plan of the
        grid with xi widened and the fp64 amplitude (octant field where both plans have the register-resident pipeline, the
        table gathered by pidx otherwise); g'(s) -> lp.gp rounded to the field type."""
        if self._wide_state is None:
            plan64 = B.get_plan(self.shape, torch.float64, 1, self.device)
            if self.stream is not None:  # a lane: the cached plan's workspace belongs to the main chain
                plan64 = B.PlanView(plan64)
            octant = self.octant_vjp and bool(L.load().nk_plan_octant_vjp(plan64.handle))
            afield64 = torch.empty(self.field_shape, dtype=torch.float64, device=self.device) if octant else None
            dense64 = None if (self.k2_dense is None or not octant) else torch.zeros_like(self.k2_dense, dtype=torch.float64)
            self._wide_state = (plan64, octant, afield64, dense64)
        plan64, octant, afield64, dense64 = self._wide_state
        xi64 = x.xi.to(torch.float64)
        g = torch.empty(self.shape, dtype=torch.float64, device=self.device)
        gp = torch.empty_like(g)
        f = self._fuse()
        f.field_octant = 1 if octant else 0
        f.pro, f.in_, f.pidx, f.amp = L.PRO_AMP, xi64.data_ptr(), self.pidx.data_ptr(), lp.amp.data_ptr()
        f.afield = self._amp_field(lp.amp, out=afield64, dense=dense64).data_ptr() if octant else None
        f.epi, f.out, f.out2, f.offset, f.nonlin = L.EPI_NONLIN, g.data_ptr(), gp.data_ptr(), self.offset_mean, self.nonlin
        B.hartley_fused(plan64, f)
        self._count("transforms", 1)
        lp.gp = gp.to(self.tdtype)
        return g

    def _weigh_data(self, u, lp):
        """u * M_d (data-space Fisher metric: N^-1, or 1/mu for counts)."""
        if self.const_wd:
            return B.axpby(self.icov_scalar, u)
        return B.binary(L.OP_MUL, u, self.icov_field if self.lh_kind == L.LH_GAUSS else lp.wd)

    def _linearize_response(self, lp, x, grad_acc, w, value, lhval, want_gradient=True):
        """energy_operators.py:517-595 / :617-640 composed with R o g o cf: mu = R g(s) lives on the data space."""
        if self.wide_response:
            # fp32 fields: everything up to the residual in fp64 (see __init__), like the reference's promoted arithmetic
            mu = self.response.times(self._forward_nonlin_wide(lp, x))
            data, icov, wide = self._data64, self._icov64, torch.float64
        else:
            self._forward_nonlin(lp, x, self.tmp)
            mu = self.response.times(self.tmp)
            data, icov, wide = self.data, self.icov_field, self.tdtype
        if self.lh_kind == L.LH_GAUSS:
            r = B.axpby(1.0, mu, -1.0, data)
            ir = B.axpby(self.icov_scalar, r) if self.const_wd else B.binary(L.OP_MUL, r, icov)
            B.vdot(r, ir, result=lhval, accumulate=False)
            B.axpby(0.5, lhval, out=lhval)
        else:
            dat = self.data.to(wide)
            B.vsum(mu, result=lhval, accumulate=False)
            ld = B.vdot(B.pointwise("log", mu), dat)
            B.axpby(1.0, lhval, -1.0, ld, out=lhval)
            lp.wd = B.pointwise("reciprocal", mu)
            ir = B.binary(L.OP_SUB, 1.0, B.binary(L.OP_MUL, dat, lp.wd))  # 1 - d / mu
            lp.wd = lp.wd.to(self.tdtype)
        ir = ir.to(self.tdtype)
        if not want_gradient:  # (lp.gp and lp.wd are what the metric needs)
            return lp
        gs = self.response.adjoint(ir, self.shape)
        return self._finish_linearize(lp, x, gs, lp.gp, grad_acc, w, value, lhval)

    def _jvp_response(self, lp, d, cg_direction=None):
        """R (g'(s) . J_cf d): latent tangent -> data space (one transform, one row-sum launch)."""
        L.check(L.load().nk_amp_jvp(self.nb, self.geo.data_ptr(), self.hyp.data_ptr(), lp.x.small.data_ptr(),
                                    lp.state.data_ptr(), d.small.data_ptr(), self.damp.data_ptr(), B._stream()), "nk_amp_jvp")
        self._amp_field(self.damp, out=self.dafield)
        f = self._fuse()
        f.pro, f.in_, f.in2 = L.PRO_AMP_JVP, d.xi.data_ptr(), lp.x.xi.data_ptr()
        f.pidx, f.amp, f.damp = self.pidx.data_ptr(), lp.amp.data_ptr(), self.damp.data_ptr()
        f.afield, f.dafield = B.ptr(lp.afield), self.dafield.data_ptr()
        f.epi, f.out, f.mul, f.mul_scalar = L.EPI_MUL, self.tmp.data_ptr(), lp.gp.data_ptr(), 1.0
        B.hartley_fused(self.plan, f)
        self._count("transforms", 1)
        return self.response.times(self.tmp)

    def lh_metric_accumulate(self, lp, d, out, scale, first, identity=0.0, dot_out=None, cg_direction=None, pipe=None,
                             addend=None, dest=None):
        """out (+)= scale * J^T M J d  (+ identity * d): the likelihood Fisher metric pulled back to latent space;
        the optional multiple of d (prior metric = 1) rides along in the VJP epilogue of the xi part.
        dest (a `_Dest`, then `out` / `first` are ignored): where a pairwise sum over samples wants this contribution.
        pipe = (chunks, wait_events or None, record_events or None): slab pipelining of the sandwich against an exchange
        on another stream (nk_fuse.pipe_chunks; the arrays are ctypes arrays of hipEvent_t handles).
        addend = (vector, factor): out += factor * vector instead of the identity term (rides in the same epilogue).
        cg_direction = (r, workspace): d.xi <- beta d.xi + r.xi first, inside the sandwich's first pass (d.small was
        updated by the caller: CgWorkspace.direction_small); the workspace scalars are rolled afterwards."""
        if cg_direction is not None and not self.fused_direction:
            raise ValueError("cg_direction needs the sandwich pipeline (FusedModel.fused_direction)")
        if addend is not None and identity:
            raise ValueError("either the identity term or an explicit addend")
        avec, afac = (d, identity) if addend is None else addend  # the vector the epilogue adds, and its factor
        if dest is None:
            dest = _Dest(out.xi, accumulate=not first)
        where = dict(accumulate=dest.accumulate, out_xi=dest.xi, carries=dest.carries)
        if self.response is not None:
            wsig = self.response.adjoint(self._weigh_data(self._jvp_response(lp, d), lp), self.shape)
            self._vjp(lp, wsig, scale, avec.xi if afac else None, afac, dot_out=dot_out, w2=lp.gp, **where)
            self._finish_metric(lp, avec, out, first, afac, dest)
            return
        L.check(L.load().nk_amp_jvp(self.nb, self.geo.data_ptr(), self.hyp.data_ptr(), lp.x.small.data_ptr(),
                                    lp.state.data_ptr(), d.small.data_ptr(), self.damp.data_ptr(), B._stream()), "nk_amp_jvp")
        # da[pindex] is expanded to an octant field once per application (1/8 of the gathers, 0.72 ms at 1024^3 fp32).
        # NK_DA_GATHER=1 lets the sandwich's first pass gather it from the table through the octant bin index instead:
        # measured 3.3 -> 8.9 ms for that pass (dependent gathers in a latency-bound row kernel) -- not the default
        gather_da = self.sandwich and self.pidx8 is not None and os.environ.get("NK_DA_GATHER", "0") == "1"
        if gather_da:
            damp_t = self.damp if self.tdtype == torch.float64 else self.damp.to(self.tdtype)
        else:
            self._amp_field(self.damp, out=self.dafield)

        def jvp_prologue(f):
            f.pro, f.in_, f.in2 = L.PRO_AMP_JVP, d.xi.data_ptr(), lp.x.xi.data_ptr()
            f.pidx, f.amp, f.damp = self.pidx.data_ptr(), lp.amp.data_ptr(), self.damp.data_ptr()
            f.afield = B.ptr(lp.afield)
            if gather_da:
                f.pidx_octant, f.dampT = self.pidx8.data_ptr(), damp_t.data_ptr()
            else:
                f.dafield = self.dafield.data_ptr()
            if cg_direction is not None:
                f.cg_r, f.cg_scal = cg_direction[0].xi.data_ptr(), cg_direction[1].scal.data_ptr()
            if pipe is not None:
                f.pipe_chunks = pipe[0]
                f.pipe_wait = None if pipe[1] is None else ctypes.cast(pipe[1], ctypes.c_void_p)
                f.pipe_record = None if pipe[2] is None else ctypes.cast(pipe[2], ctypes.c_void_p)

        if pipe is not None and not self.sandwich:
            raise ValueError("slab pipelining needs the sandwich pipeline")
        if self.sandwich:
            # H D H in five passes: the position-space field between the transforms never exists (nk_fft3.h)
            self._vjp(lp, None, scale, avec.xi if afac else None, afac, dot_out=dot_out,
                      sandwich=(jvp_prologue, self.h_dvol, lp.mid, lp.mid_scalar), **where)
            if cg_direction is not None:
                cg_direction[1].roll()
        else:
            f = self._fuse()
            jvp_prologue(f)
            f.epi, f.out, f.mul, f.mul_scalar = L.EPI_MUL, self.tmp.data_ptr(), B.ptr(lp.mid), lp.mid_scalar
            B.hartley_fused(self.plan, f)
            self._vjp(lp, self.tmp, scale, avec.xi if afac else None, afac, dot_out=dot_out, **where)
            self._count("transforms", 1)
        self._finish_metric(lp, avec, out, first, afac, dest)

    def pair_ready(self):
        """True when two samples' metric applications can share a final-pass launch (lh_metric_accumulate_pair)."""
        return bool(self.sandwich and self.octant_vjp and self.response is None and self.bin_k2 is not None
                    and len(self.shape) == 3 and self.const_mid and os.environ.get("NK_PAIR_FINAL", "1") != "0")

    def _pair_buffers(self):
        """Second set of the per-application scratch arrays (sample B of a pair), allocated on first use."""
        if getattr(self, "_pair", None) is None:
            self._pair = dict(damp=torch.empty_like(self.damp), dafield=torch.empty_like(self.dafield),
                              w8=torch.empty_like(self.w8), w8max=torch.zeros_like(self.w8max),
                              workspace=torch.empty_like(self.plan.workspace))
        return self._pair

    def lh_metric_accumulate_pair(self, lpa, lpb, d, out, scale, first, identity=0.0, dot_out=None, cg_direction=None,
                                  dests=None, identity_first=0.0):
        """lh_metric_accumulate(lpa, d, out, scale, first, cg_direction=...) followed by lh_metric_accumulate(lpb, d, out,
        scale, False, identity=..., dot_out=...) -- the same bits -- with the two final passes in one launch: sample B's
        accumulation onto `out` meets sample A's freshly written lines in the cache hierarchy (nk_hartley_sandwich_pair).
        The pending CG direction update rides in A's first pass, the identity term and the curvature dot in B's epilogue.
        dests = (`_Dest` of A, `_Dest` of B) of a pairwise sum over samples (then `out` / `first` are ignored): A stores a
        fresh vector, B joins it to its partial sums; identity_first: the multiple of d that A adds (the prior term of the
        pairwise sum sits on the first sample)."""
        if cg_direction is not None and not self.fused_direction:
            raise ValueError("cg_direction needs the sandwich pipeline (FusedModel.fused_direction)")
        pb = self._pair_buffers()
        lib = L.load()
        fuses = []
        if dests is None:
            dests = (_Dest(out.xi, accumulate=not first), _Dest(out.xi, accumulate=True))
        for lp, damp, dafield, w8, w8max, dest in ((lpa, self.damp, self.dafield, self.w8, self.w8max, dests[0]),
                                                   (lpb, pb["damp"], pb["dafield"], pb["w8"], pb["w8max"], dests[1])):
            L.check(lib.nk_amp_jvp(self.nb, self.geo.data_ptr(), self.hyp.data_ptr(), lp.x.small.data_ptr(), lp.state.data_ptr(),
                                   d.small.data_ptr(), damp.data_ptr(), B._stream()), "nk_amp_jvp")
            self._amp_field(damp, out=dafield)
            f = self._fuse()
            f.pro, f.in_, f.in2 = L.PRO_AMP_JVP, d.xi.data_ptr(), lp.x.xi.data_ptr()
            f.pidx, f.amp, f.damp = self.pidx.data_ptr(), lp.amp.data_ptr(), damp.data_ptr()
            f.afield, f.dafield = B.ptr(lp.afield), dafield.data_ptr()
            f.mul, f.mul_scalar = B.ptr(lp.mid), lp.mid_scalar
            f.epi, f.out, f.scale = L.EPI_VJP, dest.xi.data_ptr(), self.h_dvol * scale
            f.xi = lp.x.xi.data_ptr()
            f.addend, f.addend_scale, f.accumulate = None, 0.0, 1 if dest.accumulate else 0
            f.carry1, f.carry2 = (B.ptr(c) for c in (tuple(dest.carries) + (None, None))[:2])
            f.abar, f.w8 = self.abar.data_ptr(), w8.data_ptr()
            if self.scatter_fixed_point:
                f.w8max = w8max.data_ptr()
            fuses.append(f)
        if cg_direction is not None:
            fuses[0].cg_r, fuses[0].cg_scal = cg_direction[0].xi.data_ptr(), cg_direction[1].scal.data_ptr()
        if identity_first:
            fuses[0].addend, fuses[0].addend_scale = d.xi.data_ptr(), identity_first
        if identity:
            fuses[1].addend, fuses[1].addend_scale = d.xi.data_ptr(), identity
            if dot_out is not None:
                fuses[1].value = dot_out.data_ptr()
        elif dot_out is not None:
            raise ValueError("dot_out needs the identity term (the addend of the epilogue)")
        B.hartley_sandwich_pair(self.plan, fuses[0], fuses[1], self.h_dvol, pb["workspace"])
        _PairTree.settle(dests[1])
        if cg_direction is not None:
            cg_direction[1].roll()
        self._count("transforms", 4)
        shp = (ctypes.c_int64 * len(self.shape))(*self.shape)
        for lp, w8, w8max, is_first, dest, ident in ((lpa, self.w8, self.w8max, first, dests[0], identity_first),
                                                     (lpb, pb["w8"], pb["w8max"], False, dests[1], identity)):
            L.check(lib.nk_octant_scatter_k2(len(self.shape), shp, w8.data_ptr(), self.pidx.data_ptr(), self.bin_k2.data_ptr(),
                                             self.nb, self.scatter_scratch.data_ptr(), self.abar.data_ptr(),
                                             w8max.data_ptr() if self.scatter_fixed_point else 0, B._stream()),
                    "nk_octant_scatter_k2")
            self._finish_metric(lp, d, out, is_first, ident, dest)

    def _finish_metric(self, lp, d, out, first, identity, dest=None):
        self._amp_vjp(lp)
        if dest is not None:
            _PairTree.settle(dest)
            if dest.small is not None:  # a pairwise sum over samples: the caller adds the small parts
                dest.small.append(B.axpby(1.0, self.latbar, identity, d.small) if identity else B.axpby(1.0, self.latbar))
                self._count("metric", 1)
                return
        if first:
            out.small = B.axpby(1.0, self.latbar, identity, d.small) if identity else B.axpby(1.0, self.latbar)
        else:
            B.axpby(1.0, self.latbar, 1.0, out.small, out=out.small)
            if identity:
                B.axpby(identity, d.small, 1.0, out.small, out=out.small)
        self._count("metric", 1)

    def metric(self, lp, d, dot_out=None, cg_direction=None):
        """(J^T M J + 1) d at the linearisation point lp (dot_out: += d.xi . out.xi, see _vjp)."""
        out = LatentVec(torch.empty_like(d.xi), None)
        self.lh_metric_accumulate(lp, d, out, 1.0, True, identity=1.0, dot_out=dot_out, cg_direction=cg_direction)
        return out

    def lh_metric(self, lp, d, minus=None):
        """J^T M J d, optionally minus another latent vector (subtracted in the VJP epilogue: no extra pass)."""
        out = LatentVec(torch.empty_like(d.xi), None)
        self.lh_metric_accumulate(lp, d, out, 1.0, True, addend=None if minus is None else (minus, -1.0))
        return out

    # -- likelihood transformation f (geoVI; energy_operators.py:590-591, 639-640, kl_energies.py:105-124) ----
    def trafo_point(self, x):
        """Everything geoVI needs about x: amplitude tables/field, the transformed signal f(x) (Gaussian:
        N^{-1/2} g(s), Poisson: 2 sqrt(g(s))) and the weight field tf = (df/dmu) g'(s) of its Jacobian
        J_f = diag(tf) . J_cf."""
        lp = LinPoint()
        lp.x = x
        lp.amp, lp.state = self._amp_forward(x.small)
        lp.afield = self._amp_field(lp.amp)
        if self.response is not None:
            # f = N^{-1/2} mu or 2 sqrt(mu) on the data space; J_f = diag(tfd) R diag(g') J_cf
            self._forward_nonlin(lp, x, self.tmp)
            mu = self.response.times(self.tmp)
            if self.lh_kind == L.LH_GAUSS:
                lp.tfd = math.sqrt(self.icov_scalar) if self.icov_field is None else B.pointwise("sqrt", self.icov_field)
                lp.f = B.axpby(lp.tfd, mu) if self.icov_field is None else B.binary(L.OP_MUL, mu, lp.tfd)
            else:
                rt = B.pointwise("sqrt", mu)
                lp.f, lp.tfd = B.axpby(2.0, rt), B.pointwise("reciprocal", rt)
            return lp
        f = self._fuse()
        f.pro, f.in_, f.pidx, f.amp = L.PRO_AMP, x.xi.data_ptr(), self.pidx.data_ptr(), lp.amp.data_ptr()
        f.afield = lp.afield.data_ptr()
        g = torch.empty(self.shape, dtype=self.tdtype, device=self.device)
        gp = torch.empty_like(g)
        f.epi, f.out, f.out2, f.offset, f.nonlin = L.EPI_NONLIN, g.data_ptr(), gp.data_ptr(), self.offset_mean, self.nonlin
        B.hartley_fused(self.plan, f)
        self._count("transforms", 1)
        if self.lh_kind == L.LH_GAUSS:
            if self.icov_field is None:
                tw = math.sqrt(self.icov_scalar)
                lp.f, lp.tf = B.axpby(tw, g), B.axpby(tw, gp)
            else:
                tw = B.pointwise("sqrt", self.icov_field)
                lp.f, lp.tf = B.binary(L.OP_MUL, g, tw), B.binary(L.OP_MUL, gp, tw)
        else:
            rt = B.pointwise("sqrt", g)
            lp.f, lp.tf = B.axpby(2.0, rt), B.binary(L.OP_DIV, gp, rt)
        return lp

    def jvp_data(self, lp, d, out=None):
        """J_f(lp.x) d = tf * dvol_h HT(a dxi + da xi): latent tangent -> data space."""
        if self.response is not None:
            u = self._jvp_response(lp, d)
            return B.axpby(lp.tfd, u, out=out) if np.isscalar(lp.tfd) else B.binary(L.OP_MUL, u, lp.tfd, out=out)
        L.check(L.load().nk_amp_jvp(self.nb, self.geo.data_ptr(), self.hyp.data_ptr(), lp.x.small.data_ptr(),
                                    lp.state.data_ptr(), d.small.data_ptr(), self.damp.data_ptr(), B._stream()), "nk_amp_jvp")
        f = self._fuse()
        f.pro, f.in_, f.in2 = L.PRO_AMP_JVP, d.xi.data_ptr(), lp.x.xi.data_ptr()
        f.pidx, f.amp, f.damp = self.pidx.data_ptr(), lp.amp.data_ptr(), self.damp.data_ptr()
        f.afield = B.ptr(lp.afield)
        self._amp_field(self.damp, out=self.dafield)
        f.dafield = self.dafield.data_ptr()
        out = torch.empty(self.shape, dtype=self.tdtype, device=self.device) if out is None else out
        f.epi, f.out, f.mul, f.mul_scalar = L.EPI_MUL, out.data_ptr(), lp.tf.data_ptr(), 1.0
        B.hartley_fused(self.plan, f)
        self._count("transforms", 1)
        return out

    def vjp_data(self, lp, w, addend=None):
        """J_f(lp.x)^T w (+ addend): data space -> latent cotangent."""
        out = LatentVec(torch.empty_like(lp.x.xi), None)
        w2 = lp.tf
        if self.response is not None:
            u = B.axpby(lp.tfd, w) if np.isscalar(lp.tfd) else B.binary(L.OP_MUL, w, lp.tfd)
            w, w2 = self.response.adjoint(u, self.shape), lp.gp
        self._vjp(lp, w, 1.0, None if addend is None else addend.xi, 1.0, False, out.xi, w2=w2)
        self._amp_vjp(lp)
        out.small = B.axpby(1.0, self.latbar) if addend is None else B.axpby(1.0, self.latbar, 1.0, addend.small)
        return out

    # -- sampling (kl_energies.py:91-159, sampling_enabler.py:64-86) -------------------------------
    def _upload(self, arr, dtype):
        return torch.from_numpy(np.ascontiguousarray(arr)).to(dtype).to(self.device)

    def host_draws_before_xi(self, sseq):
        """What `draw_prior` takes from the HOST generator before the device draws xi -- the scalars and the (2, nb - 2)
        spectrum excitations that sort before `xi` -- drawn from a generator of its own on `sseq`, with the generator's state
        afterwards: (values by key, state).  `draw_prior(ahead=...)` continues from there; the same numbers as drawing them
        in place (same seed, same calls), but a host thread can make them while the GPU solves the previous sample
        (`_HostDrawAhead`)."""
        rng = np.random.default_rng(sseq)
        fresh = rng.bit_generator.state
        vals = {}
        for k in LATENT_KEYS:
            if k == "xi":
                break
            vals[k] = rng.normal(0.0, 1.0, (2, self.nb - 2) if k == "spectrum" else ())
        return vals, rng.bit_generator.state, fresh

    def draw_prior(self, device_rng=None, ahead=None):
        """One standard-normal latent draw, keys in alphabetical order like MultiField.from_random.  ahead: the result of
        `host_draws_before_xi` on the seed of the CURRENT random context (its draws are taken, its state continued)."""
        if device_rng is not None:
            xi = torch.randn(self.shape, dtype=self.tdtype, device=self.device, generator=device_rng)
            small = torch.randn(self.nsmall, dtype=torch.float64, device=self.device, generator=device_rng)
            return LatentVec(xi, small)
        nb = self.nb
        parts = {}
        if ahead is not None:
            # the worker drew from a FRESH generator on the context's seed: adopt its state only if the context's own generator
            # is still there (a draw made earlier in this context would otherwise be silently undone: ADVICE r5)
            if random.current_rng().bit_generator.state != ahead[2]:
                raise RuntimeError("draw_prior(ahead=...): the random context has advanced since its seed was handed to the "
                                   "draw-ahead worker")
            random.current_rng().bit_generator.state = ahead[1]
        for k in LATENT_KEYS:  # alphabetical = the reference's draw order
            if ahead is not None and k in ahead[0]:
                if k == "spectrum":
                    spectrum = self._upload(ahead[0][k], torch.float64)
                else:
                    parts[k] = ahead[0][k]
                continue
            if k == "xi":  # the numpy stream, computed on the device from the host generator's state
                xi = random.Random.normal_on_device(self.npdtype, self.shape, 0.0, 1.0, self.device)
            elif k == "spectrum":
                # (2, nb - 2) values: 6 x 10^5 on a 2048^2 grid -- 5 ms of numpy per draw with the GPU idle, four draws per
                # iteration, a fifth of a step there: small grids take the same stream from the device like xi (identical
                # values outside the ziggurat tail, tests/test_rng.py).  On large grids the host draw is noise (1.2 x 10^6
                # values = 9 ms per draw of a 5 s step at 1024^3) and stays numpy's own, bit for bit.
                if self.N <= SPECTRUM_DEVICE_DRAW_MAX_POINTS:
                    spectrum = random.Random.normal_on_device(np.float64, (2, nb - 2), 0.0, 1.0, self.device)
                else:
                    spectrum = self._upload(random.current_rng().normal(0.0, 1.0, (2, nb - 2)), torch.float64)
            else:
                parts[k] = random.current_rng().normal(0.0, 1.0, ())
        small = torch.empty(self.nsmall, dtype=torch.float64, device=self.device)
        small[:5].copy_(torch.from_numpy(np.array([parts[k] for k in SMALL_KEYS], dtype=np.float64)), non_blocking=False)
        small[5:].copy_(spectrum.reshape(-1))
        return LatentVec(xi, small)

    def draw_lh_noise(self, lp, device_rng=None):
        """J^T M_d^{1/2} eta with eta ~ N(0,1) in data space."""
        if self.response is not None:
            dshape = self.data_shape
            if device_rng is not None:
                eta = torch.randn(dshape, dtype=self.tdtype, device=self.device, generator=device_rng)
                if self.const_wd:
                    eta = B.axpby(math.sqrt(self.icov_scalar), eta)
            else:
                eta = random.Random.normal_on_device(self.npdtype, dshape, 0.0,
                                                     math.sqrt(self.icov_scalar) if self.const_wd else 1.0, self.device)
            if not self.const_wd:
                eta = B.binary(L.OP_MUL, eta, B.pointwise("sqrt", self.icov_field if self.lh_kind == L.LH_GAUSS else lp.wd))
            out = LatentVec(torch.empty(self.shape, dtype=self.tdtype, device=self.device), None)
            self._vjp(lp, self.response.adjoint(eta, self.shape), 1.0, None, 0.0, False, out.xi, w2=lp.gp)
            self._amp_vjp(lp)
            out.small = B.axpby(1.0, self.latbar)
            return out
        if device_rng is not None:
            eta = torch.randn(self.shape, dtype=self.tdtype, device=self.device, generator=device_rng)
            if self.const_mid:
                eta = B.axpby(math.sqrt(self.icov_scalar), eta)
        elif self.const_mid:
            eta = random.Random.normal_on_device(self.npdtype, self.shape, 0.0, math.sqrt(self.icov_scalar), self.device)
        else:
            eta = random.Random.normal_on_device(self.npdtype, self.shape, 0.0, 1.0, self.device)
        if not self.const_mid:
            eta = B.binary(L.OP_MUL, eta, B.pointwise("sqrt", lp.mid))
        out = LatentVec(torch.empty(self.shape, dtype=self.tdtype, device=self.device), None)
        self._vjp(lp, eta, 1.0, None, 0.0, False, out.xi)
        self._amp_vjp(lp)
        out.small = B.axpby(1.0, self.latbar)
        return out

    def draw_mgvi_sample(self, lp, controller, device_rng=None, ahead=None):
        """Returns (b, y): y solves (J^T M J + 1) y = b = s + nj by CG started at the prior draw s."""
        s = self.draw_prior(device_rng, ahead=ahead)
        nj = self.draw_lh_noise(lp, device_rng)
        b = s + nj
        g0 = self.lh_metric(lp, s, minus=nj)  # J^T M J s - nj: the subtraction rides in the transform's epilogue
        A = _Callable(lambda v, dot_out=None, cg_direction=None: self.metric(lp, v, dot_out=dot_out, cg_direction=cg_direction),
                      fused_dot=self.octant_vjp, fused_direction=self.fused_direction)
        energy = QuadraticEnergy(s, A, b, _grad=g0)
        energy.consumable = True  # s and g0 belong to this solve: the CG may update them in place
        energy, _ = ConjugateGradient(controller)(energy)
        return b, energy.position


class _Callable:
    """Operator handle for the minimisers.  ``fused_dot``: the operator can deposit the xi part of d.(A d) into a
    device scalar while it writes A d (``A(d, dot_out=slot)``) -- ConjugateGradient then skips that BLAS-1 pass."""

    def __init__(self, fn, fused_dot=False, fused_direction=False):
        self._fn = fn
        self.fused_dot = fused_dot
        # the operator can also take over the pending CG direction update: A(d, cg_direction=(r, workspace)) updates
        # d.xi in place (d <- beta d + r) before applying itself and rolls the workspace scalars
        self.fused_direction = fused_direction

    def __call__(self, x, **kw):
        return self._fn(x, **kw)


# ------------------------------------------------------------------------------------------------
# sampled KL
# ------------------------------------------------------------------------------------------------
class FusedKL(Energy):
    """SampledKLEnergyClass on the fused model (kl_energies.py:299-360).

    ``residuals``/``negs`` are the LOCAL samples of this rank; ``n_total`` the global sample count.
    ``comm`` (nifty_amd.parallel.Comm or None) sums value/gradient/metric over ranks.
    """

    def __init__(self, model, position, residuals, negs, n_total=None, comm=None, nanisinf=True, _shared=None):
        super().__init__(position)
        self.model, self.residuals, self.negs = model, residuals, negs
        self.n_total = len(residuals) if n_total is None else n_total
        self.comm, self.nanisinf = comm, nanisinf
        self.lins = []
        self._lanes = self._pick_lanes(len(residuals))
        # small grids: ONE launch set for all local samples instead of one kernel chain per sample on stream lanes
        # (nifty_amd/batched.py; single process, >= 2 samples; NK_BATCH=0: the lanes)
        self._batched = comm is None and len(residuals) >= 2 and batched.ready(model)
        # sums over samples in the pairwise order of the reference (parallel.pair_tree): the same bits for 1, 2, 4, 8 ranks.
        # Across ranks: one complete subtree per rank (equal power-of-two sample counts) -> the slice-wise tree over the rank
        # partials (`_across`); any other split -> every sample keeps its own vector and the terms are added like the
        # reference adds them, the second summand of a merge travelling to the holder of the first (`_terms`: memory of one
        # latent vector per local sample and point-to-point traffic -- correct for every split, not the fast path;
        # NK_TREE_GENERAL=0: the local tree + RCCL's all-reduce, rank-count dependent rounding, said once).
        self._shared = _shared if _shared is not None else {}
        self._tree = parallel.tree_sum_enabled()
        self._across, self._terms = False, None
        if comm is not None and self._tree:
            if "across" not in self._shared:
                counts = comm.term_counts(len(residuals))
                self._shared["across"] = comm.subtree_per_rank(counts) and comm.tree_exchange_works(model.device)
                general = (not self._shared["across"] and comm.size > 1 and sum(counts) > 0
                           and os.environ.get("NK_TREE_GENERAL", "1") != "0")
                self._shared["terms"] = counts if general else None
                if not self._shared["across"] and not general and comm.rank == 0 and comm.size > 1:
                    print("nifty_amd: the samples do not split into one power-of-two block per rank; sums over samples "
                          "depend on the rank count in the last bits", flush=True)
            self._across, self._terms = self._shared["across"], self._shared["terms"]
        self._holds_first = comm is None or comm.rank == 0  # global sample 0 carries the prior term of a pairwise sum
        value = torch.zeros(1, dtype=torch.float64, device=model.device)
        grad = None
        if self._terms is not None:
            value, grad = self._linearize_terms(position)
        elif self._batched:
            value, grad = batched.kl_linearize(self, position)
        elif len(self._lanes) > 1:
            value, grad = self._linearize_on_lanes(position, value)
        elif self._tree and len(residuals) > 0:
            value, grad = self._linearize_pairwise(position)
        else:
            for r, neg in zip(residuals, negs):
                x = position.shifted(-1.0 if neg else 1.0, r)  # p +- r and |p +- r|^2 in one pass
                lp = model.linearize(x, grad_acc=grad, n_total=self.n_total, value_acc=value)
                grad = lp.grad
                lp.grad = None
                self.lins.append(lp)
        if grad is None:  # a rank without samples
            grad = LatentVec.zeros(model)
        if self._terms is None:
            self._sum_over_ranks(grad, value)
        self._value = float(value.item())
        if math.isnan(self._value) and nanisinf:
            self._value = math.inf
        self._grad = grad

    def _sum_over_ranks(self, vec, *scalars):
        """In place: the sum over the ranks of a latent vector (and device scalars) -- the tree over the rank partials when
        every rank holds one subtree of the pairwise sum (xi slice-wise: an all-to-all and an all-gather, the bytes of an
        all-reduce), else RCCL's all-reduce."""
        comm = self.comm
        if comm is None:
            return
        if not self._across:
            comm.allreduce_sum_(list(scalars) + [vec.xi, vec.small])
            return
        comm.tree_allreduce_gathered_(list(scalars) + [vec.small])
        if vec.xi.numel() % comm.size == 0:
            comm.tree_allreduce_slices_(vec.xi)
        else:
            comm.tree_allreduce([[vec.xi]], [1] * comm.size)

    # -- any split of the samples over the ranks: one vector per sample, added like the reference adds its terms -------------
    def _tree_over_terms(self, terms):
        """terms: this rank's [tensors of sample i] in sample order -> the pair_tree sum over ALL ranks' samples, in the
        tensors of the first local term (a rank without samples: fresh zeros)."""
        like = None
        if not terms:
            m = self.model
            like = [torch.zeros(1, dtype=torch.float64, device=m.device), torch.zeros(m.shape, dtype=m.tdtype, device=m.device),
                    LatentVec.zeros(m).small][-len(self._term_shape):]
        return self.comm.tree_allreduce(terms, self._terms, like=like)

    def _linearize_terms(self, position):
        self._term_shape = ("value", "xi", "small")
        terms = []
        for r, neg in zip(self.residuals, self.negs):
            v = torch.zeros(1, dtype=torch.float64, device=self.model.device)
            lp = self.model.linearize(position.shifted(-1.0 if neg else 1.0, r), n_total=self.n_total, value_acc=v)
            terms.append([v, lp.grad.xi, lp.grad.small])
            lp.grad = None
            self.lins.append(lp)
        value, xi, small = self._tree_over_terms(terms)
        return value, LatentVec(xi, small)

    def _apply_metric_terms(self, d):
        self._term_shape = ("xi", "small")
        w, terms = 1.0 / self.n_total, []
        for i, lp in enumerate(self.lins):
            out = LatentVec(torch.empty_like(d.xi), None)
            self.model.lh_metric_accumulate(lp, d, out, w, True, identity=1.0 if (i == 0 and self._holds_first) else 0.0)
            terms.append([out.xi, out.small])
        xi, small = self._tree_over_terms(terms)
        return LatentVec(xi, small)

    def _scratch_vector(self):
        return torch.empty(self.model.shape, dtype=self.model.tdtype, device=self.model.device)

    def _linearize_pairwise(self, position):
        """Value and gradient summed over the local samples in pair_tree order: the xi part inside the VJP epilogues
        (`_PairTree`), values and small parts as lists."""
        model, nloc = self.model, len(self.residuals)
        tree = _PairTree(torch.empty_like(position.xi), self._scratch_vector)
        values, smalls = [], []
        for i, (r, neg) in enumerate(zip(self.residuals, self.negs)):
            x = position.shifted(-1.0 if neg else 1.0, r)  # p +- r and |p +- r|^2 in one pass
            values.append(torch.zeros(1, dtype=torch.float64, device=model.device))
            lp = model.linearize(x, n_total=self.n_total, value_acc=values[-1], dest=tree.place(i == nloc - 1, smalls))
            lp.grad = None
            self.lins.append(lp)
        return parallel.tree_fold(values), LatentVec(tree.total(), parallel.tree_fold(smalls))

    # -- small grids: the samples' kernel chains side by side on several streams (FusedModel.lanes) -------------------------
    def _pick_lanes(self, nloc):
        """[model] or K lanes: single process, at least two local samples, a grid small enough that one chain leaves the
        chip mostly idle (NK_LANE_MAX_POINTS, default 2^25 points; NK_LANES = number of lanes, default 4, 0 / 1 = off)."""
        want = int(os.environ.get("NK_LANES", "4"))
        small = self.model.N <= int(os.environ.get("NK_LANE_MAX_POINTS", str(1 << 25)))
        held_back = self.model.response is not None and os.environ.get("NK_LANES_RESPONSE", "1") == "0"
        if self.comm is not None or want < 2 or nloc < 2 or not small or held_back:
            return [self.model]
        return self.model.lanes(min(want, nloc))

    def _on_lanes(self, job):
        """job(lane index, lane model, sample index) for every local sample, sample i on lane i mod K, each lane on its own
        stream; the caller's stream waits for all of them afterwards.  Tensors the jobs allocate belong to their lane's stream:
        whatever the caller's stream reads later must be handed to `_adopt`."""
        lanes = self._lanes
        main = torch.cuda.current_stream(self.model.device)
        for lane in lanes[1:]:
            lane.stream.wait_stream(main)
        # (one host thread per lane was tried here as well -- the launch calls run without the interpreter lock -- and
        # changed nothing: 2048^2 fp64 Poisson 114.6 vs 115.0 ms per iteration; geoVI's fits, whose threads overlap WAITS, gain)
        for i in range(len(self.residuals)):
            k = i % len(lanes)
            if lanes[k].stream is None:
                job(k, lanes[k], i)
            else:
                with torch.cuda.stream(lanes[k].stream):
                    job(k, lanes[k], i)
        for lane in lanes[1:]:
            main.wait_stream(lane.stream)
        return main

    @staticmethod
    def _adopt(main, *tensors):
        for t in tensors:
            if t is not None:
                t.record_stream(main)

    def _linearize_on_lanes(self, position, value):
        """Every sample's weighted gradient in a vector of its own (small grids: memory is no concern), summed afterwards in
        pair_tree order on the caller's stream."""
        n = len(self.residuals)
        xs = [position.shifted(-1.0 if neg else 1.0, r) for r, neg in zip(self.residuals, self.negs)]
        values = [torch.zeros_like(value) for _ in range(n)]
        grads, lins = [None] * n, [None] * n

        def job(k, lane, i):
            lp = lane.linearize(xs[i], n_total=self.n_total, value_acc=values[i])
            grads[i], lp.grad, lins[i] = lp.grad, None, lp

        main = self._on_lanes(job)
        self.lins = lins
        for g in grads:
            self._adopt(main, g.xi, g.small)
        return parallel.tree_fold(values), self._fold_vectors(grads)

    @staticmethod
    def _fold_vectors(vecs):
        def add(a, b):
            B.axpby(1.0, a.xi, 1.0, b.xi, out=a.xi)
            B.axpby(1.0, a.small, 1.0, b.small, out=a.small)
            return a
        return parallel.tree_fold(vecs, add)

    def _apply_metric_on_lanes(self, d):
        n, w = len(self.lins), 1.0 / self.n_total
        # one vector per sample, allocated HERE (the caller's stream owns them); the first sample also adds the prior term
        outs = [LatentVec(torch.empty_like(d.xi), None) for _ in range(n)]
        prior = (1.0 if self._holds_first else 0.0) if self._tree else n * w

        def job(k, lane, i):
            lane.lh_metric_accumulate(self.lins[i], d, outs[i], w, True, identity=prior if i == 0 else 0.0)

        main = self._on_lanes(job)
        for o in outs:
            self._adopt(main, o.small)
        return self._fold_vectors(outs)

    @property
    def value(self):
        return self._value

    @property
    def gradient(self):
        return self._grad

    def at(self, position):
        return FusedKL(self.model, position, self.residuals, self.negs, self.n_total, self.comm, self.nanisinf, self._shared)

    def _apply_metric_local(self, d, dot_out=None, cg_direction=None, pipe=None):
        """This rank's share of the KL metric applied to d (no communication).  pipe = (chunks, wait, record): the FIRST
        local sample's transform waits chunk by chunk for d, the LAST one records chunk by chunk that `out` is final."""
        m = self.model
        if self._batched and pipe is None and dot_out is None and cg_direction is None:
            return batched.kl_apply_metric(self, d)
        if len(self._lanes) > 1 and pipe is None and dot_out is None and cg_direction is None:
            return self._apply_metric_on_lanes(d)
        if self._tree and len(self.lins) > 0:
            return self._apply_metric_pairwise(d, dot_out, cg_direction, pipe)
        out = LatentVec(torch.empty_like(d.xi), None)
        w = 1.0 / self.n_total
        nloc = len(self.lins)
        # the samples go two at a time with their final passes in one launch (lh_metric_accumulate_pair: the second one's
        # accumulation onto `out` meets the first one's lines in cache); an odd one out and the staged passes stay single
        paired = m.pair_ready()
        i = 0
        while i < nloc:
            lp = self.lins[i]
            # (a staged first / final pass of the pipelined exchange keeps its own launch)
            if paired and i + 1 < nloc and (pipe is None or (i > 0 and i + 1 < nloc - 1)):
                tail = i + 1 == nloc - 1
                m.lh_metric_accumulate_pair(lp, self.lins[i + 1], d, out, w, i == 0, identity=nloc * w if tail else 0.0,
                                            dot_out=dot_out if tail else None, cg_direction=cg_direction if i == 0 else None)
                i += 2
                continue
            stage = None
            if pipe is not None:
                stage = (pipe[0], pipe[1] if i == 0 else None, pipe[2] if i == nloc - 1 else None)
            # prior term (identity): every rank contributes its share nloc/n_total of d with its LAST sample, fused
            # into that sample's VJP epilogue (the sum over ranks is d); the same epilogue sees the finished local
            # q = A d and can take d.q on the way (single process only)
            last = i == nloc - 1
            m.lh_metric_accumulate(lp, d, out, w, i == 0, identity=nloc * w if last else 0.0,
                                   dot_out=dot_out if last else None, cg_direction=cg_direction if i == 0 else None,
                                   pipe=stage)
            i += 1
        if nloc == 0:
            out = LatentVec.zeros(m)
        return out

    def _apply_metric_pairwise(self, d, dot_out, cg_direction, pipe):
        """The local share with the samples added in pair_tree order (`_PairTree`): even samples store a fresh vector, odd
        ones join the partial sums that end with them -- in one launch with their even neighbour where the model can
        (lh_metric_accumulate_pair).  The prior term d rides on GLOBAL sample 0, so that the tree is the same for every
        split; the curvature dot can be taken on the way only when that is also the last sample (one local sample)."""
        m, nloc, w = self.model, len(self.lins), 1.0 / self.n_total
        if dot_out is not None and nloc > 1:
            raise ValueError("the fused curvature dot needs the prior term on the last sample (one local sample)")
        tree = _PairTree(torch.empty_like(d.xi), self._scratch_vector)
        smalls = []
        paired = m.pair_ready()
        i = 0
        while i < nloc:
            last = i == nloc - 1
            prior = 1.0 if (i == 0 and self._holds_first) else 0.0
            direction = cg_direction if i == 0 else None
            staged = pipe is not None and (i == 0 or last)
            if paired and i % 2 == 0 and not last and not staged and not (pipe is not None and i + 1 == nloc - 1):
                dests = (tree.place(False, smalls), tree.place(i + 1 == nloc - 1, smalls))
                m.lh_metric_accumulate_pair(self.lins[i], self.lins[i + 1], d, None, w, True, cg_direction=direction,
                                            dests=dests, identity_first=prior)
                i += 2
                continue
            stage = None
            if pipe is not None:
                stage = (pipe[0], pipe[1] if i == 0 else None, pipe[2] if last else None)
            dest = tree.place(last, smalls)
            if stage is not None and stage[2] is not None and dest.after:
                raise RuntimeError("a staged final pass cannot be followed by explicit partial-sum additions")
            m.lh_metric_accumulate(self.lins[i], d, None, w, True, identity=prior, dot_out=dot_out, cg_direction=direction,
                                   pipe=stage, dest=dest)
            i += 1
        return LatentVec(tree.total(), parallel.tree_fold(smalls))

    def apply_metric(self, d, dot_out=None, cg_direction=None):
        if (dot_out is not None or cg_direction is not None) and self.comm is not None:
            raise ValueError("the fused curvature dot / direction update are single-process shortcuts")
        if self._terms is not None:
            return self._apply_metric_terms(d)
        out = self._apply_metric_local(d, dot_out, cg_direction)
        self._sum_over_ranks(out)
        return out

    @property
    def metric(self):
        # (lanes / batched launches: no fused dot / direction)
        single = self.comm is None and len(self.lins) > 0 and len(self._lanes) == 1 and not self._batched
        # (pairwise sums keep the prior term on the FIRST sample; the dot needs it on the last one)
        A = _Callable(self.apply_metric, fused_dot=single and self.model.octant_vjp and (not self._tree or len(self.lins) == 1),
                      fused_direction=single and self.model.fused_direction)
        if self.comm is not None and self.comm.can_shard(self.model.N) and self._terms is None:
            A.sharded = ShardedMetric(self)  # picked up by ConjugateGradient: CG vectors sharded over the ranks
        return A

    @property
    def samples(self):
        return [(self._position - r if neg else self._position + r) for r, neg in zip(self.residuals, self.negs)]


class ShardedMetric:
    """The KL metric for a CG whose vectors are sharded over the ranks: instead of all-reducing the
    N-sized metric output and repeating every CG vector update on all ranks, the sample sum is REDUCE-SCATTERED
    (each rank receives its 1/size share of q), the CG updates run on that share, and only the new search direction
    is ALL-GATHERED for the next metric application.  Same bytes on the links as the all-reduce (which is a
    reduce-scatter followed by an all-gather), 1/size of the vector work and memory per rank.

    Chunked, overlapped exchange (round 3).  The xi part is cut along the first axis into C slab chunks
    (NK_PIPE_CHUNKS, default 8; 1 = the unchunked exchange) and a rank's share is the rank-th 1/size of EVERY chunk,
    i.e. the flat vector viewed as [C][size][m] is owned as [:, rank, :] (the CG updates are element-wise, any fixed
    subset will do).  Every chunk of the full vector is then contiguous, so chunk-wise all_gather_into_tensor /
    reduce_scatter_tensor work in place on the natural layout.  The collectives run on a side stream:
      * the all-gather of d is issued chunk by chunk in the order 0, C-1, 1, C-2, ... and the contiguous first pass of the
        first local sample's transform starts on the rows whose chunks have landed (nk_fuse.pipe_wait);
      * the final pass of the last local sample's transform finishes `out` chunk pair by chunk pair (nk_fuse.pipe_record)
        and the reduce-scatter of a pair starts while the later pairs are still computed.
    The three passes in between need the whole array, so at most the first (3.3 ms at 1024^3 fp32) and the final pass
    (4.0 ms) hide communication.  The staged passes and the overlap change no arithmetic: for a given C the results are
    bit-identical with and without them (NK_PIPE_OVERLAP=0), and on one rank for every C.  With several ranks C decides
    which elements a rank owns; since round 4 that does not enter the rounding either -- the dot products of the sharded
    vectors are sums of 64 unit sums wherever the units live (ShardedCgWorkspace, nk_red_layout) and the sample sum is the
    reference's pairwise tree (FusedKL) -- so every C and every rank count give the bits of the single-process run
    (tests/test_distributed_gloo.py; NK_TREE_SUM=0: agreement to rounding, as in rounds 1-3)."""

    def __init__(self, kl):
        self.kl, self.comm, self.model = kl, kl.comm, kl.model
        model, P = self.model, self.comm.size
        C = int(os.environ.get("NK_PIPE_CHUNKS", "8"))
        ok = (C >= 2 and model.sandwich and len(model.shape) == 3 and len(kl.lins) > 0
              and bool(L.load().nk_plan_pipe_ok(model.plan.handle, C)) and (model.N // C) % P == 0
              and not (kl._tree and len(kl.lins) > 8))  # (beyond 8 local samples the pairwise sum ends with explicit additions)
        # every rank must take the same decision (a rank without samples cannot stage its transform: it has none)
        if P > 1:
            ok = all(self.comm.allgather_object(bool(ok)))
        self.chunks = C if ok else 1
        self.m = model.N // (self.chunks * P)  # elements of this rank per chunk
        self.native = self.comm.backend_is_nccl
        self._side = self._ev_in = self._ev_out = None
        # NK_PIPE_OVERLAP=0: the same chunked ownership and collectives, but one-launch passes and a blocking exchange on
        # the compute stream (A/B for the overlap; bit-identical results)
        self.overlap = os.environ.get("NK_PIPE_OVERLAP", "1") != "0"
        if self.chunks > 1 and self.native and self.overlap:
            cache = model.__dict__.setdefault("_pipe_cache", {})  # side stream and events live as long as the model
            if self.chunks not in cache:
                ns = self.chunks // 2
                side = torch.cuda.Stream(device=model.device)
                ev_in, ev_out = [torch.cuda.Event() for _ in range(ns)], [torch.cuda.Event() for _ in range(ns)]
                for ev in ev_in + ev_out:
                    ev.record()  # creates the underlying hipEvent_t; every later record replaces this one
                cache[self.chunks] = (side, ev_in, ev_out, (ctypes.c_void_p * ns)(*[ev.cuda_event for ev in ev_in]),
                                      (ctypes.c_void_p * ns)(*[ev.cuda_event for ev in ev_out]))
            self._side, self._ev_in, self._ev_out, self._h_in, self._h_out = cache[self.chunks]

    # -- layout ---------------------------------------------------------------------------------------------------------
    def _mine(self, flat):
        """This rank's share of a full flat vector as a [C, m] view."""
        return flat.view(self.chunks, self.comm.size, self.m)[:, self.comm.rank, :]

    def shard(self, v, copy=False):
        xi = self._mine(v.xi.reshape(-1))
        if self.chunks == 1:
            xi = xi.reshape(-1)
            return LatentVec(xi.clone(), v.small.clone()) if copy else LatentVec(xi, v.small)
        return LatentVec(xi.reshape(-1) if xi.is_contiguous() else xi.contiguous().view(-1),
                         v.small.clone() if copy else v.small)

    def workspace(self):
        return ShardedCgWorkspace(self.model.device, self.comm, self._unit_layout())

    def _unit_layout(self):
        """How this rank's share [:, rank, :] of the [chunks][size][m] vector sits in the reduction units of the full vector
        (ShardedCgWorkspace), or None when it is not made of whole units (the dot products then depend on the rank count in
        their last bits, like without NK_TREE_SUM)."""
        if not self.kl._tree:
            return None
        code = L.NK_F64 if self.model.tdtype == torch.float64 else L.NK_F32
        unit, k = int(L.load().nk_red_unit(self.model.N, code)), 64
        per_segment = k // (self.chunks * self.comm.size) if k % (self.chunks * self.comm.size) == 0 else 0
        if unit == 0 or per_segment == 0 or per_segment * unit != self.m:
            return None
        return (unit, k // self.comm.size, k, per_segment, k // self.chunks, self.comm.rank * per_segment)

    def _stage_chunks(self):
        """Chunk pairs in the order the kernels' stages need them: (0, C-1), (1, C-2), ..."""
        C = self.chunks
        return [(j, C - 1 - j) for j in range(C // 2)] if C > 1 else [(0,)]

    # -- exchange -------------------------------------------------------------------------------------------------------
    def gather(self, shard_vec, out_full=None):
        """Full vector from the shares (blocking on the current stream)."""
        if out_full is None:
            out_full = LatentVec(torch.empty(self.model.shape, dtype=shard_vec.xi.dtype, device=shard_vec.xi.device),
                                 torch.empty_like(shard_vec.small))
        full = out_full.xi.reshape(-1).view(self.chunks, -1)
        mine = shard_vec.xi.view(self.chunks, self.m)
        for c in range(self.chunks):
            self.comm.all_gather(mine[c], full[c])
        out_full.small.copy_(shard_vec.small)
        return out_full

    def apply(self, d_full):
        """q share = this rank's part of (sum over ranks of the local metric applied to the FULL d): unpipelined entry
        (residual refresh; callers that hold the full vector already)."""
        staged = self.chunks > 1 and self.overlap
        timer = parallel.exchange_timer
        with timer.span("local_metric"):
            out = self.kl._apply_metric_local(d_full, pipe=(self.chunks, None, None) if staged else None)
        with timer.span("exposed_wait"), timer.span("reduce_scatter"):  # (blocking on the compute stream)
            return self._reduce(out)

    def _scatter_chunk(self, full_chunk, share):
        """share <- this rank's 1/size of the sum over ranks of one chunk: the tree over the rank partials when every rank
        holds one subtree of the pairwise sum over samples (an all-to-all: the bytes of the reduce-scatter), else RCCL's
        reduce-scatter."""
        if self.kl._across:
            self.comm.tree_reduce_slices(full_chunk, share)
        else:
            self.comm.reduce_scatter_sum(full_chunk, share)

    def _sum_small(self, small):
        if self.kl._across:
            self.comm.tree_allreduce_gathered_([small])
        else:
            self.comm.allreduce_sum_([small])

    def _reduce(self, out):
        q_xi = torch.empty(self.chunks * self.m, dtype=out.xi.dtype, device=out.xi.device)
        full = out.xi.reshape(-1).view(self.chunks, -1)
        for c in range(self.chunks):
            self._scatter_chunk(full[c], q_xi[c * self.m:(c + 1) * self.m])
        self._sum_small(out.small)
        return LatentVec(q_xi, out.small)

    def apply_shard(self, d_shard, d_full):
        """q share for the search direction held as a share: all-gather d into `d_full`, apply the local metric, reduce-
        scatter -- with the exchange overlapped chunk by chunk when RCCL runs it on the side stream."""
        timer = parallel.exchange_timer
        P, nbytes = self.comm.size, d_full.xi.numel() * d_full.xi.element_size()
        timer.count(2 * (P - 1) * nbytes // P)  # per rank: all-gather in + reduce-scatter out, (P-1)/P of the vector each
        if self._side is None:  # one chunk, or a backend that stages through the host (gloo: tests)
            with timer.span("iteration"):
                with timer.span("exposed_wait"), timer.span("all_gather"):  # (blocking: nothing of it is hidden)
                    self.gather(d_shard, d_full)
                return self.apply(d_full)
        cur = torch.cuda.current_stream(self.model.device)
        side, pairs = self._side, self._stage_chunks()
        full = d_full.xi.reshape(-1).view(self.chunks, -1)
        mine = d_shard.xi.view(self.chunks, self.m)
        with timer.span("iteration", cur):
            d_full.small.copy_(d_shard.small)
            side.wait_stream(cur)  # the share of d is final
            with torch.cuda.stream(side), timer.span("all_gather", side):
                for j, pair in enumerate(pairs):
                    for c in pair:
                        self.comm.all_gather(mine[c], full[c])
                    self._ev_in[j].record(side)
            with timer.span("local_metric", cur):
                out = self.kl._apply_metric_local(d_full, pipe=(self.chunks, self._h_in, self._h_out))
            q_xi = torch.empty(self.chunks * self.m, dtype=out.xi.dtype, device=out.xi.device)
            ofull = out.xi.reshape(-1).view(self.chunks, -1)
            with torch.cuda.stream(side):
                for j, pair in enumerate(pairs):
                    side.wait_event(self._ev_out[j])  # (outside the timed span: this is waiting for the final pass, not exchange)
                    with timer.span("reduce_scatter", side):
                        for c in pair:
                            self._scatter_chunk(ofull[c], q_xi[c * self.m:(c + 1) * self.m])
            with timer.span("exposed_wait", cur):  # what the compute stream still waits for once its own work is done
                cur.wait_stream(side)
            self._sum_small(out.small)
        return LatentVec(q_xi, out.small)


class FusedGeoEnergy(Energy):
    """0.5 |m - g(x)|^2 with g(x) = x + J_f(p)^T f(x): the geoVI sampling energy
    (EnergyAdapter(pos, GaussianEnergy(m) @ transformation, want_metric=True), kl_energies.py:115-121, 148-155).
    Every evaluation is four transforms: forward at x, J_f(p)^T, J_f(p), J_f(x)^T; so is a metric application."""

    def __init__(self, model, tp_p, m, position, nanisinf=True):
        super().__init__(position)
        self.model, self.tp_p, self.m = model, tp_p, m
        self.tp = model.trafo_point(position)
        gx = model.vjp_data(tp_p, self.tp.f, addend=position)  # x + J_f(p)^T f(x)
        self.res = gx - m
        val = 0.5 * self.res.s_vdot(self.res)
        self._value = math.inf if (math.isnan(val) and nanisinf) else val
        self._grad = self._G_T(self.res)

    def _G(self, d):    # dg(x) d = d + J_f(p)^T J_f(x) d
        return self.model.vjp_data(self.tp_p, self.model.jvp_data(self.tp, d), addend=d)

    def _G_T(self, y):  # dg(x)^T y = y + J_f(x)^T J_f(p) y
        return self.model.vjp_data(self.tp, self.model.jvp_data(self.tp_p, y), addend=y)

    @property
    def value(self):
        return self._value

    @property
    def gradient(self):
        return self._grad

    def at(self, position):
        return FusedGeoEnergy(self.model, self.tp_p, self.m, position)

    def apply_metric(self, d):
        return self._G_T(self._G(d))

    @property
    def metric(self):
        return _Callable(self.apply_metric)


def share_range(nwork, nshares, myshare):
    """reference nifty/cl/utilities.py:282-306"""
    return parallel.shareRange(nwork, nshares, myshare)


def draw_samples(model, position, n_samples, mirror_samples, controller_factory, comm=None, device_rng=None,
                 geo_minimizer=None):
    """MGVI -- or with ``geo_minimizer`` (a DescentMinimizer) geoVI -- samples of this rank
    (kl_energies.py:105-159): returns (residuals, negs, n_total).  Which rank draws which sample from which seed is
    the parallel.SamplePlan shared with the generic operator graph (kl.draw_samples); a rank may get no sample."""
    plan = parallel.SamplePlan(n_samples, mirror_samples, comm)
    plan.check_synchronised()
    cache = {}

    def linearisation():
        if "lp" not in cache:
            cache["lp"] = model.metric_point(position)  # (the sampling solves apply the metric there; no value, no gradient)
        return cache["lp"]

    # large grids take the spectrum excitations from the host generator (draw_prior): 9 ms of numpy per sample with the GPU
    # idle -- made by a host thread while the GPU solves the previous sample instead (same seed, same calls: same numbers)
    drawn_seeds = [plan.seeds[i] for i in range(plan.lo, plan.hi) if not (plan.mirror and i % 2 == 1 and i > plan.lo)]
    ahead = None
    want_ahead = (device_rng is None and model.N > SPECTRUM_DEVICE_DRAW_MAX_POINTS and bool(drawn_seeds)
                  and os.environ.get("NK_DRAW_AHEAD", "1") != "0")

    def draw(seed):
        if device_rng is not None:
            # synthetic-draw mode: still one stream per sample seed, so both members of a mirrored pair
            # see identical draws even when they live on different ranks (kl_energies.py:132-146)
            device_rng.manual_seed(int(seed.generate_state(1, np.uint64)[0] >> np.uint64(1)))
        lp = linearisation()
        return model.draw_mgvi_sample(lp, controller_factory(), device_rng, ahead=ahead.take(seed) if ahead else None)

    def linear_residual(pair, mirrored):
        return pair[1], mirrored

    def fitted_residual(pair, mirrored):
        # geoVI (kl_energies.py:105-124, 148-155): the linear sample only starts a non-linear fit of
        # g(x) = x + J_f(p)^T f(x) to  g(p) +- b
        if "tp" not in cache:
            cache["tp"] = model.trafo_point(position)
            cache["g_p"] = model.vjp_data(cache["tp"], cache["tp"].f, addend=position)
        b, y = pair
        target = cache["g_p"] - b if mirrored else cache["g_p"] + b
        start = position - y if mirrored else position + y
        fit, _ = geo_minimizer(FusedGeoEnergy(model, cache["tp"], target, start))
        return fit.position - position, False

    finish = linear_residual if geo_minimizer is None else fitted_residual
    local_pairs = len({i // 2 if plan.mirror else i for i in range(plan.lo, plan.hi)})
    lanes = _sampling_lanes(model, local_pairs)
    # the linear solves of the iteration advance together: as batched launches (nifty_amd/batched.py) or on stream lanes
    in_batch = local_pairs >= 2 and batched.ready(model)
    if lanes is None and not in_batch:
        if want_ahead:
            ahead = _HostDrawAhead(model, drawn_seeds)
        try:
            drawn = plan.run(draw, finish)
        finally:
            if ahead is not None:  # (a solve that raises must not leave the worker and its pending draw behind: ADVICE r5)
                ahead.close()
    else:
        together = geo_minimizer is not None and lanes is not None and os.environ.get("NK_GEO_THREADS", "1") != "0"
        if in_batch:
            solve = lambda jobs: batched.solve_together(model, linearisation(), jobs, controller_factory)  # noqa: E731
        else:
            solve = lambda jobs: _solve_on_lanes(model, lanes, linearisation(), jobs, controller_factory)  # noqa: E731
        drawn = plan.run_together(lambda seed: _draw_sources(model, linearisation(), seed, device_rng), solve,
                                  (lambda pair, mirrored: (pair, mirrored)) if together else finish)
        if together:  # geoVI: the non-linear fits of all local samples, one host thread and one lane per fit in flight
            if "tp" not in cache:
                cache["tp"] = model.trafo_point(position)
                cache["g_p"] = model.vjp_data(cache["tp"], cache["tp"].f, addend=position)
            fits = _fit_on_lanes(model, lanes, cache["tp"], cache["g_p"], position, drawn, geo_minimizer)
            drawn = [(f, False) for f in fits]
    return [r for r, _ in drawn], [n for _, n in drawn], plan.n_total


class _HostDrawAhead:
    """The host part of the prior draws (FusedModel.host_draws_before_xi) of a rank's samples, one sample ahead: `take(seed)`
    hands out the draws of `seed` -- started at construction for the first sample, at the previous `take` for the others --
    and starts the next sample's on a worker thread (numpy fills arrays without the interpreter lock)."""

    def __init__(self, model, seeds):
        from concurrent.futures import ThreadPoolExecutor

        self._model, self._seeds = model, list(seeds)
        self._pool = ThreadPoolExecutor(max_workers=1)
        self._pending = {}
        self._start(0)

    def _start(self, k):
        if k < len(self._seeds):
            self._pending[k] = self._pool.submit(self._model.host_draws_before_xi, self._seeds[k])

    def take(self, seed):
        k = next((j for j, sq in enumerate(self._seeds) if sq is seed), None)
        fut = self._pending.pop(k, None) if k is not None else None
        if k is not None:
            self._start(k + 1)
        if k is not None and k + 1 >= len(self._seeds):
            self._pool.shutdown(wait=False)
        return fut.result() if fut is not None else None

    def close(self):
        """Drop what was started and not taken (an exception between two samples, a seed that was never asked for)."""
        self._pending.clear()
        self._pool.shutdown(wait=False, cancel_futures=True)


def _fit_on_lanes(model, lanes, tp, g_p, position, jobs, minimizer):
    """The geoVI fits g(x) = g(p) +- b of several samples side by side (kl_energies.py:105-124, 148-155): lane k, driven by
    host thread k on its own stream, takes the samples k, k + K, ...  A fit is a whole Newton-CG minimisation with its own
    control flow (line searches, inner CGs), so the fits are interleaved by THREADS rather than by a lockstep loop: a thread
    that waits for its CG scalars lets the others enqueue (the waits release the interpreter lock).  Every lane has its own
    scratch, minimiser state and copy of the expansion point's amplitude `state`; the arithmetic of a fit is unchanged."""
    import copy
    from concurrent.futures import ThreadPoolExecutor

    main = torch.cuda.current_stream(model.device)
    K = min(len(lanes), len(jobs))
    points = []
    for _ in range(K):
        own = copy.copy(tp)
        own.state = tp.state.clone()
        points.append(own)
    for lane in lanes[1:K]:
        lane.stream.wait_stream(main)
    out = [None] * len(jobs)

    def work(k):
        lane, mine = lanes[k], copy.deepcopy(minimizer)
        torch.cuda.set_device(model.device)
        # (a worker thread starts on the DEFAULT stream: the lane without a stream of its own works on the caller's)
        with torch.cuda.stream(lane.stream if lane.stream is not None else main):
            for j in range(k, len(jobs), K):
                (b, y), mirrored = jobs[j]
                target = g_p - b if mirrored else g_p + b
                start = position - y if mirrored else position + y
                fit, _ = mine(FusedGeoEnergy(lane, points[k], target, start))
                out[j] = fit.position - position

    with ThreadPoolExecutor(max_workers=K) as pool:
        list(pool.map(work, range(K)))
    for lane in lanes[1:K]:
        main.wait_stream(lane.stream)
    for r in out:
        r.xi.record_stream(main)
        r.small.record_stream(main)
    return out


def _sampling_lanes(model, pairs):
    """Lanes for the linear solves of an MGVI iteration, or None: small grids only (NK_LANE_MAX_POINTS, default 2^25 points;
    NK_LANES lanes, default 4, 0 / 1 = off), at least two solves.  Models with a linear response take part (the sparse
    response is stateless: static CSR arrays and fresh outputs; NK_LANES_RESPONSE=0 keeps them on one stream)."""
    want = int(os.environ.get("NK_LANES", "4"))
    small = model.N <= int(os.environ.get("NK_LANE_MAX_POINTS", str(1 << 25)))
    if want < 2 or pairs < 2 or not small or (model.response is not None and os.environ.get("NK_LANES_RESPONSE", "1") == "0"):
        return None
    return model.lanes(min(want, pairs))


def _draw_sources(model, lp, seed, device_rng):
    """The random inputs of one linear sample (prior draw s, data-space noise pulled back: nj), in the sample's stream."""
    if device_rng is not None:
        device_rng.manual_seed(int(seed.generate_state(1, np.uint64)[0] >> np.uint64(1)))
    return model.draw_prior(device_rng), model.draw_lh_noise(lp, device_rng)


def _solve_on_lanes(model, lanes, lp, jobs, controller_factory):
    """FusedModel.draw_mgvi_sample for several samples at once: solve j runs on lane j mod K (own scratch, own stream), and
    minimization.ConjugateGradient.solve_many advances the solves of a wave together.  Same kernels, same arithmetic and
    the same bits as one solve after the other."""
    import copy

    main = torch.cuda.current_stream(model.device)
    pairs = []
    # the amplitude kernels keep scalars and scan aggregates of a JVP / VJP in the linearisation point's `state`: every lane
    # gets a copy of it (everything else of the point is read-only)
    points = []
    for lane in lanes:
        own = copy.copy(lp)
        own.state = lp.state.clone()
        points.append(own)
    for w0 in range(0, len(jobs), len(lanes)):
        wave = jobs[w0:w0 + len(lanes)]
        problems = []
        for lane, lp, (s, nj) in zip(lanes, points, wave):
            if lane.stream is not None:
                lane.stream.wait_stream(main)
            with (torch.cuda.stream(lane.stream) if lane.stream is not None else contextlib.nullcontext()):
                b = s + nj
                g0 = lane.lh_metric(lp, s, minus=nj)
                A = _Callable(lambda v, dot_out=None, cg_direction=None, lane=lane, lp=lp: lane.metric(lp, v, dot_out=dot_out,
                                                                                                       cg_direction=cg_direction),
                              fused_dot=lane.octant_vjp, fused_direction=lane.fused_direction)
                energy = QuadraticEnergy(s, A, b, _grad=g0)
                energy.consumable = True
            controller = controller_factory()
            if any(controller is other[1] for other in problems):
                # a caller that hands out ONE controller object (optimize_kl passes the user's sampling controller): solves that
                # run together must not share its iteration count and convergence state
                controller = copy.deepcopy(controller)
            problems.append((energy, controller, lane.stream, b))
        solved = ConjugateGradient(None).solve_many([p[:3] for p in problems])
        for (energy, _), (_, _, stream, b) in zip(solved, problems):
            if stream is not None:
                main.wait_stream(stream)
                for t in (b.xi, b.small, energy.position.xi, energy.position.small):
                    t.record_stream(main)
            pairs.append((b, energy.position))
    return pairs


def mgvi_iteration(model, mean, n_samples, controller_factory, kl_minimizer, mirror_samples=True, comm=None,
                   device_rng=None, geo_minimizer=None, on_phase=None):
    """One pass of the optimize_kl loop body (optimize_kl.py:357-451, no I/O): sample, then minimise.  on_phase(name): called
    after "sampling", "kl_construction" and "newton_cg" (bench.py hangs its per-phase clock there -- the timed code IS this
    function)."""
    residuals, negs, n_total = draw_samples(model, mean, n_samples, mirror_samples, controller_factory, comm, device_rng,
                                            geo_minimizer)
    if on_phase is not None:
        on_phase("sampling")
    kl = FusedKL(model, mean, residuals, negs, n_total, comm)
    if on_phase is not None:
        on_phase("kl_construction")
    with parallel.lockstep(comm):  # replicated minimiser: identical decisions on every rank
        kl, _ = kl_minimizer(kl)
    if on_phase is not None:
        on_phase("newton_cg")
    return kl.position, kl
