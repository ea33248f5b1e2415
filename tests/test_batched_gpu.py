"""Batched launches (include/niftyk.h "batched launches", nifty_amd/batched.py): ONE launch per kernel for all members of a
batch -- the samples of a KL evaluation / metric application (reference kl_energies.py:306-350) and the linear sampling solves
of an iteration (kl_energies.py:132-158) -- must give the BITS of the member-by-member paths: every `*_batch` entry point
against `count` single calls through the C ABI, and a whole MGVI / geoVI iteration with NK_BATCH=1 against the stream lanes
(NK_BATCH=0) and the plain sequential path (NK_LANES=0)."""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _rand(n, dtype, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randn(n, dtype=dtype, device=DEV, generator=g)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", [4099, 1 << 16, 3 * (1 << 20) + 8, 1 << 22])
def test_vector_batches_equal_single_calls(dtype, n):
    from nifty_amd import _lib as L
    from nifty_amd import backend as B
    from nifty_amd import batched as Bt

    lib = L.load()
    code = L.NK_F64 if dtype == torch.float64 else L.NK_F32
    for count in (1, 3, 8):
        xs = [_rand(n, dtype, 10 + m) for m in range(count)]
        ys = [_rand(n, dtype, 30 + m) for m in range(count)]
        alphas = [0.5 + m for m in range(count)]
        betas = [(-1.0) ** m * 0.25 for m in range(count)]
        # axpby (with and without y), binary, sqnorm, vdot
        got = Bt.axpby(alphas, xs, betas, ys)
        for m in range(count):
            assert torch.equal(got[m], B.axpby(alphas[m], xs[m], betas[m], ys[m]))
        got = Bt.axpby(alphas, xs, [0.0] * count, [None if m % 2 else ys[m] for m in range(count)])
        for m in range(count):
            assert torch.equal(got[m], B.axpby(alphas[m], xs[m], 0.0, None if m % 2 else ys[m]))
        got = Bt.binary(L.OP_MUL, xs, ys)
        for m in range(count):
            assert torch.equal(got[m], B.binary(L.OP_MUL, xs[m], ys[m]))
        res = list(torch.full((count, 1), 3.0, dtype=torch.float64, device=DEV).unbind(0))
        got = Bt.axpby_sqnorm(alphas, xs, betas, ys, res)
        for m in range(count):
            one = torch.zeros(1, dtype=torch.float64, device=DEV)
            assert torch.equal(got[m], B.axpby_sqnorm(alphas[m], xs[m], betas[m], ys[m], one))
            assert float(res[m]) == float(one)
        Bt.vdot(xs, ys, res, accumulate=True)
        for m in range(count):
            one = torch.zeros(1, dtype=torch.float64, device=DEV)
            B.axpby_sqnorm(alphas[m], xs[m], betas[m], ys[m], one)
            B.vdot(xs[m], ys[m], result=one, accumulate=True)
            assert float(res[m]) == float(one)
        # the pairwise sum over members == parallel.tree_fold with one rounded addition per merge
        from nifty_amd import parallel

        want = parallel.tree_fold([x.clone() for x in xs], lambda a, b: B.axpby(1.0, a, 1.0, b, out=a))
        out = torch.empty_like(xs[0])
        assert torch.equal(Bt.sum_tree(xs, out=out), want)
        # conjugate-gradient updates of `count` solves: scalars and vectors
        for recur in (True, False):
            X, R, D, Q, Bv = ([_rand(n, dtype, 100 * k + m) for m in range(count)] for k in range(1, 6))
            X1, R1, D1 = [t.clone() for t in X], [t.clone() for t in R], [t.clone() for t in D]
            sc = torch.rand(count, 8, dtype=torch.float64, device=DEV) + 0.5
            sc1 = sc.clone()
            rows, rows1 = [sc[m] for m in range(count)], [sc1[m] for m in range(count)]
            P = L.ptr_array
            L.check(lib.nk_cg_curv_batch(n, count, P(D), P(Q), code, P(rows), 0, B._stream()))
            if recur:
                L.check(lib.nk_cg_update_dr_batch(n, count, P(X), P(R), P(D), P(Q), code, P(rows), 0, B._stream()))
            else:
                L.check(lib.nk_cg_update_batch(n, count, P(X), P(R), P(D), P(Q), P(Bv), code, P(rows), 0, B._stream()))
            L.check(lib.nk_cg_direction_batch(n, count, P(D), P(R), code, P(rows), 1, B._stream()))
            for m in range(count):
                L.check(lib.nk_cg_curv(n, D1[m].data_ptr(), Q[m].data_ptr(), code, rows1[m].data_ptr(), 0, B._stream()))
                if recur:
                    L.check(lib.nk_cg_update_dr(n, X1[m].data_ptr(), R1[m].data_ptr(), D1[m].data_ptr(), Q[m].data_ptr(), code,
                                                rows1[m].data_ptr(), 0, B._stream()))
                else:
                    L.check(lib.nk_cg_update(n, X1[m].data_ptr(), R1[m].data_ptr(), D1[m].data_ptr(), Q[m].data_ptr(),
                                             Bv[m].data_ptr(), code, rows1[m].data_ptr(), 0, B._stream()))
                L.check(lib.nk_cg_direction(n, D1[m].data_ptr(), R1[m].data_ptr(), code, rows1[m].data_ptr(), 1, B._stream()))
                assert torch.equal(X[m], X1[m]) and torch.equal(R[m], R1[m]) and torch.equal(D[m], D1[m])
            assert torch.equal(sc, sc1)


def test_gather_and_rowsum_batches():
    from nifty_amd import _lib as L
    from nifty_amd import backend as B
    from nifty_amd import batched as Bt

    nb, n = 5000, 257 * 513
    idx = torch.randint(0, nb, (n,), dtype=torch.int32, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    tables = [_rand(nb, torch.float64, m) for m in range(5)]
    outs = [torch.empty(n, dtype=torch.float64, device=DEV) for _ in range(5)]
    L.check(L.load().nk_gather_batch(n, 5, L.ptr_array(tables), idx.data_ptr(), L.ptr_array(outs), L.NK_F64, B._stream()))
    for t, o in zip(tables, outs):
        assert torch.equal(o, B.gather(t, idx, (n,)))
    plan = B.bin_plan(idx, nb)
    ys = [torch.empty(nb, dtype=torch.float64, device=DEV) for _ in range(5)]
    Bt.rowsum(plan, outs, ys, dtype_code=L.NK_F64)
    for o, y in zip(outs, ys):
        assert torch.equal(y, B.bin_sum(o, plan))


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("weighted", [False, True])
def test_short_row_sums_of_a_batch_for_every_member_count(dtype, weighted):
    """nk_csr_rowsum_batch with ONE lane per row (short rows: the bin sums of a static index map) handles four members per
    thread, the index read once (nk_vec.hip, k_csr_rowsum_b): every count 1 .. 8 -- full and ragged slices of four -- gives
    the bits of `count` single nk_csr_rowsum launches, unweighted and with float weights (the fused multiply-add kept)."""
    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    lib = L.load()
    nrows, ncols = 70001, 50000
    g = torch.Generator(device=DEV).manual_seed(3)
    counts = torch.randint(0, 6, (nrows,), device=DEV, generator=g)  # 0 .. 5 entries per row (empty rows included)
    rowptr = torch.zeros(nrows + 1, dtype=torch.int64, device=DEV)
    rowptr[1:] = torch.cumsum(counts, 0)
    nnz = int(rowptr[-1].item())
    col = torch.randint(0, ncols, (nnz,), dtype=torch.int32, device=DEV, generator=g)
    wgt = torch.rand(nnz, dtype=torch.float32, device=DEV, generator=g) if weighted else None
    code = L.NK_F64 if dtype == torch.float64 else L.NK_F32
    xs = [_rand(ncols, dtype, 10 + m) for m in range(8)]
    single = []
    for x in xs:
        y = torch.empty(nrows, dtype=dtype, device=DEV)
        L.check(lib.nk_csr_rowsum(nrows, rowptr.data_ptr(), col.data_ptr(), B.ptr(wgt), x.data_ptr(), y.data_ptr(), code, 1,
                                  B._stream()), "nk_csr_rowsum")
        single.append(y)
    for count in range(1, 9):
        ys = [torch.full((nrows,), float("nan"), dtype=dtype, device=DEV) for _ in range(count)]
        L.check(lib.nk_csr_rowsum_batch(nrows, rowptr.data_ptr(), col.data_ptr(), B.ptr(wgt), count, L.ptr_array(xs[:count]),
                                        L.ptr_array(ys), code, 1, B._stream()), "nk_csr_rowsum_batch")
        for m in range(count):
            assert torch.equal(ys[m], single[m]), (count, m)


def test_roll_of_the_cg_scalars_can_clear_the_curvature_slot():
    """nk_cg_direction(roll): 1 leaves scal[1] (the fused direction update rolls AFTER the next d.q was deposited), 2 clears
    it as well (a caller that accumulates the next d.q afterwards: batched.solve_together) -- single and batched entry."""
    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    lib = L.load()
    start = torch.tensor([4.0, 8.0, 2.0, 0.25, 0.125, 0.0, 0.0, 0.0], dtype=torch.float64, device=DEV)
    for roll, slot1 in ((1, 8.0), (2, 0.0)):
        sc = start.clone()
        L.check(lib.nk_cg_direction(0, None, None, L.NK_F64, sc.data_ptr(), roll, B._stream()), "nk_cg_direction")
        # gamma_prev <- gamma, alpha = 4 / 8, beta = 2 / 4, reduction slots cleared
        assert sc.tolist() == [2.0, slot1, 0.0, 0.0, 0.0, 0.5, 0.5, 0.0]
        rows = [start.clone() for _ in range(3)]
        d = [_rand(4096, torch.float64, 40 + m) for m in range(3)]
        r = [_rand(4096, torch.float64, 50 + m) for m in range(3)]
        want = [0.5 * a + b for a, b in zip(d, r)]
        L.check(lib.nk_cg_direction_batch(4096, 3, L.ptr_array(d), L.ptr_array(r), L.NK_F64, L.ptr_array(rows), roll,
                                          B._stream()), "nk_cg_direction_batch")
        for row, got, ref in zip(rows, d, want):
            assert row.tolist() == [2.0, slot1, 0.0, 0.0, 0.0, 0.5, 0.5, 0.0]
            assert torch.equal(got, ref)


def test_amplitude_batches_equal_single_calls():
    from nifty_amd import _lib as L
    from nifty_amd import backend as B
    from nifty_amd.engine import FusedModel, LatentVec

    lib = L.load()
    model = FusedModel((512, 512), offset_mean=2.0, likelihood="gaussian", data=np.zeros((512, 512)), icov=1.0, device=DEV)
    nb, count = model.nb, 6
    lats = [0.3 * _rand(model.nsmall, torch.float64, 50 + m) for m in range(count)]
    dlats = [_rand(model.nsmall, torch.float64, 70 + m) for m in range(count)]
    abars = [_rand(nb, torch.float64, 90 + m) for m in range(count)]
    new = lambda size: [torch.empty(size, dtype=torch.float64, device=DEV) for _ in range(count)]  # noqa: E731
    st, amp, damp, lb = new(8 * nb + 16), new(nb), new(nb), new(model.nsmall)
    st1, amp1, damp1, lb1 = new(8 * nb + 16), new(nb), new(nb), new(model.nsmall)
    P = L.ptr_array
    geo, hyp, s_ = model.geo.data_ptr(), model.hyp.data_ptr(), B._stream()
    L.check(lib.nk_amp_forward_batch(nb, geo, hyp, count, P(lats), P(st), P(amp), s_))
    L.check(lib.nk_amp_jvp_batch(nb, geo, hyp, count, P(lats), P(st), P(dlats), P(damp), s_))
    L.check(lib.nk_amp_vjp_batch(nb, geo, hyp, count, P(lats), P(st), P(abars), P(lb), s_))
    for m in range(count):
        L.check(lib.nk_amp_forward(nb, geo, hyp, lats[m].data_ptr(), st1[m].data_ptr(), amp1[m].data_ptr(), s_))
        L.check(lib.nk_amp_jvp(nb, geo, hyp, lats[m].data_ptr(), st1[m].data_ptr(), dlats[m].data_ptr(), damp1[m].data_ptr(), s_))
        L.check(lib.nk_amp_vjp(nb, geo, hyp, lats[m].data_ptr(), st1[m].data_ptr(), abars[m].data_ptr(), lb1[m].data_ptr(), s_))
        assert torch.equal(amp[m], amp1[m]) and torch.equal(damp[m], damp1[m]) and torch.equal(lb[m], lb1[m])
    with pytest.raises(ValueError):  # members must not share the state they scribble on
        L.check(lib.nk_amp_jvp_batch(nb, geo, hyp, 2, P(lats[:2]), P([st[0], st[0]]), P(dlats[:2]), P(damp[:2]), s_))


def _kl_problem(shape, likelihood, dtype=torch.float64, response=False):
    from nifty_amd import random
    from nifty_amd.engine import FusedModel

    kw = {}
    if response:
        from nifty_amd.los_response import SparseResponse, los_matrix

        rng = np.random.default_rng(5)
        starts, ends = rng.uniform(size=(2, 300)), rng.uniform(size=(2, 300))
        rowptr, col, wgt = los_matrix(shape, tuple(1.0 / n for n in shape), starts, ends)
        kw["response"] = SparseResponse(rowptr, col, wgt, int(np.prod(shape)))
    model = FusedModel(shape, offset_mean=2.0, likelihood=likelihood, icov=100.0,
                       nonlin="exp" if likelihood == "poisson" else ("sigmoid" if response else None), dtype=dtype, device=DEV, **kw)
    random.push_sseq_from_seed(17)
    try:
        truth = model.draw_prior()
        sig = model.signal(truth)
        if response:
            model.set_data(kw["response"].times(sig), 100.0)
        elif likelihood == "poisson":
            model.set_data(torch.poisson(sig.double(), generator=torch.Generator(device=DEV).manual_seed(2)).to(torch.int64))
        else:
            model.set_data(sig, 100.0)
        mean = 0.1 * model.draw_prior()
    finally:
        random.pop_sseq()
    return model, mean


def _iteration(shape, likelihood, pairs, env, monkeypatch, geo=False, response=False, dtype=torch.float64):
    from nifty_amd import batched, random
    from nifty_amd.engine import mgvi_iteration
    from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    model, mean = _kl_problem(shape, likelihood, dtype, response)
    assert batched.ready(model) == (env.get("NK_BATCH", "1") != "0")
    random.push_sseq_from_seed(23)
    try:
        geo_min = NewtonCG(AbsDeltaEnergyController(0.5, iteration_limit=2, convergence_level=2), max_cg_iterations=4) if geo else None
        pos, kl = mgvi_iteration(model, mean, pairs, lambda: AbsDeltaEnergyController(0.05, iteration_limit=22),
                                 NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=6),
                                 geo_minimizer=geo_min)
    finally:
        random.pop_sseq()
    v = 0.5 * mean
    mv = kl.apply_metric(v)
    return dict(pos_xi=pos.xi.clone(), pos_small=pos.small.clone(), value=kl.value, grad_xi=kl.gradient.xi.clone(),
                grad_small=kl.gradient.small.clone(), mv_xi=mv.xi.clone(), mv_small=mv.small.clone(),
                residuals=[r.xi.clone() for r in kl.residuals], counters=dict(model.counters))


def _same(a, b):
    assert a["value"] == b["value"]
    for k in ("pos_xi", "pos_small", "grad_xi", "grad_small", "mv_xi", "mv_small"):
        assert torch.equal(a[k], b[k]), k
    assert len(a["residuals"]) == len(b["residuals"])
    for x, y in zip(a["residuals"], b["residuals"]):
        assert torch.equal(x, y)
    assert a["counters"]["metric"] == b["counters"]["metric"] and a["counters"]["value_grad"] == b["counters"]["value_grad"]


@pytest.mark.parametrize("shape,likelihood,pairs", [((512, 512), "poisson", 4), ((1024, 512), "gaussian", 3), ((512, 512), "gaussian", 5)])
def test_mgvi_iteration_batched_equals_lanes_and_sequential(shape, likelihood, pairs, monkeypatch):
    """sampling solves (22 CG iterations: through the residual refresh at 20), KL value / gradient / metric, a Newton-CG
    minimisation: batched == four stream lanes == one chain after the other, bit for bit; 5 pairs = 10 samples also runs a
    second wave of the batch"""
    batch = _iteration(shape, likelihood, pairs, {"NK_BATCH": "1"}, monkeypatch)
    lanes = _iteration(shape, likelihood, pairs, {"NK_BATCH": "0"}, monkeypatch)
    plain = _iteration(shape, likelihood, pairs, {"NK_BATCH": "0", "NK_LANES": "0"}, monkeypatch)
    _same(batch, lanes)
    _same(batch, plain)


def test_response_model_batched_equals_lanes(monkeypatch):
    """sigmoid(cf) -> sparse line-of-sight response -> Gaussian noise (BASELINE config 4's model), MGVI and geoVI"""
    for geo in (False, True):
        batch = _iteration((512, 512), "gaussian", 2, {"NK_BATCH": "1"}, monkeypatch, geo=geo, response=True)
        lanes = _iteration((512, 512), "gaussian", 2, {"NK_BATCH": "0"}, monkeypatch, geo=geo, response=True)
        _same(batch, lanes)


def test_fp32_model_is_batched_too(monkeypatch):
    monkeypatch.setenv("NK_WIDE_FORWARD", "0")  # (the wide forward transform of fp32 models keeps the unbatched evaluation)
    batch = _iteration((512, 512), "gaussian", 2, {"NK_BATCH": "1"}, monkeypatch, dtype=torch.float32)
    lanes = _iteration((512, 512), "gaussian", 2, {"NK_BATCH": "0"}, monkeypatch, dtype=torch.float32)
    _same(batch, lanes)


def test_transform_batch_equals_single_calls():
    """nk_hartley_fused_batch against nk_hartley_fused member by member (PLAIN -> MUL epilogue classes), on a plan with
    batched twins and on one without (member-by-member fallback inside the library)"""
    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    lib = L.load()
    for shape, twins in (((512, 1024), True), ((256, 256), False), ((64, 64, 128), False)):
        plan = B.get_plan(shape, torch.float64, 1, torch.device(DEV))
        assert bool(lib.nk_plan_batch_ok(plan.handle)) == (len(shape) == 2)
        count = 4
        views = [plan] + [B.PlanView(plan) for _ in range(count - 1)]
        xs = [_rand(int(np.prod(shape)), torch.float64, 5 + m).reshape(shape) for m in range(count)]
        muls = [_rand(int(np.prod(shape)), torch.float64, 55 + m).reshape(shape) for m in range(count)]
        outs = [torch.empty_like(x) for x in xs]
        outs1 = [torch.empty_like(x) for x in xs]
        fuses = []
        for m in range(count):
            f = L.Fuse()
            f.pro, f.in_, f.epi, f.out = L.PRO_PLAIN, xs[m].data_ptr(), L.EPI_MUL, outs[m].data_ptr()
            f.scale, f.mul, f.mul_scalar, f.addend_scale = 0.5 + m, muls[m].data_ptr(), 2.0, 1.0
            fuses.append(f)
        arr = (L.Fuse * count)(*fuses)
        L.check(lib.nk_hartley_fused_batch(plan.handle, arr, count, B._convention(), L.ptr_array([v.workspace for v in views]),
                                           B._stream()))
        for m in range(count):
            fuses[m].out = outs1[m].data_ptr()
            B.hartley_fused(views[m], fuses[m])
            assert torch.equal(outs[m], outs1[m]), (shape, m)
        with pytest.raises(ValueError):  # members must not share a workspace
            L.check(lib.nk_hartley_fused_batch(plan.handle, arr, 2, B._convention(),
                                               L.ptr_array([plan.workspace, plan.workspace]), B._stream()))
        del twins
