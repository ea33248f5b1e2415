"""LOSResponse + MaskOperator (BASELINE config 4) against vectors generated from the real reference
(tests/golden/make_golden.py::los_cases): the sparse matrices themselves, TIMES / ADJOINT_TIMES, and a config-4 style
model -- sigmoid(correlated field) -> LOS -> mask -> Gaussian noise with geoVI -- on the host and on the GPU."""
import numpy as np
import pytest

import nifty_amd as ift
from oracle import nifty_oracle as orc
from tests import goldenlib as gl
from tests.test_api_host import CF_ARGS

CASES = ("a", "b", "c", "d")


def _case(z, tag):
    sig = z[f"{tag}.sigmas"]
    return (tuple(int(i) for i in z[f"{tag}.shape"]), tuple(float(d) for d in z[f"{tag}.dist"]), z[f"{tag}.starts"],
            z[f"{tag}.ends"], None if np.isnan(sig).any() else sig)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_los_matrix_equals_reference(tag):
    z = gl.load("los")
    shape, dist, starts, ends, sig = _case(z, tag)
    dense = orc.los_dense(shape, dist, starts, ends, sig)
    assert np.max(np.abs(dense - z[f"{tag}.dense"])) <= 1e-7 * np.max(np.abs(z[f"{tag}.dense"]))
    assert gl.relerr(dense.astype(np.float64) @ z[f"{tag}.x"].reshape(-1), z[f"{tag}.times"]) < 1e-12


def test_oracle_mask_equals_reference():
    z = gl.load("los")
    assert np.array_equal(orc.mask_apply(z["mask.flags"], z["mask.x"]), z["mask.times"])
    assert np.array_equal(orc.mask_apply(z["mask.flags"], z["mask.y"], adjoint=True), z["mask.adjoint"])


def _check_los(tag, device_id):
    from scipy.sparse import csr_matrix

    z = gl.load("los")
    shape, dist, starts, ends, sig = _case(z, tag)
    sp = ift.RGSpace(shape, dist)
    R = ift.LOSResponse(sp, starts, ends, sig)
    assert R.target.shape == (starts.shape[1],)
    dense = csr_matrix((R._wgt, R._col, R._rowptr), shape=z[f"{tag}.dense"].shape).toarray()
    assert np.max(np.abs(dense - z[f"{tag}.dense"])) <= 1e-7 * np.max(np.abs(z[f"{tag}.dense"]))
    x = ift.makeField(R.domain, z[f"{tag}.x"], device_id)
    y = ift.makeField(R.target, z[f"{tag}.y"], device_id)
    rx, ry = R(x), R.adjoint(y)
    assert rx.device_id == device_id and ry.device_id == device_id
    assert gl.relerr(rx.asnumpy(), z[f"{tag}.times"]) < 1e-12
    assert gl.relerr(ry.asnumpy(), z[f"{tag}.adjoint"]) < 1e-12
    # adjointness (reference extra.py:220-231) incl. fp32 fields (float32 weights, fp64 accumulation)
    x32 = ift.makeField(R.domain, z[f"{tag}.x"].astype(np.float32), device_id)
    y32 = ift.makeField(R.target, z[f"{tag}.y"].astype(np.float32), device_id)
    a, b = float(R(x32).s_vdot(y32)), float(x32.s_vdot(R.adjoint(y32)))
    assert abs(a - b) < 1e-5 * max(abs(a), abs(b), 1e-30)


@pytest.mark.parametrize("tag", CASES)
def test_los_response_host(tag):
    _check_los(tag, -1)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_los_response_device(tag):
    _check_los(tag, 0)


def _check_mask(device_id):
    z = gl.load("los")
    sp = ift.RGSpace(z["mask.flags"].shape)
    M = ift.MaskOperator(ift.makeField(sp, z["mask.flags"]))
    assert M.target.shape == (int(np.logical_not(z["mask.flags"]).sum()),)
    assert np.array_equal(M(ift.makeField(M.domain, z["mask.x"], device_id)).asnumpy(), z["mask.times"])
    assert np.array_equal(M.adjoint(ift.makeField(M.target, z["mask.y"], device_id)).asnumpy(), z["mask.adjoint"])


def test_mask_operator_host():
    _check_mask(-1)


@pytest.mark.gpu
def test_mask_operator_device():
    _check_mask(0)


def test_los_argument_errors():
    sp = ift.RGSpace((8, 8))
    s, e = np.zeros((2, 3)), np.ones((2, 3))
    with pytest.raises(TypeError):
        ift.LOSResponse(ift.UnstructuredDomain(5), s, e)
    with pytest.raises(TypeError):
        ift.LOSResponse(sp, np.zeros((3, 3)), np.ones((3, 3)))
    with pytest.raises(TypeError):
        ift.LOSResponse(sp, s, np.ones((2, 4)))
    with pytest.raises(ValueError):  # 1/len - truncation*sigma < 0 (los_response.py:169-171)
        ift.LOSResponse(sp, s, e, sigmas=np.full(3, 10.0))
    with pytest.raises(TypeError):
        ift.MaskOperator(np.zeros(4))


def _c4_model(z, device_id):
    shape = (16, 16)
    sp = ift.RGSpace(shape)
    cfm = ift.CorrelatedFieldMaker("")
    cfm.add_fluctuations(sp, CF_ARGS["fluctuations"], CF_ARGS["flexibility"], CF_ARGS["asperity"], CF_ARGS["loglogavgslope"])
    cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
    cf = cfm.finalize()
    R = ift.LOSResponse(sp, z["c4.starts"], z["c4.ends"])
    Mk = ift.MaskOperator(ift.makeField(R.target, z["c4.flags"]))
    resp = Mk @ R @ cf.ptw("sigmoid")
    d = ift.makeField(resp.target, z["c4.data"], device_id)
    lh = ift.GaussianEnergy(d, ift.ScalingOperator(resp.target, 1.0 / 1e-3, np.float64)) @ resp
    return cf, resp, lh


def _check_c4(device_id, fuse=True):
    z = gl.load("los")
    cf, resp, lh = _c4_model(z, device_id)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "c4.x"), device_id)
    v = ift.MultiField.from_raw(cf.domain, gl.latent(z, "c4.v"), device_id)
    assert gl.relerr(resp(x).asnumpy(), z["c4.resp"]) < 1e-11
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=6)
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    hl = ham(ift.Linearization.make_var(x, want_metric=True))
    assert abs(float(hl.val.asnumpy()) - float(z["c4.ham_value"])) < 1e-10 * abs(float(z["c4.ham_value"]))
    assert gl.lat_relerr(hl.gradient.asnumpy(), gl.latent(z, "c4.ham_grad")) < 1e-9
    assert gl.lat_relerr(hl.metric(v).asnumpy(), gl.latent(z, "c4.ham_metric_v")) < 1e-9
    ift.random.push_sseq_from_seed(44)
    try:
        mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2),  # noqa: E731
                                    max_cg_iterations=8)
        nl = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=3, convergence_level=2),  # noqa: E731
                                    max_cg_iterations=6)
        nit = 2 if device_id < 0 else 1
        sl, mean = ift.optimize_kl(lh, nit, 1, mk, ic, nonlinear_sampling_minimizer=nl, output_directory=None,
                                   return_final_position=True, initial_position=x, device_id=device_id, fuse=fuse)
    finally:
        ift.random.pop_sseq()
    if device_id < 0:
        # geoVI: bounded CG lengths keep the run in the reproducible regime (see make_golden.py GEO_CG)
        assert gl.lat_relerr(mean.asnumpy(), gl.latent(z, "c4.okl_mean")) < 2e-3
        for i, s in enumerate(sl.iterator()):
            assert gl.lat_relerr(s.asnumpy(), gl.latent(z, f"c4.okl_sample{i}")) < 5e-3
        return
    # device: ONE iteration (okl1.npz) -- rounding differences between the device and the host arithmetic (1e-16) are
    # amplified to 1e-3..1e-2 by a discrete decision in the second iteration of this configuration (see
    # make_golden.py::okl1_cases); since round 3 every device sum of this model is built in a fixed order (nk_csr_rowsum)
    z1 = gl.load("okl1")
    lat = lambda pre: {k[len(pre) + 1:]: np.asarray(z1[k]) for k in z1.files if k.startswith(pre + ".")}  # noqa: E731
    assert gl.lat_relerr(mean.asnumpy(), lat("c4.mean")) < 1e-6
    for i, s in enumerate(sl.iterator()):
        assert gl.lat_relerr(s.asnumpy(), lat(f"c4.sample{i}")) < 1e-6


def test_config4_model_host():
    _check_c4(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [True, False])
def test_config4_model_device(fuse):
    """fuse=True: the fusion pass of optimize_kl recognises Mask @ LOSResponse @ sigmoid(cf) and runs the iteration on the
    fused response engine; fuse=False: the generic operator graph.  Both against the reference's golden run."""
    _check_c4(0, fuse)


@pytest.mark.gpu
def test_spmv_kernels_against_scipy_on_a_larger_response():
    """nk_spmv / nk_spmv_t on a 2000-line response over a 256 x 192 grid (rows of a few hundred entries, many lines per
    pixel) against scipy.sparse on the host; fp64 and fp32 fields."""
    from scipy.sparse import csr_matrix

    rng = np.random.default_rng(8)
    sp = ift.RGSpace((256, 192), (1.0 / 256, 1.0 / 192))
    R = ift.LOSResponse(sp, rng.uniform(size=(2, 2000)), rng.uniform(size=(2, 2000)))
    m = csr_matrix((R._wgt, R._col, R._rowptr), shape=(2000, sp.size))
    x, y = rng.normal(size=sp.shape), rng.normal(size=2000)
    for dt, tol in ((np.float64, 1e-12), (np.float32, 2e-6)):
        got = R(ift.makeField(R.domain, x.astype(dt), 0)).asnumpy()
        ref = m @ x.astype(dt).astype(np.float64).reshape(-1)
        assert gl.relerr(got, ref) < tol
        got = R.adjoint(ift.makeField(R.target, y.astype(dt), 0)).asnumpy()
        ref = (m.T @ y.astype(dt).astype(np.float64)).reshape(sp.shape)
        assert gl.relerr(got, ref) < tol


# ---- the response re-ordered by grid tiles (round 6: nk_tiled_rowsum) and the staged short-row sums -----------------------
def _random_csr(rng, n_rows, n_cols, lengths):
    rowptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    col = np.concatenate([np.sort(rng.choice(n_cols, size=k, replace=False)) for k in lengths] + [np.zeros(0, dtype=np.int64)])
    return rowptr, col.astype(np.int32), rng.normal(size=len(col)).astype(np.float32)


@pytest.mark.parametrize("shape,tile", [((150, 200), (32, 64)), ((150, 200), (16, 16)), ((37,), (None, None)),
                                        ((5, 40, 70), (32, 64)), ((64, 64), (64, 64)), ((300, 20), (8, 4))])
def test_tiled_plan_reproduces_the_matrix(shape, tile):
    """los_response.tiled_plan (the set-up side of nk_tiled_rowsum): the two-launch sum of its segments equals the CSR
    product for ragged tiles, 1-D / 3-D grids, empty rows, rows longer than a segment and tiles with more blocks than a
    wavefront step."""
    from scipy.sparse import csr_matrix

    from nifty_amd import los_response as lr

    rng = np.random.default_rng(3)
    n = int(np.prod(shape))
    lengths = rng.integers(0, min(n, 900), size=23)
    lengths[[2, 7]] = 0                     # empty rows
    lengths[3] = min(n, 5000)               # a row that fills whole tiles: segments are split, tiles overflow one window
    rowptr, col, wgt = _random_csr(rng, len(lengths), n, lengths)
    plan = lr.tiled_plan(rowptr, col, wgt, shape, *tile)
    x = rng.normal(size=shape)
    ref = csr_matrix((wgt, col, rowptr), shape=(len(lengths), n)) @ x.reshape(-1)
    assert gl.relerr(lr.tiled_rowsum_host(plan, x), ref) < 1e-13
    # every slot is used, a piece (one slot) is at most 16 consecutive blocks, padding carries weight zero
    slots = plan["blk_slot"]
    assert sorted(set(slots.tolist())) == list(range(plan["n_slots"])) and plan["row_slot"][-1] == plan["n_slots"]
    runs = np.diff(np.concatenate([[0], np.nonzero(np.diff(slots))[0] + 1, [len(slots)]]))
    assert runs.max() <= lr.TILED_SEGMAX // lr.TILED_BLOCK and len(runs) == plan["n_slots"]
    assert len(plan["loc"]) == lr.TILED_BLOCK * len(slots) == len(plan["wgt"]) and plan["item_blk"][-1] == len(slots)
    assert np.count_nonzero(plan["wgt"]) == np.count_nonzero(wgt)
    # an empty matrix is a valid plan
    empty = lr.tiled_plan(np.zeros(4, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.float32), shape, *tile)
    assert empty["n_items"] == 0 and np.array_equal(lr.tiled_rowsum_host(empty, x), np.zeros(3))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,tile", [((150, 200), (32, 64)), ((37,), (None, None)), ((5, 40, 70), (16, 32)), ((256, 192), (64, 64))])
def test_tiled_rowsum_device_equals_its_host_emulation_bit_for_bit(shape, tile):
    """nk_tiled_rowsum against tiled_rowsum_host (the same additions in the same order: identical bits in fp64), against
    scipy (fp32 fields), batched = single calls, same bits on every call."""
    import torch
    from scipy.sparse import csr_matrix

    from nifty_amd import backend as B
    from nifty_amd import los_response as lr

    rng = np.random.default_rng(4)
    n = int(np.prod(shape))
    lengths = rng.integers(0, min(n, 2500), size=57)
    lengths[5] = 0
    lengths[11] = min(n, 9000)
    rowptr, col, wgt = _random_csr(rng, len(lengths), n, lengths)
    plan = lr.tiled_plan(rowptr, col, wgt, shape, *tile)
    tm = B.TiledMatrix(plan, torch.device("cuda:0"))
    xs = [rng.normal(size=shape) for _ in range(5)]
    want = [lr.tiled_rowsum_host(plan, x) for x in xs[:2]]
    dev = [torch.from_numpy(x).cuda() for x in xs]
    single = [tm.rowsum([d.reshape(-1)], [torch.empty(len(lengths), dtype=torch.float64, device="cuda")])[0] for d in dev]
    assert np.array_equal(single[0].cpu().numpy(), want[0]) and np.array_equal(single[1].cpu().numpy(), want[1])
    outs = tm.rowsum([d.reshape(-1) for d in dev], [torch.empty(len(lengths), dtype=torch.float64, device="cuda") for _ in dev])
    assert all(torch.equal(a, b) for a, b in zip(outs, single))
    m = csr_matrix((wgt, col, rowptr), shape=(len(lengths), n))
    x32 = torch.from_numpy(xs[0].astype(np.float32)).cuda()
    y32 = tm.rowsum([x32.reshape(-1)], [torch.empty(len(lengths), dtype=torch.float32, device="cuda")])[0]
    assert gl.relerr(y32.cpu().numpy(), m @ xs[0].astype(np.float32).astype(np.float64).reshape(-1)) < 2e-6


def _fma_rowsum(rowptr, col, wgt, x):
    """Per row the sequential fused multiply-adds of k_csr_rowsum<T, 1>, exactly (rational arithmetic, one rounding each)."""
    from fractions import Fraction

    out = np.zeros(len(rowptr) - 1)
    for r in range(len(out)):
        acc = 0.0
        for j in range(rowptr[r], rowptr[r + 1]):
            w = 1.0 if wgt is None else float(wgt[j])
            acc = float(Fraction(w) * Fraction(float(x[col[j]])) + Fraction(acc))
        out[r] = acc
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("weighted", [True, False])
def test_staged_short_row_sums_keep_the_order_of_one_thread_per_row(weighted):
    """nk_csr_rowsum with lanes = 1 (the transposed response, bin sums) runs staged through LDS since round 6: the additions of
    a row are still the sequential fused multiply-adds in entry order -- checked exactly -- for rows of 0 ... 5000 entries
    (runs that span several windows and workgroups), single and batched, fp64 and fp32."""
    import torch

    from nifty_amd import _lib as L
    from nifty_amd import backend as B
    from nifty_amd import batched

    rng = np.random.default_rng(6)
    n_cols = 3000
    lengths = np.concatenate([rng.integers(0, 5, size=2600), [5000, 0, 2300], rng.integers(0, 40, size=300)])
    rowptr = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    col = rng.integers(0, n_cols, size=rowptr[-1]).astype(np.int32)
    wgt = rng.normal(size=rowptr[-1]).astype(np.float32) if weighted else None
    xs = [rng.normal(size=n_cols) for _ in range(5)]
    dev = lambda a: None if a is None else torch.from_numpy(a).cuda()  # noqa: E731
    rp, cl, wg = dev(rowptr), dev(col), dev(wgt)
    nrows = len(lengths)

    def run(x):
        y = torch.empty(nrows, dtype=x.dtype, device="cuda")
        L.check(L.load().nk_csr_rowsum(nrows, rp.data_ptr(), cl.data_ptr(), B.ptr(wg), x.data_ptr(), y.data_ptr(),
                                       B.dtype_code(x), 1, B._stream()), "nk_csr_rowsum")
        return y

    got = run(dev(xs[0]))
    assert np.array_equal(got.cpu().numpy(), _fma_rowsum(rowptr, col, wgt, xs[0]))
    singles = [run(dev(x)) for x in xs]
    ys = [torch.empty(nrows, dtype=torch.float64, device="cuda") for _ in xs]
    batched.rowsum((rp, cl, wg) if weighted else (rp, cl, 1), [dev(x) for x in xs], ys, weighted=True if weighted else None,
                   lanes=1, nrows=nrows)
    assert all(torch.equal(a, b) for a, b in zip(ys, singles))
    x32 = dev(xs[1].astype(np.float32))
    ref = _fma_rowsum(rowptr, col, wgt, xs[1].astype(np.float32).astype(np.float64)).astype(np.float32)
    assert np.array_equal(run(x32).cpu().numpy(), ref)
