"""Rank-count-independent arithmetic on the device (reference utilities.py:349-414 and
test_mpi/test_kl.py:104-114): the pieces that make 1, 2, 4 and 8 ranks produce the same bits.

* the reductions of libniftyk group a long array into 64 units; a shard made of whole units (nk_red_layout) delivers the
  unit sums, and the exchanged unit sums added in order (nk_red_finish) are the bits of the reduction of the whole array;
* the VJP epilogue adds a sample's contribution to carried partial sums as plain additions of rounded values
  (nk_fuse.carry1 / carry2), so the pairwise sum over samples built inside the epilogues -- single launches and pair
  launches -- equals the same tree built from separately stored contributions.
The end-to-end statement (one MGVI iteration on 1 / 2 / 4 ranks) is tests/test_distributed_gloo.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,n", [(torch.float32, 1 << 22), (torch.float64, 1 << 21), (torch.float32, 3 * (1 << 18))])
def test_unit_reductions_of_shards_give_the_bits_of_the_full_vector(dtype, n):
    from nifty_amd import _lib as L
    from nifty_amd import backend as B

    lib = L.load()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(n, dtype=dtype, device=dev, generator=g)
    y = torch.randn(n, dtype=dtype, device=dev, generator=g)
    code = B.dtype_code(x)
    unit = int(lib.nk_red_unit(n, code))
    assert unit == n // 64
    full_dot = B.vdot(x, y)
    full_stats = torch.empty(3, dtype=torch.float64, device=dev)
    L.check(lib.nk_stats(n, x.data_ptr(), code, full_stats.data_ptr(), B._stream()))
    assert abs(float(full_dot) - float(torch.dot(x.double(), y.double()))) < 1e-12 * n
    for ranks, chunks in ((2, 1), (2, 8), (4, 4), (8, 8), (8, 1), (64, 1)):
        m = n // (ranks * chunks)
        per_segment = 64 // (ranks * chunks)
        total_dot = torch.zeros(64, dtype=torch.float64, device=dev)
        total_stats = torch.zeros(3 * 64, dtype=torch.float64, device=dev)
        for rank in range(ranks):
            xs = x.view(chunks, ranks, m)[:, rank, :].contiguous().view(-1)
            ys = y.view(chunks, ranks, m)[:, rank, :].contiguous().view(-1)
            units = torch.full((3 * 64,), np.nan, dtype=torch.float64, device=dev)
            sink = torch.full((3,), 7.0, dtype=torch.float64, device=dev)  # untouched: the kernel delivers unit sums instead
            L.check(lib.nk_red_layout(unit, 64 // ranks, 64, per_segment, 64 // chunks, rank * per_segment, units.data_ptr()))
            try:
                L.check(lib.nk_vdot(xs.numel(), xs.data_ptr(), ys.data_ptr(), code, sink.data_ptr(), 1, B._stream()))
                total_dot += units[:64]
                assert int((units[:64] != 0).sum()) == 64 // ranks
                assert torch.equal(sink.cpu(), torch.full((3,), 7.0, dtype=torch.float64))
                L.check(lib.nk_stats(xs.numel(), xs.data_ptr(), code, sink.data_ptr(), B._stream()))
                total_stats += units
            finally:
                lib.nk_red_layout(0, 0, 0, 0, 0, 0, 0)
        got = torch.zeros(3, dtype=torch.float64, device=dev)
        L.check(lib.nk_red_finish(total_dot.data_ptr(), 64, 1, got.data_ptr(), 0, B._stream()))
        assert float(got[0]) == float(full_dot), (ranks, chunks)
        L.check(lib.nk_red_finish(total_stats.data_ptr(), 64, 3, got.data_ptr(), 0, B._stream()))
        assert torch.equal(got[:2], full_stats[:2]), (ranks, chunks)
    # a length that does not qualify is one unit: the announcement is ignored, the result lands where it always did
    odd = x[:n - 5]
    assert int(lib.nk_red_unit(odd.numel(), code)) == 0
    assert abs(float(B.vdot(odd, odd)) - float((odd.double() ** 2).sum())) < 1e-9 * odd.numel()


def _kl(shape, dtype, pairs, likelihood="gaussian"):
    from nifty_amd import random
    from nifty_amd.engine import FusedKL, FusedModel, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController

    model = FusedModel(shape, offset_mean=2.0, likelihood=likelihood, icov=100.0, nonlin="exp" if likelihood == "poisson" else None,
                       dtype=dtype, device="cuda:0")
    random.push_sseq_from_seed(17)
    try:
        truth = model.draw_prior()
        if likelihood == "poisson":
            model.set_data(torch.poisson(model.signal(truth).double()).to(torch.int64))
        else:
            model.set_data(model.signal(truth), 100.0)
        mean = 0.1 * model.draw_prior()
        res, negs, n_total = draw_samples(model, mean, pairs, True, lambda: AbsDeltaEnergyController(0.05, iteration_limit=3))
        direction = model.draw_prior()
    finally:
        random.pop_sseq()
    return model, mean, direction, FusedKL(model, mean, res, negs, n_total)


@pytest.mark.parametrize("shape,dtype,pairs", [((64, 64, 64), torch.float32, 4), ((64, 64, 64), torch.float64, 4),
                                               ((64, 64, 64), torch.float32, 3), ((64, 64, 64), torch.float32, 8),
                                               ((12, 10, 14), torch.float64, 4), ((256, 128), torch.float64, 2)])
def test_pairwise_sum_inside_the_epilogues_equals_the_tree_of_stored_terms(shape, dtype, pairs, monkeypatch):
    """KL gradient and metric application with the samples added in pair_tree order INSIDE the VJP epilogues (carried
    partial sums, pair launches on 3-D sandwich plans) against the same tree over per-sample vectors computed one by one
    and added with torch: bit-identical.  6 samples exercise the last sample's merge of unequal partial sums, 16 the
    explicit additions beyond two carries."""
    from nifty_amd import parallel
    from nifty_amd.engine import LatentVec

    monkeypatch.setenv("NK_LANES", "0")  # (the lanes of small grids keep one vector per sample anyway)
    model, mean, d, kl = _kl(shape, dtype, pairs)
    n = 2 * pairs
    assert kl._tree and len(kl.lins) == n

    def add(a, b):
        return LatentVec(a.xi + b.xi, a.small + b.small)

    # metric: sample i alone (the prior term d on sample 0), then the tree
    alone = []
    for i, lp in enumerate(kl.lins):
        out = LatentVec(torch.empty_like(d.xi), None)
        model.lh_metric_accumulate(lp, d, out, 1.0 / n, True, identity=1.0 if i == 0 else 0.0)
        alone.append(out)
    want = parallel.tree_fold(alone, add)
    got = kl.apply_metric(d)
    assert torch.equal(got.xi, want.xi) and torch.equal(got.small, want.small)
    # gradient and value
    grads, values = [], []
    for r, neg in zip(kl.residuals, kl.negs):
        v = torch.zeros(1, dtype=torch.float64, device=model.device)
        lp = model.linearize(mean.shifted(-1.0 if neg else 1.0, r), n_total=n, value_acc=v)
        grads.append(lp.grad)
        values.append(v)
    want = parallel.tree_fold(grads, add)
    assert torch.equal(kl.gradient.xi, want.xi) and torch.equal(kl.gradient.small, want.small)
    assert kl.value == float(parallel.tree_fold(values).item())
    # and the running sum of rounds 1-3 agrees to rounding
    monkeypatch.setenv("NK_TREE_SUM", "0")
    old = kl.at(mean).apply_metric(d)
    tol = 1e-5 if dtype == torch.float32 else 1e-13
    assert float((old.xi - got.xi).abs().max()) < tol * float(got.xi.abs().max())


@pytest.mark.parametrize("shape,dtype,likelihood", [((256, 128), torch.float64, "poisson"), ((256, 256, 256), torch.float32, "gaussian")])
def test_lanes_keep_the_pairwise_order(shape, dtype, likelihood, monkeypatch):
    """Small grids run the samples' chains on several streams (FusedModel.lanes) into one vector per sample: the same
    bits as the single-stream pairwise sum -- also with fp32 fields, whose fp64 forward transform needs a workspace per lane
    (the lanes once shared the cached fp64 plan's: wrong gradients on grids large enough to overlap)."""
    monkeypatch.setenv("NK_LANES", "0")
    model, mean, d, kl = _kl(shape, dtype, 4, likelihood=likelihood)
    one = (kl.value, kl.gradient.xi.clone(), kl.apply_metric(d).xi.clone())
    monkeypatch.setenv("NK_LANES", "4")
    kl4 = kl.at(mean)
    assert len(kl4._lanes) == 4
    assert kl4.value == one[0] and torch.equal(kl4.gradient.xi, one[1]) and torch.equal(kl4.apply_metric(d).xi, one[2])


@pytest.mark.parametrize("shape,dtype,likelihood", [((256, 128), torch.float64, "poisson"), ((64, 64, 64), torch.float32, "gaussian"),
                                                    ((2048, 1024), torch.float64, "poisson")])
def test_linear_samples_solved_together_equal_one_after_the_other(shape, dtype, likelihood, monkeypatch):
    """engine.draw_samples on small grids: the CGs of the linear samples advance together on up to four streams
    (ConjugateGradient.solve_many, one lane of scratch per solve) -- the same residuals, bit for bit, as one solve after
    the other, including a wave of more solves than lanes."""
    from nifty_amd import random
    from nifty_amd.engine import FusedModel, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController

    model = FusedModel(shape, offset_mean=2.0, likelihood=likelihood, icov=100.0, nonlin="exp" if likelihood == "poisson" else None,
                       dtype=dtype, device="cuda:0")
    random.push_sseq_from_seed(23)
    try:
        truth = model.draw_prior()
        if likelihood == "poisson":
            model.set_data(torch.poisson(model.signal(truth).double()).to(torch.int64))
        else:
            model.set_data(model.signal(truth), 100.0)
        mean = 0.1 * model.draw_prior()
    finally:
        random.pop_sseq()

    shared = AbsDeltaEnergyController(0.05, iteration_limit=22)  # ONE controller object for every solve, as optimize_kl passes it

    def samples(lanes):
        monkeypatch.setenv("NK_LANES", str(lanes))
        random.push_sseq_from_seed(5)
        try:
            res, negs, n = draw_samples(model, mean, 5, True, lambda: shared)
        finally:
            random.pop_sseq()
        torch.cuda.synchronize()
        return [(r.xi.clone(), r.small.clone()) for r in res], negs, n

    one, negs1, n1 = samples(0)
    together, negs4, n4 = samples(4)
    assert n1 == n4 == 10 and negs1 == negs4
    for (a_xi, a_small), (b_xi, b_small) in zip(one, together):
        assert torch.equal(a_xi, b_xi) and torch.equal(a_small, b_small)
