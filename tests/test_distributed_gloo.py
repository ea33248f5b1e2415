"""world_size-2 gloo tests of the sample-parallel path (reference test_mpi/test_kl.py:26-114): the KL
energy built from samples sharded over two ranks equals the single-process one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem():
    import nifty_amd as ift
    from tests import goldenlib as gl
    from tests.test_api_host import build

    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"))
    v = ift.MultiField.from_raw(cf.domain, gl.latent(z, "v"))
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    return ift, m, ham, x, v


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from nifty_amd import parallel

    comm, _ = parallel.init("gloo")
    ift, m, ham, x, v = _problem()
    ift.random.push_sseq_from_seed(m["seed"] + 1)
    kl = ift.SampledKLEnergy(x, ham, m["n_samples"], None, mirror_samples=True, comm=comm)
    ift.random.pop_sseq()
    assert kl.samples.n_local_samples() == 2 and kl.samples.n_samples == 4
    res = dict(value=kl.value, grad=kl.gradient.asnumpy(), met=kl.apply_metric(v).asnumpy())
    mini = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=8)
    kl2, _ = mini(kl)
    res["min_value"] = kl2.value
    res["min_pos"] = kl2.position.asnumpy()
    if rank == 0:
        torch.save(res, out)
    comm.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_kl_equals_serial(tmp_path):
    from tests import goldenlib as gl

    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    par = torch.load(out, weights_only=False)
    z = gl.load("model_g1d")
    assert abs(par["value"] - float(z["kl_value"])) < 1e-9 * abs(float(z["kl_value"]))
    assert gl.lat_relerr(par["grad"], gl.latent(z, "kl_grad")) < 1e-8
    assert gl.lat_relerr(par["met"], gl.latent(z, "kl_metric_v")) < 1e-8
    assert abs(par["min_value"] - float(z["kl_min_value"])) < 1e-7 * abs(float(z["kl_min_value"]))
    assert gl.lat_relerr(par["min_pos"], gl.latent(z, "kl_min_pos")) < 1e-6


def _plain_sum_worker(rank, world, port, out):
    os.environ["NK_TREE_SUM"] = "0"
    _worker(rank, world, port, out)


@pytest.mark.timeout(300)
def test_two_rank_kl_with_the_plain_allreduce_fallback(tmp_path):
    """NK_TREE_SUM=0 -- local running sum + ONE all-reduce, the documented fallback of rounds 1-3 --
    through the generic SampledKLEnergy (value, gradient, apply_metric, a NewtonCG run) on two ranks: equal to the serial
    run to rounding."""
    from tests import goldenlib as gl

    out = str(tmp_path / "rank0.pt")
    mp.spawn(_plain_sum_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    par = torch.load(out, weights_only=False)
    z = gl.load("model_g1d")
    assert abs(par["value"] - float(z["kl_value"])) < 1e-9 * abs(float(z["kl_value"]))
    assert gl.lat_relerr(par["grad"], gl.latent(z, "kl_grad")) < 1e-8
    assert gl.lat_relerr(par["met"], gl.latent(z, "kl_metric_v")) < 1e-8
    assert abs(par["min_value"] - float(z["kl_min_value"])) < 1e-7 * abs(float(z["kl_min_value"]))


def _uneven_kl(comm):
    """Generic-graph KL with 3 unmirrored samples (2 + 1 over two ranks), minimised inside the lockstep scope exactly
    like optimize_kl does."""
    from nifty_amd import parallel

    ift, m, ham, x, v = _problem()
    ift.random.push_sseq_from_seed(m["seed"] + 5)
    kl = ift.SampledKLEnergy(x, ham, 3, None, mirror_samples=False, comm=comm)
    ift.random.pop_sseq()
    mini = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3), max_cg_iterations=6)
    with parallel.lockstep(comm):
        kl2, _ = mini(kl)
    return kl, kl2


def _uneven_worker(rank, world, port, out):
    import faulthandler

    faulthandler.dump_traceback_later(200, exit=True)  # mispaired collectives would hang: fail instead
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from nifty_amd import parallel

    comm, _ = parallel.init("gloo")
    kl, kl2 = _uneven_kl(comm)
    assert kl.samples.n_local_samples() == (2 if (rank == 0 and world == 2) else 1) and kl.samples.n_samples == 3
    torch.save(dict(value=kl.value, min_value=kl2.value, min_pos=kl2.position.asnumpy()), f"{out}.{rank}")
    comm.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world", [2, 3])
def test_lockstep_with_unequal_sample_counts_equals_serial_bit_for_bit(tmp_path, world):
    """the dot products inside the per-sample energies must not communicate -- ranks with different
    local sample counts would mispair the broadcasts, and rank 0's per-sample values would overwrite the others'.
    Since round 4 the sums over samples follow the reference's pairwise tree over the GLOBAL sample index
    (utilities.py:349-414; reference test test_mpi/test_kl.py:104-114), whatever the split: 2 + 1 samples on two ranks and
    1 + 1 + 1 on three give the bits of the serial run, through a whole minimisation."""
    out = str(tmp_path / "rank")
    mp.spawn(_uneven_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    runs = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    kl, kl2 = _uneven_kl(None)  # serial truth
    truth = kl2.position.asnumpy()
    for r in runs:
        assert r["value"] == kl.value and r["min_value"] == kl2.value
        for k in truth:
            assert np.array_equal(r["min_pos"][k], truth[k])


def test_share_range_matches_reference_semantics():
    from nifty_amd.parallel import shareRange

    # reference utilities.py:282-306: contiguous blocks, remainder to the first shares
    assert [shareRange(8, 3, r) for r in range(3)] == [(0, 3), (3, 6), (6, 8)]
    assert [shareRange(16, 8, r) for r in range(8)] == [(2 * r, 2 * r + 2) for r in range(8)]
    assert [shareRange(2, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 2), (2, 2)]
    covered = sorted(i for r in range(5) for i in range(*shareRange(13, 5, r)))
    assert covered == list(range(13))


def _engine_worker(rank, world, port, out):
    import faulthandler

    faulthandler.dump_traceback_later(150, exit=True)  # a rank that waits for a lost partner must not hang the suite
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from nifty_amd import parallel, random
    from nifty_amd.engine import FusedKL, LatentVec, draw_samples
    from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG
    from tests import goldenlib as gl
    from tests.test_engine_gpu import _model

    comm, _ = parallel.init("gloo")  # both ranks share cuda:0; gloo stages device tensors through the host
    z = gl.load("model_g1d")
    m, model = _model(z)
    xl = LatentVec.from_dict(model, gl.latent(z, "x"))
    vl = LatentVec.from_dict(model, gl.latent(z, "v"))
    random.push_sseq_from_seed(m["seed"] + 1)
    res, negs, n_total = draw_samples(model, xl, m["n_samples"], True,
                                      lambda: AbsDeltaEnergyController(0.05, iteration_limit=m["sampling_limit"]), comm)
    random.pop_sseq()
    assert len(res) == 2 and n_total == 4
    kl = FusedKL(model, xl, res, negs, n_total, comm)
    assert getattr(kl.metric, "sharded", None) is not None  # the CG below runs on per-rank shards of the latent vector
    out_d = dict(value=kl.value, grad=kl.gradient.to_dict(), met=kl.apply_metric(vl).to_dict())
    mini = NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=8)
    with parallel.lockstep(comm):  # as engine.mgvi_iteration / optimize_kl do: rank 0's scalars steer every rank
        kl2, _ = mini(kl)
    out_d["min_value"] = kl2.value
    out_d["min_pos"] = kl2.position.to_dict()
    if rank == 0:
        torch.save(out_d, out)
    comm.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_rank_fused_engine_equals_reference(tmp_path):
    """The sample-sharded fused engine (the path bench.py --gpus N runs) with 2 ranks on one GPU."""
    from tests import goldenlib as gl

    out = str(tmp_path / "rank0.pt")
    mp.spawn(_engine_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    par = torch.load(out, weights_only=False)
    z = gl.load("model_g1d")
    assert abs(par["value"] - float(z["kl_value"])) < 1e-9 * abs(float(z["kl_value"]))
    assert gl.lat_relerr(par["grad"], gl.latent(z, "kl_grad")) < 1e-8
    assert gl.lat_relerr(par["met"], gl.latent(z, "kl_metric_v")) < 1e-8
    assert abs(par["min_value"] - float(z["kl_min_value"])) < 1e-7 * abs(float(z["kl_min_value"]))
    assert gl.lat_relerr(par["min_pos"], gl.latent(z, "kl_min_pos")) < 1e-6


def _nccl_single_worker(rank, world, port):
    import faulthandler

    faulthandler.dump_traceback_later(150, exit=True)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from nifty_amd.parallel import Comm

    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    comm = Comm()
    dev = torch.device("cuda:0")
    assert comm._native_reduce_scatter(dev)  # the RCCL collectives the sharded CG relies on exist and agree
    full = torch.arange(1024, dtype=torch.float32, device=dev)
    ref = full.clone()
    shard = torch.empty(1024, dtype=torch.float32, device=dev)
    comm.reduce_scatter_sum(full, shard)
    assert torch.equal(shard, ref)
    back = torch.empty_like(ref)
    comm.all_gather(shard, back)
    assert torch.equal(back, ref)
    s8 = torch.ones(3, dtype=torch.float64, device=dev)
    comm.allreduce_sum_([s8[1:3]])
    assert torch.equal(s8.cpu(), torch.ones(3, dtype=torch.float64))
    # host scalars / host tensors on the RCCL backend (RCCL has no host path -- they are staged on the GPU)
    assert comm.sum_float(2.5) == 2.5 and comm.max_float(-1.25) == -1.25
    h = torch.arange(4, dtype=torch.float64)
    comm.allreduce_sum_([h])
    comm.bcast_(h)
    assert torch.equal(h, torch.arange(4, dtype=torch.float64))
    from nifty_amd import parallel

    with parallel.lockstep(None):
        assert parallel.lockstep_float(3.0) == 3.0
    parallel._lockstep_stack.append(comm)  # a one-rank communicator still walks the broadcast path
    try:
        assert parallel.lockstep_float(3.5) == 3.5
    finally:
        parallel._lockstep_stack.pop()
    # the generic sampled KL builds and reduces under RCCL (value/gradient averages, sample count)
    from tests import goldenlib as gl
    from tests.test_api_host import build
    import nifty_amd as ift

    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x")).at(0)
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=m["sampling_limit"])
    ham = ift.StandardHamiltonian(lh, ic, prior_sampling_dtype=np.float64)
    ift.random.push_sseq_from_seed(m["seed"] + 1)
    kl = ift.SampledKLEnergy(x, ham, m["n_samples"], None, mirror_samples=True, comm=comm, device_id=0)
    ift.random.pop_sseq()
    assert kl.samples.n_samples == 4
    assert abs(kl.value - float(z["kl_value"])) < 1e-9 * abs(float(z["kl_value"]))
    # the rank-synchronisation guards with DEVICE fields (fingerprint from the fixed-order device reduction) and the whole
    # driver with a communicator: same result as without one
    parallel.check_MPI_equality(x, comm, hash=True)
    parallel.check_MPI_synced_random_state(comm)
    mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=1), max_cg_iterations=4)  # noqa: E731
    means = []
    for c in (comm, None):
        ift.random.push_sseq_from_seed(5)
        try:
            _, mean = ift.optimize_kl(lh, 2, 1, mk, ic, output_directory=None, return_final_position=True, device_id=0, comm=c)
        finally:
            ift.random.pop_sseq()
        means.append(mean.asnumpy())
    assert gl.lat_relerr(means[0], means[1]) < 1e-9
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rccl_collectives_of_the_sharded_cg():
    """reduce_scatter_tensor / all_gather_into_tensor / slice all_reduce on the real RCCL backend (one rank: the box has
    one GPU; the multi-rank arithmetic is covered by the gloo tests above)."""
    mp.spawn(_nccl_single_worker, args=(1, _free_port()), nprocs=1, join=True)


def _sparse_worker(rank, world, port, out):
    """Fewer samples than ranks, the distributed iterator and the rank-synchronisation guards (one process group)."""
    import faulthandler

    faulthandler.dump_traceback_later(200, exit=True)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from nifty_amd import parallel

    comm, _ = parallel.init("gloo")
    ift, m, ham, x, v = _problem()
    res = {}
    # (1) ONE unmirrored sample on two ranks: rank 1 holds nothing (shareRange(1, 2, 1) is empty; reference
    # utilities.py:349-414 sums over any split, config 2's 4 samples on an 8-GPU node leave ranks empty as well)
    ift.random.push_sseq_from_seed(m["seed"] + 9)
    kl = ift.SampledKLEnergy(x, ham, 1, None, mirror_samples=False, comm=comm)
    ift.random.pop_sseq()
    assert kl.samples.n_local_samples() == (1 if rank == 0 else 0) and kl.samples.n_samples == 1
    mini = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=5)
    with parallel.lockstep(comm):
        kl2, _ = mini(kl)
    res["one"] = dict(value=kl.value, grad=kl.gradient.asnumpy(), met=kl.apply_metric(v).asnumpy(), min_value=kl2.value,
                      mean=kl.samples.average().asnumpy())
    # (2) three samples (2 + 1): every rank iterates over ALL of them in global order (sample_list.py:186-210)
    ift.random.push_sseq_from_seed(m["seed"] + 5)
    kl3 = ift.SampledKLEnergy(x, ham, 3, None, mirror_samples=False, comm=comm)
    ift.random.pop_sseq()
    everything = [s.asnumpy() for s in kl3.samples.iterator()]
    assert len(everything) == 3
    mean, var = kl3.samples.sample_stat()
    res["three"] = dict(samples=everything, mean=mean.asnumpy(), var=var.asnumpy())
    # (3) guards (utilities.py:529-585)
    parallel.check_MPI_equality({"a": 1, "b": [1, 2]}, comm)
    parallel.check_MPI_equality(x, comm, hash=True)
    parallel.check_MPI_synced_random_state(comm)
    with pytest.raises(RuntimeError, match="not in sync"):
        parallel.check_MPI_equality(rank, comm)
    ift.random.push_sseq_from_seed(100 + rank)  # a desynchronised seed stack: the reference raises, so do we
    try:
        with pytest.raises(RuntimeError, match="not in sync"):
            ift.SampledKLEnergy(x, ham, 2, None, mirror_samples=True, comm=comm)
    finally:
        ift.random.pop_sseq()
    with pytest.raises(RuntimeError, match="boom on rank 1"):
        with parallel.ensure_all_tasks_succeed(comm):
            if rank == 1:
                raise ValueError("boom on rank 1")
    with parallel.ensure_all_tasks_succeed(comm):
        pass
    torch.save(res, f"{out}.{rank}")
    comm.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(400)
def test_ranks_without_samples_distributed_iterator_and_sync_guards(tmp_path):
    from tests import goldenlib as gl

    out = str(tmp_path / "rank")
    mp.spawn(_sparse_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0", weights_only=False), torch.load(out + ".1", weights_only=False)
    ift, m, ham, x, v = _problem()
    ift.random.push_sseq_from_seed(m["seed"] + 9)
    kl = ift.SampledKLEnergy(x, ham, 1, None, mirror_samples=False)
    ift.random.pop_sseq()
    mini = ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=5)
    kl2, _ = mini(kl)
    for r in (r0["one"], r1["one"]):
        assert abs(r["value"] - kl.value) < 1e-12 * abs(kl.value)
        assert gl.lat_relerr(r["grad"], kl.gradient.asnumpy()) < 1e-12
        assert gl.lat_relerr(r["met"], kl.apply_metric(v).asnumpy()) < 1e-12
        assert abs(r["min_value"] - kl2.value) < 1e-9 * abs(kl2.value)
        assert gl.lat_relerr(r["mean"], kl.samples.average().asnumpy()) < 1e-12
    ift.random.push_sseq_from_seed(m["seed"] + 5)
    kl3 = ift.SampledKLEnergy(x, ham, 3, None, mirror_samples=False)
    ift.random.pop_sseq()
    serial = [s.asnumpy() for s in kl3.samples.iterator()]
    mean, var = kl3.samples.sample_stat()
    for r in (r0["three"], r1["three"]):
        for a, b in zip(r["samples"], serial):
            assert gl.lat_relerr(a, b) < 1e-12
        assert gl.lat_relerr(r["mean"], mean.asnumpy()) < 1e-12 and gl.lat_relerr(r["var"], var.asnumpy()) < 1e-10


def _resume_worker(rank, world, port, base):
    import faulthandler
    import pathlib

    faulthandler.dump_traceback_later(250, exit=True)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from nifty_amd import parallel
    from tests.test_api_host import _resume_case

    comm, _ = parallel.init("gloo")
    full = _resume_case(pathlib.Path(base), -1, comm=comm, rank=rank)
    torch.save(full.asnumpy(), f"{base}/mean.{rank}")
    comm.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(400)
def test_two_rank_resume_equals_uninterrupted(tmp_path):
    """reference test/test_cl/test_mpi/test_optimize_kl.py:117-146 with two ranks: every rank writes its own sample files,
    rank 0 the mean and the markers; an interrupted and resumed run ends where the uninterrupted one does, and where the
    single-process run does."""
    import pathlib

    from tests import goldenlib as gl
    from tests.test_api_host import _resume_case

    mp.spawn(_resume_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(f"{tmp_path}/mean.{r}", weights_only=False) for r in (0, 1))
    serial = _resume_case(pathlib.Path(tmp_path) / "serial", -1).asnumpy()
    for r in (r0, r1):
        assert gl.lat_relerr(r, serial) < 1e-9


def _pipe_worker(rank, world, port, out, chunks, overlap=1, pairs=2):
    import faulthandler

    faulthandler.dump_traceback_later(200, exit=True)
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", NK_PIPE_CHUNKS=str(chunks),
                      NK_PIPE_OVERLAP=str(overlap))
    from nifty_amd import parallel

    backend = "gloo" if world > 1 else "nccl"  # one rank: the real RCCL collectives on the side stream (NK_FORCE_COMM)
    if world == 1:
        os.environ["NK_FORCE_COMM"] = "1"
    comm, _ = parallel.init(backend)
    res = _pipe_case(comm, pairs)
    assert res.pop("chunks") == chunks
    if rank == 0:
        torch.save(res, out)
    comm.barrier()
    torch.distributed.destroy_process_group()


def _pipe_case(comm, pairs=2):
    """One MGVI iteration (`pairs` mirrored sample pairs) of a 3-D model whose Newton-CG runs on sharded vectors with the
    chunked exchange."""
    from nifty_amd import random
    from nifty_amd.engine import FusedModel, mgvi_iteration
    from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG

    model = FusedModel((64, 64, 128), offset_mean=2.0, likelihood="gaussian", icov=100.0, dtype=torch.float64, device="cuda:0")
    random.push_sseq_from_seed(6)
    try:
        model.set_data(model.signal(model.draw_prior()), 100.0)
        mean = 0.1 * model.draw_prior()
        ic = lambda: AbsDeltaEnergyController(0.05, iteration_limit=4)  # noqa: E731
        mini = NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=2), max_cg_iterations=5)
        mean, kl = mgvi_iteration(model, mean, pairs, ic, mini, mirror_samples=True, comm=comm)
    finally:
        random.pop_sseq()
    sm = getattr(kl.metric, "sharded", None)
    return dict(value=kl.value, xi=mean.xi.cpu(), small=mean.small.cpu(), chunks=1 if sm is None else sm.chunks)


def _same_bits(a, b):
    return a["value"] == b["value"] and torch.equal(a["xi"], b["xi"]) and torch.equal(a["small"], b["small"])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_chunked_overlapped_exchange_of_the_sharded_cg(tmp_path):
    """SURVEY 8(e) / VERDICT r2 3a: the slab-pipelined reduce-scatter / all-gather of the sharded CG with 1, 4 and 16
    chunks.  One rank on real RCCL (side-stream overlap, staged passes) and two ranks sharing the GPU (gloo, synchronous
    chunks, staged passes): staging on / off, every chunk count and both rank counts give the bits of the plain
    single-process run -- the sums over samples follow the reference's pairwise tree (utilities.py:349-414) and the dot
    products of the sharded vectors are reduced unit by unit (nk_red_layout), so neither the ownership of the elements nor
    the number of ranks enters the rounding."""
    serial = _pipe_case(None)

    def run(world, chunks, overlap=1):
        out = str(tmp_path / f"w{world}c{chunks}o{overlap}.pt")
        mp.spawn(_pipe_worker, args=(world, _free_port(), out, chunks, overlap), nprocs=world, join=True)
        return torch.load(out, weights_only=False)

    for c in (1, 4, 16):
        assert _same_bits(run(1, c), serial), c
    assert _same_bits(run(1, 4, overlap=0), serial)
    for c in (1, 4, 16):
        assert _same_bits(run(2, c), serial), c
    assert _same_bits(run(2, 4, overlap=0), serial)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_fused_engine_bits_do_not_depend_on_the_rank_count(tmp_path):
    """Reference test_mpi/test_kl.py:104-114 (MPI result == serial result, bit for bit) on the fused engine: one MGVI
    iteration with eight samples -- sampling, KL value / gradient, Newton-CG on sharded vectors, line search -- on 1, 2 and
    4 ranks (8, 4, 2 samples per rank: the local part of the pairwise sum runs inside the VJP epilogues with two, one and
    no carried partial sums) ends at the same mean and KL value, bit for bit.  NK_TREE_SUM=0 (the running sums and
    RCCL-ordered reductions of rounds 1-3) agrees to rounding only."""
    serial = _pipe_case(None, 4)

    def run(world, chunks, env=None):
        out = str(tmp_path / f"w{world}c{chunks}.pt")
        os.environ.update(env or {})
        try:
            mp.spawn(_pipe_worker, args=(world, _free_port(), out, chunks, 1, 4), nprocs=world, join=True)
        finally:
            for k in (env or {}):
                del os.environ[k]
        return torch.load(out, weights_only=False)

    assert _same_bits(run(2, 4), serial)
    assert _same_bits(run(4, 4), serial)
    assert _same_bits(run(4, 1), serial)
    # a split that is NOT one power-of-two block per rank (3 + 3 + 2 samples): every sample keeps its own vector and the
    # terms are added like the reference adds them, summands travelling between ranks -- the same bits again
    assert _same_bits(run(3, 1), serial)
    loose = run(2, 4, env={"NK_TREE_SUM": "0"})
    assert abs(loose["value"] - serial["value"]) < 1e-9 * abs(serial["value"])
    assert float((loose["xi"] - serial["xi"]).abs().max()) < 1e-7 * float(serial["xi"].abs().max())


def _lockstep_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from nifty_amd import parallel

    comm, _ = parallel.init("gloo")
    res = {}
    # verify mode (default): values pass through unchanged, ONE collective per flush, agreement -> no error
    with parallel.lockstep(comm):
        a = [parallel.lockstep_float(v) for v in (1.5, float("nan"), -2.0, float("inf"))]
        parallel.lockstep_note([3.0, 4.0])
        parallel.lockstep_flush()
        res["agree"] = a[0] == 1.5 and a[2] == -2.0
    # a scalar that differs between the ranks is caught on EVERY rank at the flush (here: at scope exit)
    try:
        with parallel.lockstep(comm):
            parallel.lockstep_float(1.0 + 1e-16 * 0 + (2.0 ** -52) * rank)
        res["caught"] = False
    except RuntimeError as exc:
        res["caught"] = "left lockstep" in str(exc)
    # ranks that took a different NUMBER of decisions since the last flush (one more line-search probe on rank 1: ADVICE r4)
    # still meet in a collective of the same shape -- the digest is fixed-size -- and both raise instead of hanging
    try:
        with parallel.lockstep(comm):
            for v in (1.0, 2.0, 3.0)[:2 + rank]:
                parallel.lockstep_float(v)
        res["count_caught"] = False
    except RuntimeError as exc:
        res["count_caught"] = "left lockstep" in str(exc)
    # ... and the same scalars in another order are a disagreement too (the digest folds them in order)
    try:
        with parallel.lockstep(comm):
            for v in ((1.0, 2.0) if rank == 0 else (2.0, 1.0)):
                parallel.lockstep_float(v)
        res["order_caught"] = False
    except RuntimeError as exc:
        res["order_caught"] = "left lockstep" in str(exc)
    # one scalar moved by ONE ulp next to large ones (a sum / sum of squares / fold of the values rounds that
    # away; the digest of the bytes does not), and tiny values whose squares underflow
    try:
        with parallel.lockstep(comm):
            import numpy as np
            for v in (3.2e6, float(np.nextafter(117.3, 1e9)) if rank else 117.3, 2.9e6, -4.1e5, 8.8e5):
                parallel.lockstep_float(v)
        res["ulp_caught"] = False
    except RuntimeError as exc:
        res["ulp_caught"] = "left lockstep" in str(exc)
    try:
        with parallel.lockstep(comm):
            parallel.lockstep_note([1e-6, 1e-6 * (1 + 2.0 ** -52 * rank), 5e8])
        res["tiny_caught"] = False
    except RuntimeError as exc:
        res["tiny_caught"] = "left lockstep" in str(exc)
    # many decisions between two flushes need no intermediate collective
    with parallel.lockstep(comm):
        for k in range(1000):
            parallel.lockstep_float(0.5 * k)
    res["many"] = True
    # broadcast mode (rounds 1-3): rank 0's value everywhere
    os.environ["NK_LOCKSTEP"] = "broadcast"
    with parallel.lockstep(comm):
        res["forced"] = parallel.lockstep_float(10.0 + rank) == 10.0
    torch.save(res, out + str(rank))
    comm.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_lockstep_verifies_with_one_collective_per_flush(tmp_path):
    """parallel.lockstep in its default verify mode: steering scalars are compared in ONE packed MAX all-reduce per flush
    instead of one broadcast per decision; a disagreement raises on both ranks, NaN / inf included in the comparison."""
    out = str(tmp_path / "ls")
    mp.spawn(_lockstep_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in (0, 1):
        res = torch.load(out + str(rank), weights_only=False)
        assert res == dict(agree=True, caught=True, count_caught=True, order_caught=True, ulp_caught=True, tiny_caught=True,
                           many=True, forced=True), (rank, res)


def _tree_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from nifty_amd import parallel

    comm, _ = parallel.init("gloo")
    res = dict(works=comm.tree_exchange_works(torch.device("cpu")))
    # uneven split: rank r holds r + 1 terms (the last rank none when world > 2), two tensors per term
    counts = [r + 1 for r in range(world)]
    if world > 2:
        counts[-1] = 0
    first = sum(counts[:rank])
    gen = lambda i: [torch.randn(5, generator=torch.Generator().manual_seed(100 + i)),  # noqa: E731
                     torch.randn(1, dtype=torch.float64, generator=torch.Generator().manual_seed(200 + i))]
    terms = [gen(first + j) for j in range(counts[rank])]
    like = None if terms else gen(0)
    res["uneven"] = [t.clone() for t in comm.tree_allreduce(terms, counts, like=like)]
    # one subtree per rank: the slice-wise exchange
    partial = torch.randn(4 * world * 3, generator=torch.Generator().manual_seed(300 + rank))
    res["slices"] = comm.tree_allreduce_slices_(partial.clone())
    res["partial"] = partial
    torch.save(res, f"{out}.{rank}")
    comm.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("world", [2, 3, 4])
def test_tree_collectives_add_in_pair_tree_order(tmp_path, world):
    """Comm.tree_allreduce (any split of the terms, ranks without terms, point-to-point like utilities.py:349-414) and the
    slice-wise tree over rank partials: the bits of parallel.tree_fold over all terms in global order, on every rank."""
    from nifty_amd import parallel

    out = str(tmp_path / "tree")
    mp.spawn(_tree_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    runs = [torch.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    counts = [r + 1 for r in range(world)]
    if world > 2:
        counts[-1] = 0
    n = sum(counts)
    want0 = parallel.tree_fold([torch.randn(5, generator=torch.Generator().manual_seed(100 + i)) for i in range(n)])
    want1 = parallel.tree_fold([torch.randn(1, dtype=torch.float64, generator=torch.Generator().manual_seed(200 + i))
                                for i in range(n)])
    want_slices = parallel.tree_fold([r["partial"] for r in runs])
    for r in runs:
        assert r["works"]
        assert torch.equal(r["uneven"][0], want0) and torch.equal(r["uneven"][1], want1)
        assert torch.equal(r["slices"], want_slices)
