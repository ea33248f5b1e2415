"""What optimize_kl leaves on disk, against what the REFERENCE's optimize_kl left for the same model, seeds and controllers
(tests/golden/driver_io.npz, written by make_golden.py --driver-io from reference minimization/optimize_kl.py):
the file set, the random-state file (reference layout, random.py:88-110), the minisanity history pickle (:580-613), the
counting report (:716-718), the HDF5 export of operator outputs (:500-525) and a resume from a reference-written state."""
import os
import pickle
import re
import sys
import types

import numpy as np
import pytest

import nifty_amd as ift
from tests import goldenlib as gl
from tests.test_api_host import _MemoryH5Group, build


JAC_AT_ZERO = 4 * 2  # samples x Newton steps per iteration: metric applications to a zero start vector (see below)


def _controllers():
    ic_s = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=4)
    mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, iteration_limit=2), max_cg_iterations=5)  # noqa: E731
    return ic_s, mk


def _run(lh, outdir, total=3, device_id=-1, **kw):
    ic_s, mk = _controllers()
    ift.random.push_sseq_from_seed(11)
    try:
        return ift.optimize_kl(lh, total, 2, mk, ic_s, output_directory=None if outdir is None else str(outdir),
                               return_final_position=True, device_id=device_id, **kw)
    finally:
        ift.random.pop_sseq()


def _files_below(root):
    return sorted(os.path.relpath(os.path.join(r, f), root) for r, _, fs in os.walk(root) for f in fs)


def test_random_state_has_the_reference_layout():
    g = gl.load("driver_io")
    ref_bytes = g["random_state"].tobytes()
    before = ift.random.getState()
    try:
        sseqs, rngs = pickle.loads(before)  # (list of SeedSequences, list of Generators): reference random.py:96
        assert isinstance(sseqs, list) and isinstance(rngs, list) and len(sseqs) == len(rngs) >= 1
        assert all(isinstance(q, np.random.SeedSequence) for q in sseqs)
        assert all(isinstance(r, np.random.Generator) for r in rngs)
        # a file written by the reference restores the reference's stack: same depth, same draws, same children
        ift.random.setState(ref_bytes)
        assert len(pickle.loads(ift.random.getState())[0]) == int(g["state_depth"])
        np.testing.assert_array_equal(ift.random.current_rng().normal(size=4), g["state_draws"])
        np.testing.assert_array_equal(np.random.default_rng(ift.random.spawn_sseq(2)[1]).normal(size=3), g["state_child_draws"])
        # what rounds 1-4 wrote (a list of (SeedSequence, Generator) frames) is still read
        ift.random.setState(ref_bytes)
        legacy = pickle.dumps(list(zip(*pickle.loads(ref_bytes))))
        ift.random.current_rng().normal(size=7)
        ift.random.setState(legacy)
        np.testing.assert_array_equal(ift.random.current_rng().normal(size=4), g["state_draws"])
        with pytest.raises(TypeError):
            ift.random.setState(pickle.dumps({"not": "a state"}))
    finally:
        ift.random.setState(before)


def _check_run_files(outdir, g, device):
    assert _files_below(outdir) == [str(f) for f in g["files"]]
    # the state file is the reference's, byte for byte up to pickle details: same stack, same generator positions
    with open(os.path.join(outdir, "pickle", "nifty_random_state"), "rb") as f:
        ours = pickle.loads(f.read())
    theirs = pickle.loads(g["random_state"].tobytes())
    assert [q.entropy for q in ours[0]] == [q.entropy for q in theirs[0]]
    assert [q.spawn_key for q in ours[0]] == [q.spawn_key for q in theirs[0]]
    assert [q.n_children_spawned for q in ours[0]] == [q.n_children_spawned for q in theirs[0]]
    # (frame 0 is the process-wide default generator: where it stands depends on what ran before in this process)
    assert [r.bit_generator.state for r in ours[1][1:]] == [r.bit_generator.state for r in theirs[1][1:]] and len(ours[1]) >= 2
    assert ours[1][0].bit_generator.state["state"]["inc"] == theirs[1][0].bit_generator.state["state"]["inc"]
    with open(os.path.join(outdir, "pickle", "minisanity_history_latest"), "rb") as f:
        history = pickle.load(f)
    checked = 0
    for name in g.files:
        if not name.startswith("mh."):
            continue
        _, vt, cat, rest = name.split(".", 3)
        key, what = rest.rsplit(".", 1)
        got = np.array([np.nan if v is None else v for v in history[vt][cat][key][what]], dtype=np.float64)
        np.testing.assert_allclose(got, g[name], rtol=2e-5, atol=2e-6, err_msg=name)
        checked += 1
    assert checked >= 3 * 2 * 8  # index / mean / std of two value types for the residual and the seven latent keys
    with open(os.path.join(outdir, "pickle", "energy_history_latest"), "rb") as f:
        np.testing.assert_allclose(pickle.load(f).energy_values, g["energy_history"], rtol=1e-6)
    with open(os.path.join(outdir, "last_finished_iteration")) as f:
        assert f.read() == "2"
    with open(os.path.join(outdir, "minisanity.txt"), encoding="utf-8") as f:
        text = f.read()
    assert text.count("Finished index: ") == 3 and text.count("Current datetime: ") == 3 and "reduced χ²" in text
    with open(os.path.join(outdir, "counting_report.txt"), encoding="utf-8") as f:
        report = f.read()
    assert re.findall(r"Finished index: (\d+)", report) == ["0", "1", "2"] and report.count("Task 0") == 3
    counts = np.array([int(c) for c in re.findall(r"\* (?:apply|apply Linearization|Jacobian|Adjoint Jacobian): \s*(\d+)",
                                                   report)]).reshape(-1, 4)
    return counts


def test_optimize_kl_writes_what_the_reference_writes(tmp_path):
    g = gl.load("driver_io")
    m, cfm, cf, lh = build(gl.load("model_g1d"))
    sl, mean = _run(lh, tmp_path)
    assert gl.lat_relerr(mean.asnumpy(), gl.latent(g, "okl_mean")) < 1e-6
    counts = _check_run_files(tmp_path, g, -1)
    ref = g["counting"]
    # Jacobian applications: one per sample and metric application, as in the reference -- except that each of the two
    # Newton steps of an iteration starts its CG at zero and the reference still applies the metric to that zero vector
    # (quadratic_energy.py:31-39): 4 samples x 2 steps fewer here.  Linearizations / adjoint Jacobians: the reference
    # re-linearises inside every metric application (kl_energies.py:344-349); this package linearises once per sample
    # and energy (kl.py docstring).
    assert counts.shape == ref.shape == (3, 4)
    np.testing.assert_array_equal(counts[:, 2], ref[:, 2] - JAC_AT_ZERO)
    assert np.all(counts[:, 0] == 0) and np.all(counts[:, 1] < ref[:, 1]) and np.all(counts[:, 1] > 0)
    assert np.all(counts[:, 3] >= counts[:, 2])


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [True, False])
def test_optimize_kl_files_on_device(tmp_path, fuse):
    """The same files from a device run, on the fused engine (counting report from FusedModel.counters) and the generic
    graph."""
    g = gl.load("driver_io")
    m, cfm, cf, lh = build(gl.load("model_g1d"), 0)
    sl, mean = _run(lh, tmp_path, device_id=0, fuse=fuse)
    assert gl.lat_relerr(mean.asnumpy(), gl.latent(g, "okl_mean")) < 1e-6
    counts = _check_run_files(tmp_path, g, 0)
    np.testing.assert_array_equal(counts[:, 2], g["counting"][:, 2] - JAC_AT_ZERO)  # Jacobians = sample x metric applications
    if fuse:
        assert "fused engine, 4 local samples" in open(os.path.join(tmp_path, "counting_report.txt"), encoding="utf-8").read()
        np.testing.assert_array_equal(counts[:, 3], counts[:, 1] + counts[:, 2])


def test_resume_from_a_reference_written_random_state(tmp_path):
    """The run is interrupted after iteration 1; the state file in its directory is replaced by the bytes the REFERENCE
    wrote for the same run; the resumed run ends where the uninterrupted one does."""
    g = gl.load("driver_io")
    m, cfm, cf, lh = build(gl.load("model_g1d"))
    with pytest.raises(RuntimeError, match="stop"):
        _run(lh, tmp_path, terminate_callback=lambda i: (_ for _ in ()).throw(RuntimeError("stop")) if i == 1 else False)
    with open(os.path.join(tmp_path, "pickle", "nifty_random_state"), "wb") as f:
        f.write(g["random_state"].tobytes())
    ift.random.push_sseq_from_seed(999)  # a resumed process starts from whatever state; the file must decide
    try:
        ic_s, mk = _controllers()
        sl, mean = ift.optimize_kl(lh, 3, 2, mk, ic_s, output_directory=str(tmp_path), resume=True, return_final_position=True)
    finally:
        ift.random.pop_sseq()
    assert gl.lat_relerr(mean.asnumpy(), gl.latent(g, "okl_mean")) < 1e-6
    with open(os.path.join(tmp_path, "pickle", "minisanity_history_latest"), "rb") as f:
        history = pickle.load(f)
    assert history["redchisq"]["latent_variables"]["xi"]["index"] == [0, 1, 2]


def test_sample_files_follow_the_reference_convention(tmp_path):
    """No side file with the count: the number of saved samples is the run of consecutive <base>.<i>.pickle files
    (sample_list.py:333-363, 663-683); saving a shorter list over a longer one ends the run at the new length."""
    z = gl.load("model_g1d")
    m, cfm, cf, lh = build(z)
    x = ift.MultiField.from_raw(cf.domain, gl.latent(z, "x"))
    res = [ift.MultiField.from_raw(cf.domain, gl.latent(z, f"residual{i % 2}")) * (1.0 + i) for i in range(4)]
    base = str(tmp_path / "sl")
    ift.ResidualSampleList(x, res, [False, True, False, True]).save(base)
    assert sorted(os.listdir(tmp_path)) == [f"sl.{i}.pickle" for i in range(4)] + ["sl.mean.pickle"]
    with pytest.raises(RuntimeError):
        ift.ResidualSampleList(x, res[:3], [False] * 3).save(base)  # exists, no overwrite
    ift.ResidualSampleList(x, res[:3], [False, True, True]).save(base, overwrite=True)
    back = ift.ResidualSampleList.load(base)
    assert back.n_samples == 3 and not os.path.exists(base + ".3.pickle")
    for a, b in zip(back.local_iterator(), ift.ResidualSampleList(x, res[:3], [False, True, True]).local_iterator()):
        assert gl.lat_relerr(a.asnumpy(), b.asnumpy()) == 0.0
    plain = str(tmp_path / "plain")
    ift.SampleList([x, x + res[0]]).save(plain)
    assert ift.SampleList.load(plain).n_samples == 2
    os.remove(plain + ".0.pickle")
    with pytest.raises(RuntimeError):
        ift.SampleList.load(plain)


def test_export_operator_outputs(tmp_path, monkeypatch):
    """export_operator_outputs = {name: op}: <output_directory>/<name>/<latest|iteration_N>.hdf5 per iteration with
    op(sample) for every sample plus mean and standard deviation (optimize_kl.py:229-239, 310, 426, 500-525).  Without
    h5py the call fails before any work is done instead of silently writing nothing."""
    m, cfm, cf, lh = build(gl.load("model_g1d"))
    amp = cfm.amplitude
    with pytest.raises(ValueError, match="reserved"):
        _run(lh, tmp_path, export_operator_outputs={"pickle": cf})
    with pytest.raises(TypeError):
        _run(lh, tmp_path, export_operator_outputs=[cf])
    with pytest.warns(UserWarning, match="output_directory"):
        _run(lh, None, total=1, export_operator_outputs={"signal": cf})
    try:
        import h5py  # noqa: F401
    except ImportError:  # like the reference (optimize_kl.py:508-511) the run goes on without the files -- and says so
        import logging

        seen = []
        handler = logging.Handler()
        handler.emit = lambda record: seen.append(record.getMessage())
        ift.logger.addHandler(handler)
        try:
            _run(lh, tmp_path / "none", total=1, export_operator_outputs={"signal": cf})
        finally:
            ift.logger.removeHandler(handler)
        assert any("h5py" in msg for msg in seen)
        assert not os.path.exists(tmp_path / "none" / "signal" / "latest.hdf5")
    files = {}
    fake = types.ModuleType("h5py")
    fake.File = lambda fn, mode: files.setdefault(fn, _MemoryH5Group())
    monkeypatch.setitem(sys.modules, "h5py", fake)
    foreign = ift.ScalingOperator(ift.RGSpace(3), 2.0)  # not defined on the latent space: skipped (optimize_kl.py:514-515)
    sl, mean = _run(lh, tmp_path / "all", total=2, save_strategy="all",
                    export_operator_outputs={"signal": cf, "power": amp, "foreign": foreign})
    for sub in ("signal", "power", "foreign", "pickle"):
        assert os.path.isdir(tmp_path / "all" / sub)
    names = sorted(os.path.relpath(fn, tmp_path / "all") for fn in files)
    assert names == ["power/iteration_0.hdf5", "power/iteration_1.hdf5", "signal/iteration_0.hdf5", "signal/iteration_1.hdf5"]
    f = files[str(tmp_path / "all" / "signal" / "iteration_1.hdf5")]
    assert f.closed and sorted(f) == ["samples", "stats"] and sorted(f["samples"]) == ["0", "1", "2", "3"]
    assert f.attrs["nifty operator target"] == repr(cf.target)
    mu, var = sl.sample_stat(cf)
    np.testing.assert_allclose(f["stats"]["mean"], mu.asnumpy(), rtol=1e-14)
    np.testing.assert_allclose(f["stats"]["standard deviation"], np.sqrt(var.asnumpy()), rtol=1e-14)
    np.testing.assert_array_equal(f["samples"]["2"], cf.force(list(sl.iterator())[2]).asnumpy())
    p = files[str(tmp_path / "all" / "power" / "iteration_0.hdf5")]
    assert p["samples"]["0"].shape == amp.target.shape
