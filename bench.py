#!/usr/bin/env python
"""bench.py -- MGVI iterations/s of the fused MI355X path on BASELINE.json's headline workload.

A "step" is ONE global MGVI iteration (the body of the reference's optimize_kl loop,
nifty/cl/minimization/optimize_kl.py:357-451, without I/O): draw 8 mirrored MGVI samples (4 CG solves
of (J^T N^-1 J + 1) y = b) and minimise the sampled KL with NewtonCG, with the fixed recipe of
SURVEY 8(d): ic_sampling = AbsDeltaEnergyController(0.05, iteration_limit=20), kl_minimizer =
NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3), max_cg_iterations=20).
Workload (C5): 1024^3 RGSpace CorrelatedField + Gaussian likelihood, fp32 fields with fp64
accumulators, 8 samples in total; with --gpus N the samples are sharded over the ranks (strong
scaling, one all-reduce of the latent vector per KL value/gradient and per CG iteration).  Inputs are
synthetic, generated on the device before the timed region.

Prints ONE JSON line (rank 0).  NK_BENCH_SHAPE=256,256,256 / NK_BENCH_DTYPE=f64 override the workload
for quick checks (the line then names that workload and is not the headline number).
NK_BENCH_CONFIG=C2 | C3 | C4 selects another BASELINE.json config as a side measurement (C4: 4096^2
sigmoid(cf) -> masked LOSResponse(10^4 lines) -> Gaussian, geoVI, on the fused response engine).
NK_BENCH_SAMPLES=16 runs BASELINE configs[4]'s sample count; from 4 ranks on a short 16-sample leg is
appended to the line as "samples16" automatically (NK_BENCH_ALSO16=0: off, =force: from 2 ranks on).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nifty_amd import _lib as L  # noqa: E402
from nifty_amd import backend as B  # noqa: E402
from nifty_amd import minimization, parallel, random  # noqa: E402
from nifty_amd.engine import FusedKL, FusedModel, LatentVec, draw_samples, mgvi_iteration  # noqa: E402
from nifty_amd.minimization import AbsDeltaEnergyController, NewtonCG  # noqa: E402

# streams of the fused CG vector update: x, r (read + write), d, q -- plus b when the energy is re-evaluated from x.b
CG_STREAMS = 7.0 if os.environ.get("NK_CG_ENERGY_RECURRENCE", "1") == "0" else 6.0
PMC_TRAFFIC_FILE = os.environ.get("NK_PMC_FILE", "r06_pmc_traffic.json")  # latest committed PMC summary of the bench command


def kernel_source_digest():
    """sha256 over the kernel sources (nifty_amd/csrc/*.hip, *.h and include/niftyk.h, sorted by name): what a PMC summary
    under profiles/ must have been taken at to be quoted as THIS run's traffic (there is no .git on the GPU box; VERDICT r5
    item 2: the round-5 line quoted counters of a kernel generation older than the one it timed)."""
    import glob
    import hashlib

    h = hashlib.sha256()
    for name in sorted(glob.glob(os.path.join(ROOT, "nifty_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "nifty_amd", "csrc", "*.h"))
                       + [os.path.join(ROOT, "include", "niftyk.h")]):
        h.update(os.path.basename(name).encode())
        h.update(open(name, "rb").read())
    return h.hexdigest()[:16]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# transform pass kernels by profile id: strided-first pipeline (A first strided pass with prologue, B in-place strided
# pass, C final contiguous pass with epilogue) and the five-pass sandwich H D H of a metric application (S1 contiguous
# first pass with prologue, S2 in-place middle-axis passes, SM fused first-axis pass, then C)
KERNEL_NAMES = {0: "k_pass1d", 1: "k_passA", 2: "k_passB", 3: "k_passC", 4: "k_passD", 5: "k_passS1", 6: "k_passS2",
                7: "k_passSM", 8: "k_csr_rowsum", 9: "k_passC2"}  # C2: the final passes of two samples in one launch
NK_PROF_KEYS = 250
PRO_NAMES = {0: "plain", 1: "amp", 2: "amp_jvp", 3: "mul", 4: "amp_jvp+cg_direction"}
EPI_NAMES = {0: "affine", 1: "mul", 2: "vjp", 3: "likelihood", 4: "nonlin"}


class ParityFailure(AssertionError):
    """The HIP path disagrees with the oracle on the in-run sample: the bench line is printed, the exit code is 3."""


def kernel_symbol(kern, pro, epi, shape, dt_name, octant=True, wide=False, const_mid=True):
    """Device symbol behind a profile key (kernel id, prologue, epilogue) of the register-resident pipelines
    (nifty_amd/csrc/nk_fft.hip; what rocprofv3 --kernel-trace lists), for 2-D / 3-D power-of-two grids."""
    T = "float" if dt_name == "f32" else "double"
    A, M, NL = (shape[0], shape[1], shape[2]) if len(shape) == 3 else (1, shape[0], shape[-1])
    pc_first = {0: "0", 1: "4" if octant else "1", 2: "5" if octant else "3", 3: "6", 4: "8"}[pro]  # prologue class of a first pass
    ec = {0: "0", 1: "1", 2: "2", 3: "3", 4: "-1"}[epi]                                    # epilogue class of a final pass
    couples = "true" if epi == 2 else "false"
    two_level = len(shape) == 2 and M == 4096 and T == "double" and os.environ.get("NK_TWO_LEVEL", "1") != "0"  # nk_tl_split
    if kern == 1:
        if wide and pro == 1 and epi == 3:  # value / gradient forward of an fp32 model: fp64 kernels, float arrays at the ends
            return f"k2_strided<double,{M},3,9>"
        if two_level:  # the two launches of the two-level first-axis pass, timed together
            return f"k2_tl<{T},64,4,{pc_first}> + k2_tl<{T},64,5,-1>"
        return f"k2_strided<{T},{M},3,{pc_first}>"
    if kern == 2:
        return f"k2_strided<{'double' if wide and pro == 1 and epi == 3 else T},{A if len(shape) == 3 else M},0,-1>"
    if kern == 3:
        if wide and pro == 1 and epi == 3:
            return f"k2_final<double,{NL},false,5,0>"
        pair = "1" if (pro == 2 and epi == 2 and len(shape) == 3) else "0"   # a sandwich's final pass: row-mirror pairing
        if epi == 2 and len(shape) == 2 and NL * (8 if T == "double" else 4) >= 16384:
            couples = "false"  # 2-D VJP on single line pairs where the couple tile limits residency (nk_final_single_2d)
        return f"k2_final<{T},{NL},{couples},{ec},{pair}>"
    if kern == 5:
        return f"k3_contig_quad<{T},{NL // 2},{pc_first}>"  # (8 = 5 with the CG direction update: profile key pro 4)
    if kern == 6:
        return f"k2_strided<{T},{M},0,-1>"
    if kern == 7:
        return f"k3_mid<{T},{A},{'false' if const_mid else 'true'}>"
    if kern == 9:
        return f"k2_final2<{T},{NL}>"
    return KERNEL_NAMES.get(kern, str(kern))


def rowsum_bytes(nnz, nrows, ncols, b, idx_bytes=4, ptr_bytes=8, weighted=True):
    """Algorithmic bytes of one sparse product y = R x of the response: per entry its index and float32 weight as stored
    (int32 column for the CSR launches, a uint16 position inside the tile for nk_tiled_rowsum: `idx_bytes`), the operand x ONCE
    (ncols values -- round 6: until round 5 one gathered value per ENTRY was credited, which is what the wavefront-per-row
    kernel moved, not what the product needs), per row the row pointer and the stored sum."""
    return nnz * (idx_bytes + (4 if weighted else 0)) + ncols * b + nrows * (ptr_bytes + b)


def algorithmic_bytes(kernel, pro, epi, N, b, const_mid, octant=True, ndim=3):
    """Algorithmic HBM bytes of ONE launch of a transform pass kernel: the operand arrays the launch reads
    and writes, each once, at their real size.

    Every pass reads and writes the work array once (2*N*b, SURVEY 8(d) per-axis-pass model); fused operands are counted
    where they are consumed -- prologue operands in the first pass, epilogue operands in the final pass:
      * amplitude fields a[pidx], da[pidx] of the register-resident pipelines are OCTANT arrays (N*b / 2^ndim each, the 2^ndim
        sign-flip images of a coefficient share their bin) and there is NO 4-byte index stream (`octant`); the generic
        kernels gather through pidx (4*N) instead;
      * prologue key 4 = AMP_JVP with the CG direction update riding along (nk_fuse.cg_r): reads r, writes the new d;
      * the VJP epilogue reads xi and the amplitude field and writes the octant sums w8 (fp64, N*8 / 2^ndim).
    The optional addend / running sum / carried partial sums of a VJP epilogue vary from launch to launch inside a step and
    are NOT counted (conservative: the kernel moves up to 3*N*b more)."""
    if kernel == 4:  # pass D touches 2 planes only
        return 0.0
    if kernel == 9:  # nk_hartley_sandwich_pair: two final passes in one launch
        return 2.0 * algorithmic_bytes(3, pro, epi, N, b, const_mid, octant, ndim)
    field = N * b / 2 ** ndim if octant else 4.0 * N  # one amplitude field operand: octant array, or the index stream
    total = 2.0 * N * b
    if kernel == 7 and not const_mid:  # the diagonal between the two transforms of a sandwich
        total += N * b
    if kernel in (0, 1, 5):
        total += {0: 0.0, 1: field, 2: N * b + 2 * field if octant else N * b + field, 3: N * b,
                  4: 3 * N * b + 2 * field}[pro]  # (4: xi, r in, d out on top of the work array's 2 N b)
    if kernel in (0, 3):
        vjp = N * b + field + (N * 8.0 / 2 ** ndim if octant else 0.0)
        total += {0: 0.0, 1: 0.0 if const_mid else N * b, 2: vjp, 3: N * b + (0.0 if const_mid else N * b), 4: 0.0}[epi]
    return total


def collect_profile():
    lib = L.load()
    ms = (ctypes.c_double * NK_PROF_KEYS)()
    cnt = (ctypes.c_int64 * NK_PROF_KEYS)()
    lib.nk_profile_collect(ms, cnt)
    out = {}
    for key in range(NK_PROF_KEYS):
        if cnt[key]:
            out[(key // 25, (key % 25) // 5, key % 5)] = (ms[key], cnt[key])
    return out


def _cg_vector_ops_seconds(n, reps=3):
    """One CG iteration's vector work as the reference does it on the host (conjugate_gradient.py:85-126 with
    QuadraticEnergy.at_with_grad, quadratic_energy.py:44-78): d.q, r - alpha q, x - alpha d, r.r, beta d + r, A x = r + b,
    x.(Ax), b.x -- immutable Fields, i.e. every result a fresh array; np.vdot for the dots (ducc_dispatch.py:103-108)."""
    rng = np.random.default_rng(1)
    x, r, d, q, bb = (rng.standard_normal(n) for _ in range(5))
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        alpha = np.vdot(r, r) / np.vdot(d, q)
        r2 = r - q * alpha
        x2 = x - d * alpha
        gamma = np.vdot(r2, r2)
        d2 = d * max(0.0, gamma) + r2
        ax = r2 + bb
        val = 0.5 * np.vdot(x2, ax) - np.vdot(bb, x2)
        best = min(best, time.perf_counter() - t0)
        del r2, x2, d2, ax, val
    return best


def cpu_baseline(counts_per_step, shape_full, cfg="C5", bench_dtype=None, budget_s=20.0, device=None, c4=None):
    """Oracle (numpy + scipy.fft restatement of the reference path) timed on the host cores on a bounded
    sample of the workload, scaled to the full workload.  A reported baseline, not a target.

    Sample: the full grid when it is small (2-D configs), else a cube of edge NK_BENCH_CPU_EDGE (default 256 on a
    many-core host, 128 otherwise; 512 is affordable in RAM but its single-threaded set-up -- k-length table, draws --
    takes about a minute).  Seconds per step = metric applications x t_metric + value/gradient evaluations x t_vg
    + CG iterations x t_vector-ops, the first two scaled with N log N, the last with N.  C4 (geoVI with a response): every
    transform pair the step executed is priced as one oracle metric application (two transforms + both sparse products)."""
    from oracle import nifty_oracle as orc

    cores = os.cpu_count() or 1
    Nf = float(np.prod(shape_full))
    if Nf <= (1 << 25):
        sample_shape = tuple(shape_full)
    else:
        edge = int(os.environ.get("NK_BENCH_CPU_EDGE", "512" if cores >= 64 else "256" if cores >= 32 else "128"))
        sample_shape = tuple(min(edge, n) for n in shape_full)
    rng = np.random.default_rng(0)
    natural = len(sample_shape) >= 2  # the default RGSpace: bins from integer k^2, slab-parallel (same arrays)
    cf = orc.CFModel(sample_shape, None, orc.CFParams(offset_mean=2.0), workers=cores,
                     geometry=orc.power_geometry_natural(sample_shape, workers=cores) if natural else None)
    x = {k: 0.1 * v for k, v in cf.draw_latent(rng).items()}
    v = cf.draw_latent(rng)
    # fp32-representable excitations and data: the fp64 oracle, the fp64 engine and the fp32 engine see IDENTICAL inputs
    # (Field.from_random(dtype=float32) rounds its fp64 draw the same way, random.py:219-237)
    x["xi"] = x["xi"].astype(np.float32).astype(np.float64)
    v["xi"] = v["xi"].astype(np.float32).astype(np.float64)
    response = None
    if cfg == "C4":
        # the sparse matrix is INPUT DATA of the oracle here: built by the host set-up the product shares with the
        # reference's own host set-up (a numpy walk of the lines); the oracle's pure-Python walk of 2.7e7 segments would
        # take minutes.  Matrix parity against the oracle's own walk: tests/test_config4_gpu.py.
        from scipy.sparse import csr_matrix

        response = csr_matrix((c4["wgt"], c4["col"], c4["rowptr"]), shape=(len(c4["rowptr"]) - 1, int(Nf)))
        sig = orc.NONLIN["sigmoid"][0](cf.forward(x))
        data = response @ sig.ravel() + np.sqrt(1e-3) * rng.normal(size=response.shape[0])
        lh = orc.Likelihood("gaussian", data, icov=1e3, nonlin="sigmoid", response=response)
        lh_kw = dict(likelihood="gaussian", icov=1e3, nonlin="sigmoid", response=c4["response"])
    elif cfg == "C2":
        data = rng.poisson(np.exp(cf.forward(x))).astype(np.int64)
        lh = orc.Likelihood("poisson", data, nonlin="exp")
        lh_kw = dict(likelihood="poisson", nonlin="exp")
    else:
        data = (cf.forward(x) + 0.1 * rng.normal(size=sample_shape)).astype(np.float32).astype(np.float64)
        lh = orc.Likelihood("gaussian", data, icov=100.0)
        lh_kw = dict(likelihood="gaussian", icov=100.0)
    lin = orc.Linearized(cf, lh, x)
    mv = lin.metric(v)
    # parity of the HIP path against the oracle on this very sample (<= 1e-5 relative in fp64, asserted in the
    # run), in fp64 AND in the dtype the bench computes in (fp32 fields against the fp64 oracle: reported, loosely asserted)
    parity = {}
    if device is not None:
        from nifty_amd.engine import FusedModel, LatentVec

        val_o, grad_o = lin.value_grad()

        def worst(got, ref):  # max |difference| over all keys, relative to the largest entry of the whole latent vector
            top = max(float(np.max(np.abs(ref[k]))) for k in ref)
            return max(float(np.max(np.abs(np.asarray(got[k], dtype=np.float64) - ref[k]))) for k in ref) / top

        dts = [torch.float64] + ([bench_dtype] if bench_dtype not in (None, torch.float64) else [])
        for dt in dts:
            model = FusedModel(sample_shape, offset_mean=2.0, data=data, dtype=dt, device=device, **lh_kw)
            lp = model.linearize(LatentVec.from_dict(model, x))
            got = model.metric(lp, LatentVec.from_dict(model, v)).to_dict()
            parity["f64" if dt == torch.float64 else "f32"] = dict(
                value=abs(float(lp.value.item()) - val_o) / abs(val_o), gradient=worst(lp.grad.to_dict(), grad_o),
                metric=worst(got, mv))
            del model, lp
        # SURVEY 8(d) / north_star: <= 1e-5 relative, value, GRADIENT and metric application, in fp64 and in the dtype the
        # bench computes in (fp32 fields against the fp64 oracle on identical inputs)
        bad = {k: e for k, e in parity.items() if max(e.values()) >= 1e-5}
        if bad:
            raise ParityFailure(f"HIP path deviates from the oracle by more than 1e-5: {parity}")

    def time_pair(lin_, cf_, budget):
        t_met, t_vg, n = 0.0, 0.0, 0
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < budget or n < 2:
            t0 = time.perf_counter()
            lin_.metric(v)
            t1 = time.perf_counter()
            orc.Linearized(cf_, lh, x).value_grad()
            t2 = time.perf_counter()
            t_met += t1 - t0
            t_vg += t2 - t1
            n += 1
        return t_met / n, t_vg / n, n

    t_met, t_vg, n = time_pair(lin, cf, budget_s)
    Ns = float(np.prod(sample_shape))
    scale = (Nf * math.log2(Nf)) / (Ns * math.log2(Ns))
    t_cg = _cg_vector_ops_seconds(int(Ns))
    n_cg = counts_per_step.get("cg_iterations", 0.0)

    def per_step(tm, tv):
        if cfg == "C4":
            return scale * 0.5 * counts_per_step["transforms"] * tm + (Nf / Ns) * n_cg * t_cg
        return (scale * (counts_per_step["metric"] * tm + counts_per_step["value_grad"] * tv) + (Nf / Ns) * n_cg * t_cg)

    sec_per_step = per_step(t_met, t_vg)
    # the reference's default is ONE FFT thread (ducc_dispatch.py:46): same sample with workers=1 -- or, for samples beyond
    # 2^25 points (one thread needs ~30 s per evaluation of 512^3), a 256-edge corner of the same problem, scaled N log N
    if Ns > int(os.environ.get("NK_BENCH_CPU_1T_MAX", str(1 << 25))) and cfg not in ("C2", "C4"):
        shape1 = tuple(min(int(os.environ.get("NK_BENCH_CPU_1T_EDGE", "256")), n) for n in sample_shape)
        cf1 = orc.CFModel(shape1, None, orc.CFParams(offset_mean=2.0), workers=1,
                          geometry=orc.power_geometry_natural(shape1, workers=cores))
        sl = tuple(slice(0, n) for n in shape1)
        x1 = dict(x, xi=np.ascontiguousarray(x["xi"][sl]), spectrum=cf1.draw_latent(np.random.default_rng(1))["spectrum"] * 0.1)
        v1 = dict(v, xi=np.ascontiguousarray(v["xi"][sl]), spectrum=cf1.draw_latent(np.random.default_rng(2))["spectrum"])
        lh1 = orc.Likelihood("gaussian", np.ascontiguousarray(data[sl]), icov=100.0)
        N1 = float(np.prod(shape1))
        scale1 = (Nf * math.log2(Nf)) / (N1 * math.log2(N1))
    else:
        cf1, x1, v1, lh1, scale1 = orc.CFModel(sample_shape, None, orc.CFParams(offset_mean=2.0), workers=1), x, v, lh, scale
    lin1 = orc.Linearized(cf1, lh1, x1)
    lin1.metric(v1)
    t_start, n1, t1_met, t1_vg = time.perf_counter(), 0, 0.0, 0.0
    while n1 < 1 or time.perf_counter() - t_start < budget_s / 3:
        t0 = time.perf_counter()
        lin1.metric(v1)
        t1 = time.perf_counter()
        orc.Linearized(cf1, lh1, x1).value_grad()
        t1_met, t1_vg, n1 = t1_met + t1 - t0, t1_vg + time.perf_counter() - t1, n1 + 1
    sec1 = (scale1 / scale) * (per_step(t1_met / n1, t1_vg / n1) - (Nf / Ns) * n_cg * t_cg) + (Nf / Ns) * n_cg * t_cg
    priced = (f"{0.5 * counts_per_step['transforms']:.0f} transform pairs (geoVI: metric applications, energy evaluations)"
              if cfg == "C4" else
              f"{counts_per_step['metric']:.0f} metric applies + {counts_per_step['value_grad']:.0f} value/gradient "
              "evaluations")
    return dict(value=1.0 / sec_per_step, unit="MGVI iters/s", cores=cores, kind="port",
                parity_rel_err_vs_hip=max(parity["f64"].values()) if "f64" in parity else None,
                parity_rel_err_vs_hip_f32=max(parity["f32"].values()) if "f32" in parity else None,
                parity_detail=parity or None,
                value_one_thread=1.0 / sec1,
                # where the in-run parity was taken (the timed grid is larger than this sample; the timed grid itself
                # is held against the oracle by tests/test_large_oracle_gpu.py::test_config5_full_size_against_the_oracle)
                parity_sample=(f"HIP engine vs oracle (value, gradient, metric application; fp64 and the bench dtype) at "
                               f"{'x'.join(map(str, sample_shape))}"
                               + ("" if tuple(sample_shape) == tuple(shape_full) else
                                  f" -- NOT the {'x'.join(map(str, shape_full))} grid the line times")) if parity else None,
                sample=(f"oracle (numpy+scipy.fft, workers={cores}) metric apply {t_met * 1e3:.1f} ms and value+gradient "
                        f"{t_vg * 1e3:.1f} ms per sample at {'x'.join(map(str, sample_shape))} fp64 ({n} reps), "
                        f"scaled x{scale:.1f} (N log N) to {'x'.join(map(str, shape_full))} and multiplied by the {priced} "
                        f"one GPU step executed; plus {n_cg:.0f} CG iterations x {t_cg * 1e3:.1f} ms of host vector "
                        f"operations (x{Nf / Ns:.0f}, linear in N)"))


def api_leg(shape, dtype, lh_kind, noise_var, data, n_pairs, steps, device, engine_ms_per_transform):
    """NK_BENCH_API=1: the same workload through the USER-LEVEL driver, `ift.optimize_kl` with its fusion pass (reference
    minimization/optimize_kl.py:51-453: MultiField positions, ResidualSampleList, minisanity every iteration, counting
    report) instead of engine.mgvi_iteration -- after the timed region, on the same data.  Returns the time per iteration
    and per transform of the fused engine underneath, the overhead of the driver layer per transform against the engine
    leg of this run, and the peak device memory of the leg."""
    import nifty_amd as ift
    import importlib

    okl = importlib.import_module("nifty_amd.optimize_kl")  # (the package attribute of that name is the function)
    from nifty_amd.field import Field

    npdt = np.float32 if dtype == torch.float32 else np.float64
    sp = ift.RGSpace(shape)
    cfm = ift.CorrelatedFieldMaker("")
    cfm.add_fluctuations(sp, (1.0, 5e-1), (1.0, 2e-1), (5e-1, 5e-2), (-3.0, 2e-1))
    cfm.set_amplitude_total_offset(2.0, (1e-1, 3e-2))
    cf = cfm.finalize()
    d = Field(cf.target, data)
    if lh_kind == "poisson":
        lh = ift.PoissonianEnergy(d) @ cf.exp()
    else:
        lh = ift.GaussianEnergy(d, ift.ScalingOperator(cf.target, 1.0 / noise_var, npdt)) @ cf
    x0 = 0.1 * ift.from_random(cf.domain, dtype=npdt, device_id=device.index)
    ic = ift.AbsDeltaEnergyController(deltaE=0.05, iteration_limit=20)
    mk = lambda i: ift.NewtonCG(ift.AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3),  # noqa: E731
                                max_cg_iterations=20)
    run = lambda total, start: ift.optimize_kl(lh, total, n_pairs, mk, ic, output_directory=None, initial_position=start,  # noqa: E731
                                               return_final_position=True, device_id=device.index)
    torch.cuda.synchronize(device)
    torch.cuda.reset_peak_memory_stats(device)
    _, mean = run(1, x0)  # warm-up: builds the fused model of the likelihood (cached by the driver)
    del x0
    model = okl._fused_model(lh, device.index, npdt)
    if model is None:
        return {"error": "the fusion pass did not take this likelihood"}
    before = dict(model.counters)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    sl, mean = run(steps, mean)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    transforms = model.counters["transforms"] - before["transforms"]
    per_transform = 1e3 * dt / max(transforms, 1)
    out = {"what": "ift.optimize_kl (fusion pass, minisanity and reports included), same data, after the timed region",
           "steps": steps, "ms_per_step": 1e3 * dt / steps, "value": steps / dt, "unit": "MGVI iters/s",
           "transforms_per_step": transforms / steps, "ms_per_transform": per_transform,
           "engine_ms_per_transform": engine_ms_per_transform,
           "api_overhead_pct": (None if not engine_ms_per_transform
                                else round(100.0 * (per_transform / engine_ms_per_transform - 1.0), 2)),
           "peak_allocated_GB": round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
           "peak_allocated_GiB": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
           "samples": sl.n_samples}
    okl._fused_cache.clear()
    return out


RNG_LABEL = {"numpy": "the reference's numpy PCG64 + ziggurat streams (seed 42, one SeedSequence per sample), computed on the "
                      "device from the host generators' states (nk_pcg64_normal)",
             "device": "torch device generator (NK_BENCH_RNG=device)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without a launcher: run the ranks as a CHILD torchrun job (one process per GPU) and leave
        # with its exit code -- this process has not touched the GPU yet
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    comm, local_rank = parallel.init(os.environ.get("NK_DIST_BACKEND", "nccl"))
    if os.environ.get("NK_SHARE_DEVICE"):  # developer check: several ranks on one GPU (with NK_DIST_BACKEND=gloo)
        local_rank = 0
    rank = 0 if comm is None else comm.rank
    world = 1 if comm is None else comm.size
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)

    # NK_BENCH_CONFIG selects another BASELINE.json config for side measurements (never the default):
    #   C2 = 2048^2 fp64 Poisson(exp) ;  C3 = 512^3 fp64 Gaussian ;  C5 (default) = 1024^3 fp32 Gaussian
    #   C4 = 4096^2 fp64 sigmoid(cf) -> MaskOperator(LOSResponse(10^4 random lines)) -> Gaussian noise 1e-3, geoVI
    #        (reference demos/cl/getting_started_3.py:48-51, 98-100, 119-127)
    cfg = os.environ.get("NK_BENCH_CONFIG", "C5")
    preset = {"C2": ("2048,2048", "f64", "poisson"), "C3": ("512,512,512", "f64", "gaussian"),
              "C4": ("4096,4096", "f64", "gaussian"), "C5": ("1024,1024,1024", "f32", "gaussian")}[cfg]
    shape = tuple(int(s) for s in os.environ.get("NK_BENCH_SHAPE", preset[0]).split(","))
    dt_name = os.environ.get("NK_BENCH_DTYPE", preset[1])
    lh_kind = preset[2]
    dtype = torch.float32 if dt_name == "f32" else torch.float64
    b = 4 if dtype == torch.float32 else 8
    N = int(np.prod(shape))
    n_pairs = int(os.environ.get("NK_BENCH_SAMPLES", "8")) // 2  # mirrored pairs -> 8 samples in total by default
    noise_var = 1e-3 if cfg == "C4" else 0.01

    L.load()
    response, c4 = None, None
    if cfg == "C4":
        from nifty_amd.los_response import SparseResponse, los_matrix

        n_los = int(os.environ.get("NK_BENCH_NLOS", "10000"))
        lrng = np.random.default_rng(1)
        starts, ends = lrng.uniform(size=(2, n_los)), lrng.uniform(size=(2, n_los))
        flagged = np.zeros(n_los, dtype=bool)
        flagged[lrng.integers(0, n_los, n_los // 20)] = True
        rowptr, col, wgt = los_matrix(shape, tuple(1.0 / n for n in shape), starts, ends)
        from scipy.sparse import csr_matrix

        kept = csr_matrix((wgt, col, rowptr), shape=(n_los, N))[np.logical_not(flagged)].tocsr()
        response = SparseResponse(kept.indptr, kept.indices, kept.data, N)
        c4 = dict(rowptr=kept.indptr, col=kept.indices, wgt=kept.data, response=response, nnz=int(kept.nnz),
                  n_data=int(kept.shape[0]), n_los=n_los)
    model = FusedModel(shape, offset_mean=2.0, offset_std=(1e-1, 3e-2), fluctuations=(1.0, 5e-1),
                       loglogavgslope=(-3.0, 2e-1), flexibility=(1.0, 2e-1), asperity=(5e-1, 5e-2),
                       likelihood=lh_kind, icov=1.0 / noise_var,
                       nonlin="exp" if lh_kind == "poisson" else "sigmoid" if cfg == "C4" else None,
                       response=response, dtype=dtype, device=device)
    # synthetic inputs with the reference's seed discipline: push_sseq_from_seed(42); truth = from_random;
    # d = cf(truth) + N(0, noise_var) (or d ~ Poisson(exp(cf(truth)))); start = 0.1 * from_random.  Every normal field is
    # numpy's PCG64 + ziggurat stream computed ON THE DEVICE from the host generator's state (nk_pcg64_normal: the very
    # numbers `rng.normal` returns, 12 ms instead of 9 s per 1024^3 field), identical on all ranks.
    # NK_BENCH_RNG=device: torch's device generator instead (different numbers, for A/B timing).
    rng_mode = os.environ.get("NK_BENCH_RNG", "numpy")
    random.push_sseq_from_seed(42)
    gen = torch.Generator(device=device).manual_seed(42) if rng_mode == "device" else None
    truth = model.draw_prior(gen)
    data = model.signal(truth)
    if response is not None:
        data = response.times(data)
        data.add_(random.Random.normal_on_device(model.npdtype, (c4["n_data"],), 0.0, math.sqrt(noise_var), device))
        model.set_data(data, 1.0 / noise_var)
    elif lh_kind == "poisson":
        pgen = torch.Generator(device=device).manual_seed(42)
        data = torch.poisson(data.double(), generator=pgen).to(torch.int64)
        model.set_data(data)
    else:
        if gen is None:
            noise = random.Random.normal_on_device(model.npdtype, shape, 0.0, math.sqrt(noise_var), device)
        else:
            noise = torch.randn(shape, dtype=dtype, device=device, generator=gen) * math.sqrt(noise_var)
        data.add_(noise)
        del noise
        model.set_data(data, 1.0 / noise_var)
    del truth
    mean = 0.1 * model.draw_prior(gen)
    rng_draws = torch.Generator(device=device).manual_seed(1234 + rank) if rng_mode == "device" else None

    phases = {}
    phase_timing = os.environ.get("NK_BENCH_PHASES", "1" if (comm is not None or N >= (1 << 27)) else "0") == "1"

    def step(mean, pairs=n_pairs):
        ic = lambda: AbsDeltaEnergyController(0.05, iteration_limit=20)  # noqa: E731
        mini = NewtonCG(AbsDeltaEnergyController(0.5, convergence_level=2, iteration_limit=3), max_cg_iterations=20)
        # C4: geoVI, the non-linear sample fit of demos/cl/getting_started_3.py:125-127
        geo = NewtonCG(AbsDeltaEnergyController(0.5, iteration_limit=5, convergence_level=2)) if cfg == "C4" else None
        # the timed code is the engine's own entry point; with NK_BENCH_PHASES (default on for world > 1 and for grids >= 2^27
        # points, where three extra synchronisations per step are noise) a hook at its phase boundaries synchronises the
        # device and reads the clock: sampling has no exchange, the KL construction one all-reduce, the Newton-CG one exchange
        # per CG iteration
        marks = [time.perf_counter()]

        def on_phase(name):
            torch.cuda.synchronize(device)
            marks.append(time.perf_counter())
            phases[name] = phases.get(name, 0.0) + (marks[-1] - marks[-2])

        new_mean, kl = mgvi_iteration(model, mean, pairs, ic, mini, mirror_samples=True, comm=comm, device_rng=rng_draws,
                                      geo_minimizer=geo, on_phase=on_phase if phase_timing else None)
        return new_mean, kl.value

    def sync():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        mean, _ = step(mean)
    phases.clear()
    if comm is not None and os.environ.get("NK_BENCH_EXCHANGE_TIMER", "1") != "0":  # (world > 1, or NK_FORCE_COMM=1)
        parallel.exchange_timer.enable(True)  # device events around every exchange of the sharded CG (a few per iteration)
    for k in model.counters:  # in place: the lanes of small grids (FusedModel.lanes) count into the same dictionary
        model.counters[k] = 0
    minimization.counters["cg_iterations"] = 0
    lib = L.load()
    # live per-kernel HIP events (the roofline object); NK_BENCH_PROFILE=0 switches them off for side measurements
    if os.environ.get("NK_BENCH_PROFILE", "1") != "0":
        lib.nk_profile_enable(1)
    collect_profile()
    sync()
    ms = torch.cuda.memory_stats(device)
    mem_before = (int(ms.get("num_device_alloc", 0)), int(ms.get("num_device_free", 0)))
    t0 = time.perf_counter()
    energy = float("nan")
    trace = os.environ.get("NK_BENCH_TRACE", "0") == "1"  # developer: the work of every step (forces a sync per step)
    for it in range(args.steps):
        before = (dict(model.counters), minimization.counters["cg_iterations"])
        mean, energy = step(mean)
        if trace and rank == 0:
            print(f"step {it}: KL {energy!r} cg {minimization.counters['cg_iterations'] - before[1]} " +
                  " ".join(f"{k} {model.counters[k] - before[0][k]}" for k in model.counters), file=sys.stderr, flush=True)
    sync()
    elapsed = time.perf_counter() - t0
    lib.nk_profile_enable(0)
    exchange = parallel.exchange_timer.summary() if parallel.exchange_timer.on else None
    parallel.exchange_timer.enable(False)
    phase_seconds = {k: v / args.steps for k, v in phases.items()} if phases else None
    phases = {}  # (the side legs below do not add to the timed region's phases)
    # allocator behaviour of the timed region (peak footprint; hipMalloc calls after the warm-up: the peak still grows now
    # and then -- an A/B against one up-front segment showed no time difference, gpurun_out/r3q)
    ms = torch.cuda.memory_stats(device)
    device_memory = {"peak_allocated_GB": round(ms.get("allocated_bytes.all.peak", 0) / 1e9, 2),
                     "peak_reserved_GB": round(ms.get("reserved_bytes.all.peak", 0) / 1e9, 2),
                     "hipMalloc_calls_timed": int(ms.get("num_device_alloc", 0)) - mem_before[0],
                     "hipFree_calls_timed": int(ms.get("num_device_free", 0)) - mem_before[1],
                     "alloc_retries": int(ms.get("num_alloc_retries", 0))}
    prof = collect_profile()
    if comm is not None:
        elapsed = comm.max_float(elapsed, device)
    # BASELINE configs[4]: the same model with 16 samples (two per GPU on the 8-GPU node).  With >= 2 ranks a short leg of it
    # rides along (1 warm-up + up to 3 timed steps, same barriers), so that the 8-sample strong-scaling series and the
    # 16-sample case come out of one command; on one GPU 16 samples of 1024^3 do not fit next to the CG vectors.
    leg16 = None
    cg_iterations = minimization.counters["cg_iterations"]
    # Memory: 8 samples per rank peak at 232 GB of the 288 (device_memory_rank0 at N = 1) -- the leg runs where a rank holds at
    # most 4 of the 16 samples (>= 4 ranks; the decision depends on the rank count only, so every rank takes the same one),
    # after the first leg's cached blocks went back to the driver.  NK_BENCH_ALSO16=force: from 2 ranks on.
    also16 = os.environ.get("NK_BENCH_ALSO16", "1")
    if (world >= (2 if also16 == "force" else 4) and 2 * n_pairs == 8 and cfg == "C5" and also16 != "0"):
        saved = dict(model.counters)
        k16 = max(1, min(3, args.steps))
        torch.cuda.empty_cache()
        try:
            m16, _ = step(mean, 8)
            sync()
            t16 = time.perf_counter()
            for _ in range(k16):
                m16, e16 = step(m16, 8)
            sync()
            t16 = comm.max_float(time.perf_counter() - t16, device)
            leg16 = {"samples_total": 16, "steps": k16, "warmup": 1, "ms_per_step": 1e3 * t16 / k16, "value": k16 / t16,
                     "unit": "MGVI iters/s", "final_kl_energy": e16}
            del m16
        except Exception as exc:  # the side leg must never take the 8-sample line down (symmetric failures, e.g. memory)
            leg16 = {"error": repr(exc)}
        model.counters = saved
    ms_per_step = 1e3 * elapsed / args.steps
    model.counters["cg_iterations"] = cg_iterations
    counts = {k: v / args.steps for k, v in model.counters.items()}

    parity_failed = False
    if rank == 0:
        # dominant transform pass kernel of this rank, live HIP-event timing over the timed region
        by_kernel = {}
        by_symbol = {}
        spmv = {}
        for (kern, pro, epi), (ms, cnt) in prof.items():
            if kern == 8:
                # nk_csr_rowsum: the matrix (>= 16 lanes per row: thousands of pixels per line) and its transpose (<= 4 lanes
                # per row: a few lines per pixel) of the response; the unweighted launches are bin sums of static index maps
                if c4 is None or epi != 0:
                    continue
                tiled = pro >= 2 and response.tiled(device) is not None
                name = ("nk_tiled_rowsum<R>" if tiled else "k_csr_rowsum<R>") if pro >= 2 else "k_csr_rowsum<R^T>"
                ent = spmv.setdefault(name, dict(ms=0.0, cnt=0, bytes=0.0))
                ent["ms"] += ms
                ent["cnt"] += cnt
                ent["bytes"] += cnt * (rowsum_bytes(c4["nnz"], c4["n_data"], N, b, idx_bytes=2 if tiled else 4) if pro >= 2
                                       else rowsum_bytes(c4["nnz"], N, c4["n_data"], b))
                continue
            ent = by_kernel.setdefault(kern, dict(ms=0.0, cnt=0, bytes=0.0))
            ent["ms"] += ms
            ent["cnt"] += cnt
            # (the wide forward of an fp32 model, FusedModel.wide: float arrays at the ends, fp64 work array in between)
            wide_fwd = model.wide and pro == 1 and epi == 3
            nbytes = cnt * algorithmic_bytes(kern, pro, epi, N, b, model.const_mid, octant=model.octant_vjp, ndim=len(shape))
            if wide_fwd:
                # the fp64 work array (and fp64 octant amplitude field) of that transform: written by pass 1, read + written
                # by pass 2, read by the final pass -- at twice the bytes of the fp32 arrays the model above prices
                nbytes += cnt * {1: N * b + N * b / 2 ** len(shape), 2: 2 * N * b, 3: N * b}.get(kern, 0)
            ent["bytes"] += nbytes
            sym = kernel_symbol(kern, pro, epi, shape, dt_name, octant=model.octant_vjp, wide=model.wide,
                                const_mid=model.const_mid)
            se = by_symbol.setdefault(sym, dict(ms=0.0, cnt=0, bytes=0.0, families=set()))
            se["ms"] += ms
            se["cnt"] += cnt
            se["bytes"] += nbytes
            se["families"].add(KERNEL_NAMES[kern])
        roofline = None
        if by_kernel:
            kern = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
            ent = by_kernel[kern]
            avg_ms = ent["ms"] / ent["cnt"]
            achieved = ent["bytes"] / ent["cnt"] / (avg_ms * 1e-3) / 1e9
            # PMC counters cannot be read from inside the run: they come from separate rocprofv3 --pmc passes of this
            # very command (tools/pmc_bench.sh), whose summary is committed under profiles/ -- named in traffic_source
            traffic, traffic_source, traffic_stale = None, None, None
            tfile = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
            if os.path.exists(tfile) and world == 1:
                tj = json.load(open(tfile))
                if tj.get("workload") == f"{'x'.join(map(str, shape))}:{dt_name}":
                    # only counters taken on THESE kernel sources are this run's traffic
                    traffic_stale = tj.get("kernel_source_digest") != kernel_source_digest()
                    if not traffic_stale:
                        traffic = tj["kernels"].get(KERNEL_NAMES[kern], {}).get("bytes")
                    traffic_source = (f"profiles/{PMC_TRAFFIC_FILE} (separate rocprofv3 --pmc passes of this command on kernel "
                                      f"sources {tj.get('kernel_source_digest', '?')}, commit {tj.get('commit', '?')}; this run: "
                                      f"{kernel_source_digest()}" + (" -- STALE, not quoted)" if traffic_stale else ")"))
            # the same launches by DEVICE SYMBOL (what rocprofv3 --kernel-trace lists): a symbol can serve several families
            # (the in-place strided pass is one symbol for the sandwich's middle-axis passes and the second pass of a
            # value / gradient transform)
            top_sym = max(by_symbol, key=lambda k: by_symbol[k]["ms"])
            tse = by_symbol[top_sym]
            roofline = dict(bound="hbm", kernel=KERNEL_NAMES[kern], family=KERNEL_NAMES[kern],
                            kernel_symbols=sorted(sy for sy, v in by_symbol.items() if KERNEL_NAMES[kern] in v["families"]),
                            largest_symbol=dict(symbol=top_sym, families=sorted(tse["families"]), ms_total=round(tse["ms"], 2),
                                                launches=int(tse["cnt"]), avg_launch_ms=round(tse["ms"] / tse["cnt"], 4),
                                                achieved=round(tse["bytes"] / max(tse["ms"], 1e-9) / 1e6, 1),
                                                frac=round(tse["bytes"] / max(tse["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4)),
                            achieved=round(achieved, 1), peak=HBM_PEAK_GBS,
                            unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=traffic_source,
                            traffic_stale=traffic_stale,
                            avg_launch_ms=round(avg_ms, 4), launches=int(ent["cnt"]),
                            algorithmic_bytes_per_launch=ent["bytes"] / ent["cnt"],
                            all_pass_kernels={KERNEL_NAMES[k]: dict(ms_total=round(v["ms"], 2), launches=int(v["cnt"]),
                                                                    GBps=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1))
                                              for k, v in sorted(by_kernel.items())},
                            all_pass_symbols={k: dict(ms_total=round(v["ms"], 2), launches=int(v["cnt"]),
                                                      GBps=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1))
                                              for k, v in sorted(by_symbol.items(), key=lambda kv: -kv[1]["ms"])})
            if spmv:
                # sparse products of the response (C4): algorithmic bytes = per entry index + weight, the operand once, per
                # row pointer + result (rowsum_bytes); against the same 8 TB/s peak
                roofline["spmv"] = {k: dict(bound="hbm", ms_total=round(v["ms"], 2), launches=int(v["cnt"]),
                                            avg_launch_ms=round(v["ms"] / v["cnt"], 4),
                                            algorithmic_bytes_per_launch=v["bytes"] / v["cnt"],
                                            achieved=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1), peak=HBM_PEAK_GBS,
                                            unit="GB/s", frac=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6 / HBM_PEAK_GBS, 4))
                                   for k, v in sorted(spmv.items())}
        if roofline is not None:
            # "confirm on the box with a device-copy ceiling and report both" -- a 1 read + 1 write stream copy
            # of one field-sized array through the library's own chunked element-wise kernel (after the timed region)
            try:
                src = torch.empty(max(N, 1 << 28), dtype=dtype, device=device).normal_()
                dst = torch.empty_like(src)
                B.axpby(1.0, src, out=dst)
                torch.cuda.synchronize(device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    B.axpby(1.0, src, out=dst)
                e1.record()
                torch.cuda.synchronize(device)
                ceil = 10 * 2 * src.numel() * b / (e0.elapsed_time(e1) * 1e-3) / 1e9
                roofline["copy_ceiling"] = {"GBps": round(ceil, 1), "frac_of_ceiling": round(achieved / ceil, 4),
                                            "what": "measured here: 1 read + 1 write stream device copy of one field"}
                del src, dst
            except Exception as exc:
                roofline["copy_ceiling"] = {"error": repr(exc)}
        # whole-step algorithmic bytes: B_met, B_vg per sample evaluation on this rank
        D = len(shape)
        shape_label = f"{shape[0]}^{D}" if len(set(shape)) == 1 else "x".join(map(str, shape))
        B_met = (4 * D + 4) * N * b + 8 * N
        B_vg = (4 * D + 3) * N * b + 8 * N
        step_bytes = counts["metric"] * B_met + counts["value_grad"] * B_vg
        if cfg == "C4":
            # geoVI with a response: energy evaluations and metric applications of the sample fits are transform pairs that
            # are neither `metric` nor `value_grad` of the KL -- price every transform pair like a metric application
            # (2 x 2DNb + 4Nb + 8N) plus its two sparse products
            step_bytes = 0.5 * counts["transforms"] * (B_met + rowsum_bytes(c4["nnz"], c4["n_data"], N, b)
                                                       + rowsum_bytes(c4["nnz"], N, c4["n_data"], b))
        line = {
            "metric": f"MGVI iters/sec on {shape_label} RGSpace CorrelatedField, {2 * n_pairs} samples; achieved HBM GB/s",
            "value": args.steps / elapsed,
            "unit": "MGVI iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if dtype == torch.float32 else "f64",
            "data": "synthetic",
            "config": {"workload": f"{cfg}: {'x'.join(map(str, shape))} RGSpace CorrelatedField + "
                                   + (f"sigmoid -> MaskOperator(LOSResponse({c4['n_los']} lines, {c4['nnz']} entries)) -> "
                                      if c4 else "")
                                   + f"{lh_kind} likelihood, {2 * n_pairs} mirrored {'geoVI' if c4 else 'MGVI'} samples, "
                                   f"{dt_name} fields / fp64 accumulators, sampling CG limit 20, NewtonCG 3 steps x <=20 CG "
                                   "iterations" + (", sample fit NewtonCG <=5 steps" if c4 else ""),
                       "samples_total": 2 * n_pairs, "parallelism": f"sample-sharded x{world}",
                       "rng": RNG_LABEL[rng_mode]},
            "final_kl_energy": energy,
            "samples16": leg16,
            "per_step_counts_rank0": counts,
            # the WORK of a step is a property of the trajectory (532 ... 657 transforms per step between variants of the
            # library that differ in last bits only, profiles/r04_trajectory_variants.txt); the time per transform is the
            # figure that compares implementations
            "ms_per_transform_rank0": (ms_per_step / counts["transforms"]) if counts.get("transforms") else None,
            "step_algorithmic_GBps_rank0": step_bytes / (ms_per_step * 1e-3) / 1e9,
            # step_algorithmic is SURVEY 8(d)'s model: SIX passes per metric application (two three-pass transforms).  What
            # the kernels of a step really move: the operand bytes of every executed transform pass (a metric application is
            # a FIVE-pass sandwich) plus the streams of the CG vector update (CG_STREAMS) per iteration, over the same step time
            "step_hbm_GBps_rank0": ((sum(v["bytes"] for v in by_kernel.values()) / args.steps
                                     + counts.get("cg_iterations", 0.0) * CG_STREAMS * N * b / world)
                                    / (ms_per_step * 1e-3) / 1e9) if by_kernel else None,
            "roofline": roofline,
            "device_memory_rank0": device_memory,
            # seconds per step of the three phases (rank 0; None when NK_BENCH_PHASES=0): sampling runs without any exchange,
            # the KL construction ends in one all-reduce, every CG iteration of the Newton-CG in one exchange
            "phase_seconds_per_step_rank0": phase_seconds,
            # world > 1: per sharded metric application (= per CG iteration of the KL minimisation) on rank 0, device-event
            # times: all-gather of the direction, reduce-scatter of the output, the local metric application, what the
            # compute stream still waits for at the end (exposed), and exchange - exposed (hidden) -- DESIGN 5's table
            "exchange_per_cg_iteration_rank0": exchange,
            "comm_ms_per_cg_iteration": None if exchange is None else exchange["exchange_ms"],
            "overlap_ms": None if exchange is None else exchange["hidden_ms"],
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(counts, shape, cfg=cfg, bench_dtype=dtype, device=device, c4=c4)
            except ParityFailure as exc:  # the line is still printed -- and the run exits with code 3
                line["cpu_baseline"] = {"error": repr(exc)}
                parity_failed = True
            except Exception as exc:  # the baseline must never take the bench line down
                line["cpu_baseline"] = {"error": repr(exc)}
        if world == 1 and os.environ.get("NK_BENCH_API", "0") == "1" and cfg != "C4":
            # the engine leg's model, vectors and cached blocks go first: the driver builds its own fused model
            data_keep, ms_tr = model.data, line["ms_per_transform_rank0"]
            del step, model, mean
            import gc

            gc.collect()
            torch.cuda.empty_cache()
            try:
                line["api"] = api_leg(shape, dtype, lh_kind, noise_var, data_keep, n_pairs, max(1, min(args.steps, 3)), device,
                                      ms_tr)
                line["api_overhead_pct"] = line["api"].get("api_overhead_pct")
            except Exception as exc:  # the side leg must never take the bench line down
                line["api"] = {"error": repr(exc)}
    else:
        line = None
    # RCCL writes a version banner to the C-level stdout of a rank when its first communicator is created; through a pipe
    # that text sits in the C buffer until the process exits -- AFTER the JSON line.  Every rank empties its C buffers
    # first, then rank 0 prints the one line, so the line is the last thing on stdout.
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if comm is not None:
        comm.barrier()
    if line is not None:
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.barrier()
        torch.distributed.destroy_process_group()  # or ProcessGroupNCCL complains on stderr after the line
    if parity_failed:
        sys.exit(3)


if __name__ == "__main__":
    main()
