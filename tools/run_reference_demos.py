"""Run the REFERENCE's demo scripts (demos/cl/*.py) unchanged, once on the reference itself and once on nifty_amd through
nifty_amd.compat.install(), and compare everything they would have plotted (build container only: needs /root/reference; host
fields).  A third run -- nifty_amd with torch.fft instead of scipy.fft on the host, i.e. the same code with transforms that differ
in the last bit -- shows how far the long, iteration-limited minimisations of the demos amplify rounding by themselves: the
cross-implementation differences are read against that.  `ift.Plot` / `ift.single_plot` are replaced in BOTH runs by a recorder that keeps the fields handed to it (this image
has no matplotlib; plotting is out of scope for the package) -- so the comparison covers the demos' mock data, reconstructions,
power spectra and sample statistics, seed for seed.

usage: python tools/run_reference_demos.py [demo[:arg] ...]      default: the getting-started demos of the path"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMOS = "/root/reference/demos/cl"
# (getting_started_4 / polynomial_fit plot through matplotlib directly, bernoulli_map / getting_started_1:2 live on the sphere)
DEFAULT = ["getting_started_1.py:0", "getting_started_1.py:1", "getting_started_2.py", "getting_started_3.py",
           "getting_started_5_mf.py"]

RUNNER = r'''
import os, runpy, sys
import numpy as np
which, demo, arg, out = sys.argv[1:5]
if which == "reference":
    sys.path.insert(0, "/root/reference")
    sys.path.insert(0, os.path.join(%(root)r, "tests", "golden"))
    import _ref_shim
    ift = _ref_shim.load()
    sys.modules.setdefault("nifty.cl", ift)
else:
    sys.path.insert(0, %(root)r)
    import nifty_amd.compat
    ift = nifty_amd.compat.install()
    if which == "nifty_amd_torchfft":  # the same package with another host FFT library: results differ in the last bit
        import torch
        import nifty_amd.operators as _ops
        _ops._host_fftn = lambda v, axes, inverse=False: (torch.fft.ifftn if inverse else torch.fft.fftn)(v, dim=axes)
recorded = []

def keep(obj, tag):
    objs = obj if isinstance(obj, (list, tuple)) else [obj]
    for o in objs:
        if o is None:
            continue
        if hasattr(o, "keys") and hasattr(o, "asnumpy"):
            for k, v in o.asnumpy().items():
                recorded.append((tag + "." + k, np.asarray(v)))
        elif hasattr(o, "asnumpy"):
            recorded.append((tag, np.asarray(o.asnumpy())))

class Plot:
    def add(self, f, **kw):
        keep(f, "plot%%d" %% len(recorded))
    def output(self, **kw):
        pass

ift.Plot = Plot
ift.single_plot = lambda f, **kw: keep(f, "single%%d" %% len(recorded))
ift.plot_priorsamples = lambda *a, **kw: None
sys.argv = [demo] + ([arg] if arg else [])
os.chdir(os.path.dirname(out))
runpy.run_path(demo, run_name="__main__")
np.savez(out, **{"%%03d_%%s" %% (i, t): a for i, (t, a) in enumerate(recorded)})
''' % dict(root=ROOT)


def main():
    wanted = sys.argv[1:] or DEFAULT
    with tempfile.TemporaryDirectory() as tmp:
        runner = os.path.join(tmp, "runner.py")
        open(runner, "w").write(RUNNER)
        for item in wanted:
            demo, _, arg = item.partition(":")
            outs = {}
            for which in ("reference", "nifty_amd", "nifty_amd_torchfft"):
                work = os.path.join(tmp, which)
                os.makedirs(work, exist_ok=True)
                out = os.path.join(work, "rec.npz")
                if os.path.exists(out):
                    os.remove(out)
                env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", MPLBACKEND="Agg")
                r = subprocess.run([sys.executable, runner, which, os.path.join(DEMOS, demo), arg, out], capture_output=True,
                                   text=True, env=env, timeout=3600)
                outs[which] = (r, out)
            (ra, fa), (rb, fb), (rc, fc) = outs["reference"], outs["nifty_amd"], outs["nifty_amd_torchfft"]
            if ra.returncode or rb.returncode or rc.returncode:
                tail = lambda r: (r.stderr.strip().splitlines() or ["?"])[-1][:160]  # noqa: E731
                print(f"{item:40s} reference rc={ra.returncode} nifty_amd rc={rb.returncode}  "
                      f"{tail(ra) if ra.returncode else ''} | {tail(rb) if rb.returncode else ''}")
                continue
            a, b, c = np.load(fa), np.load(fb), np.load(fc)
            if sorted(a.files) != sorted(b.files) or sorted(b.files) != sorted(c.files):
                print(f"{item:40s} different records: {len(a.files)} vs {len(b.files)}")
                continue

            def errors(x, y):
                return [float(np.abs(x[k] - y[k]).max() / max(np.abs(x[k]).max(), 1e-300)) for k in sorted(x.files)]

            cross, own = errors(a, b), errors(b, c)
            fmt = lambda errs: " ".join(f"{e:.0e}" for e in errs) if len(errs) <= 12 else \
                " ".join(f"{e:.0e}" for e in errs[:6]) + f" ... median {np.median(errs):.0e}"  # noqa: E731
            print(f"{item:40s} {len(cross):3d} recorded fields, max |difference| / max |value| in script order\n"
                  f"    nifty_amd vs reference:            worst {max(cross, default=0):.1e}   {fmt(cross)}\n"
                  f"    nifty_amd vs itself (other FFT):   worst {max(own, default=0):.1e}   {fmt(own)}")


if __name__ == "__main__":
    main()
