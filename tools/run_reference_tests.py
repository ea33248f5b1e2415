"""Run the REFERENCE's own test files against nifty_amd, unmodified (build container only: needs /root/reference).

`import nifty.cl as ift` inside them resolves to nifty_amd through nifty_amd.compat.install(); the files are read where they lie,
nothing is copied into this repository and nothing is written under /root/reference (no bytecode, no pytest cache).  Host fields
(device_id = -1): what is checked is the API surface and the host arithmetic of the nifty.cl-shaped layer -- names, argument
meaning, error behaviour, adjointness / Jacobian consistency checks of the reference's extra.py -- next to the GPU parity tests
of tests/.  Test modules that need the spherical transforms (ducc0 SHT), JAX, plotting or the other VI methods are out of
scope and fail or are skipped there.

usage: python tools/run_reference_tests.py [pattern ...]     -> one line per test file, totals at the end"""
import glob
import os
import re
import subprocess
import sys
import tempfile

REF = "/root/reference/test/test_cl"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALIAS = '''import sys
sys.path.insert(0, %r)
import nifty_amd.compat
nifty_amd.compat.install()
''' % ROOT


def main():
    pats = sys.argv[1:]
    files = sorted(glob.glob(os.path.join(REF, "test_*.py")) + glob.glob(os.path.join(REF, "test_operators", "test_*.py"))
                   + glob.glob(os.path.join(REF, "test_spaces", "test_*.py")))
    files = [f for f in files if not pats or any(p in f for p in pats)]
    total = dict(passed=0, failed=0, error=0, skipped=0)
    with tempfile.TemporaryDirectory() as tmp:
        open(os.path.join(tmp, "nifty_alias_plugin.py"), "w").write(ALIAS)
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=tmp)
        for f in files:
            out = subprocess.run([sys.executable, "-m", "pytest", "-p", "nifty_alias_plugin", "-p", "no:cacheprovider", "-q", "--timeout=600", f], cwd=tmp, env=env, capture_output=True, text=True)
            lines = (out.stdout.strip() or out.stderr.strip() or "no output").splitlines()
            tail = next((l for l in reversed(lines) if re.search(r"\d+ (passed|failed|error|skipped)", l)), lines[-1]).strip("= ")
            counts = {k: int(n) for n, k in re.findall(r"(\d+) (passed|failed|error|skipped)", tail.replace("errors", "error"))}
            for k, v in counts.items():
                total[k] += v
            print(f"{os.path.relpath(f, REF):58s} {tail}")
    print("TOTAL", total)


if __name__ == "__main__":
    main()
