# GPU tests of the drop-in surface -> gpurun_out/r06s/
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out/r06s
timeout 900 python -m pytest tests/test_dropin_surface.py -x -q -m gpu > gpurun_out/r06s/surface.log 2>&1; tail -15 gpurun_out/r06s/surface.log
