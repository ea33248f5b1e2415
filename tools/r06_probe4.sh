set -u
cd "${GRAFT_REPO_ROOT:?}"
tag=r06d; mkdir -p gpurun_out/$tag
timeout 900 python3 -m pytest tests/test_los_response.py tests/test_config4_gpu.py -m gpu -x -q > gpurun_out/$tag/pytest_los.txt 2>&1
tail -5 gpurun_out/$tag/pytest_los.txt
timeout 900 python3 tools/gpu_los_probe.py > gpurun_out/$tag/los_probe.txt 2>&1
NK_ROWSUM_STAGED=0 timeout 900 python3 tools/gpu_los_probe.py 4096 10000 2>&1 | grep ADJOINT >> gpurun_out/$tag/los_probe.txt
cat gpurun_out/$tag/los_probe.txt
