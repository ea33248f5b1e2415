cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
python -m pytest tests/test_api_large_gpu.py tests/test_driver_io.py tests/test_geometry.py -m gpu -x -q -s > gpurun_out/r5b/new_tests.log 2>&1
tail -5 gpurun_out/r5b/new_tests.log
grep "peak device" gpurun_out/r5b/new_tests.log
python -m pytest tests -m gpu -q --deselect tests/test_api_large_gpu.py > gpurun_out/r5b/gputests.log 2>&1
tail -5 gpurun_out/r5b/gputests.log
NK_BENCH_API=1 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r5b/bench_api.log 2>&1
grep -o '"api": {[^}]*}' gpurun_out/r5b/bench_api.log
grep -o '"value": [0-9.]*\|"ms_per_transform_rank0": [0-9.]*' gpurun_out/r5b/bench_api.log | head -3
