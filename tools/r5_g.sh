cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r5g}; mkdir -p gpurun_out/$tag
for rep in 1 2; do
python -m pytest tests/test_batched_gpu.py tests/test_engine_gpu.py tests/test_config4_gpu.py tests/test_api_large_gpu.py tests/test_driver_io.py -x -q -m gpu --deselect tests/test_api_large_gpu.py::test_headline_workload_through_optimize_kl > gpurun_out/$tag/tests$rep.log 2>&1
tail -3 gpurun_out/$tag/tests$rep.log
done
python tools/gpu_batch_probe.py 2>&1 | grep -E "KL metric|KL value"
NK_BENCH_CONFIG=C2 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/C2.log 2>&1
grep -o '"value": [0-9.]*\|"final_kl_energy": [0-9.e+-]*' gpurun_out/$tag/C2.log
