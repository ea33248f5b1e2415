set -u
cd "${GRAFT_REPO_ROOT:?}"
tag=r06c; mkdir -p gpurun_out/$tag
timeout 900 python3 -m pytest tests/test_los_response.py tests/test_config4_gpu.py tests/test_batched_gpu.py -m gpu -x -q > gpurun_out/$tag/pytest_los.txt 2>&1
tail -5 gpurun_out/$tag/pytest_los.txt
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fence or rowsum or segment" > gpurun_out/$tag/pytest_fence.txt 2>&1
tail -5 gpurun_out/$tag/pytest_fence.txt
NK_BENCH_CONFIG=C4 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/C4_new.log 2>&1
for sh in 64,64 16,64 32,32 16,128; do
NK_TILED_SHAPE=$sh NK_BENCH_CONFIG=C4 timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/C4_tile_$sh.log 2>&1
done
grep -o '"value": [0-9.]*' gpurun_out/$tag/C*.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
NK_LANES=0 NK_BATCH=0 NK_BENCH_CONFIG=C4 NK_BENCH_PROFILE=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/prof.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_$tag/*/*.db > gpurun_out/$tag/c4_serial_kernel_stats.txt
rm -rf gpurun_out/prof_$tag
head -12 gpurun_out/$tag/c4_serial_kernel_stats.txt | cut -c1-150
