cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
python tools/gpu_batch_probe.py > gpurun_out/r5d/probe.log 2>&1
grep -E "KL metric|KL value" gpurun_out/r5d/probe.log
python -m pytest tests/test_batched_gpu.py tests/test_rank_independent_gpu.py tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
rm -rf gpurun_out/prof_probe
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_probe -- python3 tools/gpu_batch_probe.py > /dev/null 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_probe/*/*.db > gpurun_out/r5d/probe_stats.txt
head -32 gpurun_out/r5d/probe_stats.txt | cut -c1-150
NK_BENCH_CONFIG=C2 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5d/C2.log 2>&1
grep -o '"value": [0-9.]*\|"final_kl_energy": [0-9.e+-]*' gpurun_out/r5d/C2.log
