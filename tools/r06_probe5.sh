# round 6: two-level first-axis pass (NK_TWO_LEVEL) -- parity tests, per-pass probe, C2 / C4 lines with and without it
set -u
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-r06e}; mkdir -p gpurun_out/$tag
timeout 1500 python3 -m pytest tests/test_large_oracle_gpu.py tests/test_config4_gpu.py tests/test_batched_gpu.py tests/test_small_ops.py -m gpu -x -q > gpurun_out/$tag/pytest.txt 2>&1
tail -5 gpurun_out/$tag/pytest.txt
timeout 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/$tag/pytest2.txt 2>&1
tail -3 gpurun_out/$tag/pytest2.txt
for tl in 1 0; do
NK_TWO_LEVEL=$tl timeout 300 python3 tools/gpu_fused_probe.py 4096,4096 f64 > gpurun_out/$tag/probe_4096sq_f64_tl$tl.txt 2>&1
NK_TWO_LEVEL=$tl timeout 300 python3 tools/gpu_fused_probe.py 2048,2048 f64 > gpurun_out/$tag/probe_2048sq_f64_tl$tl.txt 2>&1
for cfg in C2 C4; do
NK_TWO_LEVEL=$tl NK_BENCH_CONFIG=$cfg timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/${cfg}_tl$tl.log 2>&1
done
done
grep -h "k_passA" gpurun_out/$tag/probe_4096sq_f64_tl1.txt | head -8
grep -h "k_passA" gpurun_out/$tag/probe_4096sq_f64_tl0.txt | head -8
grep -o '"value": [0-9.]*\|"ms_per_transform_rank0": [0-9.]*' gpurun_out/$tag/C*.log
