# round 6, first look: standalone pass times at the C4 / C2 grids and C4 / C2 without the sample lanes -> gpurun_out/r06a/
set -u
cd "${GRAFT_REPO_ROOT:?}"
tag=r06a; mkdir -p gpurun_out/$tag
timeout 300 python3 tools/gpu_fused_probe.py 4096,4096 f64 > gpurun_out/$tag/probe_4096sq_f64.txt 2>&1
timeout 300 python3 tools/gpu_fused_probe.py 2048,2048 f64 > gpurun_out/$tag/probe_2048sq_f64.txt 2>&1
for cfg in C4 C2; do
  NK_BENCH_CONFIG=$cfg timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/${cfg}_default.log 2>&1
  NK_LANES=0 NK_BATCH=0 NK_BENCH_CONFIG=$cfg timeout 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/${cfg}_serial.log 2>&1
done
tail -3 gpurun_out/$tag/probe_4096sq_f64.txt
grep -o '"value": [0-9.]*' gpurun_out/$tag/C*.log
