# side configurations C2 / C4 of bench.py with and without the sample lanes -> gpurun_out/<tag>/
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-r4f}; mkdir -p gpurun_out/$tag
for cfg in C2 C4; do
  for lanes in 0 4; do
    NK_LANES=$lanes NK_BENCH_CONFIG=$cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/${cfg}_lanes${lanes}.log 2>&1
  done
done
NK_LANES=2 NK_BENCH_CONFIG=C2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/C2_lanes2.log 2>&1
NK_LANES=8 NK_BENCH_CONFIG=C2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/C2_lanes8.log 2>&1
grep -o '"value": [0-9.]*\|"final_kl_energy": [0-9.e+-]*' gpurun_out/$tag/C*.log
