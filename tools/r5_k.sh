cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r5k}; mkdir -p gpurun_out/$tag
for v in "default" "NK_GEO_THREADS=0" "NK_LANES=2" "NK_LANES=8" "NK_LANES=0"; do
  name=$(echo $v | tr '=' '_')
  if [ "$v" = "default" ]; then env NK_BENCH_CONFIG=C4 NK_BENCH_PHASES=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/C4_$name.log 2>&1
  else env $v NK_BENCH_CONFIG=C4 NK_BENCH_PHASES=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/C4_$name.log 2>&1; fi
  echo "$v: $(grep -o '"value": [0-9.]*' gpurun_out/$tag/C4_$name.log | head -1) $(grep -o '"final_kl_energy": [0-9.e+-]*' gpurun_out/$tag/C4_$name.log) $(grep -o '"phase_seconds_per_step_rank0": {[^}]*}' gpurun_out/$tag/C4_$name.log)"
done
