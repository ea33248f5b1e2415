# usage: bash tools/flake_probe.sh [n]  -- repeats the short bench n times (default 4) and prints step time, work counters
# and the final KL energy of each run: a run whose energy or value/gradient count is far off the others points at
# corrupted results (this is how the out-of-bounds octant writes of a too-small VJP tile showed up)
cd "${GRAFT_REPO_ROOT:?}"
for i in $(seq 1 ${1:-4}); do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', round(d['ms_per_step']), d['per_step_counts_rank0'], d['final_kl_energy'])"
done
