# usage: bash tools/round_end_profile.sh <tag>  -- the judged command (driver style, with the NK_BENCH_API leg), the default
# command, its kernel trace and its PMC traffic -> gpurun_out/<tag>/ (copy what is to be judged to profiles/)
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-r05}
mkdir -p gpurun_out/$tag
NK_BENCH_API=1 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench_driver_style.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/bench_driver_style.log > gpurun_out/$tag/bench_driver_style_line.json
python bench.py > gpurun_out/$tag/bench_default.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/bench_default.log > gpurun_out/$tag/bench_default_line.json
bash tools/prof_bench.sh ${tag} --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/prof.log 2>&1
bash tools/pmc_bench.sh ${tag} > gpurun_out/$tag/pmc.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/$tag/*.json | head
grep -o '"api": {[^}]*}' gpurun_out/$tag/bench_driver_style_line.json
grep -o '"ms_per_transform_rank0": [0-9.]*' gpurun_out/$tag/*.json
