# usage: bash tools/round_end_profile.sh <tag>  -- the judged command (driver style), its kernel trace and its PMC traffic
cd $GRAFT_REPO_ROOT
tag=${1:-r04}
mkdir -p gpurun_out/$tag
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench_driver_style.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/bench_driver_style.log > gpurun_out/$tag/bench_driver_style_line.json
python bench.py > gpurun_out/$tag/bench_default.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/bench_default.log > gpurun_out/$tag/bench_default_line.json
bash tools/prof_bench.sh ${tag} --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/prof.log 2>&1
bash tools/pmc_bench.sh ${tag} > gpurun_out/$tag/pmc.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/$tag/*.json | head
