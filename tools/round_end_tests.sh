# usage: bash tools/round_end_tests.sh <tag> -- the driver's round-end checks on one box: pytest -m gpu (incl. the full-size
# oracle tests), smoke(), then the judged bench command
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-r05end}; mkdir -p gpurun_out/$tag
NK_REQUIRE_FULL=1 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/$tag/pytest_gpu_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/$tag/smoke_tail.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench_driver_style.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/bench_driver_style.log > gpurun_out/$tag/bench_driver_style_line.json
grep -o '"value": [0-9.]*\|"ms_per_transform_rank0": [0-9.]*\|"final_kl_energy": [0-9.e+]*' gpurun_out/$tag/bench_driver_style_line.json | head -4
