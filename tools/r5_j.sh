cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r5j}; mkdir -p gpurun_out/$tag
python -m pytest tests -q -m gpu --deselect tests/test_large_oracle_gpu.py::test_config5_full_size_against_the_oracle --deselect tests/test_api_large_gpu.py::test_headline_workload_through_optimize_kl > gpurun_out/$tag/tests.log 2>&1
tail -3 gpurun_out/$tag/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$tag/bench.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/bench.log > gpurun_out/$tag/bench_line.json
NK_BENCH_CONFIG=C2 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/C2.log 2>&1
python - <<'P'
import json,os
d=json.load(open("gpurun_out/%s/bench_line.json" % os.environ.get("TAG","r5j")))
print({k:d.get(k) for k in ("value","ms_per_step","ms_per_transform_rank0","final_kl_energy","phase_seconds_per_step_rank0")}, d["per_step_counts_rank0"])
P
grep -o '"value": [0-9.]*' gpurun_out/$tag/C2.log | head -1
