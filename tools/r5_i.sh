cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r5i}; mkdir -p gpurun_out/$tag
NK_REQUIRE_FULL=1 python -m pytest tests/test_large_oracle_gpu.py tests/test_engine_gpu.py tests/test_kernels_gpu.py -q -s -m gpu > gpurun_out/$tag/tests.log 2>&1
tail -3 gpurun_out/$tag/tests.log; grep "vs fp64 oracle\|fp32 gradient vs" gpurun_out/$tag/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/$tag/bench.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/bench.log > gpurun_out/$tag/bench_line.json
python - <<'P'
import json,os
d=json.load(open("gpurun_out/%s/bench_line.json" % os.environ.get("TAG","r5i")))
print({k:d.get(k) for k in ("value","ms_per_step","ms_per_transform_rank0","final_kl_energy")}, d["per_step_counts_rank0"])
for k,v in d["roofline"]["all_pass_symbols"].items(): print("  ",k, round(v["ms_total"]/v["launches"],3), v)
P
