cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
python -m pytest tests/test_engine_gpu.py tests/test_batched_gpu.py tests/test_kernels_gpu.py -q -x 2>&1 | tail -3
for c in C3 C2 C4; do
  NK_BENCH_CONFIG=$c timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5q/$c.log 2>&1
  python - <<P
import json
d=json.loads(open("gpurun_out/r5q/$c.log").read().strip().split("\n")[-1])
r=d["roofline"]
print("$c value", d["value"], "ms/transform", d.get("ms_per_transform_rank0"), "E", d.get("final_kl_energy"))
for k,v in r["all_pass_symbols"].items():
    if "final" in k: print("   ", k, round(v["ms_total"]/v["launches"],3), "ms", v["launches"])
P
done
