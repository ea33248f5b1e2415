# usage: bash tools/side_configs.sh <tag> [steps] -- NK_BENCH_CONFIG=C2 / C3 / C4 lines with the per-kernel averages
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-side}; steps=${2:-3}; mkdir -p gpurun_out/$tag
for c in ${NK_SIDE_CONFIGS:-C2 C3 C4}; do
  NK_BENCH_CONFIG=$c timeout 900 python bench.py --steps $steps --warmup 1 --no-cpu-baseline > gpurun_out/$tag/$c.log 2>&1
  python - <<P
import json
d=json.loads(open("gpurun_out/$tag/$c.log").read().strip().split("\n")[-1])
r=d["roofline"]
print("$c value", d["value"], "ms/transform", d.get("ms_per_transform_rank0"), "E", d.get("final_kl_energy"))
for k,v in r["all_pass_symbols"].items():
    print("   ", k, round(v["ms_total"]/v["launches"],3), "ms", v["launches"])
P
done
