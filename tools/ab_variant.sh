# usage: bash tools/ab_variant.sh <variant tag> [C3|C2|C4|main ...] -- bench lines of the product library and of build/libniftyk_<tag>.so
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
tag=$1; shift; mkdir -p gpurun_out/ab_$tag
for cfg in "$@"; do
for lib in product $tag; do
  if [ $lib = product ]; then unset NK_LIB_PATH; else export NK_LIB_PATH=build/libniftyk_$lib.so; fi
  if [ $cfg = main ]; then unset NK_BENCH_CONFIG; else export NK_BENCH_CONFIG=$cfg; fi
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$tag/${cfg}_$lib.log 2>&1
  python - <<P
import json
d=json.loads(open("gpurun_out/ab_$tag/${cfg}_$lib.log").read().strip().split("\n")[-1])
r=d["roofline"]
print("$cfg $lib value", d["value"], "ms/transform", d.get("ms_per_transform_rank0"), "E", d.get("final_kl_energy"), d.get("per_step_counts_rank0",{}).get("transforms"))
for k,v in r["all_pass_symbols"].items():
    print("   ", k, round(v["ms_total"]/v["launches"],3), "ms", v["launches"])
P
done
done
