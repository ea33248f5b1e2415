cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
for lib in product qw4; do
  if [ $lib = product ]; then unset NK_LIB_PATH; else export NK_LIB_PATH=build/libniftyk_$lib.so; fi
  timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5q/b_$lib.log 2>&1
  python - <<P
import json
d=json.loads(open("gpurun_out/r5q/b_$lib.log").read().strip().split("\n")[-1])
r=d["roofline"]
print("$lib value", d["value"], "ms/transform", d.get("ms_per_transform_rank0"), "E", d.get("final_kl_energy"))
for k,v in r["all_pass_symbols"].items():
    print("   ", k, round(v["ms_total"]/v["launches"],3), "ms", v["launches"])
P
done
