# A/B of the host draw-ahead (NK_DRAW_AHEAD): judged bench command, sampling phase seconds and final energy (must agree)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/da
for a in ${NK_AB_LIST:-1 0}; do
NK_DRAW_AHEAD=$a python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/da/ahead$a.log 2>&1
python - <<P
import json
d=json.loads(open("gpurun_out/da/ahead$a.log").read().strip().split("\n")[-1])
print("NK_DRAW_AHEAD=$a", d["value"], d["ms_per_step"], d["phase_seconds_per_step_rank0"], d["final_kl_energy"], d["per_step_counts_rank0"]["transforms"])
P
done
python -m pytest tests/test_engine_gpu.py tests/test_api_large_gpu.py tests/test_rank_independent_gpu.py -q -x 2>&1 | tail -3
