cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r5h}; mkdir -p gpurun_out/$tag
NK_REQUIRE_FULL=1 python -m pytest tests -m gpu -q -s > gpurun_out/$tag/gputests.log 2>&1
tail -4 gpurun_out/$tag/gputests.log; grep "peak device" gpurun_out/$tag/gputests.log
NK_BENCH_API=1 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench_driver_style.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/bench_driver_style.log > gpurun_out/$tag/bench_driver_style_line.json
python - <<'P'
import json,os
tag=os.environ.get("TAG","r5h")
d=json.load(open(f"gpurun_out/{tag}/bench_driver_style_line.json"))
print({k:d.get(k) for k in ("value","ms_per_step","ms_per_transform_rank0","api_overhead_pct","phase_seconds_per_step_rank0")})
print(d.get("api")); print({k:d["roofline"].get(k) for k in ("kernel","achieved","frac","avg_launch_ms")}); print(d.get("device_memory_rank0"))
P
for cfg in C2 C4 C3; do
  NK_BENCH_CONFIG=$cfg timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/$tag/bench_${cfg}.log 2>&1
  grep "^{\"metric\"" gpurun_out/$tag/bench_${cfg}.log > gpurun_out/$tag/bench_${cfg}_line.json
done
grep -o '"value": [0-9.]*' gpurun_out/$tag/bench_C*_line.json | head -3
grep -o '"step_hbm_GBps_rank0": [0-9.]*' gpurun_out/$tag/bench_C*_line.json
# two ranks sharing the GPU (gloo): exercises the sharded CG, the exchange timer and the phase timing of the bench line
NK_DIST_BACKEND=gloo NK_SHARE_DEVICE=1 NK_BENCH_SHAPE=256,256,256 NK_BENCH_DTYPE=f64 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/bench_2rank_shared.log 2>&1
grep -o '"exchange_per_cg_iteration_rank0": {[^}]*}\|"phase_seconds_per_step_rank0": {[^}]*}' gpurun_out/$tag/bench_2rank_shared.log
