cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-batched}; mkdir -p gpurun_out/$tag
# batched small-grid path after the bin-sum / roll changes: parity, then C2 twice
python -m pytest tests/test_batched_gpu.py tests/test_kernels_gpu.py tests/test_api_gpu.py tests/test_engine_gpu.py -q -x 2>&1 | tail -30
for i in 1 2; do
NK_BENCH_CONFIG=C2 timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench_c2_$i.log 2>&1
grep -o '"value": [0-9.]*\|"launches_per_step[a-z_0-9]*": [0-9.]*\|"final_kl_energy": [0-9.e+-]*' gpurun_out/$tag/bench_c2_$i.log | tr '\n' ' '; echo
done
# one-rank RCCL communicator: the exchange timer with the corrected reduce-scatter spans
NK_FORCE_COMM=1 timeout 1200 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/bench_force_comm.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/$tag/bench_force_comm.log | head -1
grep -o '"exchange_per_cg_iteration_rank0": {[^}]*}\|"phase_seconds_per_step_rank0": {[^}]*}\|"final_kl_energy": [0-9.e+]*' gpurun_out/$tag/bench_force_comm.log
