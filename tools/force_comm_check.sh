cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-force_comm}; mkdir -p gpurun_out/$tag
# a ONE-rank RCCL communicator: the sharded CG with its side-stream exchange, staged passes and the exchange timer
NK_FORCE_COMM=1 timeout 1200 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/bench_force_comm.log 2>&1
tail -3 gpurun_out/$tag/bench_force_comm.log | cut -c1-300
grep -o '"value": [0-9.]*' gpurun_out/$tag/bench_force_comm.log | head -1
grep -o '"exchange_per_cg_iteration_rank0": {[^}]*}\|"phase_seconds_per_step_rank0": {[^}]*}\|"final_kl_energy": [0-9.e+]*' gpurun_out/$tag/bench_force_comm.log
NK_REQUIRE_FULL=1 python -m pytest tests/test_large_oracle_gpu.py::test_config5_full_size_against_the_oracle tests/test_api_large_gpu.py -q -s 2>&1 | tail -4
