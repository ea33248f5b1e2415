# usage: bash tools/multi_rank_lines.sh <tag> -- `bench.py --gpus N` for N = 1, 2, 4 ranks SHARING one GPU over gloo at 256^3 fp32
# (a 1-GPU box cannot run RCCL with more than one rank): every rank count must print the same per_step_counts_rank0 and
# final_kl_energy (rank-count-independent sums), and for N > 1 the exchange fields of the sharded CG; plus the ONE-rank RCCL
# communicator on the full workload (NK_FORCE_COMM=1).  -> gpurun_out/<tag>/
set -u
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-ranks}; mkdir -p gpurun_out/$tag
export NK_BENCH_SHAPE=256,256,256 NK_DIST_BACKEND=gloo NK_SHARE_DEVICE=1
python3 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/ranks1.log 2>&1
for n in 2 4; do
  timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29520 + n)) \
    bench.py --gpus $n --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/ranks$n.log 2>&1
done
for n in 1 2 4; do
  grep "^{\"metric\"" gpurun_out/$tag/ranks$n.log > gpurun_out/$tag/bench_${n}ranks_one_gpu_gloo_256cube_line.json
  grep -o '"n_gpus": [0-9]*\|"final_kl_energy": [0-9.e+]*\|"per_step_counts_rank0": {[^}]*}\|"comm_ms_per_cg_iteration": [0-9.a-z]*' gpurun_out/$tag/ranks$n.log | tr '\n' ' '; echo
done
unset NK_BENCH_SHAPE NK_DIST_BACKEND NK_SHARE_DEVICE
NK_FORCE_COMM=1 timeout 1200 python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/force_comm.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/force_comm.log > gpurun_out/$tag/bench_force_comm_one_rank_rccl_line.json
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/$tag/force_comm.log | head -2
