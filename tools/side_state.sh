# round 6: where the three configurations stand at this commit -- the judged command (short), C2 / C4 / C3 lines and the kernel
# traces of C2 / C4 -> gpurun_out/<tag>/
set -u
cd "${GRAFT_REPO_ROOT:?}"
tag=${1:-r06i}; mkdir -p gpurun_out/$tag
timeout 900 python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/bench_main_short.log 2>&1
grep -o '"value": [0-9.]*\|"ms_per_transform_rank0": [0-9.]*\|"final_kl_energy": [0-9.e+]*' gpurun_out/$tag/bench_main_short.log | head -3
for cfg in C2 C3 C4; do
NK_BENCH_CONFIG=$cfg timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/${cfg}.log 2>&1
grep "^{\"metric\"" gpurun_out/$tag/${cfg}.log > gpurun_out/$tag/bench_${cfg}_line.json
done
grep -o '"value": [0-9.]*' gpurun_out/$tag/C*.log
bash tools/side_profile.sh C2 ${tag}_c2 > gpurun_out/$tag/c2_prof.txt 2>&1
bash tools/side_profile.sh C4 ${tag}_c4 > gpurun_out/$tag/c4_prof.txt 2>&1
head -24 gpurun_out/${tag}_c2/kernel_stats.txt | cut -c1-150
