cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r5e}; mkdir -p gpurun_out/$tag
for cfg in C2 C4; do
  NK_BENCH_CONFIG=$cfg timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/${cfg}.log 2>&1
  rm -rf gpurun_out/prof_$cfg
  NK_BENCH_CONFIG=$cfg NK_BENCH_PROFILE=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$cfg -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/${cfg}_prof.log 2>&1
  python3 tools/rocpd_summary.py gpurun_out/prof_$cfg/*/*.db > gpurun_out/$tag/${cfg}_stats.txt
  python3 tools/rocpd_gaps.py gpurun_out/prof_$cfg/*/*.db 15 0.3 > gpurun_out/$tag/${cfg}_gaps.txt
  rm -rf gpurun_out/prof_$cfg
done
grep -o '"value": [0-9.]*\|"final_kl_energy": [0-9.e+-]*\|"step_hbm_GBps_rank0": [0-9.]*' gpurun_out/$tag/C*.log
head -12 gpurun_out/$tag/C2_gaps.txt | cut -c1-150
head -30 gpurun_out/$tag/C4_stats.txt | cut -c1-150
head -14 gpurun_out/$tag/C4_gaps.txt | cut -c1-150
