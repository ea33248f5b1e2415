# batched launches: parity tests, then C2 / C4 lines with and without the batch, and the C2 kernel trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r5c}; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests/test_batched_gpu.py -x -q > gpurun_out/$tag/batched_tests.log 2>&1
tail -15 gpurun_out/$tag/batched_tests.log
for cfg in C2 C4; do
  for batch in 1 0; do
    NK_BATCH=$batch NK_BENCH_CONFIG=$cfg timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/${cfg}_batch${batch}.log 2>&1
  done
done
grep -o '"value": [0-9.]*\|"final_kl_energy": [0-9.e+-]*\|"step_hbm_GBps_rank0": [0-9.]*' gpurun_out/$tag/C*.log
rm -rf gpurun_out/prof_c2
NK_BENCH_CONFIG=C2 NK_BENCH_PROFILE=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/c2_prof.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_c2/*/*.db > gpurun_out/$tag/c2_stats.txt
head -40 gpurun_out/$tag/c2_stats.txt | cut -c1-160
