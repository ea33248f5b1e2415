# round 5, first GPU call: the whole -m gpu suite, then the C2 baseline line + kernel trace (before the sample batching)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
python -m pytest tests -m gpu -x -q > gpurun_out/r5a/gputests.log 2>&1
tail -5 gpurun_out/r5a/gputests.log
NK_BENCH_CONFIG=C2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5a/c2_line.log 2>&1
rm -rf gpurun_out/prof_c2
NK_BENCH_CONFIG=C2 NK_BENCH_PROFILE=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5a/c2_prof.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_c2/*/*.db > gpurun_out/r5a/c2_stats.txt
head -50 gpurun_out/r5a/c2_stats.txt | cut -c1-160
grep -o '"value": [0-9.]*' gpurun_out/r5a/c2_line.log gpurun_out/r5a/c2_prof.log
