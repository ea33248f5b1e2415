cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r4e
NK_BENCH_CONFIG=C2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4e/c2_line.log 2>&1
rm -rf gpurun_out/prof_c2
NK_BENCH_CONFIG=C2 NK_BENCH_PROFILE=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_c2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4e/c2_prof.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_c2/*/*.db > gpurun_out/r4e/c2_stats.txt
head -45 gpurun_out/r4e/c2_stats.txt | cut -c1-150
grep -o '"value": [0-9.]*' gpurun_out/r4e/c2_line.log gpurun_out/r4e/c2_prof.log
python -m pytest tests/test_widening.py tests/test_complex_gpu.py -q -m gpu 2>&1 | tail -3
