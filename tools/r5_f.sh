cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-r5f}; mkdir -p gpurun_out/$tag
python -m pytest tests/test_batched_gpu.py tests/test_engine_gpu.py tests/test_config4_gpu.py tests/test_api_large_gpu.py tests/test_driver_io.py -x -q -m gpu > gpurun_out/$tag/tests.log 2>&1
tail -4 gpurun_out/$tag/tests.log; grep "peak device" gpurun_out/$tag/tests.log
for cfg in C2 C4; do
  NK_BENCH_CONFIG=$cfg timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$tag/${cfg}.log 2>&1
done
grep -o '"value": [0-9.]*\|"final_kl_energy": [0-9.e+-]*\|"step_hbm_GBps_rank0": [0-9.]*' gpurun_out/$tag/C*.log
rm -rf gpurun_out/prof_C2
NK_BENCH_CONFIG=C2 NK_BENCH_PROFILE=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_C2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/C2_prof.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_C2/*/*.db > gpurun_out/$tag/C2_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/prof_C2/*/*.db 15 0.3 > gpurun_out/$tag/C2_gaps.txt
rm -rf gpurun_out/prof_C2
head -8 gpurun_out/$tag/C2_gaps.txt | cut -c1-150
python - <<'P'
tot=0;calls=0
for l in open('gpurun_out/'+__import__('sys').argv[1] if False else 'gpurun_out/TAG/C2_stats.txt'.replace('TAG', __import__('os').environ.get('TAG','r5f'))).read().splitlines()[1:]:
    p=l.split()
    try:
        c=int(p[-6]); t=float(p[-5])
    except Exception: continue
    tot+=t; calls+=c
print("C2 kernel ms/step", tot/4, "launches/step", calls/4)
P
