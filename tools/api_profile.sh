# rocprofv3 kernel trace of the headline workload through ift.optimize_kl (tools/run_c5_api.py, 2 iterations): per-kernel summary
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/apiprof; rm -rf gpurun_out/prof_api
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_api -- python3 tools/run_c5_api.py 1024 2 > gpurun_out/apiprof/run.log 2>&1
tail -2 gpurun_out/apiprof/run.log
python3 tools/rocpd_summary.py gpurun_out/prof_api/*/*.db > gpurun_out/apiprof/kernel_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/prof_api/*/*.db > gpurun_out/apiprof/idle_gaps.txt 2>&1
head -34 gpurun_out/apiprof/kernel_stats.txt | cut -c1-165
head -12 gpurun_out/apiprof/idle_gaps.txt
rm -rf gpurun_out/prof_api
