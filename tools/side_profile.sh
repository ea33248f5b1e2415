# usage: bash tools/side_profile.sh <C2|C3|C4> <tag> -- rocprofv3 kernel trace of a side configuration: per-kernel summary and
# the idle-gap analysis -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
cfg=$1; tag=${2:-side_$1}; mkdir -p gpurun_out/$tag; rm -rf gpurun_out/prof_$tag
NK_BENCH_CONFIG=$cfg NK_BENCH_PROFILE=0 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/prof.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_$tag/*/*.db > gpurun_out/$tag/kernel_stats.txt
python3 tools/rocpd_gaps.py gpurun_out/prof_$tag/*/*.db > gpurun_out/$tag/idle_gaps.txt 2>&1
head -40 gpurun_out/$tag/kernel_stats.txt | cut -c1-175
head -8 gpurun_out/$tag/idle_gaps.txt
grep -o '"value": [0-9.]*' gpurun_out/$tag/prof.log | head -1
rm -rf gpurun_out/prof_$tag
