# usage: bash tools/prof_bench.sh <tag> [bench args]  -- rocprofv3 kernel trace of bench.py, summary to gpurun_out/<tag>_stats.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
tag=$1; shift
rm -rf gpurun_out/prof_$tag
timeout 800 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python3 bench.py "$@" > gpurun_out/bench_$tag.log 2>&1
grep "^{\"metric\"" gpurun_out/bench_$tag.log > gpurun_out/bench_$tag.json
python3 tools/rocpd_summary.py gpurun_out/prof_$tag/*/*.db > gpurun_out/${tag}_stats.txt
head -32 gpurun_out/${tag}_stats.txt | cut -c1-170
