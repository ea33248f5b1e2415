set -u
cd "${GRAFT_REPO_ROOT:?}"
tag=r06f; mkdir -p gpurun_out/$tag
timeout 600 python3 -m pytest tests/test_small_ops.py tests/test_abi.py -m gpu -x -q > gpurun_out/$tag/pytest.txt 2>&1
tail -3 gpurun_out/$tag/pytest.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for tl in 1 0; do
rm -rf gpurun_out/prof_$tag
NK_TWO_LEVEL=$tl timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python3 tools/gpu_fused_probe.py 4096,4096 f64 > gpurun_out/$tag/prof_tl$tl.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_$tag/*/*.db > gpurun_out/$tag/probe_4096_tl${tl}_kernel_stats.txt
rm -rf gpurun_out/prof_$tag
head -14 gpurun_out/$tag/probe_4096_tl${tl}_kernel_stats.txt | cut -c1-160
done
