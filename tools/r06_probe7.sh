set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
tag=r06g; mkdir -p gpurun_out/$tag
for lib in build/libniftyk_tlplain.so build/libniftyk_tlplain2.so build/libniftyk_tlplain3.so; do
rm -rf gpurun_out/prof_$tag
NK_LIB_PATH=$PWD/$lib NK_TWO_LEVEL=1 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -- python3 tools/gpu_fused_probe.py 4096,4096 f64 > gpurun_out/$tag/prof.log 2>&1
python3 tools/rocpd_summary.py gpurun_out/prof_$tag/*/*.db > gpurun_out/$tag/probe_4096_$(basename $lib)_kernel_stats.txt
rm -rf gpurun_out/prof_$tag
grep "k2_tl\|k2_final" gpurun_out/$tag/probe_4096_$(basename $lib)_kernel_stats.txt | cut -c1-160
done
