# kernel-only times of the octant scatter kernels (rocprofv3 on tools/gpu_scatter_probe.py) and the digest of the bin sums
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
for shape in ${NK_SCATTER_SHAPES:-1024,1024,1024 512,512,512}; do
rm -rf gpurun_out/prof_sc
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sc -- python3 tools/gpu_scatter_probe.py $shape > gpurun_out/prof_sc.log 2>&1
echo "$shape"; grep "sha1\|bit-identical" gpurun_out/prof_sc.log; python3 tools/rocpd_summary.py gpurun_out/prof_sc/*/*.db | grep "scatter_k2\|fold_copies\|expand" | cut -c1-60,90-170
rm -rf gpurun_out/prof_sc
done
