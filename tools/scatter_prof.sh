cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for r in 1 0; do
rm -rf gpurun_out/prof_sc$r
NK_SCATTER_RANKED=$r rocprofv3 --kernel-trace --stats -d gpurun_out/prof_sc$r -- python3 tools/gpu_scatter_probe.py > /dev/null 2>&1
echo "ranked=$r"; python3 tools/rocpd_summary.py gpurun_out/prof_sc$r/*/*.db | grep "scatter\|fold_copies" | cut -c1-60,90-170
rm -rf gpurun_out/prof_sc$r
done
