# A/B of the NK_NT_LOAD build variants (NK_VARIANT_SIZES="X(512) X(1024)" tools/build_variant.sh s_nt<mask> -DNK_NT_LOAD=<mask>)
# with the per-pass probe: sandwich passes S1 (bit 2), SM (bit 16), final (bit 4)
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/r4g
for v in s_nt0 s_nt2 s_nt16 s_nt4 s_nt0 s_nt2; do
  NK_LIB_PATH=$PWD/build/libniftyk_$v.so python tools/gpu_fused_probe.py > gpurun_out/r4g/probe_$v.log 2>&1
  echo "== $v"; grep "metric acc+d\|metric 1st " gpurun_out/r4g/probe_$v.log | cut -c1-120
done
