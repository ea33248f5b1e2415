# usage: bash tools/check_spills.sh [pattern]  -- recompiles nifty_amd/csrc/*.hip with -Rpass-analysis=kernel-resource-usage and
# lists every kernel whose register allocation went to scratch (VGPR cap from __launch_bounds__ too tight).  Known and
# accepted: the 1024-thread strided kernels for line lengths 2048 / 4096 (hardware cap of 128 VGPRs) and the generic
# (run-time epilogue) scatter variant of the final pass.
cd "$(dirname "$0")/../nifty_amd/csrc"
for f in nk_fft nk_vec nk_amp nk_util; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-function \
        -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/spill_$f.o 2> /tmp/spill_$f.txt
  python3 ../../tools/kernel_resources.py /tmp/spill_$f.txt "${1:-}" 2>/dev/null |
    awk '{for(i=1;i<=NF;i++) if($i=="scratch" && $(i+1)>0) print}'
done
