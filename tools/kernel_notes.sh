#!/bin/bash
# VGPRs / spills / LDS / scratch of the kernels in libniftyk.so whose demangled name matches $1 (regex), read from the code
# object's metadata notes:  tools/kernel_notes.sh 'k2_final2<float, 1024'
set -e
LIB=${2:-$(dirname $0)/../nifty_amd/csrc/libniftyk.so}
D=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$D/fat.bin $LIB $D/copy.out  # (an output file: without one objcopy rewrites its input in place)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$D/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$D/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $D/dev.co > $D/notes.txt
python3 - "$D/notes.txt" "$1" <<'P'
import re, subprocess, sys
t = open(sys.argv[1]).read()
pat = sys.argv[2]
out = []
for k in re.split(r'\n\s+- \.agpr_count:', t)[1:]:
    g = lambda key: int(re.search(key + r':\s+(\d+)', k).group(1))
    out.append((re.search(r'\.name:\s+(\S+)', k).group(1), g(r'\.vgpr_count'), g(r'\.vgpr_spill_count'), g(r'\.sgpr_spill_count'),
                g(r'\.group_segment_fixed_size'), g(r'\.private_segment_fixed_size'), g(r'\.max_flat_workgroup_size')))
names = subprocess.run(['c++filt'], input='\n'.join(o[0] for o in out), capture_output=True, text=True).stdout.split('\n')
print("kernel | VGPRs | spilled VGPRs | spilled SGPRs | static LDS | scratch B/lane | max workgroup")
for n, o in zip(names, out):
    if re.search(pat, n):
        print(n.split('(')[0][-72:], *o[1:])
P
rm -rf $D
