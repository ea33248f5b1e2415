"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr file) per kernel."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
K = {"vgpr": r"VGPRs", "agpr": r"AGPRs", "scratch": r"ScratchSize \[bytes/lane\]", "occ": r"Occupancy \[waves/SIMD\]", "lds": r"LDS Size \[bytes/block\]"}
for b in blocks:
    name = b.split('\n')[0].split()[0]
    dem = subprocess.run(['c++filt', name.strip()], capture_output=True, text=True).stdout.strip()
    dem = dem.split('(')[0][-58:]
    if pat and not re.search(pat, dem):
        continue
    vals = {}
    for k, r in K.items():
        m = re.search(r + r': (\d+)', b)
        vals[k] = int(m.group(1)) if m else -1
    print(f"{dem:58s} VGPR {vals['vgpr']:4d} AGPR {vals['agpr']:3d} scratch {vals['scratch']:4d} occ {vals['occ']} lds {vals['lds']}")
