"""Instruction mix per kernel of a device assembly file (hipcc --cuda-device-only -S): total instructions, packed /
scalar fp32 (or fp64) VALU arithmetic, LDS and global memory instructions.
usage: python tools/asm_mix.py file.s [name-regex]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\n(_Z\w+): +; @\1\n(.*?)\n\.Lfunc_end', txt, re.S):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    if pat and not re.search(pat, dem):
        continue
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    cnt = lambda rx: sum(1 for i in ins if re.match(rx, i))
    print(f"{dem[-52:]:52s} instr {len(ins):6d}  pk_f32 {cnt(r'v_pk_(add|mul|fma)_f32'):5d}  f32 {cnt(r'v_(add|sub|mul|fma|fmac)_f32'):5d}"
          f"  f64 {cnt(r'v_(add|mul|fma)_f64'):5d}  ds {cnt(r'ds_'):4d}  global {cnt(r'global_|buffer_|scratch_'):4d}  salu {cnt(r's_'):5d}"
          f"  mov {cnt(r'v_(mov|accvgpr)'):5d}")
