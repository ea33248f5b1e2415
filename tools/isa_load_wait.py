"""Per kernel of an AMDGPU assembly listing (hipcc -S --cuda-device-only ... -o x.s): the sequence of global loads (L), stores (S),
scratch accesses (X), s_waitcnt vmcnt(n) (wn), barriers (|B|) and branches (<br>).  A healthy memory phase reads "LLLL...w7w6...";
"Lw0<br>Lw0<br>" is a loop of dependent load -> use chains.  Usage: python tools/isa_load_wait.py x.s '<regex on the demangled name>'"""
import re, subprocess, sys
txt=open(sys.argv[1]).read()
pat=sys.argv[2]
for m in re.finditer(r'\n(_Z\w+): +; @\1\n(.*?)\n\.Lfunc_end', txt, re.S):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    if not re.search(pat, dem): continue
    lines=[l.strip() for l in body.split('\n') if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    seq=[]
    for l in lines:
        op=l.split()[0]
        if op.startswith("global_load") or op.startswith("buffer_load"): seq.append("L")
        elif op.startswith("global_store") or op.startswith("buffer_store"): seq.append("S")
        elif op.startswith("scratch_"): seq.append("X")
        elif op=="s_waitcnt":
            a=l.split(None,1)[1]
            mm=re.search(r"vmcnt\((\d+)\)",a)
            seq.append("w%s"%mm.group(1) if mm else "")
        elif op=="s_barrier": seq.append("|B|")
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): seq.append("<br>")
    print(dem); print("  "+"".join(seq)[:1600])
