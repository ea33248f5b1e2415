# sweep of the generic kernels' launch parameters (tile / threads of the contiguous pass A and of the strided passes) on one
# mixed-radix grid.  usage: bash tools/generic_sweep.sh "768,768,768" f32
cd "${GRAFT_REPO_ROOT:?}"
shape=${1:-768,768,768}; dt=${2:-f32}
run() { env "$@" python - <<PY 2>/dev/null
import sys, torch
sys.path.insert(0, ".")
from nifty_amd import backend as B
shape = tuple(int(s) for s in "$shape".split(","))
x = torch.randn(shape, dtype=torch.float32 if "$dt" == "f32" else torch.float64, device="cuda")
out = torch.empty_like(x)
B.hartley(x, out=out); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): B.hartley(x, out=out)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / 5:8.3f} ms")
PY
}
echo "default: $(run NK_X=0)"
for ta in 4 8 12 16 24; do for tha in 256 512 768 1024; do echo "TILE_A=$ta THREADS_A=$tha: $(run NK_TILE_A=$ta NK_THREADS_A=$tha)"; done; done
for ths in 128 256 512 1024; do echo "THREADS_S=$ths: $(run NK_THREADS_S=$ths)"; done
for tb in 8 16 32; do echo "TILE_B=$tb TILE_C=$tb: $(run NK_TILE_B=$tb NK_TILE_C=$tb)"; done
