cd "${GRAFT_REPO_ROOT:?}"
run() { shape=$1; dt=$2; shift 2; env "$@" python - <<PY 2>/dev/null
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
for case in "768,768,768 f32" "960,960,960 f32" "640,640,640 f32" "3072,3072 f32" "1000,1000 f64" "3000,3000 f64" "1536,1536 f64" "5000,5000 f32"; do
  set -- $case
  echo "== $1 $2 default: $(run $1 $2 NK_X=0)"
  for ta in 8 16; do for tha in 256 512; do for ths in 256 512; do for tb in 16 32; do
    echo "A=$ta/$tha S=$ths TILE=$tb: $(run $1 $2 NK_TILE_A=$ta NK_THREADS_A=$tha NK_THREADS_S=$ths NK_TILE_B=$tb NK_TILE_C=$tb)"
  done; done; done; done
done
