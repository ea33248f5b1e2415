#!/bin/bash
# per-pass kernel times of the fused probe for every block-order variant (NK_XMAP bitmask: 1 first pass, 2 in-place, 4 final)
for m in ${@:-0 1 2 4}; do
  echo "== NK_XMAP=$m"
  NK_XMAP=$m python tools/gpu_fused_probe.py 2>/dev/null | grep -v "^/opt"
done
