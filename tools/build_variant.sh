#!/bin/bash
# usage: tools/build_variant.sh <tag> [extra hipcc -D flags...]   -> build/libniftyk_<tag>.so with only the 1024-point fast kernels
# (for A/B timing with NK_LIB_PATH; the product library is built by nifty_amd/csrc/Makefile)
set -e
tag=$1; shift
cd "$(dirname "$0")/../nifty_amd/csrc"
mkdir -p ../../build/var_$tag
FL="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-function"
SIZES=${NK_VARIANT_SIZES:-X(1024)}
hipcc $FL "-DNK_FAST_SIZES(X)=$SIZES" "$@" -Rpass-analysis=kernel-resource-usage -c nk_fft.hip -o ../../build/var_$tag/nk_fft.o 2> ../../build/var_$tag/remarks.txt
# (the batched twins and the pair final pass are explicit instantiations for the same size list: rebuilt alongside)
hipcc $FL "-DNK_FAST_SIZES(X)=$SIZES" "$@" -c nk_fft_b.hip -o ../../build/var_$tag/nk_fft_b.o &
hipcc $FL "-DNK_FAST_SIZES(X)=$SIZES" "$@" -c nk_fft_p.hip -o ../../build/var_$tag/nk_fft_p.o &
wait
for f in nk_util nk_vec nk_amp nk_prod nk_rng; do [ -f $f.o ] || hipcc $FL -c $f.hip -o $f.o; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libniftyk_$tag.so ../../build/var_$tag/nk_fft.o ../../build/var_$tag/nk_fft_b.o ../../build/var_$tag/nk_fft_p.o nk_util.o nk_vec.o nk_amp.o nk_prod.o nk_rng.o
echo built build/libniftyk_$tag.so
