#!/bin/bash
# usage: tools/build_pair_variant.sh <tag> [extra hipcc -D flags...]  -> build/libniftyk_<tag>.so whose pair final pass
# (nk_fft_p.hip) is compiled with the given flags, every other object taken from the product build (A/B timing with NK_LIB_PATH)
set -e
tag=$1; shift
cd "$(dirname "$0")/../nifty_amd/csrc"
mkdir -p ../../build/var_$tag
FL="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Wno-unused-function"
SIZES=${NK_VARIANT_SIZES:-X(64) X(128) X(256) X(512) X(1024) X(2048) X(4096)}
hipcc $FL "-DNK_FAST_SIZES(X)=$SIZES" "$@" -c nk_fft_p.hip -o ../../build/var_$tag/nk_fft_p.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/libniftyk_$tag.so ../../build/var_$tag/nk_fft_p.o nk_util.o nk_fft.o nk_fft_b.o nk_vec.o nk_amp.o nk_prod.o nk_rng.o
echo built build/libniftyk_$tag.so
