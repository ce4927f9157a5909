#!/bin/bash
# usage: tools/build_variant.sh NAME "-DFLAG ..."  -> ab_libs/libarcvae_NAME.so (all kernel files rebuilt with the extra flags)
# A/B of two builds on one box: ARCVAE_HIP_LIB=$PWD/ab_libs/libarcvae_NAME.so python bench.py ...
set -e
NAME=$1; FLAGS=$2
cd "$(dirname "$0")/../mlx-vae_amd/csrc"
mkdir -p ../../ab_libs build/$NAME
for f in gemm misc lstm decoder latent; do
  /opt/rocm/bin/hipcc -I../../include -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function $FLAGS -c $f.hip -o build/$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_libs/libarcvae_$NAME.so build/$NAME/*.o
ls -la ../../ab_libs/libarcvae_$NAME.so
