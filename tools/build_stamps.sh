#!/bin/bash
# Diagnostic library with eight stamps per tick in the persistent sweeps (-DARCVAE_PS_STAMPS): ab_libs/libarcvae_stamps.so
# (ab_libs/ is git-ignored and travels to the GPU box); used by tools/tick_stamps.py through ARCVAE_HIP_LIB.
set -e
cd "$(dirname "$0")/../mlx-vae_amd/csrc"
mkdir -p ../../ab_libs build
/opt/rocm/bin/hipcc -I../../include -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -DARCVAE_PS_STAMPS -c lstm.hip -o build/lstm_stamps.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../ab_libs/libarcvae_stamps.so build/gemm.o build/misc.o build/lstm_stamps.o build/decoder.o build/latent.o
ls -la ../../ab_libs/libarcvae_stamps.so
