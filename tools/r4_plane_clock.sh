#!/bin/bash
# The clock the chip holds inside the plane GEMM's K loop (diagnostic build -DARCVAE_PLANE_CLOCK: s_memtime / s_memrealtime around the
# loop of every block; never the shipped library).  GPU box, repo root: tools/r4_plane_clock.sh
set -e
tools/build_variant.sh clock "-DARCVAE_PLANE_CLOCK" > /dev/null
mkdir -p gpurun_out
ARCVAE_HIP_LIB=$PWD/ab_libs/libarcvae_clock.so ARCVAE_PLANE_CLOCK_PRINT=1 python tools/bench_planes.py 2>&1 | grep "plane clock\|^planes" | tail -4 | tee gpurun_out/r4_plane_clock.txt
