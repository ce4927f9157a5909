#!/bin/bash
# usage: tools/kernel_regs.sh FILE.hip [grep pattern]  -> VGPR / AGPR / SGPR / spill counts per kernel (device-only compile)
F=$1; PAT=${2:-.}
cd "$(dirname "$0")/../mlx-vae_amd/csrc"
/opt/rocm/bin/hipcc -I../../include -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -c $F -o /tmp/kr_dev.o || exit 1
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/kr_dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/kr_gfx950.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/kr_gfx950.o | grep -E "^\s+\.name:|\.vgpr_count|\.sgpr_count|\.agpr_count|\.vgpr_spill" | paste - - - - - | sed 's/ \+/ /g' \
  | awk '{n="";for(i=1;i<=NF;i++){if($i==".name:")n=$(i+1)} ; a="";v="";s="";sp=""; for(i=1;i<=NF;i++){if($i==".agpr_count:")a=$(i+1); if($i==".vgpr_count:")v=$(i+1); if($i==".sgpr_count:")s=$(i+1); if($i==".vgpr_spill_count:")sp=$(i+1)} print "vgpr",v,"agpr",a,"sgpr",s,"spill",sp,n}' | c++filt | grep -E "$PAT"
