#!/bin/bash
# usage: profiles/build_ablation.sh <name> <hipcc -D flags...>   -> pixelbox_amd/abl/libpixelbox_hip_<name>.so
# (pb_embed.hip recompiled with the flags, linked with the other objects of the in-tree build; probes take it with PIXELBOX_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p pixelbox_amd/abl
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -c pixelbox_amd/csrc/pb_embed.hip -o pixelbox_amd/abl/pb_embed_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pixelbox_amd/abl/libpixelbox_hip_$name.so pixelbox_amd/abl/pb_embed_$name.o \
  pixelbox_amd/csrc/pb_scan.o pixelbox_amd/csrc/pb_gemm_p3.o pixelbox_amd/csrc/pb_sharded.o pixelbox_amd/csrc/pb_phash.o -ldl
rm -f pixelbox_amd/abl/pb_embed_$name.o
echo pixelbox_amd/abl/libpixelbox_hip_$name.so
