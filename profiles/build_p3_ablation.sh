#!/bin/bash
# usage: profiles/build_p3_ablation.sh <name> <-DPB_P3_ABL=bits>  -> pixelbox_amd/abl/libpixelbox_hip_p3_<name>.so  (pb_gemm_p3.hip recompiled with the
# flags, linked with the other objects of the in-tree build; probes take it with PIXELBOX_LIB=...; results of these builds are INVALID, timing only)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p pixelbox_amd/abl
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -c pixelbox_amd/csrc/pb_gemm_p3.hip -o pixelbox_amd/abl/pb_gemm_p3_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pixelbox_amd/abl/libpixelbox_hip_p3_$name.so pixelbox_amd/abl/pb_gemm_p3_$name.o \
  pixelbox_amd/csrc/pb_scan.o pixelbox_amd/csrc/pb_embed.o pixelbox_amd/csrc/pb_sharded.o pixelbox_amd/csrc/pb_phash.o -ldl
rm -f pixelbox_amd/abl/pb_gemm_p3_$name.o
echo pixelbox_amd/abl/libpixelbox_hip_p3_$name.so
