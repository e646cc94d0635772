#!/bin/bash
# usage: profiles/build_scan_ablation.sh <name> <hipcc -D flags...>   -> pixelbox_amd/abl/libpixelbox_hip_scan_<name>.so
# (pb_scan.hip recompiled with the flags -- PB_MQ_ABL=1/2/3: the collect kernel without tests / MFMAs / tests and LDS operand reads,
# PB_MQ_STAMP: its per-phase clocks -- linked with the other objects of the in-tree build; probes take it with PIXELBOX_LIB=...;
# results of the ABL builds are INVALID, timing only)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p pixelbox_amd/abl
/opt/rocm/bin/hipcc "$@" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -c pixelbox_amd/csrc/pb_scan.hip -o pixelbox_amd/abl/pb_scan_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pixelbox_amd/abl/libpixelbox_hip_scan_$name.so pixelbox_amd/abl/pb_scan_$name.o \
  pixelbox_amd/csrc/pb_embed.o pixelbox_amd/csrc/pb_gemm_p3.o pixelbox_amd/csrc/pb_sharded.o pixelbox_amd/csrc/pb_phash.o -ldl
rm -f pixelbox_amd/abl/pb_scan_$name.o
echo pixelbox_amd/abl/libpixelbox_hip_scan_$name.so
