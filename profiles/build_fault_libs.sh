#!/bin/bash
# Fault-injection builds of the library for the certificate's rows-seen / tile-sum clauses (run HERE: hipcc cross-compiles):
#   build/libpb_fault_skip.so    -DPB_FAULT_SKIP_TILE=60000    one tile is evaluated but counted as nobody's (round 4's check)
#   build/libpb_fault_double.so  -DPB_FAULT_DOUBLE_TILE=60000  tile 60001 is never read, tile 60000 is read twice: the row counts still add up
# Then on the GPU box: PIXELBOX_LIB=build/libpb_fault_double.so python tests/_stress_one_query.py 40   (every form must report
# "0 of 40 calls differ from the oracle; certified 0/40": the certificate refuses, the exhaustive pass answers).
set -e
cd "$(dirname "$0")/.."
mkdir -p build
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
C=pixelbox_amd/csrc
for v in skip:PB_FAULT_SKIP_TILE double:PB_FAULT_DOUBLE_TILE; do
  name=${v%%:*}; def=${v##*:}
  hipcc $F -D$def=60000 -c $C/pb_scan.hip -o build/pb_scan_fault_$name.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o build/libpb_fault_$name.so build/pb_scan_fault_$name.o $C/pb_embed.o $C/pb_gemm_p3.o $C/pb_sharded.o $C/pb_phash.o -ldl
done
ls -la build/libpb_fault_*.so
