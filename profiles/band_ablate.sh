#!/bin/bash
# runs profiles/band_probe.py (forced band kernel only) against every library under pixelbox_amd/abl/ and the in-tree one;
# prints the per-block timings of the band kernel (bands 1, batch 512) from the tuner trace
out=${1:-gpurun_out/abl}
mkdir -p $out
for lib in pixelbox_amd/libpixelbox_hip.so pixelbox_amd/abl/*.so; do
  name=$(basename $lib .so | sed 's/libpixelbox_hip_\?//'); name=${name:-base}
  PIXELBOX_LIB=$PWD/$lib PB_PROBE_VARIANTS=forced PB_TRACE_TUNE=1 timeout 300 python profiles/band_probe.py > $out/$name.txt 2> $out/$name.err
  echo "== $name: $(grep 'forced:' $out/$name.txt)"
  grep -E "n512.*band kernel, bands 1 " $out/$name.err | sed 's/front //; s/LDS-ring band kernel, //; s/(separate.*//' | sort -u | awk '{a[$1" "$2" "$3]=a[$1" "$2" "$3]" "$(NF-1)} END{for(k in a) print "   "k": "a[k]}' | sort
done
