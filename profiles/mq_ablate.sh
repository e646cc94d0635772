#!/bin/bash
# The collect kernel of a 1 024-query burst (k_scan_multi_wg) under rocprofv3 --kernel-trace --stats: the product build and the timing-only
# ablations of profiles/build_scan_ablation.sh (abl1: no survivor tests; abl3: no tests and no LDS operand reads) -- what the tests and the
# operand reads cost beside the MFMAs.  usage (GPU box, repo root): bash profiles/mq_ablate.sh [out dir]
O=${1:-gpurun_out/mq_ablate}; R=$PWD; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in product abl1 abl3; do
  if [ $v = product ]; then unset PIXELBOX_LIB; else export PIXELBOX_LIB=$R/pixelbox_amd/abl/libpixelbox_hip_scan_$v.so; [ -f $PIXELBOX_LIB ] || continue; fi
  rm -rf $R/$O/$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$v -o t -- python3 $R/profiles/mq_probe.py > $R/$O/$v.txt 2>&1
  f=$(find $R/$O/$v -name t_kernel_stats.csv | head -1)
  echo "== $v"; python3 -c "
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_scan_multi_wg' in r['Name']: print('k_scan_multi_wg calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3), 'min us %.1f' % (float(r['MinNs']) / 1e3))
" $f
  rm -rf $R/$O/$v
done
