#!/bin/bash
# Same-box A/B of the burst collect kernel: the in-tree library against another build of pb_scan.hip (profiles/build_scan_ablation.sh <name> [-D...]
# from a tree with the variant applied -> pixelbox_amd/abl/libpixelbox_hip_scan_<name>.so), kernel times of every k_scan_multi_wg launch of
# profiles/mq_probe.py under rocprofv3, two rounds (the first launch of a run is the 128-query warm-up call).  This is how round 6's
# scalar-address and barrier-free variants were held against the product (profiles/r06_burst_collect.txt).
# usage (GPU box, repo root): bash profiles/mq_ab.sh <name>
N=${1:-base}; R=$PWD; O=gpurun_out/ab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
for v in product $N; do
  if [ $v = product ]; then unset PIXELBOX_LIB; else export PIXELBOX_LIB=$R/pixelbox_amd/abl/libpixelbox_hip_scan_$v.so; [ -f $PIXELBOX_LIB ] || { echo "no $PIXELBOX_LIB"; continue; }; fi
  rm -rf $R/$O/$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$v -o t -- python3 $R/profiles/mq_probe.py > $R/$O/$v.txt 2>&1
  f=$(find $R/$O/$v -name t_kernel_trace.csv | head -1)
  python3 -c "
import csv, sys
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'k_scan_multi_wg' in r['Kernel_Name']]
print('$v', ' '.join('%.0f'%x for x in d), 'us')
" $f
  rm -rf $R/$O/$v
done; done
