#!/bin/bash
# Same-box A/B of the burst collect kernel's two step forms: the default (counted arrivals, no barrier in the loop) against PB_MQ_BARRIER=1
# (a workgroup barrier per step), kernel times of every k_scan_multi_wg launch of profiles/mq_probe.py under rocprofv3 (the first is the
# 128-query warm-up call).  usage (GPU box, repo root): bash profiles/mq_ab.sh
R=$PWD; O=gpurun_out/ab; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
for v in relaxed barrier; do
  if [ $v = relaxed ]; then unset PB_MQ_BARRIER; else export PB_MQ_BARRIER=1; fi
  rm -rf $R/$O/$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$v -o t -- python3 $R/profiles/mq_probe.py > $R/$O/$v.txt 2>&1
  f=$(find $R/$O/$v -name t_kernel_trace.csv | head -1)
  python3 -c "
import csv, sys
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'k_scan_multi_wg' in r['Kernel_Name']]
print('$v', ' '.join('%.0f'%x for x in d), 'us')
" $f
  grep burst $R/$O/$v.txt | tail -1
  rm -rf $R/$O/$v
done; done
