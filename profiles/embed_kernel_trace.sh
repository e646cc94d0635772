#!/bin/bash
# kernel trace of profiles/embed_probe.py (no counter passes); prints the last forward per kernel.  usage: embed_kernel_trace.sh <outdir>
O=$PWD/${1:-gpurun_out/kt}; R=$PWD; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PB_PROBE_REPS=3
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/profiles/embed_probe.py > $O/kt.out 2> $O/kt.err
python3 $R/profiles/embed_layers.py $(find $O/kt -name kt_kernel_trace.csv) > $O/layers.txt 2>&1
rm -rf $O/kt
cd $R
