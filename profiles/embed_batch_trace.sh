#!/bin/bash
# per-kernel table of the last forward at another batch size: usage embed_batch_trace.sh <batch> <outdir>
B=${1:-128}; O=$PWD/${2:-gpurun_out/bt}; R=$PWD; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PB_PROBE_REPS=3 PB_PROBE_BATCH=$B
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/profiles/embed_probe.py > $O/kt.out 2> $O/kt.err
python3 $R/profiles/embed_layers.py $(find $O/kt -name kt_kernel_trace.csv) $B > $O/layers_b$B.txt 2>&1
rm -rf $O/kt
cat $O/layers_b$B.txt
