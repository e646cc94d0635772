O=$PWD/gpurun_out/r03s; R=$PWD; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PB_PROBE_REPS=3 PB_PROBE_BATCH=1
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/profiles/embed_probe.py > $O/kt.out 2> $O/kt.err
python3 $R/profiles/embed_layers.py $(find $O/kt -name kt_kernel_trace.csv) 1 > $O/layers_b1.txt 2>&1
rm -rf $O/kt
cat $O/layers_b1.txt
