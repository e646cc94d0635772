#!/bin/bash
# kernel trace + SQ counter passes over profiles/embed_probe.py; prints the last forward per kernel.  usage: embed_pmc_pass.sh <outdir>
O=$PWD/${1:-gpurun_out/pmc}; R=$PWD; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PB_PROBE_REPS=3
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python3 $R/profiles/embed_probe.py > $O/kt.out 2> $O/kt.err
python3 $R/profiles/embed_layers.py $(find $O/kt -name kt_kernel_trace.csv) > $O/layers.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -o p -- python3 $R/profiles/embed_probe.py > /dev/null 2> $O/p1.err
python3 $R/profiles/pmc_last_forward.py $(find $O/p1 -name p_counter_collection.csv) > $O/pmc1.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/p2 -o p -- python3 $R/profiles/embed_probe.py > /dev/null 2> $O/p2.err
python3 $R/profiles/pmc_last_forward.py $(find $O/p2 -name p_counter_collection.csv) > $O/pmc2.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
rm -rf $O/kt $O/p1 $O/p2
cd $R
