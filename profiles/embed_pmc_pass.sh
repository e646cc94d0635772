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
# pass 3 (round 6): the instruction mix the issue-floor table needs -- matrix instructions by input type, transcendentals, scalar
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -o p -- python3 $R/profiles/embed_probe.py > /dev/null 2> $O/p3.err
python3 $R/profiles/pmc_last_forward.py $(find $O/p3 -name p_counter_collection.csv) > $O/pmc3.txt 2>&1
python3 $R/profiles/issue_floor.py $O/layers.txt $O/pmc1.txt $O/pmc2.txt $O/pmc3.txt > $O/issue_floor.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
rm -rf $O/kt $O/p1 $O/p2 $O/p3
cd $R
