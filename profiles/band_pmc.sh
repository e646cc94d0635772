#!/bin/bash
# usage: band_pmc.sh <outdir>  (PIXELBOX_LIB from the environment)
O=$PWD/$1; R=$PWD; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export PB_PROBE_REPS=2
timeout 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/p1 -o p -- python3 $R/profiles/embed_probe.py > /dev/null 2> $O/p1.err
python3 $R/profiles/pmc_last_forward.py $(find $O/p1 -name p_counter_collection.csv) > $O/pmc1.txt 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -o p -- python3 $R/profiles/embed_probe.py > /dev/null 2> $O/p2.err
python3 $R/profiles/pmc_last_forward.py $(find $O/p2 -name p_counter_collection.csv) > $O/pmc2.txt 2>&1
timeout 200 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum --output-format csv -d $O/p3 -o p -- python3 $R/profiles/embed_probe.py > /dev/null 2> $O/p3.err
python3 $R/profiles/pmc_last_forward.py $(find $O/p3 -name p_counter_collection.csv) > $O/pmc3.txt 2>&1
rm -rf $O/p1 $O/p2 $O/p3
cd $R
