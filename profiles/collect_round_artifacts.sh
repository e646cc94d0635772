#!/bin/bash
# Regenerates the round's evidence on a GPU box (run from the repo root: bash profiles/collect_round_artifacts.sh):
# GPU tests, bench.py plain and under rocprofv3 --kernel-trace --stats, separate --pmc passes (FETCH_SIZE, WRITE_SIZE,
# MFMA busy, SQ wave-cycle split) summarised on the box, embed error and mlhash latency.  Outputs: gpurun_out/r01z/;
# the summaries are then copied to profiles/r01_*.
set -x
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r01z; mkdir -p $O
cd $R && python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/gpu_tests.txt
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/full -o full -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/full.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o m -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2> $O/pmc_mfma.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o s -- python3 $R/bench.py --steps 2 --warmup 1 > /dev/null 2> $O/pmc_sq.err
python3 $R/profiles/summarize_pmc.py $O/scan_pmc.json FETCH_SIZE=$O/pmc_fetch/f_counter_collection.csv WRITE_SIZE=$O/pmc_write/w_counter_collection.csv > $O/summarize_pmc.txt 2>&1
python3 $R/profiles/summarize_mfma.py $O/mfma_pmc.json $O/pmc_mfma/m_counter_collection.csv > $O/summarize_mfma.txt 2>&1
python3 $R/profiles/summarize_sq.py $O/sq_pmc.json $O/pmc_sq/s_counter_collection.csv > $O/summarize_sq.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_sq
cd $R
python3 profiles/embed_error.py 64 > $O/embed_error.txt 2>&1
python3 profiles/mlhash_latency.py > $O/mlhash_latency.txt 2>&1
du -sh $O; ls $O
