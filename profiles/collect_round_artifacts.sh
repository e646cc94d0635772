#!/bin/bash
# Regenerates the round's evidence on a GPU box (run from the repo root: bash profiles/collect_round_artifacts.sh r02):
# GPU tests, bench.py plain and under rocprofv3 --kernel-trace --stats, separate --pmc passes (FETCH_SIZE, WRITE_SIZE for the
# scan launches and for the embed forward; MFMA busy and the SQ wave-cycle split for the embed forward), summarised on the
# box.  Outputs: gpurun_out/<tag>/; the summaries are then copied to profiles/<tag>_*.
# PMC passes profile SHORT commands (scan legs only / profiles/embed_probe.py): a counter pass over the whole bench, with
# its thousands of dispatches (1M-image end-to-end leg, tuning loops), takes tens of minutes.  Every command runs under `timeout`
# (a hung command would otherwise hold the box until gpurun's own limit).
set -x
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O
# the plain bench FIRST, on a device that streams at its idle rate (wait_quiet.py: a box can arrive, or be left by our own test suite, with
# minutes of driver scrubbing ahead of it); then the tests; then the profiled runs, each behind the same wait
cd /tmp; export TMPDIR=/tmp
run_tests() { cd $R && timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests_full.txt 2>&1; grep -E "passed|failed" $O/gpu_tests_full.txt > $O/gpu_tests.txt; cd /tmp; }
TESTS_DONE=0
timeout 400 python3 $R/profiles/wait_quiet.py > $O/wait_quiet.txt 2>&1
if [ $? -ne 0 ]; then  # a box that streams 2-3 % under its idle rate for minutes on end (seen: 250 s at 0.876, then 0.902 for the rest of the hour): tests first, then ask again
  run_tests; TESTS_DONE=1
  timeout 400 python3 $R/profiles/wait_quiet.py >> $O/wait_quiet.txt 2>&1
fi
T0=$SECONDS; timeout 900 python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo "plain bench.py: $((SECONDS - T0)) s wall" > $O/bench_wall.txt
[ $TESTS_DONE = 1 ] || run_tests
timeout 400 python3 $R/profiles/wait_quiet.py >> $O/wait_quiet.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/full -o full -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/full.err
# the headline legs alone (default steps): the LOOPQ filter instance's average in this stats file is the 10M-row launches only
timeout 400 python3 $R/profiles/wait_quiet.py >> $O/wait_quiet.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/head -o head -- python3 $R/bench.py --no-embed --e2e-images 0 --no-sweep --no-cpu-baseline > $O/bench_headline_under_rocprof.json 2> $O/head.err
cp $(find $O/head -name "head_kernel_stats.csv") $O/headline_kernel_stats.csv; rm -rf $O/head
SCAN="--no-embed --e2e-images 0 --no-sweep --no-cpu-baseline --steps 2 --warmup 1"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py $SCAN > /dev/null 2> $O/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py $SCAN > /dev/null 2> $O/pmc_write.err
python3 $R/profiles/summarize_pmc.py $O/scan_pmc.json FETCH_SIZE=$(find $O/pmc_fetch -name f_counter_collection.csv) WRITE_SIZE=$(find $O/pmc_write -name w_counter_collection.csv) > $O/summarize_pmc.txt 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/epmc_fetch -o f -- python3 $R/profiles/embed_probe.py > /dev/null 2> $O/epmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/epmc_write -o w -- python3 $R/profiles/embed_probe.py > /dev/null 2> $O/epmc_write.err
python3 $R/profiles/summarize_embed_pmc.py $O/embed_pmc.json FETCH_SIZE=$(find $O/epmc_fetch -name f_counter_collection.csv) WRITE_SIZE=$(find $O/epmc_write -name w_counter_collection.csv) > $O/summarize_embed_pmc.txt 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o m -- python3 $R/profiles/embed_probe.py > /dev/null 2> $O/pmc_mfma.err
python3 $R/profiles/summarize_mfma.py $O/mfma_pmc.json $(find $O/pmc_mfma -name m_counter_collection.csv) > $O/summarize_mfma.txt 2>&1
rm -rf $O/pmc_fetch $O/pmc_write $O/epmc_fetch $O/epmc_write $O/pmc_mfma
cp $(find $O/full -name "full_kernel_stats.csv") $O/full_kernel_stats.csv; rm -rf $O/full
cd $R
timeout 300 python3 profiles/embed_error.py 64 > $O/embed_error.txt 2>&1
timeout 300 python3 profiles/mlhash_latency.py > $O/mlhash_latency.txt 2>&1
timeout 300 python3 profiles/single_call_probe.py > $O/single_call.txt 2>&1
timeout 300 python3 profiles/exact_probe.py > $O/exact_probe.txt 2>&1
timeout 200 bash profiles/embed_kernel_trace.sh gpurun_out/$TAG/layers512 > /dev/null 2>&1; cp $O/layers512/layers.txt $O/embed_layers.txt
timeout 200 bash profiles/embed_batch_trace.sh 1 gpurun_out/$TAG/layers1 > /dev/null 2>&1; cp $O/layers1/layers_b1.txt $O/embed_layers_batch1.txt
timeout 300 python3 profiles/embed_f64.py 512 bench > $O/embed_f64.txt 2>&1; timeout 300 python3 profiles/embed_f64.py 512 parity >> $O/embed_f64.txt 2>&1
# round 6: per-dispatch SQ counters of the last forward (three --pmc passes) + the issue-floor table built from them; the layer table of the
# same run with the counters appended is the round's embed_layers.txt
timeout 1500 bash profiles/embed_pmc_pass.sh gpurun_out/$TAG/pmc_layers > /dev/null 2>&1
{ cat $O/pmc_layers/layers.txt; echo; echo "# SQ counters per dispatch of the same forward (separate --pmc passes; profiles/embed_pmc_pass.sh, profiles/pmc_last_forward.py)"; cat $O/pmc_layers/pmc1.txt; echo; cat $O/pmc_layers/pmc2.txt; echo; cat $O/pmc_layers/pmc3.txt; } > $O/embed_layers.txt
cp $O/pmc_layers/issue_floor.txt $O/embed_issue_floor.txt
timeout 300 python3 profiles/dual_probe.py 2>&1 | grep batch > $O/dual.txt
timeout 200 ./build/block_small_bench p3 > $O/block_small.txt 2>&1
# round 6: the burst collect kernel -- ablation builds (pixelbox_amd/abl/libpixelbox_hip_scan_abl{1,3}.so from profiles/build_scan_ablation.sh, if present)
# and its per-dispatch counters
timeout 600 bash profiles/mq_ablate.sh gpurun_out/$TAG/mq_ablate > $O/burst_ablate.txt 2>&1
timeout 600 bash profiles/mq_pmc_pass.sh gpurun_out/$TAG/mq_pmc > /dev/null 2>&1; cp $O/mq_pmc/mq_issue_account.txt $O/burst_issue_account.txt
du -sh $O; ls $O
