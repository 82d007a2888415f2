#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun; outputs under gpurun_out/prof_$1):
#   kernel stats of the default bench, FETCH_SIZE / WRITE_SIZE passes (separate --pmc runs, eager steps), per-layer table.
set -o pipefail
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $out/bench_default.json 2> $out/bench_default.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-detect --no-roofline > $out/bench_prof.json 2> $out/bench_prof.err || exit 2
python tools/trace_by_layer.py $out/kt/kt_kernel_trace.csv > $out/by_layer.txt 2>&1
cp $out/kt/kt_kernel_stats.csv $out/kernel_stats.csv
cp $out/by_layer.txt $out/by_layer_keep.txt; rm -f $out/kt/kt_kernel_trace.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pf -o f -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline --no-detect > $out/pmc_f.log 2>&1 || exit 3
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pw -o w -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-roofline --no-cpu-baseline --no-detect > $out/pmc_w.log 2>&1 || exit 4
python tools/pmc_traffic.py $out/pf/f_counter_collection.csv $out/pw/w_counter_collection.csv 3 $out/hbm_traffic_pmc.json > $out/pmc_traffic.txt 2>&1
rm -rf $out/pf $out/pw $out/kt
ls -la $out
# SQ counters of the grouped weight gradient (its own run: counters never share a run with other trace domains)
KB_ITERS=3 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $out/sq -o wg -- python3 tools/wgbench.py > $out/sq_wg.log 2>&1 && python tools/pmc_summary.py $out/sq/wg_counter_collection.csv > $out/wgrad_sq_counters.txt 2>&1
rm -rf $out/sq
python bench.py --input-size 512 --k 7 --max-num-bboxes 100 --no-cpu-baseline --no-detect > $out/bench_512.json 2> $out/bench_512.err
ls -la $out
