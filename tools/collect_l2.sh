#!/bin/bash
# L2 (TCC) counters of the grouped weight-gradient launch: its own short rocprofv3 pass under a timeout (a TCC pass has hung
# inside rocprofv3 on this pool before); output under gpurun_out/prof_$1.
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--steps 1 --warmup 1 --no-graph --no-roofline --no-cpu-baseline --no-detect --no-configs"
echo "tcc pass start" > $out/wg2.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr --output-format csv -d $out/wg2 -o s -- python3 bench.py $B >> $out/wg2.log 2>&1
echo "rc $?" >> $out/wg2.log
python tools/pmc_summary.py $out/wg2/s_counter_collection.csv | grep -A 6 "conv_wgrad_grouped" > $out/wgrad_l2_counters.txt 2>&1
rm -rf $out/wg2
tail -3 $out/wg2.log; cat $out/wgrad_l2_counters.txt
