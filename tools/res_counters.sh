# SQ / LDS counters of the resident-image launches on block17's shapes (kbench driver).  usage (through gpurun): bash tools/res_counters.sh
set -o pipefail
out=gpurun_out/res_ctr
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export KB_ONLY=${KB_ONLY:-b17_} KB_NO_WGRAD=1 KB_CFG=${KB_CFG:-98} KB_ITERS=5
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS --output-format csv -d $out/a -o a -- python3 tools/kbench.py > $out/a.log 2>&1 || exit 2
python tools/pmc_summary.py $out/a/a_counter_collection.csv > $out/lds.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d $out/b -o b -- python3 tools/kbench.py > $out/b.log 2>&1 || exit 3
python tools/pmc_summary.py $out/b/b_counter_collection.csv > $out/sq.txt 2>&1
rm -rf $out/a $out/b
grep -A 12 "conv_resident" $out/lds.txt | head -60; grep -A 12 "conv_resident" $out/sq.txt | head -60
