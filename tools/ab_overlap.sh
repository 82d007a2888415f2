set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "pair or igemm5 or dgrad" 2>&1 | tail -4
MBX_DETERMINISTIC=1 python tools/overlap_check.py 96 2 16 2>&1 | tail -2
bash tools/ab_env.sh - "MBX_CONV_PAIR=0" | grep rep
python tools/step_trace.py gpurun_out/step_trace_r4g.tsv 2>/dev/null | head -8
