set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_postproc.py tests/test_gpu_nnops.py tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -4
python tools/step_trace.py gpurun_out/step_trace_r4h.tsv 2>/dev/null | grep -i "match\|avgpool\|step wall"
bash tools/ab_env.sh - | grep rep
