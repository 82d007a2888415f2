set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "igemm5" 2>&1 | tail -3
bash tools/ab_builds.sh "-" "-DMBX_I5_NO_EARLY_REM" "-" "-DMBX_I5_NO_EARLY_REM" 2>&1 | grep "defs"
