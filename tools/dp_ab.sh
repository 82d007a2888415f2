# A/B of the data-parallel form of the step on ONE rank (MBX_FORCE_DIST=1: process group, buckets, graphs per bucket) against the
# single-GPU form, same box.  usage (through gpurun): bash tools/dp_ab.sh
set -o pipefail
run() { # $1 = tag, rest = env
  env "${@:2}" HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --no-cpu-baseline --no-detect --no-roofline --no-configs 2>gpurun_out/dp_$1.err | grep "^{" > gpurun_out/dp_$1.json
  python - <<PY
import json
j=json.loads(open("gpurun_out/dp_$1.json").read().strip().splitlines()[-1])
print("$1", j["ms_per_step"], j.get("kernels",{}).get("backward_segments"), j.get("kernels",{}).get("graphs"), j.get("data_parallel",{}).get("allreduce_exposed_ms"))
PY
}
single() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-detect --no-roofline --no-configs 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single', j['ms_per_step'])"; }
single
run seg1notail MBX_FORCE_DIST=1 MBX_DP_SEGMENTS=1 MBX_DP_TAIL_PARAMS=0
run seg1notail_noreduce MBX_FORCE_DIST=1 MBX_DP_SEGMENTS=1 MBX_DP_TAIL_PARAMS=0 MBX_DP_NO_ALLREDUCE=1
run seg2 MBX_FORCE_DIST=1 MBX_DP_SEGMENTS=2
run seg2_noreduce MBX_FORCE_DIST=1 MBX_DP_SEGMENTS=2 MBX_DP_NO_ALLREDUCE=1
single
