# rocprofv3 kernel statistics of the step in its data-parallel form on ONE rank (process group from the environment, no launcher:
# the profiler's preloaded library must not see an exec) beside the single-GPU form.  usage (through gpurun): bash tools/dp_trace.sh
set -o pipefail
out=gpurun_out/dp_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
A="--steps 10 --warmup 3 --no-cpu-baseline --no-detect --no-roofline --no-configs"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s -o s -- python3 bench.py $A > $out/single.json 2> $out/single.err || exit 2
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 MBX_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 ${DP_ENV:-MBX_DP_SEGMENTS=1 MBX_DP_TAIL_PARAMS=0}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/d -o d -- python3 bench.py $A > $out/dist.json 2> $out/dist.err || exit 3
python - <<PY
import csv, json
def load(p):
    r = {}
    for row in csv.DictReader(open(p)):
        n = row["Name"].split("(")[0].split("<")[0][-48:]
        r[n] = r.get(n, 0.0) + float(row["TotalDurationNs"]) / 1e6 / 13.0
    return r
a, b = load("$out/s/s_kernel_stats.csv"), load("$out/d/d_kernel_stats.csv")
print("ms per step (13 steps profiled): single %.3f dist %.3f" % (sum(a.values()), sum(b.values())))
for k in sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, 0) - a.get(k, 0)))[:14]:
    print("%-50s %8.3f %8.3f  %+.3f" % (k, a.get(k, 0), b.get(k, 0), b.get(k, 0) - a.get(k, 0)))
for f in ("single", "dist"):
    j = json.loads([l for l in open("$out/%s.json" % f) if l.startswith("{")][-1]); print(f, j["ms_per_step"])
PY
rm -rf $out/s $out/d
