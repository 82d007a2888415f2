# A/B of environment knobs on ONE box: tools/ab_env.sh "<VAR=val ...>" "<...>" ...   ("-" = defaults); two repetitions each
cd "$(dirname "$0")/.."
B="--steps 30 --warmup 5 --no-cpu-baseline --no-detect --no-roofline --no-configs"
for rep in 1 2; do
for envs in "$@"; do
  e="$envs"; [ "$e" = "-" ] && e=""
  env $e python bench.py $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s] rep $rep: %.3f ms/step %.1f img/s' % ('$envs', j['ms_per_step'], j['value']))"
done
done
