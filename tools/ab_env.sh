# same-box A/B of one environment switch: bash tools/ab_env.sh VAR=VALUE [bench args]
kv=$1; shift
for i in 1 2 3; do
for mode in on off; do
if [ $mode = on ]; then pre=""; else pre="$kv"; fi
env $pre python bench.py --no-other-configs --no-cpu-baseline --no-cli-loop "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', '(default)' if '$mode'=='on' else '$kv', d['ms_per_step'], d['roofline']['kernels'])"
done; done
