# round 6 final: whole GPU suite, smoke, default bench line
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()"
time python bench.py > gpurun_out/r06_g_bench.json 2> gpurun_out/r06_g_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r06_g_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric', 'value', 'ms_per_step', 'n_gpus', 'steps')})
print(d['roofline'])
print(d['cpu_baseline'])
print({k: (v.get('us_per_step') or v.get('ms_per_step')) for k, v in d['other_configs'].items() if isinstance(v, dict)})
print(d['dp_form'])
PY
