# 4 ranks sharing the GPU, weight-norm hand-off shape (cfl pcd D=4096 L=36 K=4): error words / NaN and wall time per run, 60 s bound
bad=0
for i in $(seq 1 ${N:-6}); do
  s=$(date +%s)
  r=$(STEPS=4 L=36 K=4 python tools/r06_dp_repro4.py 4 2>&1 | grep "NaN count" | sed 's/.*one-shot engine \([0-9]*\) .*, \([0-9.na]*\)\]$/\1:\2/' | tr '\n' ' ')
  e=$(date +%s)
  case "$r" in *"nan"*|*":1.0"*|*":2.0"*|*":3.0"*|*":4.0"*) bad=$((bad+1));; esac
  echo "run $i: $((e-s)) s  NaN count:error word per step = $r"
done
echo "$bad of ${N:-6} runs with an error word or NaN"
