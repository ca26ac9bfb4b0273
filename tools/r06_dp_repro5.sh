# failure rate of the 4-rank shared-GPU run (cfl pcd 4096 / L K from env), default spin limit vs 2^30
for lim in 0 1073741824; do
  bad=0
  for i in $(seq 1 ${N:-10}); do
    r=$(CFL_DEBUG_SPIN_LIMIT=$lim STEPS=4 python tools/r06_dp_repro4.py ${W:-4} 2>&1 | grep "NaN count" | sed 's/.*one-shot engine \([0-9]*\) .*, \([0-9.na]*\)\]$/\1:\2/' | tr '\n' ' ')
    case "$r" in *"nan"*|*":1.0"*|*":2.0"*|*":3.0"*|*":4.0"*) bad=$((bad+1)); echo "  limit $lim run $i: $r";; esac
  done
  echo "limit $lim: $bad of ${N:-10} runs with an error word or NaN"
done
