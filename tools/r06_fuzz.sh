# round 6: the randomised sweeps of earlier rounds against the final library (new seeds)
mkdir -p gpurun_out
for job in "fuzz_parity.py 150 606" "fuzz_conv.py 150 606" "fuzz_halo.py 100 606" "fuzz_gan.py 20 606"; do
  set -- $job
  echo "== $job" >> gpurun_out/r06_fuzz.txt
  timeout 900 python tools/$1 $2 $3 2>&1 | tail -6 >> gpurun_out/r06_fuzz.txt
done
cat gpurun_out/r06_fuzz.txt
