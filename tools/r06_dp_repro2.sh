C1="[('cfl', 'pcd', 4096, 36, 4, 4, {'reg_const': 0.0001}, False, 5)]"
for w in 4 2; do
  B=$w
  FUZZ_CASES="[('cfl', 'pcd', 4096, 36, 4, $B, {'reg_const': 0.0001}, False, 5)]" timeout 300 python tools/fuzz_dp_ranks.py $w 1 1 2>&1 | grep "RuntimeError:\|dp ranks\|FAIL" | head -3
  FUZZ_CASES="[('cfl', 'pcd', 4096, 36, 4, $B, {}, False, 5)]" timeout 300 python tools/fuzz_dp_ranks.py $w 1 1 2>&1 | grep "RuntimeError:\|dp ranks\|FAIL" | head -3
  FUZZ_CASES="[('dist', 'pcd', 4096, 36, 4, $B, {}, False, 5)]" timeout 300 python tools/fuzz_dp_ranks.py $w 1 1 2>&1 | grep "RuntimeError:\|dp ranks\|FAIL" | head -3
  FUZZ_CASES="[('cfl', 'pcd', 4096, 36, 4, $((64*w)), {}, False, 5)]" timeout 300 python tools/fuzz_dp_ranks.py $w 1 1 2>&1 | grep "RuntimeError:\|dp ranks\|FAIL" | head -3
done
