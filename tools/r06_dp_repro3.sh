run() {  # world, case, env...
  w=$1; c=$2; shift 2
  echo "world $w $c $*: $(env "$@" FUZZ_CASES="[$c]" timeout 300 python tools/fuzz_dp_ranks.py $w 1 1 2>&1 | grep "RuntimeError:\|dp ranks\|FAIL" | head -1 | cut -c1-230)"
}
C="('cfl', 'pcd', 4096, 36, 4, 256, {}, False, 5)"
run 4 "$C" X=1
run 4 "$C" CFL_DP_PUSH_SEPARATE=1
run 4 "$C" CFL_DP_SPLIT_ADAM=1
run 4 "$C" CFL_DP_PUSH_SEPARATE=1 CFL_DP_SPLIT_ADAM=1
run 3 "('cfl', 'pcd', 4096, 36, 4, 192, {}, False, 5)" X=1
run 4 "('cfl', 'pcd', 4096, 20, 3, 256, {}, False, 5)" X=1
run 4 "('cfl', 'pcd', 4096, 16, 3, 256, {}, False, 5)" X=1
run 4 "('cfl', 'pcd', 2048, 20, 5, 256, {}, False, 5)" X=1
run 4 "('cfl', 'monomer', 4096, 20, 3, 256, {}, False, 5)" X=1
run 8 "('cfl', 'pcd', 4096, 20, 3, 512, {}, False, 5)" X=1
