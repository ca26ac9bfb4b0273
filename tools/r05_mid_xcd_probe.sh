# (the CFL_DEBUG_MID_XCD switch this script drove was removed after the measurement: commit history, profiles/r05_mid_xcd_ab.txt)
# round 5, VERDICT item 3: row tiles dealt over the XCDs by proj (CFL_DEBUG_NOXCD=1) + mid taking rows from its own XCD's L2
# (CFL_DEBUG_MID_XCD=1), against the default (d slices dealt over the XCDs).  Same box, interleaved, bench medians.
run() {
  env $2 python bench.py --timed-seconds 1.5 --no-other-configs --no-cpu-baseline --no-cli-loop --no-dp-form $3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('%-46s %8.3f us/step   proj %.2f mid %.2f grad %.2f' % ('$1', 1e3*d['ms_per_step'], k['proj']['avg_us'], k['mid']['avg_us'], k['grad']['avg_us']))"
}
for i in 1 2 3; do
run "default (d slices per XCD)" "CFL_X=0" "$@"
run "row tiles per XCD" "CFL_DEBUG_NOXCD=1" "$@"
run "row tiles per XCD + mid on the slab XCD" "CFL_DEBUG_NOXCD=1 CFL_DEBUG_MID_XCD=1" "$@"
done
