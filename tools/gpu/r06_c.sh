#!/bin/bash
# round 6, call c: partial prefetch (first 2 / 4 profile rows a step ahead) at 4 and 5 wavefronts per SIMD
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6c; mkdir -p $O
run() { # name lib lds
  local out; out=$(env ${2:+PG_LIB=$2} ${3:+PG_X_FILL_LDS=$3} python tools/fill_probe.py 200000 2>$O/err_$1.txt | tail -1)
  echo "{\"variant\": \"$1\", \"lds_per_wave\": \"${3:-}\", \"probe\": $out}" | tee -a $O/fill_occupancy_ab2.jsonl
}
for rep in 1 2 3; do
  run full_two_code_profile_lds10240 tools/variants/lib_full2code.so 10240
  run partial2_4_waves tools/variants/lib_leanp2.so 10240
  run partial2_5_waves tools/variants/lib_leanp2.so 5120
  run partial4_4_waves tools/variants/lib_leanp4.so 10240
  run partial4_5_waves tools/variants/lib_leanp4.so 5120
  run lean_5_waves tools/variants/lib_lean.so 5120
done
