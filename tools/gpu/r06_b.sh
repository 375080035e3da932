#!/bin/bash
# round 6, call b: what a fifth wavefront per SIMD buys the fill (tools/experiments/fill_lean_occupancy.py): the lean build
# (93 VGPRs, 5 KB of two-code profile) at 4 and at 5 wavefronts per SIMD (dynamic LDS padded), beside the production kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6b; mkdir -p $O
run() { # name lib lds
  local out; out=$(env ${2:+PG_LIB=$2} ${3:+PG_X_FILL_LDS=$3} python tools/fill_probe.py 200000 2>$O/err_$1.txt | tail -1)
  echo "{\"variant\": \"$1\", \"lds_per_wave\": \"${3:-}\", \"probe\": $out}" | tee -a $O/fill_occupancy_ab.jsonl
}
for rep in 1 2 3; do
  run production "" ""
  run full_two_code_profile_lds10240 tools/variants/lib_full2code.so 10240
  run lean_4_waves_lds10240 tools/variants/lib_lean.so 10240
  run lean_5_waves_lds8192 tools/variants/lib_lean.so 8192
  run lean_lds5120 tools/variants/lib_lean.so 5120
done
