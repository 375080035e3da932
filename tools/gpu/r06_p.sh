#!/bin/bash
# round 6, call p: the fill's recurrence at 1 - 8 wavefronts per SIMD (what a fifth / sixth wavefront could buy); the path tests with the index made on the device
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6p; mkdir -p $O
for i in 1 2 3; do tools/ubench/valu_rate occupancy 3000 > $O/occupancy_$i.json; done
tail -12 $O/occupancy_3.json
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_path") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
