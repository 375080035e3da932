#!/bin/bash
# round 4, call F: the tree with the stages' length limits as skips and the bisecting SiteBatcher: whole -m gpu suite, three
# salts of the fuzz tests, two stress seeds, the wide / long-node probes on the integer-addition kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_f
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$O/pytest.log" 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for salt in 3111 3222 3333; do
  PG_SEED_SALT=$salt timeout 900 python -m pytest tests/test_gpu_klib.py tests/test_gpu_kmer.py tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_path.py tests/test_gpu_general.py -m gpu -q -p no:cacheprovider > $O/salt_$salt.log 2>&1
  echo "salt $salt rc=$? $(tail -1 $O/salt_$salt.log)"
done
for seed in 1111 1212; do
  timeout 900 python tests/stress_parity.py 2000 $seed > $O/stress_$seed.log 2>&1; echo "stress $seed rc=$? $(tail -1 $O/stress_$seed.log)"
done
timeout 600 python tools/readlen_probe.py > "$O/readlen_probe.json" 2> "$O/readlen_probe.err"; echo "readlen rc=$?"; tail -3 "$O/readlen_probe.json" | cut -c1-600
timeout 300 python tools/config5_probe.py > "$O/config5_probe.json" 2> "$O/config5_probe.err"; echo "config5 rc=$?"; tail -2 "$O/config5_probe.json" | cut -c1-400
timeout 300 python tools/stage_probe.py > "$O/stage_probe.json" 2> "$O/stage_probe.err"; echo "stage rc=$?"; tail -2 "$O/stage_probe.json" | cut -c1-600
