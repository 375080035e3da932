#!/bin/bash
# round 4: parity beyond the committed seeds on the tree as it is: the -m gpu suite, three salts of the fuzz tests (every
# stage), two stress seeds against the reference's gssw.c, and the probes (read lengths, long nodes, cascade stages)
# usage: r04_validate.sh [salt ...]   (default 3111 3222 3333)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_validate
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
SALTS=${*:-3111 3222 3333}
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$O/pytest.log" 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for salt in $SALTS; do
  PG_SEED_SALT=$salt timeout 900 python -m pytest tests/test_gpu_klib.py tests/test_gpu_kmer.py tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_path.py tests/test_gpu_general.py -m gpu -q -p no:cacheprovider > $O/salt_$salt.log 2>&1
  echo "salt $salt rc=$? $(tail -1 $O/salt_$salt.log)"
done
for seed in 1111 1212; do
  timeout 900 python tests/stress_parity.py 2000 $seed > $O/stress_$seed.log 2>&1; echo "stress $seed rc=$? $(tail -1 $O/stress_$seed.log)"
done
timeout 600 python tools/readlen_probe.py > "$O/readlen_probe.json" 2> "$O/readlen_probe.err"; echo "readlen rc=$?"
timeout 300 python tools/config5_probe.py > "$O/config5_probe.json" 2> "$O/config5_probe.err"; echo "config5 rc=$?"
timeout 300 python tools/stage_probe.py > "$O/stage_probe.json" 2> "$O/stage_probe.err"; echo "stage rc=$?"
