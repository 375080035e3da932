#!/bin/bash
# round 4, call E: the traceback's sparse diagonal probes (H loaded at depths 4 / 8 / 12 / 16 of a run instead of at every cell):
# A/B against the committed tree (fill + traceback alone, then the bench's step with the traceback under the next fill),
# the whole -m gpu suite and two stress salts on the new walk
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_e
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
for round in 1 2; do for v in head sparse; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$v\", /" | tee -a "$O/ab.jsonl"
done; done
for round in 1 2; do for v in head sparse; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_${v}_$round.json" 2> "$O/bench_${v}_$round.err"
  python - <<PY
import json
d = json.loads(open("$O/bench_${v}_$round.json").readline()); r = d["roofline"]
print("$v", d["value"], d["ms_per_step"], r["avg_launch_ms"], d["kernel_ms"])
PY
done; done
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$O/pytest.log" 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
for salt in 2111 2222; do
  PG_SEED_SALT=$salt timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_general.py -m gpu -q -p no:cacheprovider > $O/salt_$salt.log 2>&1
  echo "salt $salt rc=$? $(tail -1 $O/salt_$salt.log)"
done
timeout 900 python tests/stress_parity.py 2000 1010 > $O/stress_1010.log 2>&1; echo "stress rc=$? $(tail -1 $O/stress_1010.log)"
