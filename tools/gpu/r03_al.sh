#!/bin/bash
# round 3, call AL: final kernels -- kernel stats of the sites workload (config 3), three more salts and two more stress seeds
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_al
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o c3 -- python $R/bench.py --workload config3 --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench_config3.json" 2> "$O/prof.err"
echo "rc=$?"
head -6 "$O/prof/c3_kernel_stats.csv" | cut -c1-150
python - <<PY
import json
d = json.loads(open("$O/bench_config3.json").readline())
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"])
PY
for salt in 1377 1488 1599; do
  PG_SEED_SALT=$salt timeout 900 python -m pytest tests/test_gpu_klib.py tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_path.py tests/test_gpu_general.py -m gpu -q -p no:cacheprovider > $O/salt_$salt.log 2>&1
  echo "salt $salt rc=$? $(tail -1 $O/salt_$salt.log)"
done
for seed in 808 909; do
  timeout 900 python tests/stress_parity.py 2000 $seed > $O/stress_$seed.log 2>&1; echo "stress $seed rc=$? $(tail -1 $O/stress_$seed.log)"
done
