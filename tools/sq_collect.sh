#!/bin/bash
# SQ counters of the fill kernel (separate rocprofv3 passes, --kernel-trace only).  Runs on the GPU box via gpurun.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq
mkdir -p "$OUT"
echo "${PG_HEAD:-unknown}" > "$OUT/head.txt"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/set$i" -o sq -- python $R/bench.py --steps 1 --warmup 0 --reads 200000 --no-cpu-baseline --stream-batches 0 --sites-steps 0 --e2e-steps 0 --collective off --exact-shortcut-steps 0 --plain-steps 0 --config5-graphs 0 > /dev/null 2> "$OUT/set$i.err"
done
find "$OUT" -name "*counter_collection.csv"
