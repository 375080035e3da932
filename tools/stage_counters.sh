#!/bin/bash
# Counters of the kernels besides the fill / traceback (seed stages, klib kernels, count path, cascade hand-over): FETCH_SIZE and
# WRITE_SIZE in passes of their own (with the 1 GiB calibration kernels, as tools/pmc_collect.sh), SQ sets, and a plain
# --kernel-trace --stats pass for the durations.  rocprofv3 with --kernel-trace only.  Runs on the GPU box via gpurun.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/stage_counters
N=${1:-200000}
mkdir -p "$OUT"
echo "${PG_HEAD:-unknown}" > "$OUT/head.txt"
cd /tmp && export TMPDIR=/tmp
RUN="python $R/tools/stage_counters_run.py $N"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o st -- $RUN > "$OUT/trace.out" 2> "$OUT/trace.err"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/$C" -o st -- $RUN > "$OUT/$C.out" 2> "$OUT/$C.err"
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/cal_$C" -o cal -- python $R/tools/pmc_calibrate.py > "$OUT/cal_$C.out" 2> "$OUT/cal_$C.err"
done
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/sq$i" -o st -- $RUN > /dev/null 2> "$OUT/sq$i.err"
done
find "$OUT" -name "*.csv" | head -40
