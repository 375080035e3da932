#!/bin/bash
# SQ counters of the path kernel alone (19 200 reads = a workflow batch; 1 M reads)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for N in 1000000; do
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR" "SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_FLAT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$O/n$N/set$i" -o sq -- python $R/tools/path_probe.py $N 2 > /dev/null 2> "$O/n${N}_set$i.err"
done
done
cd $R
python - <<PY
import csv, glob, collections, json
for N in (1000000,):
    agg = collections.defaultdict(float); calls = 0
    for f in glob.glob('$O/n%d/set*/**/*counter_collection.csv' % N, recursive=True):
        for row in csv.DictReader(open(f)):
            if 'pg_path_kernel' in row['Kernel_Name']:
                agg[row['Counter_Name']] += float(row['Counter_Value'])
    # 4 launches per run (warm-up, 2 timed, flags)
    print(N, json.dumps({k: v / 4 for k, v in sorted(agg.items())}))
PY
