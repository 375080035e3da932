#!/bin/bash
# klib stage: kernel split under rocprofv3 (kernel stats only)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/m; mkdir -p $O
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stage_kernels -o stage -- python $R/tools/stage_probe.py 1000000 > $O/stage.json 2> $O/stage.err
cat $O/stage.json | cut -c1-700
for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; head -14 $f | cut -c1-260; cp $f $O/stage_kernel_stats.csv; done
find $O -name "*.db" -delete; find $O -name "*trace.csv" -delete
