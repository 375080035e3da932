#!/bin/bash
# config-3 headline (sites leg as the main workload) under rocprofv3 kernel stats
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/q; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --workload config3 --steps 5 --warmup 1 --no-cpu-baseline > $O/config3.json 2> $O/config3.err; tail -c 1200 $O/config3.json; echo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o c3 -- python bench.py --workload config3 --steps 5 --warmup 1 --no-cpu-baseline > $O/config3_prof.json 2> $O/config3_prof.err
for f in $(find $O/k -name "*kernel_stats.csv"); do head -12 $f | cut -c1-220; cp $f $O/config3_kernel_stats.csv; done
rm -rf $O/k
