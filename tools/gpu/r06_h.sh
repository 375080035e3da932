#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6h; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_path or test_gpu_workflow or host_cpp or hand_over or cascade or kmer_filter or test_gpu_counts") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
for n in 1000000 19200; do python tools/path_probe.py $n | tee -a $O/path_probe.jsonl; done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o st -- python $R/tools/path_probe.py 19200 2 > /dev/null 2> "$O/trace.err"
cd $R; grep -h "pg_path_kernel" $O/trace/*/*kernel_stats.csv 2>/dev/null | head -3 || find $O/trace -name "*stats*" | head
