#!/bin/bash
# the fragment kernel with LDS sized to the graphs: parity suites that count, then its duration under the fill (rocprofv3 --stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/frag_lds; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_counts.py tests/test_gpu_scale.py tests/test_gpu_workflow.py tests/test_gpu_configs.py -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"; echo "rocprof rc=$?"
head -5 $O/prof/bench_kernel_stats.csv | cut -c1-150
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof3" -o bench -- python $R/bench.py --workload config3 --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench3_under_rocprof.json" 2> "$O/prof3.err"; echo "rocprof3 rc=$?"
head -6 $O/prof3/bench_kernel_stats.csv | cut -c1-150
