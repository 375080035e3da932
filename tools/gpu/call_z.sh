#!/bin/bash
# ThreadSanitizer pass of the host workflow (lanes, device slots, pooled staging) on the 10 000-site data set, 1 000 sites
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/z; mkdir -p $O
W=$R/tools/e2e/_data
head -1500 $W/graphs.txt > $O/graphs_1500.txt
export PG_E2E_REPS=2
export TSAN_OPTIONS="report_signal_unsafe=0 halt_on_error=0 history_size=4 log_path=$O/tsan"
( time timeout 900 setarch x86_64 -R $R/tools/variants/grmpy_batch_tsan $W/ref.fa $W/manifest.txt $O/graphs_1500.txt 16 $O/g_tsan.json 0 0 1 ) > $O/run.json 2> $O/run.err; echo "rc=$?"; tail -c 400 $O/run.json; echo
ls $O | head; for f in $O/tsan.*; do [ -f "$f" ] && { grep -c "WARNING: ThreadSanitizer" $f; grep -A12 "WARNING: ThreadSanitizer" $f | grep -E "paragraph|grmpy|pghost|common::|#0|#1|#2" | head -40; }; done | head -80
