#!/bin/bash
# what the device does in a pass of `paragraph`'s default cascade (path + gssw) and of the gssw-only workflow: kernel traces
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5m; mkdir -p $O
export TMPDIR=/tmp
D=/dev/shm/pg_mode_trace
for mode in ${MODES:-path gssw}; do
  opt=""; [ $mode = path ] && opt="path_sequence_matching=1"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/$mode -o t -- python $R/tools/e2e/mode_trace.py run $D 4 $opt > $O/$mode.json 2> $O/$mode.err)
  f=$(find $O/$mode -name '*kernel_trace.csv' | head -1)
  s=$(python -c "import json;print(json.load(open('$O/$mode.json'))['s_per_pass'])")
  python tools/e2e/mode_trace.py summary $f 5 $s > $O/${mode}_device.json
  cat $O/$mode.json; cat $O/${mode}_device.json
  rm -rf $O/$mode
done
