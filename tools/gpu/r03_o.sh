#!/bin/bash
# round 3, call O: the traceback launch bounded to N wavefronts in flight (PG_TRACE_BLOCKS) -- does spreading its reads over the
# next chunk's fill take the tax off the fill?  Parity first, then the headline loop per setting.
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_o; mkdir -p $O
PG_TRACE_BLOCKS=512 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
for B in 0 8192 4096 2048 1024 512 256; do
  PG_TRACE_BLOCKS=$B timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --stream-batches 0 --sites-steps 0 --collective off > $O/bench_$B.json 2> $O/bench_$B.err
  python - $O/bench_$B.json $B <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("blocks", sys.argv[2], "value %.3f M" % (d["value"] / 1e6), "ms/step %.2f" % d["ms_per_step"], "fill ms", d["roofline"].get("avg_launch_ms"))
except Exception as e:
    print("blocks", sys.argv[2], "failed", e)
PY
done
