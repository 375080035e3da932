#!/bin/bash
# stream priorities: the count / copy streams one level up (default) against all three at the default level (PG_STREAM_PRIORITY=0)
mkdir -p gpurun_out; : > gpurun_out/prio_ab.jsonl
for round in 1 2; do for pr in default 0; do
  if [ $pr = 0 ]; then export PG_STREAM_PRIORITY=0; else unset PG_STREAM_PRIORITY; fi
  timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --sites-steps 0 --stream-batches 0 2>/dev/null | PR=$pr python -c '
import json,sys,os
d=json.loads(sys.stdin.readline())
print(json.dumps({"stream_priority": os.environ["PR"], "Mreads_s": round(d["value"]/1e6,3), "ms_per_step": round(d["ms_per_step"],3), "fill_ms": round(d["roofline"]["avg_launch_ms"],3), "trace_ms_total": round(d["kernel_ms"]["trace"],1)}))' | tee -a gpurun_out/prio_ab.jsonl
done; done
