#!/bin/bash
# A/B of the bench's workspace size (launches per 1 M-read step): tools/gpu/ws_ab.sh  -> gpurun_out/ws_ab.jsonl
mkdir -p gpurun_out; : > gpurun_out/ws_ab.jsonl
for ws in 128 192 128 192; do
  timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --workspace-gib $ws 2>gpurun_out/ws_ab.err | WS=$ws python -c '
import json,sys,os
d=json.loads(sys.stdin.readline())
print(json.dumps({"workspace_gib": int(os.environ["WS"]), "Mreads_s": round(d["value"]/1e6,3), "ms_per_step": round(d["ms_per_step"],3), "launches": d["roofline"].get("launches"), "avg_launch_ms": d["roofline"].get("avg_launch_ms")}))' | tee -a gpurun_out/ws_ab.jsonl
done
tail -3 gpurun_out/ws_ab.err
