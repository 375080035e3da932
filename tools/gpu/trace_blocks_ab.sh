#!/bin/bash
# A/B of the traceback's wavefronts in flight (PG_TRACE_BLOCKS; default 8 per CU = 2 048): tools/gpu/trace_blocks_ab.sh -> gpurun_out/trace_blocks_ab.jsonl
mkdir -p gpurun_out; : > gpurun_out/trace_blocks_ab.jsonl
for round in 1 2; do for tb in ${TBS:-2048 1024 1536 3072}; do
  PG_TRACE_BLOCKS=$tb timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --sites-steps 0 --stream-batches 0 2>/dev/null | TB=$tb python -c '
import json,sys,os
d=json.loads(sys.stdin.readline())
print(json.dumps({"trace_blocks": int(os.environ["TB"]), "Mreads_s": round(d["value"]/1e6,3), "ms_per_step": round(d["ms_per_step"],3), "fill_ms": round(d["roofline"]["avg_launch_ms"],3), "trace_ms_total": round(d["kernel_ms"]["trace"],1)}))' | tee -a gpurun_out/trace_blocks_ab.jsonl
done; done
